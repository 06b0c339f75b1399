// Radix-2^k number-theoretic transforms over BN254 Fr and the H-polynomial pipeline.
//
// Replaces bellman_ce 0.3.0 `EvaluationDomain::{fft, ifft, coset_fft, icoset_fft, mul_assign,
// sub_assign, divide_by_z_on_coset}` as driven by `create_proof` (the reference reaches it through
// zinc-vm/src/core/circuit/facade.rs:153).  Results are exact field values, so they do not depend
// on the decomposition used here.
//
// Decomposition: N = R1*R2*R3 (each a power of two, at most 1024).  Index i = i1*R2*R3 + i2*R3 + i3,
// frequency k = k1 + k2*R1 + k3*R1*R2.  Pass A transforms over i1 for a tile of adjacent i3 and
// multiplies by w_N^((i2*R3+i3)*k1); pass B transforms over i2 and multiplies by w_N^(i3*k2*R1);
// pass C transforms over i3 (contiguous) for a tile of adjacent k1 and scatters to natural order.
// Each workgroup stages a T x R tile (T*R <= 1024 elements, 32 KiB) in LDS and runs the R-point
// transforms there, so every pass is one read and one write of the array.
#include <memory>

#include "common.hpp"
#include "fp29.hpp"

namespace zg {

static constexpr unsigned MAX_RADIX_BITS = 10;
static constexpr unsigned TILE_ELEMS = 1024;
// Domains up to this size have their whole H computation done by workgroups that keep a full vector in LDS
// (2^12 elements of nine words = 144 KiB of the CU's 160).
static constexpr unsigned SMALL_MAX_LOG = 12;
static constexpr unsigned SMALL_THREADS = 1024;

struct PassParams {
  uint32_t logR, logT;
  uint32_t in_jstride, in_tstride, out_jstride, out_tstride;
  uint32_t bdiv;
  uint32_t in_hi, in_lo, out_hi, out_lo, tw_lo_mul;
  uint32_t tw_scale;  // inter-pass twiddle exponent scale (k_pass_twiddles), 0: none
  uint32_t load_j_fast;
  uint32_t nq;  // non-zero: logical tile L splits as q = L % nq, rem = L / nq (q fastest) instead of by bdiv
  // Chunked operands (a sharded proof's exchange buffers): the vector's natural index i lives at
  // (i >> rlog) * rstride + (i & (2^rlog - 1)) -- chunk i >> rlog of 2^rlog elements starts rstride elements after
  // the previous one.  rstride = 0: the vector is contiguous.  Factor tables, mulvec and subvec stay natural.
  uint32_t in_rlog, in_rstride, out_rlog, out_rstride;
};

// N = R1*R2*R3: the passes of one transform and their launch sizes
struct NttPlan {
  unsigned r1 = 0, r2 = 0, r3 = 0;
  PassParams a = {}, b = {}, c = {};
  uint32_t grid_a = 0, grid_b = 0, grid_c = 0;
};

// Every table is a full one (N entries): with 288 GB of HBM a few multiples of the domain size cost
// nothing, and a twiddle or a coset factor is then one streamed load instead of a field product.
// Tables the transform kernel multiplies by are kept as nine 29-bit limbs in radix-2^261 Montgomery
// form (Fr29): the data stays in the ABI's radix-2^256 form, and x_256 * w_261 * 2^-261 is again in it.
struct NttDomain {
  unsigned log_n = 0;
  NttPlan plan;
  Fr29* tw_a[2] = {};      // [dir][N]  pass A's inter-pass twiddles in its store order (dir 1 = inverse root)
  Fr29* tw_b[2] = {};      // [dir][N]  pass B's (three-pass transforms only)
  Fr29* gm = nullptr;      // g^i / m
  Fr29* ginvm = nullptr;   // g^-i / m
  Fr29* hz = nullptr;      // 32 * g^-i / (m * (g^m - 1)): the trailing factor of the fused H pipeline
  Fr29* kz = nullptr;      // [1]  1 / (m * (g^m - 1))
  Fr29* w_small[2] = {};   // [dir][512]  w_1024^e
  Fr29* w_dom[2] = {};     // [dir][N/2]  w_N^e, only for domains the one-workgroup kernels take (N <= 2^SMALL_MAX_LOG)
  Fr* g = nullptr;         // g^i (ABI form, for the stand-alone coset_fft pre-scaling)
  Fr minv, zinv;           // 1/m, 1/(g^m - 1)
  void* block = nullptr;   // one allocation behind all of the above
};

// out[i] = c * base^i.  A workgroup raises base to its first index once and then each lane walks
// 16 consecutive powers, so the table costs ~1 product per entry.
__global__ __launch_bounds__(256) void k_pow_table(Fr* out, uint32_t count, Fr base, Fr c) {
  const uint32_t first = (blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (first >= count) return;
  Fr v = mul(c, pow_u64(base, first));
  for (uint32_t k = 0; k < 16 && first + k < count; k++) {
    out[first + k] = v;
    v = mul(v, base);
  }
}

// out[i] = c * base^i for i < count, device memory, on the context's stream
zg16_status pow_table_device(zg16_ctx* ctx, Fr* out, size_t count, const Fr& base, const Fr& c) {
  if (!count) return ZG16_OK;
  k_pow_table<<<(unsigned)((count + 4095) / 4096), 256, 0, ctx->stream>>>(out, (uint32_t)count, base, c);
  ZG_HIP(ctx, hipGetLastError());
  return ZG16_OK;
}

// ABI-form table -> Fr29 table (radix change by one product)
__global__ __launch_bounds__(256) void k_to_fr29(const Fr* __restrict__ in, Fr29* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = from_r256<Fr29Params>(in[i]);
}

// LDS tile of nine-word elements.  Element i lives at slot(i) * 9: the odd word stride maps 64
// distinct slots mod 64 to 64 distinct banks, and slot() folds index bits 6..9 into bits 0..5 so that
// the lanes of a wave hit distinct slots mod 64 in every access pattern of the pass -- stride 4, 16
// and 64 units of the radix-4 rounds, the bit-reversed load, the linear store -- instead of piling
// up four to sixteen deep on the same banks.
ZG_D uint32_t slot(uint32_t i) { return i ^ (((i >> 6) & 3) * 21) ^ ((i >> 8) & 3); }
ZG_D Fr29 lds_get(const uint32_t* s, uint32_t i) {
  Fr29 r;
  const uint32_t* e = s + slot(i) * 9;
#pragma unroll
  for (int k = 0; k < 9; k++) r.v[k] = e[k];
  return r;
}
ZG_D void lds_put(uint32_t* s, uint32_t i, const Fr29& x) {
  uint32_t* e = s + slot(i) * 9;
#pragma unroll
  for (int k = 0; k < 9; k++) e[k] = x.v[k];
}

// Workgroups go to the eight XCDs round-robin (workgroup b runs on XCD b mod 8) and every XCD has its own L2.  A
// tile narrower than a 128-byte line (four elements) shares its lines with the tiles next to it, so tiles that are
// neighbours in memory are given to the same XCD at about the same time: XCD x works through the contiguous run of
// logical tiles [x G/8, (x+1) G/8).  The partial lines then meet in that L2 instead of being fetched from (and
// written back to) HBM once per tile.
ZG_D uint32_t xcd_tile(uint32_t b, uint32_t grid) { return (grid & 7) ? b : (b & 7) * (grid >> 3) + (b >> 3); }

// One decimation-in-time butterfly on lazy values: (x0, x1) <- (x0 + t, x0 - t) with t = x1 * w, or
// t = x1 when the twiddle is one (TRIVIAL).  The subtrahend is a fresh product (below 2p) or, in the
// trivial case, a value below `KSUB` p, so the result grows by an additive constant per stage instead
// of doubling: ten stages keep every value below 40p (products need operand bounds whose product
// stays below 169).  NORM_ADD: normalise the sum too (it is about to be stored or multiplied).
template <bool TRIVIAL, bool NORM_ADD>
ZG_D void butterfly29(Fr29& x0, Fr29& x1, const Fr29& w, const uint32_t (&K)[9]) {
  const Fr29 t = TRIVIAL ? x1 : mul(x1, w);
  Fr29 d;
#pragma unroll
  for (int k = 0; k < 9; k++) d.v[k] = x0.v[k] + K[k] - t.v[k];
  x1 = norm(d);
  Fr29 a = add_lazy(x0, t);
  x0 = NORM_ADD ? norm(a) : a;
}

// One pass of the transform over a T x R tile staged in LDS (see the file comment).  Elements are
// unpacked to nine 29-bit limbs on load, go through log2(R) decimation-in-time stages two at a time
// (a lane holds the four elements of a radix-4 unit in registers, so LDS is touched every second
// stage), and every output takes exactly one product -- the inter-pass twiddle, the trailing scale
// factor, another vector's element (mulvec) or one -- which also brings it back below 2p for the
// canonical packed store.  subvec, when given, is subtracted element-wise from what is stored.
struct PassExtras {
  const Fr29* factor;  // per-element factor in store order, or a single constant (factor_const)
  const Fr* mulvec;    // used when factor is null: out[o] = v * mulvec[o] * 2^-261 (mulvec may alias out)
  const Fr* subvec;
  uint32_t factor_const;
};

// Up to three vectors of one size go through the same pass in one launch (blockIdx.y picks the vector): below 2^20
// elements a pass has fewer workgroups than the GPU has slots, and the a, b, c transforms of a proof are independent.
static constexpr int PASS_BATCH = 3;
struct PassBatch {
  const Fr* in[PASS_BATCH];
  Fr* out[PASS_BATCH];
  PassExtras ex[PASS_BATCH];
};

__global__ __launch_bounds__(256, 4) void k_ntt_pass(PassBatch pb, PassParams p, const Fr29* __restrict__ w_small) {
  __shared__ uint32_t s[TILE_ELEMS * 9];
  const Fr* in = pb.in[blockIdx.y];
  Fr* out = pb.out[blockIdx.y];
  const PassExtras ex = pb.ex[blockIdx.y];
  const uint32_t tid = threadIdx.x;
  const uint32_t logR = p.logR, logT = p.logT;
  const uint32_t R = 1u << logR, T = 1u << logT, E = R << logT;
  const uint32_t tile = xcd_tile(blockIdx.x, gridDim.x);
  const uint32_t q = p.nq ? tile % p.nq : tile / p.bdiv, rem = p.nq ? tile / p.nq : tile % p.bdiv;
  const uint32_t in_base = q * p.in_hi + rem * p.in_lo;
  const uint32_t out_base = q * p.out_hi + rem * p.out_lo;
  const uint32_t wshift = MAX_RADIX_BITS - logR;  // w_R^e = w_1024^(e << wshift)

  // load in bit-reversed position: decimation in time then leaves natural order
  for (uint32_t idx = tid; idx < E; idx += 256) {
    uint32_t j, t;
    if (p.load_j_fast) {
      j = idx & (R - 1);
      t = idx >> logR;
    } else {
      t = idx & (T - 1);
      j = idx >> logT;
    }
    uint32_t src = in_base + j * p.in_jstride + t * p.in_tstride;
    if (p.in_rstride) src = (src >> p.in_rlog) * p.in_rstride + (src & ((1u << p.in_rlog) - 1));
    const Fr x = in[src];
    lds_put(s, (t << logR) + (__brev(j) >> (32 - logR)), unpack29<Fr29Params>(x));
  }
  __syncthreads();

  uint32_t st = 0;
  for (; st + 1 < logR; st += 2) {
    const uint32_t h = 1u << st;
    for (uint32_t u = tid; u < (E >> 2); u += 256) {
      const uint32_t t = u >> (logR - 2), qq = u & ((R >> 2) - 1);
      const uint32_t grp = qq >> st, pos = qq & (h - 1);
      const uint32_t i0 = (t << logR) + (grp << (st + 2)) + pos;
      Fr29 a = lds_get(s, i0), b = lds_get(s, i0 + h), c = lds_get(s, i0 + 2 * h), d = lds_get(s, i0 + 3 * h);
      // stage st: pairs (a, b) and (c, d), twiddle w_(2h)^pos; stage st+1: pairs (a, c) with w_(4h)^pos and
      // (b, d) with w_(4h)^(pos + h)
      if (st == 0) {
        // w_2^0 = w_4^0 = 1: three of the four twiddles are one (inputs are canonical, below p)
        butterfly29<true, false>(a, b, a, Fr29Params::K2);
        butterfly29<true, false>(c, d, c, Fr29Params::K2);
        butterfly29<true, true>(a, c, a, Fr29Params::K4);      // c < 2p + ... below 4p
        const Fr29 w3 = w_small[(R >> 2) << wshift];
        butterfly29<false, true>(b, d, w3, Fr29Params::K2);
      } else {
        const Fr29 w1 = w_small[(pos << (logR - 1 - st)) << wshift];
        butterfly29<false, false>(a, b, w1, Fr29Params::K2);
        butterfly29<false, false>(c, d, w1, Fr29Params::K2);
        const Fr29 w2 = w_small[(pos << (logR - 2 - st)) << wshift];
        butterfly29<false, true>(a, c, w2, Fr29Params::K2);
        const Fr29 w3 = w_small[((pos + h) << (logR - 2 - st)) << wshift];
        butterfly29<false, true>(b, d, w3, Fr29Params::K2);
      }
      lds_put(s, i0, a);
      lds_put(s, i0 + h, b);
      lds_put(s, i0 + 2 * h, c);
      lds_put(s, i0 + 3 * h, d);
    }
    __syncthreads();
  }
  if (st < logR) {  // odd number of stages: one radix-2 round
    const uint32_t h = 1u << st;
    for (uint32_t u = tid; u < (E >> 1); u += 256) {
      const uint32_t t = u >> (logR - 1), qq = u & ((R >> 1) - 1);
      const uint32_t grp = qq >> st, pos = qq & (h - 1);
      const uint32_t i0 = (t << logR) + (grp << (st + 1)) + pos;
      Fr29 a = lds_get(s, i0), b = lds_get(s, i0 + h);
      const Fr29 w1 = w_small[(pos << (logR - 1 - st)) << wshift];
      butterfly29<false, true>(a, b, w1, Fr29Params::K2);
      lds_put(s, i0, a);
      lds_put(s, i0 + h, b);
    }
    __syncthreads();
  }

  for (uint32_t idx = tid; idx < E; idx += 256) {
    const uint32_t t = idx & (T - 1), kk = idx >> logT;
    const Fr29 v = lds_get(s, (t << logR) + kk);
    const uint32_t o = out_base + kk * p.out_jstride + t * p.out_tstride;
    // the inter-pass twiddle or the trailing scale factor, laid out in store order (a streaming read)
    Fr29 f;
    if (ex.factor)
      f = ex.factor[ex.factor_const ? 0 : o];
    else if (ex.mulvec)
      f = unpack29<Fr29Params>(ex.mulvec[o]);
    else
      f = Fr29::one();
    Fr r = pack29<Fr29Params, FrParams>(mul(v, f));
    if (ex.subvec) r = sub(r, ex.subvec[o]);
    out[p.out_rstride ? (o >> p.out_rlog) * p.out_rstride + (o & ((1u << p.out_rlog) - 1)) : o] = r;
  }
}

// ---- small domains: a whole vector in one workgroup's LDS ---------------------------------------------
// A proof of a small circuit is a chain of launches with almost no work in each, and the transforms were twelve of
// them at 2^11 rows.  Up to 2^SMALL_MAX_LOG rows a vector fits one workgroup's LDS, so a workgroup runs BOTH
// transforms of a chain without leaving the kernel (k_small_chains: one workgroup each for a, b and c) and a
// second launch multiplies, transforms back and subtracts (k_small_finish): two launches for the H polynomial.
// Same butterflies, same lazy bounds as k_ntt_pass (values grow by 2p per stage: below 30p after twelve).

// log2(n) decimation-in-time stages over the n elements in LDS: bit-reversed positions in, natural order out.
// w[e] = w_n^e for e < n / 2.  Ends with a barrier.
// wsh: the table belongs to a domain 2^wsh times as large (its every 2^wsh-th entry is this domain's root power).
ZG_D void small_dit(uint32_t* s, uint32_t logN, const Fr29* __restrict__ w, uint32_t wsh = 0) {
  const uint32_t tid = threadIdx.x, N = 1u << logN;
  uint32_t st = 0;
  for (; st + 1 < logN; st += 2) {
    const uint32_t h = 1u << st;
    for (uint32_t u = tid; u < (N >> 2); u += SMALL_THREADS) {
      const uint32_t grp = u >> st, pos = u & (h - 1);
      const uint32_t i0 = (grp << (st + 2)) + pos;
      Fr29 a = lds_get(s, i0), b = lds_get(s, i0 + h), c = lds_get(s, i0 + 2 * h), d = lds_get(s, i0 + 3 * h);
      if (st == 0) {
        butterfly29<true, false>(a, b, a, Fr29Params::K2);
        butterfly29<true, false>(c, d, c, Fr29Params::K2);
        butterfly29<true, true>(a, c, a, Fr29Params::K4);
        const Fr29 w3 = w[(N >> 2) << wsh];
        butterfly29<false, true>(b, d, w3, Fr29Params::K2);
      } else {
        const Fr29 w1 = w[(pos << (logN - 1 - st)) << wsh];
        butterfly29<false, false>(a, b, w1, Fr29Params::K2);
        butterfly29<false, false>(c, d, w1, Fr29Params::K2);
        const Fr29 w2 = w[(pos << (logN - 2 - st)) << wsh];
        butterfly29<false, true>(a, c, w2, Fr29Params::K2);
        const Fr29 w3 = w[((pos + h) << (logN - 2 - st)) << wsh];
        butterfly29<false, true>(b, d, w3, Fr29Params::K2);
      }
      lds_put(s, i0, a);
      lds_put(s, i0 + h, b);
      lds_put(s, i0 + 2 * h, c);
      lds_put(s, i0 + 3 * h, d);
    }
    __syncthreads();
  }
  if (st < logN) {  // odd number of stages: one radix-2 round
    const uint32_t h = 1u << st;
    for (uint32_t u = tid; u < (N >> 1); u += SMALL_THREADS) {
      const uint32_t grp = u >> st, pos = u & (h - 1);
      const uint32_t i0 = (grp << (st + 1)) + pos;
      Fr29 a = lds_get(s, i0), b = lds_get(s, i0 + h);
      const Fr29 w1 = w[(pos << (logN - 1 - st)) << wsh];
      butterfly29<false, true>(a, b, w1, Fr29Params::K2);
      lds_put(s, i0, a);
      lds_put(s, i0 + h, b);
    }
    __syncthreads();
  }
}

struct SmallChains {
  const Fr* in[3];
  Fr* out[3];
};

// Workgroup 0: a <- coset_fft(ifft(a)); workgroup 1: the same for b; workgroup 2: c <- ifft(c) / Z.
// gm[i] = g^i / m, kz[0] = 1 / (m Z); w_inv / w_fwd: the domain's roots for the two directions.  1 <= logN.
__global__ __launch_bounds__(SMALL_THREADS) void k_small_chains(SmallChains v, uint32_t logN,
                                                                const Fr29* __restrict__ w_inv, const Fr29* __restrict__ w_fwd,
                                                                const Fr29* __restrict__ gm, const Fr29* __restrict__ kz) {
  __shared__ uint32_t s[(1u << SMALL_MAX_LOG) * 9];
  const uint32_t which = blockIdx.x, N = 1u << logN, tid = threadIdx.x, rs = 32 - logN;
  const Fr* in = v.in[which];
  Fr* out = v.out[which];
  for (uint32_t i = tid; i < N; i += SMALL_THREADS) lds_put(s, __brev(i) >> rs, unpack29<Fr29Params>(in[i]));
  __syncthreads();
  small_dit(s, logN, w_inv);
  if (which == 2) {
    const Fr29 f = kz[0];
    for (uint32_t i = tid; i < N; i += SMALL_THREADS) out[i] = pack29<Fr29Params, FrParams>(mul(lds_get(s, i), f));
    return;
  }
  // times g^i / m, and into bit-reversed position for the forward transform: positions i and brev(i) swap, so every
  // lane works on its own pair and no barrier is needed inside the step
  for (uint32_t i = tid; i < N; i += SMALL_THREADS) {
    const uint32_t j = __brev(i) >> rs;
    if (i < j) {
      const Fr29 xi = mul(lds_get(s, i), gm[i]), xj = mul(lds_get(s, j), gm[j]);
      lds_put(s, j, xi);
      lds_put(s, i, xj);
    } else if (i == j) {
      lds_put(s, i, mul(lds_get(s, i), gm[i]));
    }
  }
  __syncthreads();
  small_dit(s, logN, w_fwd);
  const Fr29 one = Fr29::one();
  for (uint32_t i = tid; i < N; i += SMALL_THREADS) out[i] = pack29<Fr29Params, FrParams>(mul(lds_get(s, i), one));
}

// out <- icoset_fft(a * b) / Z - c for the coset evaluations a, b and c = ifft(c) / Z (out may be a).
// hz[i] = 32 g^-i / (m Z): the product of two radix-2^256 values reduced by 2^261 is a * b / 32.
__global__ __launch_bounds__(SMALL_THREADS) void k_small_finish(const Fr* a, const Fr* __restrict__ b, const Fr* __restrict__ c,
                                                                Fr* out, uint32_t logN, const Fr29* __restrict__ w_inv,
                                                                const Fr29* __restrict__ hz) {
  __shared__ uint32_t s[(1u << SMALL_MAX_LOG) * 9];
  const uint32_t N = 1u << logN, tid = threadIdx.x, rs = 32 - logN;
  for (uint32_t i = tid; i < N; i += SMALL_THREADS)
    lds_put(s, __brev(i) >> rs, mul(unpack29<Fr29Params>(a[i]), unpack29<Fr29Params>(b[i])));
  __syncthreads();
  small_dit(s, logN, w_inv);
  for (uint32_t i = tid; i < N; i += SMALL_THREADS)
    out[i] = sub(pack29<Fr29Params, FrParams>(mul(lds_get(s, i), hz[i])), c[i]);
}

// The same two launches with every transform cut in two halves on two workgroups (a small circuit's proof waits
// for these kernels, and one CU per vector is compute-bound: the halves take half as long).  With M = N / 2:
//   an inverse transform splits by OUTPUT parity (decimation in frequency): X[2k + h] is the half-size transform of
//   u_0[i] = x[i] + x[i + M] or u_1[i] = (x[i] - x[i + M]) v^i, both of which a workgroup forms on load from the
//   whole input;
//   a forward transform splits by INPUT parity (decimation in time): the workgroup that holds a'[2k + h] computes
//   E = fft_M(even part) or O = fft_M(odd part) and stores that; the closing butterflies A[k] = E[k] + w^k O[k],
//   A[k + M] = E[k] - w^k O[k] are done by whoever reads A next -- the finish kernel, on load.
// So nothing is exchanged between workgroups inside a launch.  Inputs and outputs must be different arrays (both
// workgroups of a vector read all of it).
//
// k_small_chains_split, workgroup 2 * which + h: a, b: out[h M + k] = E[k] (h = 0) or O[k] (h = 1) of
// coset_fft(ifft(in)); c: out[2 k + h] = (ifft(c) / Z)[2 k + h].  2 <= logN.
__global__ __launch_bounds__(SMALL_THREADS) void k_small_chains_split(SmallChains v, uint32_t logN,
                                                                      const Fr29* __restrict__ w_inv, const Fr29* __restrict__ w_fwd,
                                                                      const Fr29* __restrict__ gm, const Fr29* __restrict__ kz) {
  __shared__ uint32_t s[(1u << (SMALL_MAX_LOG - 1)) * 9];
  const uint32_t which = blockIdx.x >> 1, h = blockIdx.x & 1, tid = threadIdx.x;
  const uint32_t logM = logN - 1, M = 1u << logM, rs = 32 - logM;
  const Fr* in = v.in[which];
  Fr* out = v.out[which];
  for (uint32_t i = tid; i < M; i += SMALL_THREADS) {
    const Fr29 x0 = unpack29<Fr29Params>(in[i]), x1 = unpack29<Fr29Params>(in[i + M]);
    const Fr29 u = h ? mul(sub_k(x0, x1, Fr29Params::K2), w_inv[i]) : norm(add_lazy(x0, x1));
    lds_put(s, __brev(i) >> rs, u);
  }
  __syncthreads();
  small_dit(s, logM, w_inv, 1);
  if (which == 2) {
    const Fr29 f = kz[0];
    for (uint32_t k = tid; k < M; k += SMALL_THREADS) out[2 * k + h] = pack29<Fr29Params, FrParams>(mul(lds_get(s, k), f));
    return;
  }
  // times g^j / m at j = 2 k + h, and into bit-reversed position for the forward half-size transform
  for (uint32_t k = tid; k < M; k += SMALL_THREADS) {
    const uint32_t j = __brev(k) >> rs;
    if (k < j) {
      const Fr29 xk = mul(lds_get(s, k), gm[2 * k + h]), xj = mul(lds_get(s, j), gm[2 * j + h]);
      lds_put(s, j, xk);
      lds_put(s, k, xj);
    } else if (k == j) {
      lds_put(s, k, mul(lds_get(s, k), gm[2 * k + h]));
    }
  }
  __syncthreads();
  small_dit(s, logM, w_fwd, 1);
  const Fr29 one = Fr29::one();
  for (uint32_t k = tid; k < M; k += SMALL_THREADS) out[h * M + k] = pack29<Fr29Params, FrParams>(mul(lds_get(s, k), one));
}

// k_small_finish_split, workgroup h: out[2 k + h] = (icoset_fft(A * B) / Z - c)[2 k + h] from the halves E | O of A
// and of B that k_small_chains_split left (ea[0 .. M) = E, ea[M .. N) = O).  With t = w^i O[i]:
//   (P[i] + P[i + M]) / 2 = Ea Eb + ta tb,   (P[i] - P[i + M]) / 2 = Ea tb + ta Eb   (one fused product each),
// and the 2 goes into the closing product with hz.
__global__ __launch_bounds__(SMALL_THREADS) void k_small_finish_split(const Fr* __restrict__ ea, const Fr* __restrict__ eb,
                                                                      const Fr* __restrict__ c, Fr* __restrict__ out, uint32_t logN,
                                                                      const Fr29* __restrict__ w_inv, const Fr29* __restrict__ w_fwd,
                                                                      const Fr29* __restrict__ hz) {
  __shared__ uint32_t s[(1u << (SMALL_MAX_LOG - 1)) * 9];
  const uint32_t h = blockIdx.x, tid = threadIdx.x;
  const uint32_t logM = logN - 1, M = 1u << logM, rs = 32 - logM;
  for (uint32_t i = tid; i < M; i += SMALL_THREADS) {
    const Fr29 a0 = unpack29<Fr29Params>(ea[i]), b0 = unpack29<Fr29Params>(eb[i]);
    const Fr29 w = w_fwd[i];
    const Fr29 ta = mul(unpack29<Fr29Params>(ea[i + M]), w), tb = mul(unpack29<Fr29Params>(eb[i + M]), w);
    const Fr29 na = sub_k(Fr29::zero(), ta, Fr29Params::K2);  // -ta
    const Fr29 u = h ? mul(mul_sub(a0, tb, na, b0, Fr29Params::KS4), w_inv[i]) : mul_sub(a0, b0, na, tb, Fr29Params::KS4);
    lds_put(s, __brev(i) >> rs, u);
  }
  __syncthreads();
  small_dit(s, logM, w_inv, 1);
  for (uint32_t k = tid; k < M; k += SMALL_THREADS) {
    const uint32_t j = 2 * k + h;
    const Fr29 y = lds_get(s, k);
    out[j] = sub(pack29<Fr29Params, FrParams>(mul(add_lazy(y, y), hz[j])), c[j]);
  }
}

// factor[o] = w^e for the inter-pass twiddle exponent e of every element a pass stores, in store order.
__global__ __launch_bounds__(256) void k_pass_twiddles(PassParams p, const Fr29* __restrict__ w, Fr29* __restrict__ factor) {
  const uint32_t logR = p.logR, logT = p.logT;
  const uint32_t T = 1u << logT, E = (1u << logR) << logT;
  const uint32_t q = blockIdx.x / p.bdiv, rem = blockIdx.x % p.bdiv;
  const uint32_t out_base = q * p.out_hi + rem * p.out_lo;
  const uint32_t m0 = rem * p.tw_lo_mul;
  for (uint32_t idx = threadIdx.x; idx < E; idx += 256) {
    const uint32_t t = idx & (T - 1), kk = idx >> logT;
    factor[out_base + kk * p.out_jstride + t * p.out_tstride] = w[(m0 + t) * kk * p.tw_scale];
  }
}

// x[i] *= table[i]
__global__ __launch_bounds__(256) void k_scale_powers(Fr* x, uint32_t n, const Fr* __restrict__ table) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    x[i] = mul(x[i], table[i]);
}

__global__ __launch_bounds__(256) void k_scale_const(Fr* x, uint32_t n, Fr c) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    x[i] = mul(x[i], c);
}

// a[i] = (a[i]*b[i] - c[i]) * zinv
__global__ __launch_bounds__(256) void k_mul_sub_scale(Fr* a, const Fr* __restrict__ b,
                                                       const Fr* __restrict__ c, uint32_t n, Fr zinv) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    a[i] = mul(sub(mul(a[i], b[i]), c[i]), zinv);
}

static Fr host_root_of_unity() {
  // 7^((r-1) / 2^28)
  uint32_t e[8];
  for (int i = 0; i < 8; i++) e[i] = FrParams::mod(i);
  e[0] -= 1;
  uint32_t sh[8];
  for (int i = 0; i < 8; i++) {
    uint64_t lo = e[i] >> 28;
    uint64_t hi = (i + 1 < 8) ? ((uint64_t)e[i + 1] << 4) : 0;
    sh[i] = (uint32_t)(lo | hi);
  }
  return pow_limbs(from_u64<FrParams>(7), sh, 8);
}

static unsigned max_radix_bits() {
  static unsigned v = 0;
  if (!v) {
    const char* e = getenv("ZG16_NTT_RADIX_BITS");
    v = e ? (unsigned)atoi(e) : MAX_RADIX_BITS;
    if (v < 2) v = 2;
    if (v > MAX_RADIX_BITS) v = MAX_RADIX_BITS;
  }
  return v;
}

static NttPlan make_plan(unsigned n) {
  NttPlan pl;
  const unsigned rb = max_radix_bits();
  unsigned r1, r2, r3;
  if (n <= rb) {
    r1 = 0; r2 = 0; r3 = n;
  } else if (n <= 2 * rb) {
    r1 = (n + 1) / 2; r2 = 0; r3 = n - r1;
  } else {
    r1 = (n + 2) / 3; r2 = (n - r1 + 1) / 2; r3 = n - r1 - r2;
  }
  pl.r1 = r1; pl.r2 = r2; pl.r3 = r3;
  const uint32_t R1 = 1u << r1, R2 = 1u << r2, R3 = 1u << r3;
  auto tile_bits = [](unsigned rbits, unsigned limit_bits) {
    unsigned t = 10 - rbits;  // TILE_ELEMS / R
    return t < limit_bits ? t : limit_bits;
  };
  if (r1) {  // pass A: over i1, tile of adjacent (i2,i3)
    PassParams& p = pl.a;
    p.logR = r1;
    p.logT = tile_bits(r1, r2 + r3);
    p.in_jstride = p.out_jstride = R2 * R3;
    p.in_tstride = p.out_tstride = 1;
    p.bdiv = 0x80000000u;
    p.in_lo = p.out_lo = p.tw_lo_mul = 1u << p.logT;
    p.tw_scale = 1;
    pl.grid_a = (R2 * R3) >> p.logT;
  }
  if (r2) {  // pass B: over i2, tile of adjacent i3, one k1 per block row
    PassParams& p = pl.b;
    p.logR = r2;
    p.logT = tile_bits(r2, r3);
    p.in_jstride = p.out_jstride = R3;
    p.in_tstride = p.out_tstride = 1;
    p.bdiv = R3 >> p.logT;
    p.in_hi = p.out_hi = R2 * R3;
    p.in_lo = p.out_lo = p.tw_lo_mul = 1u << p.logT;
    p.tw_scale = R1;
    pl.grid_b = R1 * (R3 >> p.logT);
  }
  {  // pass C: over i3 (contiguous), tile of adjacent k1, scatter to natural order
    PassParams& p = pl.c;
    p.logR = r3;
    p.logT = tile_bits(r3, r1);
    p.in_jstride = 1;
    p.in_tstride = R2 * R3;
    p.out_jstride = R1 * R2;
    p.out_tstride = 1;
    p.bdiv = R2;
    p.in_hi = (R2 * R3) << p.logT;
    p.in_lo = R3;
    p.out_hi = 1u << p.logT;
    p.out_lo = R1;
    p.tw_scale = 0;
    p.load_j_fast = 1;
    p.nq = R1 >> p.logT;  // tiles of adjacent k1 are neighbours in the output
    pl.grid_c = (R1 >> p.logT) * R2;
  }
  return pl;
}

static zg16_status get_domain(zg16_ctx* ctx, unsigned log_n, NttDomain** out) {
  auto it = ctx->domains.find(log_n);
  if (it != ctx->domains.end()) {
    *out = it->second;
    return ZG16_OK;
  }
  if (log_n >= 28) return fail(ctx, ZG16_ERR_POLYNOMIAL_DEGREE_TOO_LARGE, "domain 2^%u exceeds 2^27", log_n);
  if (ctx->capturing) return fail(ctx, ZG16_ERR_DEVICE, "NTT tables for 2^%u would have to be built during a stream capture", log_n);
  NttDomain* d = new NttDomain();
  d->log_n = log_n;
  d->plan = make_plan(log_n);
  if (d->plan.r1 > MAX_RADIX_BITS) {
    delete d;
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "2^%u does not split into three radix-2^%u passes", log_n, max_radix_bits());
  }
  const size_t N = (size_t)1 << log_n;
  const size_t n_tw = (d->plan.r1 ? 2 : 0) + (d->plan.r2 ? 2 : 0);
  const size_t n_dom = log_n >= 1 && log_n <= SMALL_MAX_LOG ? N / 2 : 0;
  const size_t bytes29 = ((3 + n_tw) * N + 2 * 512 + 8 + 2 * n_dom) * sizeof(Fr29), bytes = bytes29 + N * sizeof(Fr);
  // staging that is released again: the exponent-indexed power tables the store-order tables are built from
  const size_t stage_bytes = N * sizeof(Fr29) + (N + 512) * sizeof(Fr);
  void* stage = nullptr;
  hipError_t e = hipMalloc(&d->block, bytes);
  if (e == hipSuccess) e = hipMalloc(&stage, stage_bytes);
  if (e != hipSuccess) {
    (void)hipFree(d->block);
    delete d;
    return fail(ctx, ZG16_ERR_OUT_OF_MEMORY, "hipMalloc of NTT tables failed: %s", hipGetErrorString(e));
  }
  Fr29* p = (Fr29*)d->block;
  if (d->plan.r1) { d->tw_a[0] = p; p += N; d->tw_a[1] = p; p += N; }
  if (d->plan.r2) { d->tw_b[0] = p; p += N; d->tw_b[1] = p; p += N; }
  d->gm = p; p += N;
  d->ginvm = p; p += N;
  d->hz = p; p += N;
  d->kz = p; p += 8;
  d->w_small[0] = p; p += 512;
  d->w_small[1] = p; p += 512;
  if (n_dom) { d->w_dom[0] = p; p += n_dom; d->w_dom[1] = p; p += n_dom; }
  d->g = (Fr*)((char*)d->block + bytes29);
  Fr29* w_full = (Fr29*)stage;
  Fr* tmp = (Fr*)((char*)stage + N * sizeof(Fr29));

  Fr omega = host_root_of_unity();
  for (unsigned i = log_n; i < 28; i++) omega = sqr(omega);
  const Fr omega_inv = inv(omega);
  Fr w1024 = host_root_of_unity();
  for (unsigned i = MAX_RADIX_BITS; i < 28; i++) w1024 = sqr(w1024);
  const Fr w1024_inv = inv(w1024);
  const Fr g = from_u64<FrParams>(7), g_inv = inv(g);
  const Fr one = Fr::one();
  d->minv = inv(from_u64<FrParams>((uint64_t)1 << log_n));
  d->zinv = inv(sub(pow_u64(g, (uint64_t)1 << log_n), one));
  hipStream_t st = ctx->stream;
  auto launch = [&](Fr* dst, size_t count, const Fr& base, const Fr& c) {
    k_pow_table<<<(unsigned)((count + 4095) / 4096), 256, 0, st>>>(dst, (uint32_t)count, base, c);
  };
  auto table29 = [&](Fr29* dst, size_t count, const Fr& base, const Fr& c) {
    launch(tmp, count, base, c);
    k_to_fr29<<<(unsigned)((count + 255) / 256), 256, 0, st>>>(tmp, dst, (uint32_t)count);
  };
  for (int dir = 0; dir < 2; dir++) {
    if (!d->plan.r1) break;
    table29(w_full, N, dir ? omega_inv : omega, one);
    k_pass_twiddles<<<d->plan.grid_a, 256, 0, st>>>(d->plan.a, w_full, d->tw_a[dir]);
    if (d->plan.r2) k_pass_twiddles<<<d->plan.grid_b, 256, 0, st>>>(d->plan.b, w_full, d->tw_b[dir]);
  }
  table29(d->gm, N, g, d->minv);
  table29(d->ginvm, N, g_inv, d->minv);
  const Fr kz = mul(d->minv, d->zinv);
  table29(d->hz, N, g_inv, mul(from_u64<FrParams>(32), kz));
  table29(d->kz, 1, one, kz);
  table29(d->w_small[0], 512, w1024, one);
  table29(d->w_small[1], 512, w1024_inv, one);
  if (n_dom) {
    table29(d->w_dom[0], n_dom, omega, one);
    table29(d->w_dom[1], n_dom, omega_inv, one);
  }
  launch(d->g, N, g, one);
  hipError_t le = hipGetLastError();
  hipError_t se = hipStreamSynchronize(st);
  (void)hipFree(stage);
  if (le != hipSuccess || se != hipSuccess) {
    (void)hipFree(d->block);
    delete d;
    return fail(ctx, ZG16_ERR_DEVICE, "building NTT tables failed: %s", hipGetErrorString(le != hipSuccess ? le : se));
  }
  ctx->domains[log_n] = d;
  *out = d;
  return ZG16_OK;
}

void ntt_free_domains(zg16_ctx* ctx) {
  for (auto& kv : ctx->domains) {
    (void)hipFree(kv.second->block);
    delete kv.second;
  }
  ctx->domains.clear();
}

// Unnormalised transform: dst[i] = DFT_{w^+-1}(src)[i] * (post_scale ? post_scale[i] : 1).  dst may
// alias src.  The factor rides on the last pass's store, so `ifft` followed by a coset scaling (or
// `icoset_fft`'s trailing g^-i / m) costs no extra product and no extra pass.
// nvec (at most PASS_BATCH) transforms of the domain's size, every pass one launch over all of them: vector v goes
// from src[v] to dst[v] (in place allowed) and takes last[v] on its last store.  split_last: the last pass is
// launched vector by vector, in order (a vector's last store may then read an earlier vector's result: mulvec).
static zg16_status ntt_raw_batch(zg16_ctx* ctx, NttDomain* d, int nvec, const Fr* const* src, Fr* const* dst, bool inverse,
                                 const PassExtras* last, const ChunkLayout* src_chunks = nullptr,
                                 const ChunkLayout* dst_chunks = nullptr, bool split_last = false) {
  const unsigned n = d->log_n;
  if (nvec < 1 || nvec > PASS_BATCH || n == 0) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad transform batch");
  const NttPlan& pl = d->plan;
  const int dir = inverse ? 1 : 0;
  Fr* tmp = nullptr;
  if (pl.r1) ZG_TRY(scratch_get(ctx, 0, (size_t)nvec * (sizeof(Fr) << n), (void**)&tmp));
  hipStream_t st = ctx->stream;
  const double bytes = 64.0 * (double)((size_t)1 << n);
  // a chunked source is read by the first pass, a chunked destination written by the last
  PassParams first = pl.r1 ? pl.a : pl.c, last_pass = pl.c;
  if (src_chunks && src_chunks->stride) {
    first.in_rlog = src_chunks->log_blk;
    first.in_rstride = (uint32_t)src_chunks->stride;
    if (!pl.r1) last_pass = first;
  }
  if (dst_chunks && dst_chunks->stride) {
    last_pass.out_rlog = dst_chunks->log_blk;
    last_pass.out_rstride = (uint32_t)dst_chunks->stride;
  }
  PassBatch pa = {}, pbb = {}, pc = {};
  for (int v = 0; v < nvec; v++) {
    Fr* t = tmp ? tmp + ((size_t)v << n) : nullptr;
    pa.in[v] = src[v]; pa.out[v] = t; pa.ex[v] = PassExtras{d->tw_a[dir], nullptr, nullptr, 0};
    pbb.in[v] = t; pbb.out[v] = t; pbb.ex[v] = PassExtras{d->tw_b[dir], nullptr, nullptr, 0};
    pc.in[v] = pl.r1 ? t : src[v]; pc.out[v] = dst[v]; pc.ex[v] = last[v];
  }
  if (pl.r1) {
    KernelTimer kt(ctx, K_NTT_PASS, nvec * bytes);
    k_ntt_pass<<<dim3(pl.grid_a, nvec), 256, 0, st>>>(pa, first, d->w_small[dir]);
  }
  if (pl.r2) {
    KernelTimer kt(ctx, K_NTT_PASS, nvec * bytes);
    k_ntt_pass<<<dim3(pl.grid_b, nvec), 256, 0, st>>>(pbb, pl.b, d->w_small[dir]);
  }
  if (split_last && nvec > 1) {
    for (int v = 0; v < nvec; v++) {
      PassBatch one = {};
      one.in[0] = pc.in[v]; one.out[0] = pc.out[v]; one.ex[0] = pc.ex[v];
      KernelTimer kt(ctx, K_NTT_PASS, bytes);
      k_ntt_pass<<<dim3(pl.grid_c, 1), 256, 0, st>>>(one, last_pass, d->w_small[dir]);
    }
  } else {
    KernelTimer kt(ctx, K_NTT_PASS, nvec * bytes);
    k_ntt_pass<<<dim3(pl.grid_c, nvec), 256, 0, st>>>(pc, last_pass, d->w_small[dir]);
  }
  ZG_HIP(ctx, hipGetLastError());
  return ZG16_OK;
}

static zg16_status ntt_raw(zg16_ctx* ctx, NttDomain* d, const Fr* src, Fr* dst, bool inverse,
                           const Fr29* post_scale = nullptr, const PassExtras* last = nullptr,
                           const ChunkLayout* src_chunks = nullptr, const ChunkLayout* dst_chunks = nullptr) {
  if (d->log_n == 0) {
    if (dst != src) ZG_HIP(ctx, hipMemcpyAsync(dst, src, sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
    if (post_scale) k_scale_const<<<1, 256, 0, ctx->stream>>>(dst, 1, d->minv);  // g^0 / m
    return ZG16_OK;
  }
  const PassExtras ex = last ? *last : PassExtras{post_scale, nullptr, nullptr, 0};
  return ntt_raw_batch(ctx, d, 1, &src, &dst, inverse, &ex, src_chunks, dst_chunks);
}

static unsigned ew_grid(size_t n) {
  size_t g = (n + 255) / 256;
  return (unsigned)(g > 4096 ? 4096 : (g ? g : 1));
}

zg16_status ntt_device(zg16_ctx* ctx, Fr* data, unsigned log_n, bool inverse, bool coset) {
  NttDomain* d;
  ZG_TRY(get_domain(ctx, log_n, &d));
  const uint32_t n = 1u << log_n;
  hipStream_t st = ctx->stream;
  if (!inverse) {
    if (coset) k_scale_powers<<<ew_grid(n), 256, 0, st>>>(data, n, d->g);
    ZG_TRY(ntt_raw(ctx, d, data, data, false));
  } else if (coset) {
    ZG_TRY(ntt_raw(ctx, d, data, data, true, d->ginvm));
  } else {
    ZG_TRY(ntt_raw(ctx, d, data, data, true));
    k_scale_const<<<ew_grid(n), 256, 0, st>>>(data, n, d->minv);
  }
  ZG_HIP(ctx, hipGetLastError());
  return ZG16_OK;
}

// ifft then coset_fft of one vector of row evaluations, in place.
zg16_status coset_evals_device(zg16_ctx* ctx, Fr* p, unsigned log_m) {
  NttDomain* d;
  ZG_TRY(get_domain(ctx, log_m, &d));
  // the 1/m of the ifft and the coset factors g^i are one table (g^i / m) applied as the inverse
  // transform stores its result
  ZG_TRY(ntt_raw(ctx, d, p, p, true, d->gm));
  ZG_TRY(ntt_raw(ctx, d, p, p, false));
  return ZG16_OK;
}

// a <- icoset_fft((a*b - c) / Z) for coset evaluation vectors a, b, c.
zg16_status h_finish_device(zg16_ctx* ctx, Fr* a, const Fr* b, const Fr* c, unsigned log_m) {
  NttDomain* d;
  ZG_TRY(get_domain(ctx, log_m, &d));
  const uint32_t m = 1u << log_m;
  {
    KernelTimer kt(ctx, K_NTT_POINTWISE, 128.0 * m);
    k_mul_sub_scale<<<ew_grid(m), 256, 0, ctx->stream>>>(a, b, c, m, d->zinv);
  }
  ZG_TRY(ntt_raw(ctx, d, a, a, true, d->ginvm));
  ZG_HIP(ctx, hipGetLastError());
  return ZG16_OK;
}

// The whole H computation in six transforms instead of upstream's seven.  icoset_fft is linear and
// undoes coset_fft, so  icoset_fft((A*B - C) / Z) = icoset_fft(A*B / Z) - ifft(c) / Z  with A, B, C the
// coset evaluations: c needs one inverse transform and no coset evaluation at all.  The element-wise
// product rides on the last store of b's coset transform and the subtraction on the last store of
// the final inverse transform, so there is no separate element-wise kernel either.  Field
// arithmetic is exact: the coefficients equal upstream's.
zg16_status h_pipeline_device(zg16_ctx* ctx, Fr* a, Fr* b, Fr* c, unsigned log_m) {
  if (log_m == 0) {
    for (Fr* p : {a, b, c}) ZG_TRY(coset_evals_device(ctx, p, log_m));
    return h_finish_device(ctx, a, b, c, log_m);
  }
  NttDomain* d;
  ZG_TRY(get_domain(ctx, log_m, &d));
  // ZG16_SMALL_H=0 sends small domains through the general passes too (read per call: tests compare the two)
  const char* small_knob = getenv("ZG16_SMALL_H");
  if (d->w_dom[0] && !(small_knob && atoi(small_knob) == 0)) {
    const double bytes = 64.0 * (double)((size_t)1 << log_m);
    // every transform on two workgroups (ZG16_SMALL_SPLIT=0: on one); the halves pass through a block of their own
    const char* split_knob = getenv("ZG16_SMALL_SPLIT");  // (read per call, like ZG16_SMALL_H: tests compare the two)
    if (!(split_knob && atoi(split_knob) == 0) && log_m >= 2) {
      const size_t m = (size_t)1 << log_m;
      Fr* t;
      ZG_TRY(scratch_get(ctx, 0, 3 * m * sizeof(Fr), (void**)&t));
      const SmallChains v = {{a, b, c}, {t, t + m, t + 2 * m}};
      {
        KernelTimer kt(ctx, K_NTT_PASS, 5 * bytes);
        k_small_chains_split<<<6, SMALL_THREADS, 0, ctx->stream>>>(v, log_m, d->w_dom[1], d->w_dom[0], d->gm, d->kz);
      }
      {
        KernelTimer kt(ctx, K_NTT_PASS, bytes);
        k_small_finish_split<<<2, SMALL_THREADS, 0, ctx->stream>>>(t, t + m, t + 2 * m, a, log_m, d->w_dom[1], d->w_dom[0], d->hz);
      }
      ZG_HIP(ctx, hipGetLastError());
      return ZG16_OK;
    }
    const SmallChains v = {{a, b, c}, {a, b, c}};
    {
      KernelTimer kt(ctx, K_NTT_PASS, 5 * bytes);
      k_small_chains<<<3, SMALL_THREADS, 0, ctx->stream>>>(v, log_m, d->w_dom[1], d->w_dom[0], d->gm, d->kz);
    }
    {
      KernelTimer kt(ctx, K_NTT_PASS, bytes);
      k_small_finish<<<1, SMALL_THREADS, 0, ctx->stream>>>(a, b, c, a, log_m, d->w_dom[1], d->hz);
    }
    ZG_HIP(ctx, hipGetLastError());
    return ZG16_OK;
  }
  const PassExtras scale_c = {d->kz, nullptr, nullptr, 1};
  // Below 2^20 rows a pass has fewer workgroups than the GPU has room for (and a proof is a chain of launches): the
  // three inverse transforms go through their passes together, so do the first passes of a's and b's forward
  // transforms (b's last store reads a's result).  Seven launches instead of twelve.  ZG16_NTT_BATCH=0: one by one.
  static const bool batch_wanted = !(getenv("ZG16_NTT_BATCH") && atoi(getenv("ZG16_NTT_BATCH")) == 0);
  if (batch_wanted && d->plan.r1 && d->plan.grid_c < 1024) {
    const Fr* src3[3] = {c, a, b};
    Fr* dst3[3] = {c, a, b};
    const PassExtras inv3[3] = {scale_c, {d->gm, nullptr, nullptr, 0}, {d->gm, nullptr, nullptr, 0}};
    ZG_TRY(ntt_raw_batch(ctx, d, 3, src3, dst3, true, inv3));
    const PassExtras fwd2[2] = {{nullptr, nullptr, nullptr, 0}, {nullptr, a, nullptr, 0}};
    ZG_TRY(ntt_raw_batch(ctx, d, 2, src3 + 1, dst3 + 1, false, fwd2, nullptr, nullptr, true));
    const PassExtras finish = {d->hz, nullptr, c, 0};
    return ntt_raw(ctx, d, b, a, true, nullptr, &finish);
  }
  // c <- ifft(c) / Z
  ZG_TRY(ntt_raw(ctx, d, c, c, true, nullptr, &scale_c));
  ZG_TRY(coset_evals_device(ctx, a, log_m));
  ZG_TRY(ntt_raw(ctx, d, b, b, true, d->gm));
  // b <- A * B / 32 (a radix-2^256 value times another one, reduced by 2^261)
  const PassExtras times_a = {nullptr, a, nullptr, 0};
  ZG_TRY(ntt_raw(ctx, d, b, b, false, nullptr, &times_a));
  // a <- icoset_fft(A * B) / Z - c
  const PassExtras finish = {d->hz, nullptr, c, 0};
  return ntt_raw(ctx, d, b, a, true, nullptr, &finish);
}

// ---- pieces of an H computation whose transforms are spread over G GPUs (shard.hip drives them) -------
// GPU s of G holds element s + G j of every coefficient vector at slot j (j < L = m / G).  Factors that the
// single-GPU pipeline takes from its natural-order tables are needed at those strided indices here, so each
// (domain, shard count, shard) gets its own tables of L entries:
//   gm[j] = g^(s + G j) / L            (1/L of the local inverse transform and the coset factor in one)
//   hz[j] = g^-(s + G j) / (L Z)       (trailing factor of the final inverse transform, Z = g^m - 1)
//   kz    = 1 / (L Z)                  (a single constant)
// The 1/G of every inverse transform is part of ntt_cross_device's twiddles.
struct HShardTables {
  Fr29 *gm = nullptr, *hz = nullptr, *kz = nullptr;
  void* block = nullptr;
};

zg16_status hshard_tables(zg16_ctx* ctx, unsigned log_m, unsigned log_g, unsigned shard, const HShardTables** out) {
  const uint64_t key = ((uint64_t)log_m << 32) | (log_g << 16) | shard;
  auto it = ctx->hshard_tables.find(key);
  if (it != ctx->hshard_tables.end()) {
    *out = (const HShardTables*)it->second;
    return ZG16_OK;
  }
  const unsigned log_l = log_m - log_g;
  const size_t L = (size_t)1 << log_l;
  HShardTables* t = new HShardTables();
  void* stage = nullptr;
  hipError_t e = hipMalloc(&t->block, (2 * L + 8) * sizeof(Fr29));
  if (e == hipSuccess) e = hipMalloc(&stage, L * sizeof(Fr));
  if (e != hipSuccess) {
    (void)hipFree(t->block);
    delete t;
    return fail(ctx, ZG16_ERR_OUT_OF_MEMORY, "hipMalloc of the sharded-transform tables failed: %s", hipGetErrorString(e));
  }
  t->gm = (Fr29*)t->block;
  t->hz = t->gm + L;
  t->kz = t->hz + L;
  Fr* tmp = (Fr*)stage;
  const Fr g = from_u64<FrParams>(7), g_inv = inv(g), one = Fr::one();
  const Fr linv = inv(from_u64<FrParams>((uint64_t)1 << log_l));
  const Fr zinv = inv(sub(pow_u64(g, (uint64_t)1 << log_m), one));
  const uint64_t G = (uint64_t)1 << log_g;
  hipStream_t st = ctx->stream;
  auto table29 = [&](Fr29* dst, size_t count, const Fr& base, const Fr& c) {
    k_pow_table<<<(unsigned)((count + 4095) / 4096), 256, 0, st>>>(tmp, (uint32_t)count, base, c);
    k_to_fr29<<<(unsigned)((count + 255) / 256), 256, 0, st>>>(tmp, dst, (uint32_t)count);
  };
  table29(t->gm, L, pow_u64(g, G), mul(pow_u64(g, shard), linv));
  table29(t->hz, L, pow_u64(g_inv, G), mul(pow_u64(g_inv, shard), mul(linv, zinv)));
  table29(t->kz, 1, one, mul(linv, zinv));
  const hipError_t le = hipGetLastError(), se = hipStreamSynchronize(st);
  (void)hipFree(stage);
  if (le != hipSuccess || se != hipSuccess) {
    (void)hipFree(t->block);
    delete t;
    return fail(ctx, ZG16_ERR_DEVICE, "building the sharded-transform tables failed: %s", hipGetErrorString(le != hipSuccess ? le : se));
  }
  ctx->hshard_tables[key] = t;
  *out = t;
  return ZG16_OK;
}

void ntt_free_hshard_tables(zg16_ctx* ctx) {
  for (auto& kv : ctx->hshard_tables) {
    HShardTables* t = (HShardTables*)kv.second;
    (void)hipFree(t->block);
    delete t;
  }
  ctx->hshard_tables.clear();
}

// One local L-point transform of the sharded pipeline, unnormalised, with what rides on its last store:
// which = 0: nothing; 1: times gm[j]; 2: times the constant kz; 3: times hz[j], minus subvec[j].
zg16_status hshard_local_ntt(zg16_ctx* ctx, unsigned log_l, const HShardTables* t, const Fr* src, Fr* dst, bool inverse,
                             int which, const Fr* subvec, const ChunkLayout* src_chunks, const ChunkLayout* dst_chunks) {
  NttDomain* d;
  ZG_TRY(get_domain(ctx, log_l, &d));
  if (log_l == 0) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "a sharded transform needs at least two elements per GPU");
  PassExtras ex = {nullptr, nullptr, nullptr, 0};
  if (which == 1) ex.factor = t->gm;
  if (which == 2) { ex.factor = t->kz; ex.factor_const = 1; }
  if (which == 3) { ex.factor = t->hz; ex.subvec = subvec; }
  return ntt_raw(ctx, d, src, dst, inverse, nullptr, &ex, src_chunks, dst_chunks);
}

// nvec (at most 3) of those transforms with every pass one launch over all of them; the vectors share the direction
// and the layouts, which[v] and subvec[v] are per vector.
zg16_status hshard_local_ntt_batch(zg16_ctx* ctx, unsigned log_l, const HShardTables* t, int nvec, const Fr* const* src,
                                   Fr* const* dst, bool inverse, const int* which, const Fr* const* subvec,
                                   const ChunkLayout* src_chunks, const ChunkLayout* dst_chunks) {
  NttDomain* d;
  ZG_TRY(get_domain(ctx, log_l, &d));
  if (log_l == 0 || nvec < 1 || nvec > PASS_BATCH) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad sharded transform batch");
  PassExtras ex[PASS_BATCH];
  for (int v = 0; v < nvec; v++) {
    ex[v] = PassExtras{nullptr, nullptr, nullptr, 0};
    if (which[v] == 1) ex[v].factor = t->gm;
    if (which[v] == 2) { ex[v].factor = t->kz; ex[v].factor_const = 1; }
    if (which[v] == 3) { ex[v].factor = t->hz; ex[v].subvec = subvec ? subvec[v] : nullptr; }
  }
  return ntt_raw_batch(ctx, d, nvec, src, dst, inverse, ex, src_chunks, dst_chunks);
}

__global__ __launch_bounds__(256) void k_mul_vec(Fr* a, const Fr* __restrict__ b, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) a[i] = mul(a[i], b[i]);
}

// a[i] *= b[i]
zg16_status mul_vec_device(zg16_ctx* ctx, Fr* a, const Fr* b, size_t n) {
  k_mul_vec<<<ew_grid(n), 256, 0, ctx->stream>>>(a, b, (uint32_t)n);
  ZG_HIP(ctx, hipGetLastError());
  return ZG16_OK;
}

// ---- one transform over several GPUs ----------------------------------------------------------
// N = G * L points over G GPUs (G a power of two, at most 8), one exchange.  GPU g starts with the
// decimated sequence x[g + G j], j < L.  With k = k2 + L k1:
//     X[k2 + L k1] = sum_g w_G^(g k1) * ( w_N^(g k2) * Y_g[k2] ),   Y_g = the L-point transform of GPU g's part,
// so every GPU runs its own L-point transform (ntt_device), the all-to-all hands GPU h the slice
// k2 in [h L/G, (h+1) L/G) of every Y_g as G blocks, and this kernel applies the twiddle and the G-point
// transform across the blocks: block k1 of GPU h then holds X[(h L/G + i) + L k1].  The inverse runs
// the same steps backwards (inverse G-point transform, conjugate twiddle with the 1/G folded in,
// all-to-all, inverse L-point transform with its 1/L).
struct CrossRoots {
  Fr w[4];  // w_G^j (or its inverse), j < G/2
};

// The G-point transform across the blocks, in registers: y in bit-reversed order in, natural order out.
template <int LG>
ZG_D void cross_stages(Fr (&y)[1 << LG], const CrossRoots& roots) {
  constexpr int G = 1 << LG;
#pragma unroll
  for (int s = 0; s < LG; s++) {
    const int h = 1 << s;
#pragma unroll
    for (int u = 0; u < G / 2; u++) {
      const int pos = u & (h - 1), i0 = ((u >> s) << (s + 1)) + pos;
      const Fr t = pos ? mul(y[i0 + h], roots.w[pos << (LG - 1 - s)]) : y[i0 + h];
      y[i0 + h] = sub(y[i0], t);
      y[i0] = add(y[i0], t);
    }
  }
}

// in / out: block g of the vector at g * in_stride / g * out_stride elements (blk when contiguous; in may be out).
// mulvec (forward only): the result is multiplied element-wise by the natural-order vector mulvec (block g at g blk).
template <int LG>
__global__ __launch_bounds__(256) void k_ntt_cross(const Fr* in, size_t in_stride, Fr* out, size_t out_stride, uint32_t blk,
                                                   const Fr* __restrict__ tw, CrossRoots roots, Fr scale0, int inverse,
                                                   const Fr* __restrict__ mulvec, size_t in_vstride, size_t out_vstride) {
  constexpr int G = 1 << LG;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= blk) return;
  in += blockIdx.y * in_vstride;    // vector blockIdx.y of a launch over several vectors
  out += blockIdx.y * out_vstride;
  Fr y[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    Fr x = in[(size_t)g * in_stride + i];
    if (!inverse && g) x = mul(x, tw[(size_t)(g - 1) * blk + i]);
    y[__brev((unsigned)g) >> (32 - LG)] = x;
  }
  cross_stages<LG>(y, roots);
#pragma unroll
  for (int g = 0; g < G; g++) {
    Fr x = y[g];
    if (inverse) x = mul(x, g ? tw[(size_t)(g - 1) * blk + i] : scale0);
    if (mulvec) x = mul(x, mulvec[(size_t)g * blk + i]);
    out[(size_t)g * out_stride + i] = x;
  }
}

// The twiddle table of one (size, shard count, shard, direction): [G - 1][L / G] entries
// w_N^(+-g k2) for k2 in this shard's slice, times 1/G when inverse.  Built on first use and kept.
// The middle of a sharded H computation in one launch: forward cross step of a and of b, their element-wise product,
// inverse cross step of the product.  All three touch, for a given i, only the elements i of the G blocks, so a
// thread carries its 2 G elements through in registers (three launches and two trips through memory before).
template <int LG>
__global__ __launch_bounds__(256) void k_ntt_cross_product(const Fr* __restrict__ a, const Fr* __restrict__ b, size_t in_stride,
                                                           Fr* __restrict__ out, size_t out_stride, uint32_t blk,
                                                           const Fr* __restrict__ tw_f, CrossRoots roots_f,
                                                           const Fr* __restrict__ tw_i, CrossRoots roots_i, Fr scale0_i) {
  constexpr int G = 1 << LG;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= blk) return;
  Fr ya[G], yb[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    Fr xa = a[(size_t)g * in_stride + i], xb = b[(size_t)g * in_stride + i];
    if (g) {
      const Fr t = tw_f[(size_t)(g - 1) * blk + i];
      xa = mul(xa, t);
      xb = mul(xb, t);
    }
    ya[__brev((unsigned)g) >> (32 - LG)] = xa;
    yb[__brev((unsigned)g) >> (32 - LG)] = xb;
  }
  cross_stages<LG>(ya, roots_f);
  cross_stages<LG>(yb, roots_f);
  Fr z[G];
#pragma unroll
  for (int k = 0; k < G; k++) z[__brev((unsigned)k) >> (32 - LG)] = mul(ya[k], yb[k]);
  cross_stages<LG>(z, roots_i);
#pragma unroll
  for (int g = 0; g < G; g++) out[(size_t)g * out_stride + i] = mul(z[g], g ? tw_i[(size_t)(g - 1) * blk + i] : scale0_i);
}

struct CrossTable {
  Fr* tw = nullptr;
  CrossRoots roots;
  Fr scale0;
};

void ntt_free_cross_tables(zg16_ctx* ctx) {
  for (auto& kv : ctx->cross_tables) {
    CrossTable* t = (CrossTable*)kv.second;
    (void)hipFree(t->tw);
    delete t;
  }
  ctx->cross_tables.clear();
}

// The twiddles and roots of one (size, shard count, shard, direction), built on first use and kept.
static zg16_status cross_table(zg16_ctx* ctx, unsigned log_n, unsigned log_g, unsigned shard, bool inverse, CrossTable** out) {
  const uint32_t G = 1u << log_g, blk = 1u << (log_n - 2 * log_g);
  const uint64_t key = ((uint64_t)log_n << 32) | (log_g << 16) | (shard << 1) | (inverse ? 1 : 0);
  hipStream_t st = ctx->stream;
  CrossTable* t;
  auto it = ctx->cross_tables.find(key);
  if (it != ctx->cross_tables.end()) {
    t = (CrossTable*)it->second;
  } else {
    // (the roots are host field arithmetic -- an inversion among it -- and are kept with the table: a proof calls
    // this six times)
    Fr omega = host_root_of_unity();
    for (unsigned i = log_n; i < 28; i++) omega = sqr(omega);
    if (inverse) omega = inv(omega);
    const Fr g_inv = inv(from_u64<FrParams>(G));
    std::unique_ptr<CrossTable> fresh(new CrossTable());
    ZG_HIP(ctx, hipMalloc((void**)&fresh->tw, (size_t)(G - 1) * blk * sizeof(Fr)));
    for (uint32_t g = 1; g < G; g++) {
      const Fr base = pow_u64(omega, g);
      Fr c = pow_u64(base, (uint64_t)shard * blk);
      if (inverse) c = mul(c, g_inv);
      k_pow_table<<<(blk + 4095) / 4096, 256, 0, st>>>(fresh->tw + (size_t)(g - 1) * blk, blk, base, c);
    }
    const Fr wg = pow_u64(omega, (uint64_t)1 << (log_n - log_g));  // w_G (or its inverse)
    Fr p = Fr::one();
    for (uint32_t j = 0; j < 4; j++) {
      fresh->roots.w[j] = p;
      p = mul(p, wg);
    }
    fresh->scale0 = inverse ? g_inv : Fr::one();
    t = fresh.release();
    ctx->cross_tables[key] = t;
  }
  *out = t;
  return ZG16_OK;
}

zg16_status ntt_cross_device(zg16_ctx* ctx, Fr* data, unsigned log_n, unsigned log_g, unsigned shard, bool inverse) {
  return ntt_cross_chunked(ctx, data, 0, data, 0, log_n, log_g, shard, inverse, nullptr);
}

// The same with the blocks of the input and of the output `in_stride` / `out_stride` elements apart (0: contiguous)
// and, for a forward step, an element-wise product with `mulvec` riding on the store.  nvec > 1: that many vectors in
// one launch, vector v at in + v in_vstride / out + v out_vstride (no mulvec then).
zg16_status ntt_cross_chunked(zg16_ctx* ctx, const Fr* in, size_t in_stride, Fr* out, size_t out_stride, unsigned log_n,
                              unsigned log_g, unsigned shard, bool inverse, const Fr* mulvec, int nvec, size_t in_vstride,
                              size_t out_vstride) {
  const uint32_t G = 1u << log_g, blk = 1u << (log_n - 2 * log_g);
  if (nvec < 1 || (nvec > 1 && mulvec)) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad cross-step batch");
  if (log_g == 0) {
    for (int v = 0; v < nvec; v++)
      if (in != out) ZG_HIP(ctx, hipMemcpyAsync(out + v * out_vstride, in + v * in_vstride, (sizeof(Fr) << log_n), hipMemcpyDeviceToDevice, ctx->stream));
    return ZG16_OK;
  }
  if (!in_stride) in_stride = blk;
  if (!out_stride) out_stride = blk;
  hipStream_t st = ctx->stream;
  CrossTable* t;
  ZG_TRY(cross_table(ctx, log_n, log_g, shard, inverse, &t));
  const dim3 grid((blk + 255) / 256, (unsigned)nvec);
  switch (log_g) {
    case 1: k_ntt_cross<1><<<grid, 256, 0, st>>>(in, in_stride, out, out_stride, blk, t->tw, t->roots, t->scale0, inverse, mulvec, in_vstride, out_vstride); break;
    case 2: k_ntt_cross<2><<<grid, 256, 0, st>>>(in, in_stride, out, out_stride, blk, t->tw, t->roots, t->scale0, inverse, mulvec, in_vstride, out_vstride); break;
    default: k_ntt_cross<3><<<grid, 256, 0, st>>>(in, in_stride, out, out_stride, blk, t->tw, t->roots, t->scale0, inverse, mulvec, in_vstride, out_vstride); break;
  }
  ZG_HIP(ctx, hipGetLastError());
  return ZG16_OK;
}

 // (see k_ntt_cross_product) a, b: blocks in_stride apart; out: blocks out_stride apart (0: contiguous)
zg16_status ntt_cross_product(zg16_ctx* ctx, const Fr* a, const Fr* b, size_t in_stride, Fr* out, size_t out_stride,
                              unsigned log_n, unsigned log_g, unsigned shard) {
  const uint32_t blk = 1u << (log_n - 2 * log_g);
  if (log_g < 1 || log_g > 3) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad shard count for the fused cross step");
  if (!in_stride) in_stride = blk;
  if (!out_stride) out_stride = blk;
  CrossTable *f, *v;
  ZG_TRY(cross_table(ctx, log_n, log_g, shard, false, &f));
  ZG_TRY(cross_table(ctx, log_n, log_g, shard, true, &v));
  const unsigned grid = (blk + 255) / 256;
  hipStream_t st = ctx->stream;
  switch (log_g) {
    case 1: k_ntt_cross_product<1><<<grid, 256, 0, st>>>(a, b, in_stride, out, out_stride, blk, f->tw, f->roots, v->tw, v->roots, v->scale0); break;
    case 2: k_ntt_cross_product<2><<<grid, 256, 0, st>>>(a, b, in_stride, out, out_stride, blk, f->tw, f->roots, v->tw, v->roots, v->scale0); break;
    default: k_ntt_cross_product<3><<<grid, 256, 0, st>>>(a, b, in_stride, out, out_stride, blk, f->tw, f->roots, v->tw, v->roots, v->scale0); break;
  }
  ZG_HIP(ctx, hipGetLastError());
  return ZG16_OK;
}

}  // namespace zg
