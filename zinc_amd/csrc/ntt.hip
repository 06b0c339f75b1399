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
#include "common.cuh"

namespace zg {

static constexpr unsigned MAX_RADIX_BITS = 10;
static constexpr unsigned TILE_ELEMS = 1024;

// Every table is a full one (N entries): with 288 GB of HBM a few multiples of the domain size cost
// nothing, and a twiddle or a coset factor is then one 32-byte load instead of a field product.
struct NttDomain {
  unsigned log_n = 0;
  Fr* w[2] = {};           // [dir][N]  w^e  (dir 1 = inverse root)
  Fr* g = nullptr;         // g^i
  Fr* gm = nullptr;        // g^i / m
  Fr* ginvm = nullptr;     // g^-i / m
  Fr* w_small[2] = {};     // [dir][512]  w_1024^e
  Fr minv, zinv;           // 1/m, 1/(g^m - 1)
  Fr* block = nullptr;     // one allocation behind all of the above
};

struct PassParams {
  uint32_t logR, logT;
  uint32_t in_jstride, in_tstride, out_jstride, out_tstride;
  uint32_t bdiv;
  uint32_t in_hi, in_lo, out_hi, out_lo, tw_lo_mul;
  uint32_t tw_scale;  // 0: no inter-pass twiddle
  uint32_t load_j_fast;
  uint32_t post_scale;  // multiply output element i by scale[i] (last pass only)
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

__global__ __launch_bounds__(256) void k_ntt_pass(const Fr* __restrict__ in, Fr* __restrict__ out,
                                                  PassParams p, const Fr* __restrict__ w_small,
                                                  const Fr* __restrict__ tw, const Fr* __restrict__ scale) {
  __shared__ Fr s[TILE_ELEMS];
  const uint32_t tid = threadIdx.x;
  const uint32_t logR = p.logR, logT = p.logT;
  const uint32_t R = 1u << logR, T = 1u << logT, E = R << logT;
  const uint32_t q = blockIdx.x / p.bdiv, rem = blockIdx.x % p.bdiv;
  const uint32_t in_base = q * p.in_hi + rem * p.in_lo;
  const uint32_t out_base = q * p.out_hi + rem * p.out_lo;
  const uint32_t m0 = rem * p.tw_lo_mul;

  for (uint32_t idx = tid; idx < E; idx += 256) {
    uint32_t j, t;
    if (p.load_j_fast) {
      j = idx & (R - 1);
      t = idx >> logR;
    } else {
      t = idx & (T - 1);
      j = idx >> logT;
    }
    s[(t << logR) + j] = in[in_base + j * p.in_jstride + t * p.in_tstride];
  }
  __syncthreads();

  // decimation in frequency: natural order in, bit-reversed order out
  for (int st = (int)logR - 1; st >= 0; --st) {
    const uint32_t half = 1u << st;
    for (uint32_t b = tid; b < (E >> 1); b += 256) {
      const uint32_t t = b >> (logR - 1), jj = b & ((R >> 1) - 1);
      const uint32_t grp = jj >> st, pos = jj & (half - 1);
      const uint32_t i0 = (t << logR) + (grp << (st + 1)) + pos, i1 = i0 + half;
      const Fr x = s[i0], y = s[i1];
      s[i0] = add(x, y);
      Fr d = sub(x, y);
      if (pos) d = mul(d, w_small[(pos << (logR - 1 - st)) << (MAX_RADIX_BITS - logR)]);
      s[i1] = d;
    }
    __syncthreads();
  }

  for (uint32_t idx = tid; idx < E; idx += 256) {
    const uint32_t t = idx & (T - 1), kk = idx >> logT;
    Fr v = s[(t << logR) + (__brev(kk) >> (32 - logR))];
    if (p.tw_scale) {
      const uint32_t e = (m0 + t) * kk * p.tw_scale;
      if (e) v = mul(v, tw[e]);
    }
    const uint32_t o = out_base + kk * p.out_jstride + t * p.out_tstride;
    if (p.post_scale) v = mul(v, scale[o]);
    out[o] = v;
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

static zg16_status get_domain(zg16_ctx* ctx, unsigned log_n, NttDomain** out) {
  auto it = ctx->domains.find(log_n);
  if (it != ctx->domains.end()) {
    *out = it->second;
    return ZG16_OK;
  }
  if (log_n >= 28) return fail(ctx, ZG16_ERR_POLYNOMIAL_DEGREE_TOO_LARGE, "domain 2^%u exceeds 2^27", log_n);
  NttDomain* d = new NttDomain();
  d->log_n = log_n;
  const size_t N = (size_t)1 << log_n;
  const size_t total = 5 * N + 2 * 512;
  hipError_t e = hipMalloc(&d->block, total * sizeof(Fr));
  if (e != hipSuccess) {
    delete d;
    return fail(ctx, ZG16_ERR_OUT_OF_MEMORY, "hipMalloc of NTT tables failed: %s", hipGetErrorString(e));
  }
  Fr* p = d->block;
  d->w[0] = p; p += N;
  d->w[1] = p; p += N;
  d->g = p; p += N;
  d->gm = p; p += N;
  d->ginvm = p; p += N;
  d->w_small[0] = p; p += 512;
  d->w_small[1] = p; p += 512;

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
  launch(d->w[0], N, omega, one);
  launch(d->w[1], N, omega_inv, one);
  launch(d->g, N, g, one);
  launch(d->gm, N, g, d->minv);
  launch(d->ginvm, N, g_inv, d->minv);
  launch(d->w_small[0], 512, w1024, one);
  launch(d->w_small[1], 512, w1024_inv, one);
  ZG_HIP(ctx, hipGetLastError());
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

// Unnormalised transform: dst[i] = DFT_{w^+-1}(src)[i] * (post_scale ? post_scale[i] : 1).  dst may
// alias src.  The factor rides on the last pass's store, so `ifft` followed by a coset scaling (or
// `icoset_fft`'s trailing g^-i / m) costs one extra product per element and no extra pass.
static zg16_status ntt_raw(zg16_ctx* ctx, NttDomain* d, const Fr* src, Fr* dst, bool inverse,
                           const Fr* post_scale = nullptr) {
  const unsigned n = d->log_n;
  if (n == 0) {
    if (dst != src) ZG_HIP(ctx, hipMemcpyAsync(dst, src, sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
    if (post_scale) k_scale_powers<<<1, 256, 0, ctx->stream>>>(dst, 1, post_scale);
    return ZG16_OK;
  }
  const unsigned rb = max_radix_bits();
  unsigned r1, r2, r3;
  if (n <= rb) {
    r1 = 0; r2 = 0; r3 = n;
  } else if (n <= 2 * rb) {
    r1 = (n + 1) / 2; r2 = 0; r3 = n - r1;
  } else {
    r1 = (n + 2) / 3; r2 = (n - r1 + 1) / 2; r3 = n - r1 - r2;
  }
  if (r1 > MAX_RADIX_BITS) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "2^%u does not split into three radix-2^%u passes", n, rb);
  const uint32_t R1 = 1u << r1, R2 = 1u << r2, R3 = 1u << r3;
  const int dir = inverse ? 1 : 0;
  Fr* tmp = nullptr;
  if (r1) ZG_TRY(scratch_get(ctx, 0, sizeof(Fr) << n, (void**)&tmp));
  hipStream_t st = ctx->stream;
  auto tile_bits = [](unsigned rbits, unsigned limit_bits) {
    unsigned t = 10 - rbits;  // TILE_ELEMS / R
    return t < limit_bits ? t : limit_bits;
  };
  if (r1) {  // pass A: over i1, tile of adjacent (i2,i3)
    PassParams p = {};
    p.logR = r1;
    p.logT = tile_bits(r1, r2 + r3);
    p.in_jstride = p.out_jstride = R2 * R3;
    p.in_tstride = p.out_tstride = 1;
    p.bdiv = 0x80000000u;
    p.in_lo = p.out_lo = p.tw_lo_mul = 1u << p.logT;
    p.tw_scale = 1;
    const uint32_t grid = (R2 * R3) >> p.logT;
    KernelTimer kt(ctx, K_NTT_PASS, 64.0 * (double)((size_t)1 << n));
    k_ntt_pass<<<grid, 256, 0, st>>>(src, tmp, p, d->w_small[dir], d->w[dir], nullptr);
  }
  if (r2) {  // pass B: over i2, tile of adjacent i3, one k1 per block row
    PassParams p = {};
    p.logR = r2;
    p.logT = tile_bits(r2, r3);
    p.in_jstride = p.out_jstride = R3;
    p.in_tstride = p.out_tstride = 1;
    p.bdiv = R3 >> p.logT;
    p.in_hi = p.out_hi = R2 * R3;
    p.in_lo = p.out_lo = p.tw_lo_mul = 1u << p.logT;
    p.tw_scale = R1;
    const uint32_t grid = R1 * (R3 >> p.logT);
    KernelTimer kt(ctx, K_NTT_PASS, 64.0 * (double)((size_t)1 << n));
    k_ntt_pass<<<grid, 256, 0, st>>>(tmp, tmp, p, d->w_small[dir], d->w[dir], nullptr);
  }
  {  // pass C: over i3 (contiguous), tile of adjacent k1, scatter to natural order
    PassParams p = {};
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
    p.post_scale = post_scale ? 1 : 0;
    const uint32_t grid = (R1 >> p.logT) * R2;
    KernelTimer kt(ctx, K_NTT_PASS, 64.0 * (double)((size_t)1 << n));
    k_ntt_pass<<<grid, 256, 0, st>>>(r1 ? tmp : src, dst, p, d->w_small[dir], d->w[dir], post_scale);
  }
  ZG_HIP(ctx, hipGetLastError());
  return ZG16_OK;
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

zg16_status h_pipeline_device(zg16_ctx* ctx, Fr* a, Fr* b, Fr* c, unsigned log_m) {
  for (Fr* p : {a, b, c}) ZG_TRY(coset_evals_device(ctx, p, log_m));
  return h_finish_device(ctx, a, b, c, log_m);
}

}  // namespace zg
