// Fixed-base scalar multiplication: out[i] = scalars[i] * base, affine results.
//
// This is the group half of bellman_ce 0.3.0 `generate_parameters` (called by the reference at
// zinc-vm/src/core/circuit/facade.rs:126): every point of a proving key is a known scalar times
// one generator.  Used to build 2^20..2^24-row keys in seconds; not part of the prove path.
//
// Method: byte windows.  T[w][d-1] = d * 256^w * base for w < 32, d in 1..255 is built once per
// base (sixteen lanes per window; the context keeps the table of the last base), then every lane sums the 32 table entries its scalar selects with
// mixed additions and converts its own result to affine.
#include <string.h>

#include <vector>

#define ZG_OUTLINE_FIELD_MUL 1
#include "r1cs.hpp"

namespace zg {

static constexpr int FB_WINDOWS = 32, FB_DIGITS = 255;

// Lane (w, j), j < 16, writes the multiples 16 j + 1 .. 16 j + 16 of 256^w * base: 8 w doublings to the window's
// base, four more to its sixteenfold, j additions of that to the segment's start and then the segment's sixteen
// additions -- a chain of at most 252 doublings and 31 additions instead of 248 and 255 with one lane per window.
template <class F>
__global__ void k_fb_table_rows(Affine<F> base, XYZZ<F>* table) {
  const int w = threadIdx.x >> 4, j = threadIdx.x & 15;
  if (w >= FB_WINDOWS) return;
  XYZZ<F> b = XYZZ<F>::from_affine(base);
  for (int i = 0; i < 8 * w; i++) b = dbl(b);
  XYZZ<F> b16 = b;
  for (int i = 0; i < 4; i++) b16 = dbl(b16);
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int i = 0; i < j; i++) add_full(acc, b16);
  for (int d = 16 * j; d < 16 * j + 16 && d < FB_DIGITS; d++) {
    add_full(acc, b);
    table[w * FB_DIGITS + d] = acc;
  }
}

template <class F>
__global__ void k_fb_table_affine(const XYZZ<F>* table, Affine<F>* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = to_affine(table[i]);
}

template <class F>
__global__ __launch_bounds__(256) void k_fb_mul(const Affine<F>* __restrict__ table, const Fr* __restrict__ scalars,
                                                Affine<F>* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fr s = from_mont(scalars[i]);
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int w = 0; w < FB_WINDOWS; w++) {
    const uint32_t d = (s.v[w >> 2] >> (8 * (w & 3))) & 0xff;
    if (d) add_mixed(acc, table[w * FB_DIGITS + d - 1]);
  }
  out[i] = to_affine(acc);
}

// d_scalars: n Montgomery Fr in device memory (not inside scratch arena 5); out: n affine points in host memory.
template <class F>
static zg16_status fixed_base_device(zg16_ctx* ctx, const Affine<F>& b, const Fr* d_scalars, size_t n, void* out) {
  if (n >= 0xffffffffull) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "too many scalars");
  if (!n) return ZG16_OK;
  const int tn = FB_WINDOWS * FB_DIGITS;
  constexpr int g = sizeof(F) > sizeof(Fq) ? 1 : 0;
  char* buf;
  const size_t t_x = (size_t)tn * sizeof(XYZZ<F>), t_a = (size_t)tn * sizeof(Affine<F>), o_b = n * sizeof(Affine<F>);
  ZG_TRY(scratch_get(ctx, 5, t_x + o_b + 1024, (void**)&buf));
  XYZZ<F>* tx = (XYZZ<F>*)buf;
  Affine<F>* dout = (Affine<F>*)(buf + t_x);
  hipStream_t st = ctx->stream;
  // the table of this base, kept with the context from one call to the next
  if (!ctx->fb_table[g]) ZG_HIP(ctx, hipMalloc(&ctx->fb_table[g], t_a));
  Affine<F>* ta = (Affine<F>*)ctx->fb_table[g];
  static_assert(sizeof(Affine<F>) <= sizeof(ctx->fb_base[g]), "base point larger than its cache key");
  if (!ctx->fb_valid[g] || memcmp(ctx->fb_base[g], &b, sizeof b) != 0) {
    ctx->fb_valid[g] = false;
    k_fb_table_rows<F><<<1, 16 * FB_WINDOWS, 0, st>>>(b, tx);
    k_fb_table_affine<F><<<(tn + 255) / 256, 256, 0, st>>>(tx, ta, tn);
    ZG_HIP(ctx, hipGetLastError());
    memcpy(ctx->fb_base[g], &b, sizeof b);
    ctx->fb_valid[g] = true;
  }
  k_fb_mul<F><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(ta, d_scalars, dout, (uint32_t)n);
  ZG_HIP(ctx, hipGetLastError());
  ZG_HIP(ctx, hipMemcpyAsync(out, dout, o_b, hipMemcpyDeviceToHost, st));
  ZG_HIP(ctx, hipStreamSynchronize(st));
  return ZG16_OK;
}

template <class F>
static zg16_status fixed_base(zg16_ctx* ctx, const uint64_t* base, const uint64_t* scalars, size_t n, uint64_t* out) {
  if (!ctx || !base || (n && (!scalars || !out))) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to fixed-base multiply");
  if (!n) return ZG16_OK;
  Affine<F> b;
  memcpy(&b, base, sizeof b);
  if (n <= 16) {
    // a handful of scalars (the generators k1 * G, k2 * G of a key generation): a table of 8160 multiples, or even a
    // kernel launch, costs more than the host's double-and-add
    for (size_t i = 0; i < n; i++) {
      Fr s;
      memcpy(s.v, scalars + 4 * i, 32);
      const Fr sc = from_mont(s);
      const Affine<F> r = to_affine(scalar_mul_w4(XYZZ<F>::from_affine(b), sc.v));
      memcpy(out + i * (sizeof r / 8), &r, sizeof r);
    }
    return ZG16_OK;
  }
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  Fr* ds;
  ZG_TRY(scratch_get(ctx, 7, n * sizeof(Fr), (void**)&ds));
  ZG_HIP(ctx, hipMemcpyAsync(ds, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
  return fixed_base_device<F>(ctx, b, ds, n, out);
}

// ---- the field half of key generation ------------------------------------------------------------------
// bellman_ce 0.3.0 `generate_parameters` (the reference calls `generate_random_parameters` at
// zinc-vm/src/core/circuit/facade.rs:126 and core/contract/facade.rs:299): with L_j(tau) the Lagrange values of the
// m-point domain (the inverse transform of the powers of tau), every variable v gets
//   a_v = sum over A's entries (row, v) of coeff * L_row(tau),  b_v, c_v likewise,
// i.e. the TRANSPOSED matrices times the vector L -- the same sparse mat-vec as the prover's row evaluation, on
// the transposed structure.  Then ic_v / l_v = (beta a_v + alpha b_v + c_v) / gamma (inputs) or / delta (aux), the
// h scalars tau^i (tau^m - 1) / delta, and every key point is one of those scalars times g1 or g2.

// out[v] = (beta a[v] + alpha b[v] + c[v]) * (v < n_in ? gamma_inv : delta_inv)
__global__ __launch_bounds__(256) void k_setup_combine(const Fr* __restrict__ a, const Fr* __restrict__ b,
                                                       const Fr* __restrict__ c, Fr* __restrict__ out, uint32_t nv,
                                                       uint32_t n_in, Fr alpha, Fr beta, Fr gamma_inv, Fr delta_inv) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  const Fr t = add(add(mul(beta, a[v]), mul(alpha, b[v])), c[v]);
  out[v] = mul(t, v < n_in ? gamma_inv : delta_inv);
}

// A variable that occurs in very many rows (ONE does, in about every boolean row) would be one endless "row" of the
// transposed matrix for a single wavefront; such a row is cut into segments of SEGMENT terms, the segments after the
// first become extra rows behind the nv real ones, and k_fold_segments adds them back.
static constexpr uint32_t SEGMENT = 8192;
struct SegmentList {
  std::vector<uint32_t> owner;  // owner[i]: the variable extra row nv + i belongs to (ascending)
};

// out[owner] += out[nv + i] for every extra row, one thread per matrix (there are few extra rows)
__global__ void k_fold_segments(Fr* a, Fr* b, Fr* c, const uint32_t* __restrict__ owner_a, uint32_t na,
                                const uint32_t* __restrict__ owner_b, uint32_t nb_, const uint32_t* __restrict__ owner_c,
                                uint32_t nc, uint32_t nv) {
  Fr* outs[3] = {a, b, c};
  const uint32_t* owners[3] = {owner_a, owner_b, owner_c};
  const uint32_t counts[3] = {na, nb_, nc};
  const uint32_t k = threadIdx.x;
  if (k >= 3) return;
  for (uint32_t i = 0; i < counts[k]; i++) outs[k][owners[k][i]] = add(outs[k][owners[k][i]], outs[k][nv + i]);
}

// The transpose of one CSR matrix as a device-resident R1csMatrix whose "rows" are the variables (inputs then aux,
// then the extra segment rows) and whose "columns" are the constraint rows.
static zg16_status upload_transposed(zg16_ctx* ctx, const zg16_csr& m, size_t n_rows, size_t n_in, size_t n_aux, int which,
                                     R1csMatrix* out, SegmentList* segs) {
  const size_t nv = n_in + n_aux;
  if (!m.row_ptr) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "matrix %d has no row_ptr", which);
  const size_t nnz = (size_t)m.row_ptr[n_rows];
  if (nnz >= 0xffffffffull) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "matrix %d has too many entries", which);
  if (nnz && (!m.col || !m.coeff)) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "matrix %d has no col/coeff", which);
  std::vector<uint32_t> vp(nv + 1, 0), rows(nnz), longs;
  std::vector<uint64_t> co(4 * nnz);
  auto flat = [&](uint32_t c, size_t* v) {
    const bool aux = (c & 0x80000000u) != 0;
    const size_t i = c & 0x7fffffffu;
    if (i >= (aux ? n_aux : n_in)) return false;
    *v = aux ? n_in + i : i;
    return true;
  };
  for (size_t e = 0; e < nnz; e++) {
    size_t v;
    if (!flat(m.col[e], &v)) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "matrix %d names a variable the circuit does not have", which);
    vp[v + 1]++;
  }
  // entries stay grouped by variable in the transposed arrays; a row boundary is added after every SEGMENT terms
  std::vector<uint32_t> count(nv);
  for (size_t v = 0; v < nv; v++) {
    count[v] = vp[v + 1];
    vp[v + 1] += vp[v];
  }
  std::vector<uint32_t> cursor(vp.begin(), vp.end() - 1);
  for (size_t r = 0; r < n_rows; r++) {
    if (m.row_ptr[r + 1] < m.row_ptr[r] || m.row_ptr[r + 1] > nnz) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "matrix %d row_ptr is not monotone", which);
    for (uint64_t e = m.row_ptr[r]; e < m.row_ptr[r + 1]; e++) {
      size_t v;
      flat(m.col[e], &v);
      const uint32_t pos = cursor[v]++;
      rows[pos] = (uint32_t)r;
      memcpy(&co[4 * (size_t)pos], m.coeff + 4 * e, 32);
    }
  }
  // row offsets: the nv real rows (each at most SEGMENT terms long), then the extra segments.  The entries of one
  // variable are contiguous, so a real row followed by its extra rows covers them in order -- but extra rows sit
  // behind ALL real rows, so the entries are regrouped: real parts first, segments after.
  std::vector<uint32_t> rp(1, 0), src_lo, src_hi;
  for (size_t v = 0; v < nv; v++) {
    const uint32_t head = count[v] < SEGMENT ? count[v] : SEGMENT;
    src_lo.push_back(vp[v]);
    src_hi.push_back(vp[v] + head);
    rp.push_back(rp.back() + head);
  }
  for (size_t v = 0; v < nv; v++)
    for (uint32_t done = SEGMENT; done < count[v]; done += SEGMENT) {
      const uint32_t len = count[v] - done < SEGMENT ? count[v] - done : SEGMENT;
      src_lo.push_back(vp[v] + done);
      src_hi.push_back(vp[v] + done + len);
      rp.push_back(rp.back() + len);
      segs->owner.push_back((uint32_t)v);
    }
  if (!segs->owner.empty()) {
    std::vector<uint32_t> rows2(nnz);
    std::vector<uint64_t> co2(4 * nnz);
    for (size_t r = 0; r + 1 < rp.size(); r++) {
      memcpy(&rows2[rp[r]], &rows[src_lo[r]], (size_t)(src_hi[r] - src_lo[r]) * 4);
      memcpy(&co2[4 * (size_t)rp[r]], &co[4 * (size_t)src_lo[r]], (size_t)(src_hi[r] - src_lo[r]) * 32);
    }
    rows.swap(rows2);
    co.swap(co2);
  }
  vp.swap(rp);
  for (size_t r = 0; r + 1 < vp.size(); r++)
    if (vp[r + 1] - vp[r] > SPMV_LONG_ROW) longs.push_back((uint32_t)r);
  std::vector<uint16_t> codes;
  std::vector<Fr> table;
  bool needs_full = false;
  spmv_encode_coeffs(co.data(), nnz, codes, table, &needs_full);
  auto up = [&](auto** dst, const auto* src, size_t count) -> zg16_status {
    *dst = nullptr;
    if (!count) return ZG16_OK;
    ZG_HIP(ctx, hipMalloc((void**)dst, count * sizeof(**dst)));
    ZG_HIP(ctx, hipMemcpy(*dst, src, count * sizeof(**dst), hipMemcpyHostToDevice));
    return ZG16_OK;
  };
  out->nnz = nnz;
  out->n_long = (uint32_t)longs.size();
  out->n_table = (uint32_t)table.size();
  ZG_TRY(up(&out->row_ptr, vp.data(), vp.size()));
  ZG_TRY(up(&out->col, rows.data(), nnz));
  ZG_TRY(up(&out->code, codes.data(), nnz));
  ZG_TRY(up(&out->table, table.data(), table.size()));
  ZG_TRY(up(&out->long_rows, longs.data(), longs.size()));
  if (needs_full) ZG_TRY(up(&out->coeff, (const Fr*)co.data(), nnz));
  return ZG16_OK;
}

// (this file outlines the device-side field product; host code reaches the product through a host/device wrapper)
ZG_HD Fr mul_hd(const Fr& a, const Fr& b) { return mul(a, b); }

static Fr load_fr4(const uint64_t* p) {
  Fr r;
  memcpy(r.v, p, 32);
  return r;
}

static zg16_status generate_parameters(zg16_ctx* ctx, const zg16_r1cs_view* v, const zg16_toxic_waste* tw, zg16_parameters_out* out) {
  const size_t n_in = v->n_inputs, n_aux = v->n_aux, nv = n_in + n_aux, n_rows = v->n_rows;
  if (n_in < 1) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "n_inputs must include the constant ONE");
  if (nv >= 0x7fffffffull) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "too many variables");
  const unsigned log_m = ceil_log2(n_rows ? n_rows : 1);
  if (log_m >= 28) return fail(ctx, ZG16_ERR_POLYNOMIAL_DEGREE_TOO_LARGE, "%zu rows need a domain of 2^%u", n_rows, log_m);
  const size_t m = (size_t)1 << log_m;
  if (!out->ic || (m > 1 && !out->h) || (n_aux && !out->l) || !out->a || !out->b_g1 || !out->b_g2)
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "zg16_generate_parameters: an output array is missing");
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  const Fr tau = load_fr4(tw->tau), alpha = load_fr4(tw->alpha), beta = load_fr4(tw->beta), gamma = load_fr4(tw->gamma),
           delta = load_fr4(tw->delta);
  if (gamma.is_zero() || delta.is_zero()) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "gamma and delta must be invertible");
  const Fr gamma_inv = inv(gamma), delta_inv = inv(delta), one = Fr::one();
  G1Affine g1;
  G2Affine g2;
  memcpy(&g1, tw->g1, sizeof g1);
  memcpy(&g2, tw->g2, sizeof g2);

  // transposed matrices (a temporary zg16_r1cs whose rows are the variables), released on every exit
  std::unique_ptr<zg16_r1cs, void (*)(zg16_r1cs*)> t(new zg16_r1cs(), r1cs_destroy);
  t->n_in = m;  // (the "witness" of the transposed product is the vector L of m entries)
  t->n_aux = 0;
  t->log_m = ceil_log2(nv);
  const zg16_csr* mats[3] = {&v->a, &v->b, &v->c};
  SegmentList segs[3];
  size_t rows_t = nv;  // rows of the widest transposed matrix: the variables plus its extra segments
  for (int k = 0; k < 3; k++) {
    ZG_TRY(upload_transposed(ctx, *mats[k], n_rows, n_in, n_aux, k, &t->m[k], &segs[k]));
    rows_t = std::max(rows_t, nv + segs[k].owner.size());
  }
  // (a matrix with fewer extra rows than the widest gets empty rows up to rows_t: its row offsets are padded)
  for (int k = 0; k < 3; k++) {
    const size_t have = nv + segs[k].owner.size();
    if (have == rows_t) continue;
    std::vector<uint32_t> rp(rows_t + 1);
    ZG_HIP(ctx, hipMemcpy(rp.data(), t->m[k].row_ptr, (have + 1) * 4, hipMemcpyDeviceToHost));
    for (size_t r = have + 1; r <= rows_t; r++) rp[r] = rp[have];
    (void)hipFree(t->m[k].row_ptr);
    t->m[k].row_ptr = nullptr;
    ZG_HIP(ctx, hipMalloc((void**)&t->m[k].row_ptr, (rows_t + 1) * 4));
    ZG_HIP(ctx, hipMemcpy(t->m[k].row_ptr, rp.data(), (rows_t + 1) * 4, hipMemcpyHostToDevice));
  }
  t->n_rows = rows_t;

  // device work area: L (m), a_s, b_s, c_s (rows_t each), icl_s (nv), h_s (m), the owners of the extra rows
  const size_t n_extra = segs[0].owner.size() + segs[1].owner.size() + segs[2].owner.size();
  Fr* work = nullptr;
  ZG_HIP(ctx, hipMalloc((void**)&work, (2 * m + 3 * rows_t + nv) * sizeof(Fr) + (n_extra + 1) * 4));
  std::unique_ptr<Fr, void (*)(Fr*)> work_guard(work, [](Fr* p) { (void)hipFree(p); });
  Fr *lag = work, *a_s = lag + m, *b_s = a_s + rows_t, *c_s = b_s + rows_t, *icl_s = c_s + rows_t, *h_s = icl_s + nv;
  uint32_t* owners = (uint32_t*)(h_s + m);
  uint32_t* owner_k[3];
  {
    size_t at = 0;
    for (int k = 0; k < 3; k++) {
      owner_k[k] = owners + at;
      if (!segs[k].owner.empty())
        ZG_HIP(ctx, hipMemcpy(owner_k[k], segs[k].owner.data(), segs[k].owner.size() * 4, hipMemcpyHostToDevice));
      at += segs[k].owner.size();
    }
  }
  ZG_TRY(pow_table_device(ctx, lag, m, tau, one));
  ZG_TRY(ntt_device(ctx, lag, log_m, true, false));  // L_j(tau)
  ZG_TRY(spmv_device(ctx, t.get(), lag, a_s, b_s, c_s));
  if (n_extra)
    k_fold_segments<<<1, 64, 0, ctx->stream>>>(a_s, b_s, c_s, owner_k[0], (uint32_t)segs[0].owner.size(), owner_k[1],
                                                (uint32_t)segs[1].owner.size(), owner_k[2], (uint32_t)segs[2].owner.size(),
                                                (uint32_t)nv);
  k_setup_combine<<<(unsigned)((nv + 255) / 256), 256, 0, ctx->stream>>>(a_s, b_s, c_s, icl_s, (uint32_t)nv, (uint32_t)n_in, alpha,
                                                                       beta, gamma_inv, delta_inv);
  // h[i] = g1 * tau^i (tau^m - 1) / delta
  const Fr zt = mul_hd(sub(pow_u64(tau, (uint64_t)m), one), delta_inv);
  ZG_TRY(pow_table_device(ctx, h_s, m - 1, tau, zt));
  ZG_HIP(ctx, hipGetLastError());

  // every key point is a known scalar times g1 or g2
  std::vector<G1Affine> pa(nv), pb1(nv);
  std::vector<G2Affine> pb2(nv);
  std::vector<G1Affine> icl(nv);
  ZG_TRY(fixed_base_device<Fq>(ctx, g1, a_s, nv, pa.data()));
  ZG_TRY(fixed_base_device<Fq>(ctx, g1, b_s, nv, pb1.data()));
  ZG_TRY(fixed_base_device<Fq2>(ctx, g2, b_s, nv, pb2.data()));
  ZG_TRY(fixed_base_device<Fq>(ctx, g1, icl_s, nv, icl.data()));
  if (m > 1) ZG_TRY(fixed_base_device<Fq>(ctx, g1, h_s, m - 1, out->h));
  // bellman refuses a circuit with a variable that no constraint mentions (its l point would be the identity)
  for (size_t i = 0; i < n_aux; i++)
    if (icl[n_in + i].is_inf()) return fail(ctx, ZG16_ERR_UNCONSTRAINED_VARIABLE, "aux variable %zu occurs in no constraint", i);
  memcpy(out->ic, icl.data(), n_in * sizeof(G1Affine));
  if (n_aux) memcpy(out->l, icl.data() + n_in, n_aux * sizeof(G1Affine));
  // a, b_g1, b_g2 keep only the variables whose polynomial is non-zero (inputs first, then aux, order kept)
  size_t na = 0, nb = 0;
  for (size_t i = 0; i < nv; i++) {
    if (!pa[i].is_inf()) memcpy(out->a + 8 * na++, &pa[i], sizeof(G1Affine));
    if (!pb1[i].is_inf()) {
      memcpy(out->b_g1 + 8 * nb, &pb1[i], sizeof(G1Affine));
      memcpy(out->b_g2 + 16 * nb, &pb2[i], sizeof(G2Affine));
      nb++;
    }
  }
  out->n_a = na;
  out->n_b = nb;
  // the verifying key's own points
  Fr* vk_s;
  ZG_TRY(scratch_get(ctx, 7, 4 * sizeof(Fr), (void**)&vk_s));
  const Fr vks[4] = {alpha, beta, delta, gamma};
  ZG_HIP(ctx, hipMemcpyAsync(vk_s, vks, sizeof vks, hipMemcpyHostToDevice, ctx->stream));
  G1Affine v1[3];
  G2Affine v2[3];
  ZG_TRY(fixed_base_device<Fq>(ctx, g1, vk_s, 3, v1));
  ZG_TRY(fixed_base_device<Fq2>(ctx, g2, vk_s + 1, 3, v2));
  memcpy(out->alpha_g1, &v1[0], 64); memcpy(out->beta_g1, &v1[1], 64); memcpy(out->delta_g1, &v1[2], 64);
  memcpy(out->beta_g2, &v2[0], 128); memcpy(out->delta_g2, &v2[1], 128); memcpy(out->gamma_g2, &v2[2], 128);
  return ZG16_OK;
}

}  // namespace zg

extern "C" {
zg16_status zg16_fixed_base_g1(zg16_ctx* ctx, const uint64_t base[8], const uint64_t* scalars, size_t n, uint64_t* out) {
  return zg::fixed_base<zg::Fq>(ctx, base, scalars, n, out);
}
zg16_status zg16_fixed_base_g2(zg16_ctx* ctx, const uint64_t base[16], const uint64_t* scalars, size_t n, uint64_t* out) {
  return zg::fixed_base<zg::Fq2>(ctx, base, scalars, n, out);
}
zg16_status zg16_generate_parameters(zg16_ctx* ctx, const zg16_r1cs_view* circuit, const zg16_toxic_waste* toxic,
                                     zg16_parameters_out* out) {
  if (!ctx || !circuit || !toxic || !out) return zg::fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_generate_parameters");
  return zg::generate_parameters(ctx, circuit, toxic, out);
}
}
