// Proving keys straight from the bytes `zvm setup` writes.
//
// zinc-vm reads its key with `Parameters::<Bn256>::read(file, true)`
// (zinc-vm/src/zvm/arguments/command/prove.rs:73-76): every point is parsed from the uncompressed
// pairing_ce encoding, checked to be on the curve and in the prime-order subgroup, and the query
// vectors may not contain the point at infinity.  On one host thread that is minutes for a 2^24-row
// key; here the byte ranges go to the GPU as they are and one kernel decodes, converts to Montgomery
// form and checks each point (G2: on the twist and psi(P) = [6x^2] P, which holds exactly when r P = O -- see
// g2_in_subgroup in curve.hpp; `checked` = 2 runs the plain multiplication by r instead, kept as its checker).
//
// Layout (bellman_ce 0.3.0 / pairing_ce 0.17.0, not part of the reference tree):
//   alpha_g1 beta_g1 (G1) beta_g2 gamma_g2 (G2) delta_g1 (G1) delta_g2 (G2) | u32-BE n, ic[n] (G1)
//   | u32-BE n, h[n] | l | a | b_g1 (G1) | b_g2 (G2)
//   G1 = x || y, G2 = x.c1 || x.c0 || y.c1 || y.c0, 32-byte big-endian canonical coordinates;
//   byte 0 bit 6 = point at infinity (all other bits zero), bit 7 must be clear (compression flag).
#include <string.h>

#include <vector>

#include <memory>

#define ZG_OUTLINE_FIELD_MUL 1
#include "r1cs.hpp"
#include "msm_ops.hpp"

namespace zg {

enum DecodeError : uint32_t { DEC_OK = 0, DEC_ENCODING = 1, DEC_NOT_ON_CURVE = 2, DEC_NOT_IN_SUBGROUP = 3, DEC_INFINITY = 4 };

// 32 big-endian bytes -> Montgomery Fq; false if the integer is not below q
ZG_HD bool read_fq(const uint8_t* b, bool mask_flags, Fq* out) {
  Fq x;
  for (int i = 0; i < 8; i++) {
    const uint8_t* p = b + 28 - 4 * i;
    uint32_t hi = p[0];
    if (mask_flags && i == 7) hi &= 0x3f;
    x.v[i] = (hi << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
  }
  if (!is_canonical<FqParams>(x.v)) return false;
  *out = to_mont(x);
  return true;
}

ZG_HD bool all_zero(const uint8_t* b, int n) {
  uint8_t o = b[0] & 0x3f;
  for (int i = 1; i < n; i++) o |= b[i];
  return o == 0;
}

ZG_HD uint32_t decode_g1(const uint8_t* b, bool checked, bool allow_inf, Affine<Fq>* out) {
  *out = Affine<Fq>::inf();
  if (b[0] & 0x80) return DEC_ENCODING;
  if (b[0] & 0x40) {
    if (!all_zero(b, 64)) return DEC_ENCODING;
    return allow_inf ? DEC_OK : DEC_INFINITY;
  }
  Affine<Fq> p;
  if (!read_fq(b, true, &p.x) || !read_fq(b + 32, false, &p.y)) return DEC_ENCODING;
  // G1 has cofactor one: on the curve means in the subgroup
  if (checked && !on_curve(p, from_u64<FqParams>(3))) return DEC_NOT_ON_CURVE;
  *out = p;
  return DEC_OK;
}

// r, the group order, little-endian words
ZG_HD uint32_t order_word(int i) { return FrParams::mod(i); }

// checked: 0 none; 1 on the twist and in the subgroup (psi test); 2 the same with the plain r P = O; 3 on the twist
// only (the caller runs the subgroup test in its own kernel, k_g2_subgroup)
ZG_HD uint32_t decode_g2(const uint8_t* b, int checked, bool allow_inf, const G2SubgroupConsts& k, Affine<Fq2>* out) {
  *out = Affine<Fq2>::inf();
  if (b[0] & 0x80) return DEC_ENCODING;
  if (b[0] & 0x40) {
    if (!all_zero(b, 128)) return DEC_ENCODING;
    return allow_inf ? DEC_OK : DEC_INFINITY;
  }
  Affine<Fq2> p;
  if (!read_fq(b, true, &p.x.c1) || !read_fq(b + 32, false, &p.x.c0) || !read_fq(b + 64, false, &p.y.c1) ||
      !read_fq(b + 96, false, &p.y.c0))
    return DEC_ENCODING;
  if (checked) {
    if (!on_curve(p, k.curve_b)) return DEC_NOT_ON_CURVE;
    if (checked == 2) {
      uint32_t r[8];
      for (int i = 0; i < 8; i++) r[i] = order_word(i);
      if (!scalar_mul(XYZZ<Fq2>::from_affine(p), r).is_inf()) return DEC_NOT_IN_SUBGROUP;
    } else if (checked == 1 && !g2_in_subgroup(p, k)) {
      return DEC_NOT_IN_SUBGROUP;
    }
  }
  *out = p;
  return DEC_OK;
}

__global__ __launch_bounds__(256) void k_decode_g1(const uint8_t* __restrict__ src, uint32_t n, int checked,
                                                   Affine<Fq>* __restrict__ dst, uint32_t* __restrict__ err) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<Fq> p;
  const uint32_t e = decode_g1(src + (size_t)i * 64, checked != 0, false, &p);
  dst[i] = p;
  if (e) atomicMax(err, e);
}

__global__ __launch_bounds__(256, 1) void k_decode_g2(const uint8_t* __restrict__ src, uint32_t n, int checked, G2SubgroupConsts curve_b,
                                                      Affine<Fq2>* __restrict__ dst, uint32_t* __restrict__ err) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<Fq2> p;
  const uint32_t e = decode_g2(src + (size_t)i * 128, checked, false, curve_b, &p);
  dst[i] = p;
  if (e) atomicMax(err, e);
}

// The subgroup test of g2_in_subgroup -- psi(P) = [6x^2] P -- for a whole query vector, in the arithmetic of the
// multiexp kernels: a lane pair per point (even lane: real components, odd lane: imaginary ones), 29-bit limbs,
// three waves per SIMD, where the element-wise form above is one lane per point in saturated limbs at one wave per
// SIMD.  pts: decoded points in the ABI form (the identity is skipped: the decoder has already judged it).
// psi: the constants (psi_x, psi_y) of G2SubgroupConsts as one "point" in the multiexp kernels' packed form.
__global__ __launch_bounds__(256, 2) void k_g2_subgroup(const Affine<Fq2>* __restrict__ pts, uint32_t n,
                                                        const Affine<Fq2>* __restrict__ psi, uint32_t* __restrict__ err) {
  using O = MsmOps<Fq2>;
  using P = Fq29Params;
  const uint32_t j = (blockIdx.x * blockDim.x + threadIdx.x) / 2;
  if (j >= n) return;  // (the lanes of a pair share j)
  const int im = O::sub();
  const Fq* c = (const Fq*)(pts + j);
  const Fq x = c[im], y = c[2 + im];
  if (pair_both(x.is_zero() && y.is_zero())) return;
  // ABI form (Montgomery radix 2^256) -> the 29-bit code's radix 2^261, canonical as a stored base is: one product
  // and a conditional subtraction.  (Five modular doublings of the packed words, the way the table builder converts,
  // are compiled into a loop over partly overlapping vector loads here that corrupts words of the value: hipcc 7.2.)
  const O::Aff base = {unpack29<P>(pack29<P, FqParams>(from_r256<P>(x))), unpack29<P>(pack29<P, FqParams>(from_r256<P>(y)))};
  O::Aff k;
  O::load_base(psi, false, &k);
  const O::Pt acc = g2_times_6x2_pair(base);
  // the identity is not psi(P), which is a finite point
  const bool inf = O::is_inf(acc);
  const bool eq = pair_equals_affine(acc, g2_psi_pair(base, k.x, k.y));
  if ((inf || !eq) && !im) atomicMax(err, (uint32_t)DEC_NOT_IN_SUBGROUP);
}

static const char* decode_text(uint32_t e) {
  switch (e) {
    case DEC_ENCODING: return "malformed point encoding";
    case DEC_NOT_ON_CURVE: return "point is not on the curve";
    case DEC_NOT_IN_SUBGROUP: return "point is not in the prime-order subgroup";
    case DEC_INFINITY: return "point at infinity in a query vector";
    default: return "ok";
  }
}

struct Cursor {
  const uint8_t* p;
  size_t left;
  bool take(size_t n, const uint8_t** out) {
    if (left < n) return false;
    *out = p;
    p += n;
    left -= n;
    return true;
  }
  bool u32be(size_t* v) {
    const uint8_t* b;
    if (!take(4, &b)) return false;
    *v = ((size_t)b[0] << 24) | ((size_t)b[1] << 16) | ((size_t)b[2] << 8) | b[3];
    return true;
  }
};

// The direct table (small queries) and the other windows of a decoded query.
template <class F>
static zg16_status query_tables(zg16_ctx* ctx, Affine<F>* pts, size_t count, uint32_t table_c, uint32_t table_step,
                                Affine<F>** direct) {
  if (!count) return ZG16_OK;
  if (table_c) ZG_TRY(msm_build_direct<F>(ctx, pts, count, direct));
  return msm_build_table<F>(ctx, pts, count, table_c, table_step);
}

// The packed-form psi constants k_g2_subgroup takes (the ABI values times 32), at an address that outlives any copy.
static const Affine<Fq2>* psi_consts_internal() {
  static const Affine<Fq2> packed = MsmOps<Fq2>::to_internal(Affine<Fq2>{g2_subgroup_consts().psi_x, g2_subgroup_consts().psi_y});
  return &packed;
}

// Queue, on `st`, the copy of points [lo, hi) of a vector of encodings to the GPU and their decoding into window 0
// of a fresh table (*dst, allocated here with room for every window), with the checks `checked` asks for.  Nothing
// waits: a violation raises the flag *err (device memory, one word per query, cleared by the caller) and the caller
// reads the flags once everything is queued.  raw: device scratch for the encodings (count * sizeof(Affine<F>) bytes
// rounded up to 256, and 256 more).
template <class F>
static zg16_status decode_enqueue(zg16_ctx* ctx, hipStream_t st, const uint8_t* bytes, size_t lo, size_t hi, int checked,
                                  uint32_t table_c, uint32_t table_step, const G2SubgroupConsts& consts, uint8_t* raw,
                                  uint32_t* err, Affine<F>** dst) {
  *dst = nullptr;
  const size_t count = hi - lo;
  if (!count) return ZG16_OK;
  constexpr size_t PB = sizeof(Affine<F>);  // the encoding is as long as the affine point
  const size_t windows = table_c ? msm_table_windows(table_c, table_step) : 1;
  ZG_HIP(ctx, hipMalloc((void**)dst, windows * count * sizeof(Affine<F>)));
  ZG_HIP(ctx, hipMemcpyAsync(raw, bytes + lo * PB, count * PB, hipMemcpyHostToDevice, st));
  const unsigned grid = (unsigned)((count + 255) / 256);
  if constexpr (sizeof(F) == sizeof(Fq)) {
    k_decode_g1<<<grid, 256, 0, st>>>(raw, (uint32_t)count, checked, *dst, err);
  } else {
    // checked = 1: on the twist in the decoder, the subgroup test in the lane-pair kernel; 2: the plain r P = O
    // inside the decoder (the slow form, kept as the checker of the other)
    k_decode_g2<<<grid, 256, 0, st>>>(raw, (uint32_t)count, checked == 1 ? 3 : checked, consts, *dst, err);
    if (checked == 1) {
      // (the constants ride in the slack behind the encodings)
      Affine<Fq2>* psi = (Affine<Fq2>*)(raw + ((count * PB + 255) & ~(size_t)255));
      ZG_HIP(ctx, hipMemcpyAsync(psi, psi_consts_internal(), sizeof(Affine<Fq2>), hipMemcpyHostToDevice, st));
      k_g2_subgroup<<<(unsigned)((2 * count + 255) / 256), 256, 0, st>>>(*dst, (uint32_t)count, psi, err);
    }
  }
  ZG_HIP(ctx, hipGetLastError());
  return ZG16_OK;
}

}  // namespace zg

using namespace zg;

extern "C" {

zg16_status zg16_pk_upload_bytes(zg16_ctx* ctx, const uint8_t* data, size_t len, int checked, int shard_index,
                                 int shard_count, zg16_pk** out) {
  if (!ctx || !data || !out || shard_count < 1 || shard_index < 0 || shard_index >= shard_count)
    return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to zg16_pk_upload_bytes");
  *out = nullptr;
  if (len >= ((size_t)1 << 40)) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "key data too large");
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  const G2SubgroupConsts& curve_b = g2_subgroup_consts();
  Cursor cur = {data, len};
  auto eof = [&]() { return fail(ctx, ZG16_ERR_MALFORMED_KEY, "unexpected end of key data"); };
  std::unique_ptr<zg16_pk, void (*)(zg16_pk*)> pk(new zg16_pk(), zg::pk_destroy);
  pk->ctx = ctx;
  pk->shard_index = shard_index;
  pk->shard_count = shard_count;

  // verifying key: six points, then the ic vector (not used by the prover, still validated).  bellman's
  // VerifyingKey::read converts every one of them with into_affine() -- on the curve and in the subgroup -- whatever
  // `checked` says (that flag only governs the five query vectors), and refuses the identity inside ic.
  const uint8_t* b;
  uint32_t e;
  auto g1 = [&](G1Affine* dst, const char* what, bool allow_inf = true) -> zg16_status {
    if (!cur.take(64, &b)) return eof();
    if ((e = decode_g1(b, true, allow_inf, dst))) return fail(ctx, ZG16_ERR_MALFORMED_KEY, "%s: %s", what, decode_text(e));
    return ZG16_OK;
  };
  auto g2 = [&](G2Affine* dst, const char* what) -> zg16_status {
    if (!cur.take(128, &b)) return eof();
    if ((e = decode_g2(b, checked == 2 ? 2 : 1, true, curve_b, dst))) return fail(ctx, ZG16_ERR_MALFORMED_KEY, "%s: %s", what, decode_text(e));
    return ZG16_OK;
  };
  G2Affine gamma2;
  ZG_TRY(g1(&pk->alpha_g1, "alpha_g1"));
  ZG_TRY(g1(&pk->beta_g1, "beta_g1"));
  ZG_TRY(g2(&pk->beta_g2, "beta_g2"));
  ZG_TRY(g2(&gamma2, "gamma_g2"));
  ZG_TRY(g1(&pk->delta_g1, "delta_g1"));
  ZG_TRY(g2(&pk->delta_g2, "delta_g2"));
  pk->build_delta_tables();
  size_t n_ic;
  if (!cur.u32be(&n_ic)) return eof();
  for (size_t i = 0; i < n_ic; i++) {
    G1Affine ic;
    ZG_TRY(g1(&ic, "ic", false));
  }

  // the five query vectors: locate them first, then decode this shard's ranges on the GPU
  const uint8_t* vec[5];
  size_t cnt[5];
  static const char* const names[5] = {"h", "l", "a", "b_g1", "b_g2"};
  for (int k = 0; k < 5; k++) {
    if (!cur.u32be(&cnt[k])) return eof();
    if (!cur.take(cnt[k] * (k == 4 ? 128 : 64), &vec[k])) return eof();
  }
  if (cnt[3] != cnt[4]) return fail(ctx, ZG16_ERR_MALFORMED_KEY, "b_g1 and b_g2 lengths differ");
  pk->n_h = cnt[0]; pk->n_l = cnt[1]; pk->n_a = cnt[2]; pk->n_b1 = cnt[3]; pk->n_b2 = cnt[4];
  auto range = [&](size_t n, size_t* lo, size_t* hi) {
    *lo = n * (size_t)shard_index / (size_t)shard_count;
    *hi = n * (size_t)(shard_index + 1) / (size_t)shard_count;
  };
  range(pk->n_h, &pk->lo_h, &pk->hi_h);
  range(pk->n_l, &pk->lo_l, &pk->hi_l);
  range(pk->n_a, &pk->lo_a, &pk->hi_a);
  range(pk->n_b1, &pk->lo_b, &pk->hi_b);
  size_t largest = pk->hi_l - pk->lo_l;
  for (size_t q : {pk->hi_a - pk->lo_a, pk->hi_b - pk->lo_b}) largest = q > largest ? q : largest;
  pk->table_c = msm_table_window(largest, 1);
  pk->table_c_h = msm_table_window(pk->hi_h - pk->lo_h, 0);
  pk->table_c_g2 = msm_table_window(pk->hi_b - pk->lo_b, 2);
  {
    const TableQuery tq[5] = {{pk->hi_h - pk->lo_h, pk->table_c_h, 64}, {pk->hi_l - pk->lo_l, pk->table_c, 64},
                              {pk->hi_a - pk->lo_a, pk->table_c, 64}, {pk->hi_b - pk->lo_b, pk->table_c, 64},
                              {pk->hi_b - pk->lo_b, pk->table_c_g2, 128}};
    pk->table_step = msm_table_step(tq, 5);
  }
  // Two chains side by side.  The copy / decode / check chain runs on the context's second stream in the order h, l,
  // a, b_g1, b_g2 and records an event behind each query; the table builders follow on the main stream (small keys:
  // on all four streams, each query's chain of short kernels beside the others'), each behind its query's event.  So
  // the PCIe copy and the checks of the later queries -- the G2 subgroup test is the longest -- hide under the table
  // expansion of the earlier ones, and the host waits once, at the end.
  size_t most = pk->hi_h - pk->lo_h;
  for (size_t q : {pk->hi_l - pk->lo_l, pk->hi_a - pk->lo_a, pk->hi_b - pk->lo_b}) most = q > most ? q : most;
  const bool side_by_side = most <= ((size_t)1 << 16) && ctx->stream == ctx->own_stream && !ctx->deferred_frees;
  const size_t cnt_q[5] = {pk->hi_h - pk->lo_h, pk->hi_l - pk->lo_l, pk->hi_a - pk->lo_a, pk->hi_b - pk->lo_b, pk->hi_b - pk->lo_b};
  const size_t lo_q[5] = {pk->lo_h, pk->lo_l, pk->lo_a, pk->lo_b, pk->lo_b};
  size_t raw_off[6] = {0};
  for (int k = 0; k < 5; k++) raw_off[k + 1] = raw_off[k] + ((cnt_q[k] * (k == 4 ? 128 : 64) + 255) & ~(size_t)255) + 256;
  uint8_t* raw = nullptr;
  uint32_t* err = nullptr;
  hipEvent_t decoded[5] = {};
  std::vector<void*> scratch_to_free;
  hipStream_t const own = ctx->stream;
  hipStream_t const feed = ctx->stream == ctx->own_stream ? ctx->aux_stream : ctx->stream;
  auto run = [&]() -> zg16_status {
    ZG_HIP(ctx, hipMalloc((void**)&raw, raw_off[5] + 256));
    ZG_HIP(ctx, hipMalloc((void**)&err, 256));
    ZG_HIP(ctx, hipMemsetAsync(err, 0, 256, feed));
    for (auto& e : decoded) ZG_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // the builders are queued without waiting (their scratch is freed below, after the streams have drained), each
    // right behind its own query's decoding: a copy from pageable memory holds the host until it is staged, so the
    // tables of query k are already being built while the host feeds query k + 1
    auto on = [&](hipStream_t s, int k) -> zg16_status {
      ZG_HIP(ctx, hipEventRecord(decoded[k], feed));
      ctx->stream = side_by_side ? s : own;
      ZG_HIP(ctx, hipStreamWaitEvent(ctx->stream, decoded[k], 0));
      return ZG16_OK;
    };
    ZG_TRY(decode_enqueue<Fq>(ctx, feed, vec[0], lo_q[0], lo_q[0] + cnt_q[0], checked, pk->table_c_h, pk->table_step, curve_b, raw + raw_off[0], err + 0, &pk->h));
    ZG_TRY(on(own, 0));
    ZG_TRY(query_tables<Fq>(ctx, pk->h, cnt_q[0], pk->table_c_h, pk->table_step, &pk->h_d));
    ZG_TRY(decode_enqueue<Fq>(ctx, feed, vec[1], lo_q[1], lo_q[1] + cnt_q[1], checked, pk->table_c, pk->table_step, curve_b, raw + raw_off[1], err + 1, &pk->l));
    ZG_TRY(on(ctx->early_stream, 1));
    ZG_TRY(query_tables<Fq>(ctx, pk->l, cnt_q[1], pk->table_c, pk->table_step, &pk->l_d));
    ZG_TRY(decode_enqueue<Fq>(ctx, feed, vec[2], lo_q[2], lo_q[2] + cnt_q[2], checked, pk->table_c, pk->table_step, curve_b, raw + raw_off[2], err + 2, &pk->a));
    ZG_TRY(on(ctx->side_stream, 2));
    ZG_TRY(query_tables<Fq>(ctx, pk->a, cnt_q[2], pk->table_c, pk->table_step, &pk->a_d));
    ZG_TRY(decode_enqueue<Fq>(ctx, feed, vec[3], lo_q[3], lo_q[3] + cnt_q[3], checked, pk->table_c, pk->table_step, curve_b, raw + raw_off[3], err + 3, &pk->b1));
    ZG_TRY(on(own, 3));
    ZG_TRY(query_tables<Fq>(ctx, pk->b1, cnt_q[3], pk->table_c, pk->table_step, &pk->b1_d));
    ZG_TRY(decode_enqueue<Fq2>(ctx, feed, vec[4], lo_q[4], lo_q[4] + cnt_q[4], checked, pk->table_c_g2, pk->table_step, curve_b, raw + raw_off[4], err + 4, &pk->b2));
    ZG_TRY(on(feed, 4));
    ZG_TRY(query_tables<Fq2>(ctx, pk->b2, cnt_q[4], pk->table_c_g2, pk->table_step, &pk->b2_d));
    return ZG16_OK;
  };
  std::vector<void*>* const outer_frees = ctx->deferred_frees;
  if (!outer_frees) ctx->deferred_frees = &scratch_to_free;
  const zg16_status queued = run();
  ctx->stream = own;
  ctx->deferred_frees = outer_frees;
  // every stream, also after a failure: the buffers below are in use until then
  const hipError_t drained = hipDeviceSynchronize();
  uint32_t herr[5] = {};
  hipError_t read = hipSuccess;
  if (err) read = hipMemcpy(herr, err, sizeof herr, hipMemcpyDeviceToHost);
  for (void* p : scratch_to_free) (void)hipFree(p);
  (void)hipFree(raw);
  (void)hipFree(err);
  for (auto& e : decoded)
    if (e) (void)hipEventDestroy(e);
  ZG_TRY(queued);
  ZG_HIP(ctx, drained);
  ZG_HIP(ctx, read);
  for (int k = 0; k < 5; k++)
    if (herr[k]) return fail(ctx, ZG16_ERR_MALFORMED_KEY, "%s query: %s", names[k], decode_text(herr[k]));
  *out = pk.release();
  return ZG16_OK;
}

// 32 big-endian bytes of the canonical value of a Montgomery Fq
static void write_fq(const Fq& m, uint8_t* b) {
  const Fq x = from_mont(m);
  for (int i = 0; i < 8; i++) {
    uint8_t* p = b + 28 - 4 * i;
    p[0] = (uint8_t)(x.v[i] >> 24); p[1] = (uint8_t)(x.v[i] >> 16); p[2] = (uint8_t)(x.v[i] >> 8); p[3] = (uint8_t)x.v[i];
  }
}
// is the canonical value of a greater than that of b (both Montgomery)?
static bool fq_gt(const Fq& a, const Fq& b) {
  const Fq x = from_mont(a), y = from_mont(b);
  for (int i = 7; i >= 0; i--)
    if (x.v[i] != y.v[i]) return x.v[i] > y.v[i];
  return false;
}

// bellman `Proof::write` (zinc-vm/src/zvm/arguments/command/prove.rs:171-174): A, B, C in pairing_ce's
// compressed encoding -- x big-endian (G2: c1 then c0), bit 7 of byte 0 set when y is the larger of
// (y, -y) (Fq2 compared by c1, then c0), bit 6 for the point at infinity.
zg16_status zg16_proof_encode(const zg16_proof* proof, uint8_t out[128]) {
  if (!proof || !out) return ZG16_ERR_INVALID_ARGUMENT;
  memset(out, 0, 128);
  auto g1 = [](const uint64_t* limbs, uint8_t* dst) {
    G1Affine p;
    memcpy(&p, limbs, sizeof p);
    if (p.is_inf()) { dst[0] = 0x40; return; }
    write_fq(p.x, dst);
    if (fq_gt(p.y, neg(p.y))) dst[0] |= 0x80;
  };
  g1(proof->a, out);
  {
    G2Affine p;
    memcpy(&p, proof->b, sizeof p);
    uint8_t* dst = out + 32;
    if (p.is_inf()) {
      dst[0] = 0x40;
    } else {
      write_fq(p.x.c1, dst);
      write_fq(p.x.c0, dst + 32);
      const Fq2 ny = neg(p.y);
      const bool larger = p.y.c1 != ny.c1 ? fq_gt(p.y.c1, ny.c1) : fq_gt(p.y.c0, ny.c0);
      if (larger) dst[0] |= 0x80;
    }
  }
  g1(proof->c, out + 96);
  return ZG16_OK;
}

// The verifying-key points a key handle holds, as a view for zg16_prove_finish (query fields stay null).
zg16_status zg16_pk_vk_view(const zg16_pk* pk, zg16_pk_view* out) {
  if (!pk || !out) return ZG16_ERR_INVALID_ARGUMENT;
  memset(out, 0, sizeof *out);
  out->alpha_g1 = (const uint64_t*)&pk->alpha_g1;
  out->beta_g1 = (const uint64_t*)&pk->beta_g1;
  out->beta_g2 = (const uint64_t*)&pk->beta_g2;
  out->delta_g1 = (const uint64_t*)&pk->delta_g1;
  out->delta_g2 = (const uint64_t*)&pk->delta_g2;
  out->n_h = pk->n_h; out->n_l = pk->n_l; out->n_a = pk->n_a; out->n_b_g1 = pk->n_b1; out->n_b_g2 = pk->n_b2;
  return ZG16_OK;
}
}
