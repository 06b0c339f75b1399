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
    } else if (!g2_in_subgroup(p, k)) {
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

// Decode points [lo, hi) of a vector of `point_bytes`-byte encodings into window 0 of a fresh table and
// fill the other windows.
template <class F>
static zg16_status decode_query(zg16_ctx* ctx, const uint8_t* bytes, size_t lo, size_t hi, int checked, uint32_t table_c,
                                uint32_t table_step, const G2SubgroupConsts& curve_b, const char* name, Affine<F>** dst,
                                Affine<F>** direct, bool build_tables) {
  *dst = nullptr;
  *direct = nullptr;
  const size_t count = hi - lo;
  if (!count) return ZG16_OK;
  constexpr size_t PB = sizeof(Affine<F>);  // the encoding is as long as the affine point
  const size_t windows = table_c ? msm_table_windows(table_c, table_step) : 1;
  ZG_HIP(ctx, hipMalloc((void**)dst, windows * count * sizeof(Affine<F>)));
  uint8_t* raw;
  ZG_TRY(scratch_get(ctx, 5, count * PB + 256, (void**)&raw));
  uint32_t* err;
  ZG_TRY(scratch_get(ctx, 7, 256, (void**)&err));
  ZG_HIP(ctx, hipMemsetAsync(err, 0, 4, ctx->stream));
  ZG_HIP(ctx, hipMemcpyAsync(raw, bytes + lo * PB, count * PB, hipMemcpyHostToDevice, ctx->stream));
  const unsigned grid = (unsigned)((count + 255) / 256);
  if constexpr (sizeof(F) == sizeof(Fq))
    k_decode_g1<<<grid, 256, 0, ctx->stream>>>(raw, (uint32_t)count, checked, *dst, err);
  else
    k_decode_g2<<<grid, 256, 0, ctx->stream>>>(raw, (uint32_t)count, checked, curve_b, *dst, err);
  ZG_HIP(ctx, hipGetLastError());
  uint32_t herr = 0;
  ZG_HIP(ctx, hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, ctx->stream));
  ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (herr) return fail(ctx, ZG16_ERR_MALFORMED_KEY, "%s query: %s", name, decode_text(herr));
  return build_tables ? query_tables<F>(ctx, *dst, count, table_c, table_step, direct) : ZG16_OK;
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
  // Small keys: decode the five queries, then build their tables side by side on the context's four streams (each
  // is a chain of short kernels on a handful of wavefronts); large keys fill the GPU one query at a time.
  size_t most = pk->hi_h - pk->lo_h;
  for (size_t q : {pk->hi_l - pk->lo_l, pk->hi_a - pk->lo_a, pk->hi_b - pk->lo_b}) most = q > most ? q : most;
  const bool side_by_side = most <= ((size_t)1 << 16) && ctx->stream == ctx->own_stream && !ctx->deferred_frees;
  const bool now = !side_by_side;
  ZG_TRY(decode_query<Fq>(ctx, vec[0], pk->lo_h, pk->hi_h, checked, pk->table_c_h, pk->table_step, curve_b, names[0], &pk->h, &pk->h_d, now));
  ZG_TRY(decode_query<Fq>(ctx, vec[1], pk->lo_l, pk->hi_l, checked, pk->table_c, pk->table_step, curve_b, names[1], &pk->l, &pk->l_d, now));
  ZG_TRY(decode_query<Fq>(ctx, vec[2], pk->lo_a, pk->hi_a, checked, pk->table_c, pk->table_step, curve_b, names[2], &pk->a, &pk->a_d, now));
  ZG_TRY(decode_query<Fq>(ctx, vec[3], pk->lo_b, pk->hi_b, checked, pk->table_c, pk->table_step, curve_b, names[3], &pk->b1, &pk->b1_d, now));
  ZG_TRY(decode_query<Fq2>(ctx, vec[4], pk->lo_b, pk->hi_b, checked, pk->table_c_g2, pk->table_step, curve_b, names[4], &pk->b2, &pk->b2_d, now));
  if (side_by_side) {
    std::vector<void*> scratch_to_free;
    hipStream_t const own = ctx->stream;
    auto build_all = [&]() -> zg16_status {
      ctx->stream = ctx->aux_stream;
      ZG_TRY(query_tables<Fq2>(ctx, pk->b2, pk->hi_b - pk->lo_b, pk->table_c_g2, pk->table_step, &pk->b2_d));
      ctx->stream = own;
      ZG_TRY(query_tables<Fq>(ctx, pk->h, pk->hi_h - pk->lo_h, pk->table_c_h, pk->table_step, &pk->h_d));
      ctx->stream = ctx->early_stream;
      ZG_TRY(query_tables<Fq>(ctx, pk->l, pk->hi_l - pk->lo_l, pk->table_c, pk->table_step, &pk->l_d));
      ctx->stream = ctx->side_stream;
      ZG_TRY(query_tables<Fq>(ctx, pk->a, pk->hi_a - pk->lo_a, pk->table_c, pk->table_step, &pk->a_d));
      ctx->stream = own;
      ZG_TRY(query_tables<Fq>(ctx, pk->b1, pk->hi_b - pk->lo_b, pk->table_c, pk->table_step, &pk->b1_d));
      return ZG16_OK;
    };
    ctx->deferred_frees = &scratch_to_free;
    const zg16_status built = build_all();
    ctx->stream = own;
    ctx->deferred_frees = nullptr;
    const hipError_t drained = hipDeviceSynchronize();  // also after a failure: the scratch is in use until then
    for (void* p : scratch_to_free) (void)hipFree(p);
    ZG_TRY(built);
    ZG_HIP(ctx, drained);
  }
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
