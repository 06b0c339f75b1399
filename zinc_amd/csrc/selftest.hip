// Element-wise device evaluation of the field and group primitives, so that parity tests can pin
// them (edge cases included) against golden vectors through the C ABI.  Not used while proving.
#include <string.h>

#define ZG_OUTLINE_FIELD_MUL 1
#include "common.hpp"
#include "msm_ops.hpp"

namespace zg {

template <class T>
ZG_D T field_op(int op, const T& a, const T& b) {
  switch (op) {
    case 0: return add(a, b);
    case 1: return sub(a, b);
    case 2: return mul(a, b);
    case 3: return sqr(a);
    case 4: return neg(a);
    default: return a.is_zero() ? a : inv(a);
  }
}

template <class T>
__global__ void k_field_op(int op, const T* a, const T* b, T* out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = field_op(op, a[i], b[i]);
}

template <class F>
__global__ void k_group_op(int op, const Affine<F>* p, const void* q, Affine<F>* out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  XYZZ<F> acc = XYZZ<F>::from_affine(p[i]);
  if (op == 0) {
    acc = dbl(acc);
  } else if (op == 1) {
    add_mixed(acc, ((const Affine<F>*)q)[i]);
  } else if (op == 2) {
    // route the second operand through a non-trivial XYZZ representation (2q - q)
    const Affine<F> qa = ((const Affine<F>*)q)[i];
    XYZZ<F> t = dbl(XYZZ<F>::from_affine(qa));
    add_mixed(t, neg(qa));
    add_full(acc, t);
  } else {
    const Fr k = from_mont(((const Fr*)q)[i]);
    acc = scalar_mul(acc, k.v);
  }
  out[i] = to_affine(acc);
}

// The same group operations through the arithmetic the multiexp kernels use (MsmOps: 29-bit-limb
// field, G2 spread over lane pairs).  p and q arrive in the kernels' packed base form; the result is
// written as an XYZZ point in the library-wide form.
template <class F>
__global__ __launch_bounds__(64) void k_group_op_msm(int op, const Affine<F>* p, const void* q, XYZZ<F>* out, uint32_t n) {
  using O = MsmOps<F>;
  const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) / O::LP;
  if (i >= n) return;
  typename O::Pt acc = O::inf();
  typename O::Aff a;
  if (O::load_base(p + i, false, &a)) O::madd(acc, a);
  if (op == 0) {
    acc = O::dbl(acc);
  } else if (op == 1) {
    if (O::load_base((const Affine<F>*)q + i, false, &a)) O::madd(acc, a);
  } else if (op == 2) {
    // route the second operand through a non-trivial XYZZ representation (2q - q)
    typename O::Pt t = O::inf();
    if (O::load_base((const Affine<F>*)q + i, false, &a)) {
      O::madd(t, a);
      t = O::dbl(t);
      O::load_base((const Affine<F>*)q + i, true, &a);
      O::madd(t, a);
    }
    O::add(acc, t);
  } else {
    const Fr k = from_mont(((const Fr*)q)[i]);
    typename O::Pt r = O::inf();
    for (int w = 7; w >= 0; w--)
      for (int b = 31; b >= 0; b--) {
        r = O::dbl(r);
        if ((k.v[w] >> b) & 1) O::add(r, acc);
      }
    acc = r;
  }
  O::store_std(out + i, acc);
}

// op 7: P + (2q - q), op 8: 2 P + 2 (2q - q), through the spread form of the sums that end a multiexp (MsmOps::own_add): SL lanes per point,
// each holding its part of both operands.
template <class F>
__global__ __launch_bounds__(64) void k_group_op_spread(int op, const Affine<F>* p, const Affine<F>* q, XYZZ<F>* out, uint32_t n) {
  using O = MsmOps<F>;
  const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) / O::SL;
  if (i >= n) return;  // (whole groups leave)
  typename O::Pt acc = O::inf(), t = O::inf();
  typename O::Aff a;
  if (O::load_base(p + i, false, &a)) O::madd(acc, a);
  if (O::load_base(q + i, false, &a)) {
    O::madd(t, a);
    t = O::dbl(t);
    O::load_base(q + i, true, &a);
    O::madd(t, a);
  }
  typename O::Own x = O::own_of(acc);
  O::own_add(x, O::own_of(t));
  if (op == 8) {  // (p + q) + q, then + p: sums of the spread form as operands
    O::own_add(x, O::own_of(t));
    typename O::Own y = O::own_of(acc);
    O::own_add(y, x);
    x = y;
  }
  O::own_store_std(out + i, x);
}

// The two sides of the G2 subgroup test in the lane-pair arithmetic: op 4: [6x^2] P, op 5: psi(P) (q = the two psi
// constants as one affine "point" (psi_x, psi_y) in the packed base form).
__global__ __launch_bounds__(64) void k_g2_endo_msm(int op, const Affine<Fq2>* p, const Affine<Fq2>* consts, XYZZ<Fq2>* out, uint32_t n) {
  using O = MsmOps<Fq2>;
  const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) / 2;
  if (i >= n) return;
  O::Aff a, k;
  O::Pt acc = O::inf();
  if (O::load_base(p + i, false, &a)) {
    if (op == 4) {
      acc = g2_times_6x2_pair(a);
    } else {
      O::load_base(consts, false, &k);
      const O::Aff q = g2_psi_pair(a, k.x, k.y);
      acc = {q.x, q.y, pair_one<Fq29Params>(), pair_one<Fq29Params>()};
      // the comparison the key decoder makes, exercised on the way: psi(P) must equal itself and differ from P
      if (!pair_equals_affine(acc, q) || pair_equals_affine(acc, a)) acc = O::inf();
    }
  }
  O::store_std(out + i, acc);
}

// keyfile.hip: the subgroup test as the key-file decoder launches it (points in the ABI form)
__global__ void k_g2_subgroup(const Affine<Fq2>* __restrict__ pts, uint32_t n, const Affine<Fq2>* __restrict__ psi,
                              uint32_t* __restrict__ err);

// op 6: out[0] of every point = 1 when the decoder's kernel accepts the point alone, else 0 (one launch per point)
static zg16_status run_g2_subgroup_kernel(zg16_ctx* ctx, const uint64_t* p, size_t n, uint64_t* out) {
  char* buf;
  ZG_TRY(scratch_get(ctx, 5, n * sizeof(Affine<Fq2>) + 512 + 4 * n, (void**)&buf));
  Affine<Fq2>* psi = (Affine<Fq2>*)(buf + ((n * sizeof(Affine<Fq2>) + 255) & ~(size_t)255));
  uint32_t* err = (uint32_t*)(psi + 2);
  ZG_HIP(ctx, hipMemcpyAsync(buf, p, n * sizeof(Affine<Fq2>), hipMemcpyHostToDevice, ctx->stream));
  ZG_HIP(ctx, hipMemsetAsync(err, 0, 4 * n, ctx->stream));
  static const Affine<Fq2> k = MsmOps<Fq2>::to_internal(Affine<Fq2>{g2_subgroup_consts().psi_x, g2_subgroup_consts().psi_y});
  ZG_HIP(ctx, hipMemcpyAsync(psi, &k, sizeof k, hipMemcpyHostToDevice, ctx->stream));
  for (size_t i = 0; i < n; i++)
    k_g2_subgroup<<<1, 256, 0, ctx->stream>>>((const Affine<Fq2>*)buf + i, 1, psi, err + i);
  ZG_HIP(ctx, hipGetLastError());
  std::vector<uint32_t> flags(n);
  ZG_HIP(ctx, hipMemcpyAsync(flags.data(), err, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
  ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memset(out, 0, n * sizeof(Affine<Fq2>));
  for (size_t i = 0; i < n; i++) out[16 * i] = flags[i] ? 0 : 1;
  return ZG16_OK;
}

template <class F>
static zg16_status run_group_msm(zg16_ctx* ctx, int op, const uint64_t* p, const uint64_t* q, size_t n, uint64_t* out) {
  char* buf;
  const size_t pb = n * sizeof(Affine<F>), qb = op == 3 ? n * sizeof(Fr) : pb, ob = n * sizeof(XYZZ<F>);
  const size_t pa = (pb + 255) & ~(size_t)255, qa = (qb + 255) & ~(size_t)255;
  ZG_TRY(scratch_get(ctx, 5, pa + qa + ob + 256, (void**)&buf));
  ZG_HIP(ctx, hipMemcpyAsync(buf, p, pb, hipMemcpyHostToDevice, ctx->stream));
  if ((op != 0 && op < 4) || op >= 7) ZG_HIP(ctx, hipMemcpyAsync(buf + pa, q, qb, hipMemcpyHostToDevice, ctx->stream));
  ZG_TRY(msm_build_table<F>(ctx, (Affine<F>*)buf, n, 0));
  if (op == 1 || op == 2 || op >= 7) ZG_TRY(msm_build_table<F>(ctx, (Affine<F>*)(buf + pa), n, 0));
  const unsigned lanes = (unsigned)n * MsmOps<F>::LP;
  if constexpr (sizeof(F) > sizeof(Fq)) {
    if (op >= 4 && op <= 6) {
      // the psi constants, as one packed "point"
      Affine<Fq2> k = {g2_subgroup_consts().psi_x, g2_subgroup_consts().psi_y};
      k = MsmOps<Fq2>::to_internal(k);
      ZG_HIP(ctx, hipMemcpyAsync(buf + pa, &k, sizeof k, hipMemcpyHostToDevice, ctx->stream));
      ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (k is a local)
      k_g2_endo_msm<<<(lanes + 63) / 64, 64, 0, ctx->stream>>>(op, (const Affine<Fq2>*)buf, (const Affine<Fq2>*)(buf + pa),
                                                              (XYZZ<Fq2>*)(buf + pa + qa), (uint32_t)n);
    }
  }
  if (op >= 7) {
    const unsigned spread_lanes = (unsigned)n * MsmOps<F>::SL;
    k_group_op_spread<F><<<(spread_lanes + 63) / 64, 64, 0, ctx->stream>>>(op, (const Affine<F>*)buf, (const Affine<F>*)(buf + pa),
                                                                         (XYZZ<F>*)(buf + pa + qa), (uint32_t)n);
  }
  if (op < 4)
    k_group_op_msm<F><<<(lanes + 63) / 64, 64, 0, ctx->stream>>>(op, (const Affine<F>*)buf, buf + pa,
                                                               (XYZZ<F>*)(buf + pa + qa), (uint32_t)n);
  ZG_HIP(ctx, hipGetLastError());
  std::vector<XYZZ<F>> host(n);
  ZG_HIP(ctx, hipMemcpyAsync(host.data(), buf + pa + qa, ob, hipMemcpyDeviceToHost, ctx->stream));
  ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < n; i++) {
    const Affine<F> r = to_affine(host[i]);
    memcpy((char*)out + i * sizeof r, &r, sizeof r);
  }
  return ZG16_OK;
}

template <class T>
static zg16_status run_field(zg16_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  char* buf;
  const size_t bytes = n * sizeof(T);
  ZG_TRY(scratch_get(ctx, 5, 3 * bytes + 768, (void**)&buf));
  T *da = (T*)buf, *db = (T*)(buf + ((bytes + 255) & ~(size_t)255)), *dout = (T*)(buf + 2 * ((bytes + 255) & ~(size_t)255));
  ZG_HIP(ctx, hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, ctx->stream));
  ZG_HIP(ctx, hipMemcpyAsync(db, b ? b : a, bytes, hipMemcpyHostToDevice, ctx->stream));
  k_field_op<T><<<(unsigned)((n + 63) / 64), 64, 0, ctx->stream>>>(op, da, db, dout, (uint32_t)n);
  ZG_HIP(ctx, hipGetLastError());
  ZG_HIP(ctx, hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZG16_OK;
}

template <class F>
static zg16_status run_group(zg16_ctx* ctx, int op, const uint64_t* p, const uint64_t* q, size_t n, uint64_t* out) {
  char* buf;
  const size_t pb = n * sizeof(Affine<F>), qb = op == 3 ? n * sizeof(Fr) : pb;
  const size_t pa = (pb + 255) & ~(size_t)255, qa = (qb + 255) & ~(size_t)255;
  ZG_TRY(scratch_get(ctx, 5, 2 * pa + qa + 256, (void**)&buf));
  ZG_HIP(ctx, hipMemcpyAsync(buf, p, pb, hipMemcpyHostToDevice, ctx->stream));
  if (op != 0) ZG_HIP(ctx, hipMemcpyAsync(buf + pa, q, qb, hipMemcpyHostToDevice, ctx->stream));
  k_group_op<F><<<(unsigned)((n + 63) / 64), 64, 0, ctx->stream>>>(op, (const Affine<F>*)buf, buf + pa,
                                                                  (Affine<F>*)(buf + pa + qa), (uint32_t)n);
  ZG_HIP(ctx, hipGetLastError());
  ZG_HIP(ctx, hipMemcpyAsync(out, buf + pa + qa, pb, hipMemcpyDeviceToHost, ctx->stream));
  ZG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZG16_OK;
}

}  // namespace zg

extern "C" {

zg16_status zg16_selftest_field(zg16_ctx* ctx, int field, int op, const uint64_t* a, const uint64_t* b, size_t n,
                                uint64_t* out) {
  if (!ctx || !a || !out || !n || op < 0 || op > 5 || n > (1u << 24)) return zg::fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad selftest arguments");
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  if (field == 0) return zg::run_field<zg::Fr>(ctx, op, a, b, n, out);
  if (field == 1) return zg::run_field<zg::Fq>(ctx, op, a, b, n, out);
  if (field == 2) return zg::run_field<zg::Fq2>(ctx, op, a, b, n, out);
  return zg::fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "unknown field %d", field);
}

zg16_status zg16_selftest_group(zg16_ctx* ctx, int group, int op, const uint64_t* p, const uint64_t* q, size_t n,
                                uint64_t* out) {
  const bool endo = group == 4 && (op >= 4 && op <= 6);  // the G2 subgroup test: its two sides, its kernel
  const bool spread = (group == 3 || group == 4) && (op == 7 || op == 8);  // the addition of the sums that end a multiexp
  if (!ctx || !p || !out || !n || op < 0 || (op > 3 && !endo && !spread) || (op && !endo && !q) || n > (1u << 24))
    return zg::fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad selftest arguments");
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  if (group == 1) return zg::run_group<zg::Fq>(ctx, op, p, q, n, out);
  if (group == 2) return zg::run_group<zg::Fq2>(ctx, op, p, q, n, out);
  if (group == 3) return zg::run_group_msm<zg::Fq>(ctx, op, p, q, n, out);
  if (group == 4 && op == 6) return zg::run_g2_subgroup_kernel(ctx, p, n, out);
  if (group == 4) return zg::run_group_msm<zg::Fq2>(ctx, op, p, q, n, out);
  return zg::fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "unknown group %d", group);
}
}
