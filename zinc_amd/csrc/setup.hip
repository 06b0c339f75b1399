// Fixed-base scalar multiplication: out[i] = scalars[i] * base, affine results.
//
// This is the group half of bellman_ce 0.3.0 `generate_parameters` (called by the reference at
// zinc-vm/src/core/circuit/facade.rs:126): every point of a proving key is a known scalar times
// one generator.  Used to build 2^20..2^24-row keys in seconds; not part of the prove path.
//
// Method: byte windows.  T[w][d-1] = d * 256^w * base for w < 32, d in 1..255 is built once per
// call (one lane per window), then every lane sums the 32 table entries its scalar selects with
// mixed additions and converts its own result to affine.
#include <string.h>

#define ZG_OUTLINE_FIELD_MUL 1
#include "common.cuh"

namespace zg {

static constexpr int FB_WINDOWS = 32, FB_DIGITS = 255;

template <class F>
__global__ void k_fb_table_rows(Affine<F> base, XYZZ<F>* table) {
  const int w = threadIdx.x;
  if (w >= FB_WINDOWS) return;
  XYZZ<F> b = XYZZ<F>::from_affine(base);
  for (int i = 0; i < 8 * w; i++) b = dbl(b);
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int d = 0; d < FB_DIGITS; d++) {
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

template <class F>
static zg16_status fixed_base(zg16_ctx* ctx, const uint64_t* base, const uint64_t* scalars, size_t n, uint64_t* out) {
  if (!ctx || !base || (n && (!scalars || !out))) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "bad arguments to fixed-base multiply");
  if (n >= 0xffffffffull) return fail(ctx, ZG16_ERR_INVALID_ARGUMENT, "too many scalars");
  if (!n) return ZG16_OK;
  ZG_HIP(ctx, hipSetDevice(ctx->device));
  Affine<F> b;
  memcpy(&b, base, sizeof b);
  const int tn = FB_WINDOWS * FB_DIGITS;
  char* buf;
  const size_t t_x = (size_t)tn * sizeof(XYZZ<F>), t_a = (size_t)tn * sizeof(Affine<F>);
  const size_t s_b = n * sizeof(Fr), o_b = n * sizeof(Affine<F>);
  ZG_TRY(scratch_get(ctx, 5, t_x + t_a + s_b + o_b + 1024, (void**)&buf));
  XYZZ<F>* tx = (XYZZ<F>*)buf;
  Affine<F>* ta = (Affine<F>*)(buf + t_x);
  Fr* ds = (Fr*)(buf + t_x + t_a);
  Affine<F>* dout = (Affine<F>*)(buf + t_x + t_a + ((s_b + 255) & ~(size_t)255));
  hipStream_t st = ctx->stream;
  ZG_HIP(ctx, hipMemcpyAsync(ds, scalars, s_b, hipMemcpyHostToDevice, st));
  k_fb_table_rows<F><<<1, 64, 0, st>>>(b, tx);
  k_fb_table_affine<F><<<(tn + 255) / 256, 256, 0, st>>>(tx, ta, tn);
  k_fb_mul<F><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(ta, ds, dout, (uint32_t)n);
  ZG_HIP(ctx, hipGetLastError());
  ZG_HIP(ctx, hipMemcpyAsync(out, dout, o_b, hipMemcpyDeviceToHost, st));
  ZG_HIP(ctx, hipStreamSynchronize(st));
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
}
