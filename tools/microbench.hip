// Developer tool: VALU integer-multiply rates on gfx950 and throughput of the field / curve
// primitives.  Not part of the shipped library.
//   hipcc -O3 --offload-arch=gfx950 -I zinc_amd/csrc tools/microbench.hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "curve.hpp"

using namespace zg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int ILP>
__global__ void k_mad64(uint64_t* out, uint32_t a0, uint32_t b0, int iters) {
  uint64_t acc[ILP];
  uint32_t a = a0 + threadIdx.x, b = b0 + blockIdx.x;
  for (int k = 0; k < ILP; k++) acc[k] = k + threadIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < ILP; k++) acc[k] = (uint64_t)a * (uint32_t)(b + k) + acc[k];
    a += (uint32_t)acc[0];
  }
  uint64_t s = 0;
  for (int k = 0; k < ILP; k++) s ^= acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// v_mul_lo_u32 alone (the q = t * inv step of a Montgomery round), and a mix of 9 multiply-adds to 1 mul_lo as
// the field product has it
template <int ILP>
__global__ void k_mullo(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
  uint32_t lo[ILP];
  uint32_t a = a0 + threadIdx.x, b = b0 + blockIdx.x;
  for (int k = 0; k < ILP; k++) lo[k] = k + 1;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < ILP; k++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo[k]) : "v"(a + k));
    b += lo[0];
  }
  uint32_t s = b;
  for (int k = 0; k < ILP; k++) s ^= lo[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
__global__ void k_mullohi(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
  uint32_t lo[ILP], hi[ILP];
  uint32_t a = a0 + threadIdx.x, b = b0 + blockIdx.x;
  for (int k = 0; k < ILP; k++) { lo[k] = k; hi[k] = k; }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < ILP; k++) {
      lo[k] += a * (b + k);
      hi[k] += __umulhi(a, b + k);
    }
    a += lo[0];
  }
  uint32_t s = 0;
  for (int k = 0; k < ILP; k++) s ^= lo[k] ^ hi[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
__global__ void k_mad24(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
  uint32_t lo[ILP];
  uint32_t a = (a0 + threadIdx.x) & 0xffffff, b = (b0 + blockIdx.x) & 0xffff;
  for (int k = 0; k < ILP; k++) lo[k] = k;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < ILP; k++) lo[k] = __umul24(a, b + k) + lo[k];
    a = (a + lo[0]) & 0xffffff;
  }
  uint32_t s = 0;
  for (int k = 0; k < ILP; k++) s ^= lo[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
__global__ void k_fma64(double* out, double a0, double b0, int iters) {
  double acc[ILP];
  double a = a0 + threadIdx.x, b = b0;
  for (int k = 0; k < ILP; k++) acc[k] = k;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < ILP; k++) acc[k] = __builtin_fma(a, b, acc[k]);
  }
  double s = 0;
  for (int k = 0; k < ILP; k++) s += acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
__global__ void k_fma32(float* out, float a0, float b0, int iters) {
  float acc[ILP];
  float a = a0 + threadIdx.x, b = b0;
  for (int k = 0; k < ILP; k++) acc[k] = k;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < ILP; k++) acc[k] = __builtin_fmaf(a, b, acc[k]);
  }
  float s = 0;
  for (int k = 0; k < ILP; k++) s += acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
__global__ void k_mulchain(F* io, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  F x = io[t], y = x;
  for (int i = 0; i < iters; i++) x = mul(x, y);
  io[t] = x;
}

template <class F>
__global__ void k_mulchain2(F* io, int iters) {  // two independent chains per lane
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  F x = io[t], y = x, z = add(x, x);
  for (int i = 0; i < iters; i++) { x = mul(x, y); z = mul(z, y); }
  io[t] = add(x, z);
}

template <class F>
__global__ void k_addchain(F* io, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  F x = io[t], y = x;
  for (int i = 0; i < iters; i++) { x = add(x, y); y = sub(y, x); }
  io[t] = add(x, y);
}

template <class F>
__global__ void k_maddchain(XYZZ<F>* acc, const Affine<F>* pts, int npts, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  XYZZ<F> a = acc[t];
  for (int i = 0; i < iters; i++) add_mixed(a, pts[(t + i * 977) % npts]);
  acc[t] = a;
}

template <class K>
float time_kernel(K launch, int reps = 3) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

template <class F>
static F host_rand(uint64_t& s) {
  F r;
  for (int i = 0; i < 8; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; r.v[i] = (uint32_t)(s >> 32); }
  r.v[7] &= 0x0fffffff;
  return r;
}

int main() {
  const int blocks = 256 * 8, threads = 256, iters = 2000;
  const double nthr = (double)blocks * threads;
  void* buf;
  CK(hipMalloc(&buf, (size_t)blocks * threads * 256));
  CK(hipMemset(buf, 1, (size_t)blocks * threads * 256));
  {
    float ms = time_kernel([&] { k_mad64<8><<<blocks, threads>>>((uint64_t*)buf, 3, 5, iters); });
    printf("v_mad_u64_u32  : %.2f Gop/s\n", nthr * iters * 8 / ms / 1e6);
    ms = time_kernel([&] { k_mullo<8><<<blocks, threads>>>((uint32_t*)buf, 3, 5, iters); });
    printf("v_mul_lo_u32   : %.2f Gop/s\n", nthr * iters * 8 / ms / 1e6);
    ms = time_kernel([&] { k_mullohi<8><<<blocks, threads>>>((uint32_t*)buf, 3, 5, iters); });
    printf("mul_lo+mul_hi  : %.2f Gpair/s\n", nthr * iters * 8 / ms / 1e6);
    ms = time_kernel([&] { k_mad24<8><<<blocks, threads>>>((uint32_t*)buf, 3, 5, iters); });
    printf("mad_u32_u24    : %.2f Gop/s\n", nthr * iters * 8 / ms / 1e6);
    ms = time_kernel([&] { k_fma64<8><<<blocks, threads>>>((double*)buf, 1.0001, 0.999, iters); });
    printf("v_fma_f64      : %.2f Gop/s\n", nthr * iters * 8 / ms / 1e6);
    ms = time_kernel([&] { k_fma32<8><<<blocks, threads>>>((float*)buf, 1.0001f, 0.999f, iters); });
    printf("v_fma_f32      : %.2f Gop/s\n", nthr * iters * 8 / ms / 1e6);
  }
  // field mul correctness (device vs host, same header) and throughput
  {
    const int n = blocks * threads;
    std::vector<Fq> h(n), ref(64);
    uint64_t seed = 42;
    for (int i = 0; i < n; i++) h[i] = host_rand<Fq>(seed);
    for (int i = 0; i < 64; i++) { Fq x = h[i], y = x; for (int k = 0; k < 50; k++) x = mul(x, y); ref[i] = x; }
    CK(hipMemcpy(buf, h.data(), n * sizeof(Fq), hipMemcpyHostToDevice));
    k_mulchain<Fq><<<blocks, threads>>>((Fq*)buf, 50);
    std::vector<Fq> out(64);
    CK(hipMemcpy(out.data(), buf, 64 * sizeof(Fq), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 64; i++) bad += !(out[i] == ref[i]);
    printf("Fq mul device-vs-host mismatches: %d\n", bad);
    float ms = time_kernel([&] { k_mulchain<Fq><<<blocks, threads>>>((Fq*)buf, iters); });
    printf("Fq mul chain   : %.2f Gmul/s\n", nthr * iters / ms / 1e6);
    ms = time_kernel([&] { k_mulchain2<Fq><<<blocks, threads>>>((Fq*)buf, iters); });
    printf("Fq mul 2chains : %.2f Gmul/s\n", nthr * iters * 2 / ms / 1e6);
    ms = time_kernel([&] { k_addchain<Fq><<<blocks, threads>>>((Fq*)buf, iters); });
    printf("Fq add/sub     : %.2f Gop/s\n", nthr * iters * 2 / ms / 1e6);
  }
  {
    const int n = blocks * threads, npts = 4096;
    // points k*G on G1 built on the host
    std::vector<G1Affine> pts(npts);
    G1Affine g = {from_u64<FqParams>(1), from_u64<FqParams>(2)};
    G1XYZZ a = G1XYZZ::from_affine(g);
    for (int i = 0; i < npts; i++) { pts[i] = to_affine(a); add_mixed(a, g); }
    G1Affine* dp; G1XYZZ* dacc;
    CK(hipMalloc(&dp, npts * sizeof(G1Affine)));
    CK(hipMalloc(&dacc, n * sizeof(G1XYZZ)));
    CK(hipMemcpy(dp, pts.data(), npts * sizeof(G1Affine), hipMemcpyHostToDevice));
    CK(hipMemset(dacc, 0, n * sizeof(G1XYZZ)));
    // correctness: thread t sums pts[(t + i*977) % npts], i < 40
    k_maddchain<Fq><<<blocks, threads>>>(dacc, dp, npts, 40);
    std::vector<G1XYZZ> out(8);
    CK(hipMemcpy(out.data(), dacc, 8 * sizeof(G1XYZZ), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int t = 0; t < 8; t++) {
      G1XYZZ r = G1XYZZ::inf();
      for (int i = 0; i < 40; i++) add_mixed(r, pts[(t + i * 977) % npts]);
      G1Affine x = to_affine(r), y = to_affine(out[t]);
      bad += !(x.x == y.x && x.y == y.y);
    }
    printf("G1 madd device-vs-host mismatches: %d\n", bad);
    float ms = time_kernel([&] { k_maddchain<Fq><<<blocks, threads>>>(dacc, dp, npts, 200); });
    printf("G1 mixed add   : %.3f Gadd/s\n", nthr * 200 / ms / 1e6);
  }
  printf("done\n");
  return 0;
}
