// Developer tool: throughput and correctness of the 29-bit-limb field / curve code against the
// saturated 32-bit-limb code.  Not part of the shipped library.
//   hipcc -O3 --offload-arch=gfx950 -I zinc_amd/csrc tools/microbench29.hip -o bin/microbench29
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "curve29.hpp"

using namespace zg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_mul29_chain(Fq* io, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq29 x = unpack29<Fq29Params>(io[t]), y = x;
  for (int i = 0; i < iters; i++) x = mul(x, y);
  io[t] = pack29<Fq29Params, FqParams>(x);
}

__global__ void k_mul29_check(const Fq* a, const Fq* b, Fq* out, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  // inputs and output in radix-2^256 Montgomery form
  Fq29 x = from_r256<Fq29Params>(a[t]), y = from_r256<Fq29Params>(b[t]);
  out[t] = to_r256<Fq29Params, FqParams>(t & 1 ? sqr(add_lazy(x, Fq29::zero())) : mul(x, y));
}

__global__ void k_madd29_chain(XYZZ<Fq>* out, const Affine<Fq>* pts261, int npts, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  G1XYZZ29 a = G1XYZZ29::inf();
  for (int i = 0; i < iters; i++) add_mixed29(a, unpack_affine<Fq29Params>(pts261[(t + i * 977) % npts]));
  out[t] = export_xyzz<Fq29Params, FqParams>(a);
}

// a lone wave walking a chain of dependent full additions, as the bucket-fold kernels do
__global__ void k_add_full_chain(XYZZ<Fq>* out, const Affine<Fq>* pts261, int npts, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  G1XYZZ29 a = G1XYZZ29::inf(), b = G1XYZZ29::inf();
  add_mixed29(a, unpack_affine<Fq29Params>(pts261[t % npts]));
  add_mixed29(b, unpack_affine<Fq29Params>(pts261[(t + 7) % npts]));
  add_mixed29(b, unpack_affine<Fq29Params>(pts261[(t + 9) % npts]));
  for (int i = 0; i < iters; i++) add_full29(a, b);
  out[t] = export_xyzz<Fq29Params, FqParams>(a);
}


// the shuffle fold of the bucket-fold kernels in isolation: every lane starts from one point, `levels` levels of
// shuffle + full addition under `if (lane < d)`.  mode 0: as the kernels do it; 1: the shuffles only; 2: the additions
// only (the other operand stays in the lane); 3: the addition for every lane (no branch)
__global__ void k_fold(XYZZ<Fq>* out, const Affine<Fq>* pts261, int npts, int levels, int mode) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, lt = threadIdx.x & 63;
  G1XYZZ29 acc = G1XYZZ29::inf(), other = G1XYZZ29::inf();
  add_mixed29(acc, unpack_affine<Fq29Params>(pts261[t % npts]));
  add_mixed29(other, unpack_affine<Fq29Params>(pts261[(t * 7 + 3) % npts]));
#pragma unroll 1
  for (int d = 1 << (levels - 1); d >= 1; d >>= 1) {
    G1XYZZ29 o = other;
    if (mode != 2) {
      const uint32_t* src = (const uint32_t*)&acc;
      uint32_t* dst = (uint32_t*)&o;
#pragma unroll
      for (int i = 0; i < (int)(sizeof(G1XYZZ29) / 4); i++) dst[i] = __shfl_down(src[i], d);
    }
    if (mode == 1) {
      acc.x.v[0] ^= o.x.v[0] & 1;
    } else if (mode == 3) {
      add_full29(acc, other);
      acc.x.v[0] ^= o.x.v[0] & 0;
    } else if (mode == 4) {   // every lane adds; lanes outside the fold add the identity
      if (lt >= d) o = G1XYZZ29::inf();
      add_full29(acc, o);
    } else if (mode == 5) {   // every lane adds what the shuffle gave it (its own point at the wave's end: a doubling)
      add_full29(acc, o);
    } else if (mode == 6) {   // lanes outside the fold add into a scratch copy
      G1XYZZ29 tmp = acc;
      add_full29(tmp, lt < d ? o : other);
      if (lt < d) acc = tmp;
    } else if (lt < d) {
      add_full29(acc, o);
    }
  }
  out[t] = export_xyzz<Fq29Params, FqParams>(acc);
}

// a lone wave running dependent full additions with only the first `active` lanes switched on
__global__ void k_add_active(XYZZ<Fq>* out, const Affine<Fq>* pts261, int npts, int iters, int active) {
  const int t = threadIdx.x;
  G1XYZZ29 a = G1XYZZ29::inf(), b = G1XYZZ29::inf();
  add_mixed29(a, unpack_affine<Fq29Params>(pts261[t % npts]));
  add_mixed29(b, unpack_affine<Fq29Params>(pts261[(t + 7) % npts]));
  add_mixed29(b, unpack_affine<Fq29Params>(pts261[(t + 9) % npts]));
  if (t < active)
    for (int i = 0; i < iters; i++) add_full29(a, b);
  out[t] = export_xyzz<Fq29Params, FqParams>(a);
}

__global__ void k_madd_chain(XYZZ<Fq>* acc, const Affine<Fq>* pts, int npts, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  XYZZ<Fq> a = XYZZ<Fq>::inf();
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

static Fq host_rand(uint64_t& s) {
  Fq r;
  for (int i = 0; i < 8; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; r.v[i] = (uint32_t)(s >> 32); }
  r.v[7] &= 0x0fffffff;
  return r;
}

int main() {
  const int blocks = 256 * 4, threads = 256;
  const double nthr = (double)blocks * threads;
  const int n = blocks * threads;
  // field products
  {
    std::vector<Fq> a(n), b(n), ref(n), out(n);
    uint64_t seed = 7;
    for (int i = 0; i < n; i++) { a[i] = host_rand(seed); b[i] = host_rand(seed); }
    // edge operands
    a[0] = Fq::zero(); a[2] = Fq::one(); b[4] = Fq::zero();
    for (int i = 0; i < 8; i++) a[6].v[i] = FqParams::mod(i);
    a[6].v[0] -= 1;  // p - 1
    a[7] = a[6];
    for (int i = 0; i < 4096; i++) ref[i] = (i & 1) ? mul(a[i], a[i]) : mul(a[i], b[i]);
    Fq *da, *db, *dout;
    CK(hipMalloc(&da, n * sizeof(Fq))); CK(hipMalloc(&db, n * sizeof(Fq))); CK(hipMalloc(&dout, n * sizeof(Fq)));
    CK(hipMemcpy(da, a.data(), n * sizeof(Fq), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), n * sizeof(Fq), hipMemcpyHostToDevice));
    k_mul29_check<<<blocks, threads>>>(da, db, dout, n);
    CK(hipMemcpy(out.data(), dout, n * sizeof(Fq), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 4096; i++) bad += !(out[i] == ref[i]);
    printf("Fq29 mul/sqr mismatches vs host (4096 cases): %d\n", bad);
    const int iters = 2000;
    float ms = time_kernel([&] { k_mul29_chain<<<blocks, threads>>>(da, iters); });
    printf("Fq29 mul chain : %.2f Gmul/s\n", nthr * iters / ms / 1e6);
  }
  // mixed additions
  {
    const int npts = 4096;
    std::vector<G1Affine> pts(npts), pts261(npts);
    G1Affine g = {from_u64<FqParams>(1), from_u64<FqParams>(2)};
    G1XYZZ acc = G1XYZZ::from_affine(g);
    for (int i = 0; i < npts; i++) {
      pts[i] = to_affine(acc);
      add_mixed(acc, g);
      pts261[i] = pts[i];
      for (int k = 0; k < 5; k++) { pts261[i].x = dbl(pts261[i].x); pts261[i].y = dbl(pts261[i].y); }
    }
    // make the chain hit the exceptional cases: equal and opposite neighbours
    pts[977 % npts] = pts[0]; pts261[977 % npts] = pts261[0];                  // thread 0: P + P
    pts[(1 + 977) % npts] = neg(pts[1]);                                        // thread 1: P - P
    pts261[(1 + 977) % npts] = pts261[1];
    pts261[(1 + 977) % npts].y = neg(pts261[1].y);
    G1Affine *dp, *dp261; G1XYZZ *d0, *d1;
    CK(hipMalloc(&dp, npts * sizeof(G1Affine))); CK(hipMalloc(&dp261, npts * sizeof(G1Affine)));
    CK(hipMalloc(&d0, n * sizeof(G1XYZZ))); CK(hipMalloc(&d1, n * sizeof(G1XYZZ)));
    CK(hipMemcpy(dp, pts.data(), npts * sizeof(G1Affine), hipMemcpyHostToDevice));
    CK(hipMemcpy(dp261, pts261.data(), npts * sizeof(G1Affine), hipMemcpyHostToDevice));
    k_madd29_chain<<<blocks, threads>>>(d1, dp261, npts, 40);
    std::vector<G1XYZZ> out(64);
    CK(hipMemcpy(out.data(), d1, 64 * sizeof(G1XYZZ), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int t = 0; t < 64; t++) {
      G1XYZZ r = G1XYZZ::inf();
      for (int i = 0; i < 40; i++) add_mixed(r, pts[(t + i * 977) % npts]);
      G1Affine x = to_affine(r), y = to_affine(out[t]);
      bad += !(x.x == y.x && x.y == y.y);
    }
    printf("G1 madd29 device-vs-host mismatches (64 chains incl. P+P, P-P): %d\n", bad);
    float ms = time_kernel([&] { k_madd_chain<<<blocks, threads>>>(d0, dp, npts, 200); });
    printf("G1 mixed add (32-bit limbs) : %.3f Gadd/s\n", nthr * 200 / ms / 1e6);
    ms = time_kernel([&] { k_madd29_chain<<<blocks, threads>>>(d1, dp261, npts, 200); });
    printf("G1 mixed add (29-bit limbs) : %.3f Gadd/s\n", nthr * 200 / ms / 1e6);
  }
  {
    const int npts = 4096;
    G1Affine* dp261; G1XYZZ* d1;
    CK(hipMalloc(&dp261, npts * sizeof(G1Affine))); CK(hipMalloc(&d1, n * sizeof(G1XYZZ)));
    std::vector<G1Affine> pts261(npts);
    G1Affine g = {from_u64<FqParams>(1), from_u64<FqParams>(2)};
    G1XYZZ acc = G1XYZZ::from_affine(g);
    for (int i = 0; i < npts; i++) {
      pts261[i] = to_affine(acc);
      add_mixed(acc, g);
      for (int k = 0; k < 5; k++) { pts261[i].x = dbl(pts261[i].x); pts261[i].y = dbl(pts261[i].y); }
    }
    CK(hipMemcpy(dp261, pts261.data(), npts * sizeof(G1Affine), hipMemcpyHostToDevice));
    float ms = time_kernel([&] { k_add_full_chain<<<1, 64>>>(d1, dp261, npts, 200); });
    printf("lone wave, dependent full additions (hot code): %.2f us each\n", ms * 1e3 / 200);
    ms = time_kernel([&] { k_add_full_chain<<<1, 64>>>(d1, dp261, npts, 1); });
    printf("lone wave, kernel with ONE full addition + 3 mixed: %.2f us\n", ms * 1e3);
    for (int active : {64, 48, 32, 17, 16, 8, 4, 1}) {
      ms = time_kernel([&] { k_add_active<<<1, 64>>>(d1, dp261, npts, 200, active); });
      printf("lone wave, %2d lanes switched on: %.2f us per dependent full addition\n", active, ms * 1e3 / 200);
    }
    for (int active : {64, 8})
      for (int it : {1, 2, 4, 8}) {
        ms = time_kernel([&] { k_add_active<<<16, 256>>>(d1, dp261, npts, it, active); });
        printf("64 waves, %2d of every 256 threads switched on, %d additions: %.2f us\n", active, it, ms * 1e3);
      }
    for (int mode = 0; mode < 7; mode++)
      for (int levels : {4, 6}) {
        ms = time_kernel([&] { k_fold<<<16, 256>>>(d1, dp261, npts, levels, mode); });
        printf("fold: 64 waves, mode %d, %d levels: %.2f us\n", mode, levels, ms * 1e3);
      }
  }
  printf("done\n");
  return 0;
}
