// Developer probe: why a shuffle fold with the group addition under `if (lane < d)` is slow.  Times a few ways of
// writing the same loop on 64 waves.   hipcc -O3 --offload-arch=gfx950 -I zinc_amd/csrc -I include tools/fold_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "curve29.hpp"
using namespace zg;

template <int V>
__global__ void k_probe(G1XYZZ29* io, int levels) {
  const int lt = threadIdx.x & 63;
  G1XYZZ29 acc = io[threadIdx.x], other = io[threadIdx.x + 256];
  if (V == 0) {  // the addition under the changing mask
#pragma unroll 1
    for (int d = 1 << (levels - 1); d >= 1; d >>= 1)
      if (lt < d) add_full29(acc, other);
  } else if (V == 1) {  // every lane adds, the result is kept under the mask
#pragma unroll 1
    for (int d = 1 << (levels - 1); d >= 1; d >>= 1) {
      G1XYZZ29 x = acc;
      add_full29(x, other);
      if (lt < d) acc = x;
    }
  } else if (V == 2) {  // fixed mask
#pragma unroll 1
    for (int d = 1 << (levels - 1); d >= 1; d >>= 1)
      if (lt < 8) add_full29(acc, other);
  } else if (V == 3) {  // changing mask, but growing
#pragma unroll 1
    for (int d = 1; d < (1 << levels); d <<= 1)
      if (lt < d) add_full29(acc, other);
  }
  io[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  G1XYZZ29* d;
  const int n = 16 * 256 + 512;
  std::vector<G1XYZZ29> h(n);
  // distinct multiples of a point are not needed for timing: any non-exceptional operands do
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 9; k++) {
      h[i].x.v[k] = (i * 977 + k * 13 + 5) & 0x0fffffff; h[i].y.v[k] = (i * 131 + k * 7 + 3) & 0x0fffffff;
      h[i].zz.v[k] = (i * 17 + k + 1) & 0x0fffffff; h[i].zzz.v[k] = (i * 19 + k + 2) & 0x0fffffff;
    }
  (void)hipMalloc(&d, n * sizeof(G1XYZZ29));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](auto kern, const char* what) {
    for (int levels : {4, 6}) {
      float best = 1e9f;
      for (int r = 0; r < 4; r++) {
        (void)hipMemcpy(d, h.data(), n * sizeof(G1XYZZ29), hipMemcpyHostToDevice);
        (void)hipEventRecord(e0); kern<<<16, 256>>>(d, levels); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("%-40s %d levels: %7.2f us\n", what, levels, best * 1e3);
    }
  };
  run(k_probe<0>, "addition under the shrinking mask");
  run(k_probe<1>, "every lane adds, masked keep");
  run(k_probe<2>, "fixed mask of 8 lanes");
  run(k_probe<3>, "growing mask");
  return 0;
}
