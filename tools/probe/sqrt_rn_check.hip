// Brute-force check of gs_sqrt_rn / gs_div_count (csrc/common.hpp) against hipcc's own sqrtf() / f32 division:
// every float of four binades around PyG's variance threshold plus 2^26 random ones in [1e-5, 1e6]; counts for
// edge counts 1..31.  Build: hipcc --offload-arch=gfx950 -O3 -I include -I gnn-epc-saft_amd/csrc tools/probe/sqrt_rn_check.hip -o /tmp/sqrt_rn_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "common.hpp"

__global__ void k_check(uint32_t first, uint32_t count, unsigned long long *bad_sqrt, unsigned long long *bad_div) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float x = __uint_as_float(first + i);
  if (gs::gs_sqrt_rn(x) != sqrtf(x)) atomicAdd(bad_sqrt, 1ull);
  for (int c = 1; c < 32; ++c) {
    const float fc = (float)c, inv = 1.f / fc;
    if (gs::gs_div_count(x, fc, inv) != x / fc) atomicAdd(bad_div, 1ull);
    if (gs::gs_div_count(-x, fc, inv) != -x / fc) atomicAdd(bad_div, 1ull);
  }
}

int main() {
  unsigned long long *d, h[2] = {0, 0};
  hipMalloc(&d, 16);
  hipMemset(d, 0, 16);
  // binades from 2^-17 (7.6e-6) to 2^20 (1e6): 37 binades x 2^23 floats, in launches of 2^24
  const uint32_t lo = 0x37000000u, hi = 0x49800000u;
  unsigned long long total = 0;
  for (uint64_t f = lo; f < hi; f += (1u << 24)) {
    const uint32_t n = (uint32_t)((hi - f) < (1u << 24) ? (hi - f) : (1u << 24));
    hipLaunchKernelGGL(k_check, dim3((n + 255) / 256), dim3(256), 0, 0, (uint32_t)f, n, d, d + 1);
    total += n;
  }
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("checked %llu floats in [%.3g, %.3g): gs_sqrt_rn != sqrtf: %llu; gs_div_count != '/' (counts 1..31, both signs): %llu\n",
         total, 7.6e-6, 1.05e6, h[0], h[1]);
  return (h[0] || h[1]) ? 1 : 0;
}
