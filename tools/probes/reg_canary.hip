// VGPR / SGPR canary: every lane keeps NR known values in vector registers (pinned with empty asm), idles, and checks them.
#include <hip/hip_runtime.h>
#include <cstdint>
template <int NR>
__global__ __launch_bounds__(512, 4) void reg_canary_kernel(int spin, unsigned* bad, unsigned* first) {
  unsigned r[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) { r[i] = 0xA5000000u ^ (threadIdx.x << 8) ^ i; asm volatile("" : "+v"(r[i])); }
  unsigned s0 = 0x5EED0000u ^ blockIdx.x;
  asm volatile("" : "+s"(s0));
  for (int it = 0; it < spin; ++it) {
    __builtin_amdgcn_s_sleep(30);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      asm volatile("" : "+v"(r[i]));
      const unsigned w = 0xA5000000u ^ (threadIdx.x << 8) ^ i;
      if (r[i] != w) { if (atomicAdd(bad, 1u) == 0) { first[0] = i; first[1] = r[i]; first[2] = w; first[3] = threadIdx.x; } r[i] = w; }
    }
    asm volatile("" : "+s"(s0));
    if (s0 != (0x5EED0000u ^ blockIdx.x)) { atomicAdd(bad + 1, 1u); s0 = 0x5EED0000u ^ blockIdx.x; }
  }
}
extern "C" int reg_canary_launch(int blocks, int spin, unsigned* d_bad, unsigned* d_first, void* stream) {
  hipLaunchKernelGGL(reg_canary_kernel<112>, dim3(blocks), dim3(512), 0, static_cast<hipStream_t>(stream), spin, d_bad, d_first);
  return (int)hipGetLastError();
}
