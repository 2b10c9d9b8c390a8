// LDS canary: a workgroup fills `bytes` of dynamic LDS with a pattern, idles for `spin` iterations and counts the words that changed.
// Run next to another kernel on a second stream: a non-zero count means somebody else wrote into this workgroup's LDS.
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ void lds_canary_kernel(int words, int spin, unsigned* bad, unsigned* first) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0xC0DE0000u ^ (unsigned)i ^ (blockIdx.x << 20);
  __syncthreads();
  unsigned n = 0;
  for (int it = 0; it < spin; ++it) {
    for (int i = threadIdx.x; i < words; i += blockDim.x) {
      const unsigned v = lds[i], w = 0xC0DE0000u ^ (unsigned)i ^ (blockIdx.x << 20);
      if (v != w) { ++n; if (atomicAdd(bad, 1u) == 0) { first[0] = (unsigned)i; first[1] = v; first[2] = w; first[3] = blockIdx.x; } lds[i] = w; }
    }
    __builtin_amdgcn_s_sleep(20);
  }
}
extern "C" int lds_canary_launch(int blocks, int threads, int bytes, int spin, unsigned* d_bad, unsigned* d_first, void* stream) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(lds_canary_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(threads), bytes, static_cast<hipStream_t>(stream), bytes / 4, spin, d_bad, d_first);
  return (int)hipGetLastError();
}
