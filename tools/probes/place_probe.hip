// Where do the workgroups of a 512-workgroup launch (256 threads, 64 KB LDS, 2 per CU) land, and when do they start?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned* out, int spin_ticks) {
  extern __shared__ unsigned char lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)spin_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    out[4 * blockIdx.x] = hw;
    out[4 * blockIdx.x + 1] = xcc;
    out[4 * blockIdx.x + 2] = (unsigned)t0;
    out[4 * blockIdx.x + 3] = lds[5];
  }
}
int main() {
  const int n = 512;
  unsigned* d;
  hipMalloc(&d, n * 16);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(n), dim3(256), 65536, 0, d, 2000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(n * 4);
  hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> by;
  unsigned tmin = ~0u;
  for (int i = 0; i < n; ++i) tmin = h[4 * i + 2] < tmin ? h[4 * i + 2] : tmin;
  for (int i = 0; i < n; ++i) by[((h[4 * i + 1] & 15) << 8) | ((h[4 * i] >> 8) & 0xff)].push_back(i);
  printf("distinct keys %zu\n", by.size());
  int shown = 0, hist[8] = {0};
  for (auto& kv : by) {
    hist[kv.second.size() < 7 ? kv.second.size() : 7]++;
    if (shown++ < 12) {
      printf("key %03x:", kv.first);
      for (int i : kv.second) printf(" bid %d (hw %08x tg %u wave %u simd %u t+%u)", i, h[4 * i], (h[4 * i] >> 16) & 15, h[4 * i] & 15, (h[4 * i] >> 4) & 3, h[4 * i + 2] - tmin);
      printf("\n");
    }
  }
  printf("workgroups per key histogram:");
  for (int i = 0; i < 8; ++i) printf(" %d:%d", i, hist[i]);
  printf("\n");
  return 0;
}
