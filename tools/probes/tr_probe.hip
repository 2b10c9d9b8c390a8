// ds_read_b64_tr_b16 semantics: every lane supplies its own 8-byte-aligned LDS address; what does lane l receive?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const int* addr, unsigned short* out) {
  __shared__ unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned a = (unsigned)(size_t)lds + (unsigned)addr[threadIdx.x];
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
  int* d_a; unsigned short* d_o;
  hipMalloc(&d_a, 256); hipMalloc(&d_o, 512);
  for (int test = 0; test < 3; ++test) {
    std::vector<int> a(64);
    for (int l = 0; l < 64; ++l) {
      const int i = l & 15, g = l >> 4, r = i >> 2, q = i & 3;
      if (test == 0) a[l] = 8 * l;                                   // natural: one contiguous 512-byte block
      if (test == 1) a[l] = 1024 * r + 2 * (16 * g + 4 * q);          // rows 1024 bytes apart, group g = columns 16 g .. 16 g + 15
      if (test == 2) a[l] = 1024 * (3 - r) + 2 * (16 * (3 - g) + 4 * (q ^ 1));   // a scrambled assignment
    }
    hipMemcpy(d_a, a.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_a, d_o);
    std::vector<unsigned short> o(256);
    hipMemcpy(o.data(), d_o, 512, hipMemcpyDeviceToHost);
    printf("test %d\n", test);
    for (int l = 0; l < 64; ++l) {
      printf("  l%02d addr %4d ->", l, a[l]);
      for (int j = 0; j < 4; ++j) printf(" %5d", o[l * 4 + j]);
      if (l % 2) printf("\n");
    }
  }
  return 0;
}
