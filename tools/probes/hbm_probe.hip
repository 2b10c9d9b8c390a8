// HBM copy ceiling probe (dev tool, not part of the library): y = x over n float4, in the shapes the depthwise kernels use.
#include <hip/hip_runtime.h>
#include <stdint.h>
using v4f = __attribute__((ext_vector_type(4))) float;

// mode 0: one float4 per thread, plain; 1: nontemporal load+store; 2: nt store only; 3: nt load only
template <int MODE>
__global__ __launch_bounds__(256) void copy1(const v4f* __restrict__ x, v4f* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  v4f v = (MODE == 1 || MODE == 3) ? __builtin_nontemporal_load(x + i) : x[i];
  if (MODE == 1 || MODE == 2) __builtin_nontemporal_store(v, y + i); else y[i] = v;
}
// U float4 per thread, all loads first (U in flight), block-contiguous
template <int U, int MODE>
__global__ __launch_bounds__(256) void copyU(const v4f* __restrict__ x, v4f* __restrict__ y, int64_t n) {
  const int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
  v4f v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { const int64_t i = base + 256 * u; v[u] = i < n ? ((MODE & 1) ? __builtin_nontemporal_load(x + i) : x[i]) : v4f{0, 0, 0, 0}; }
#pragma unroll
  for (int u = 0; u < U; ++u) { const int64_t i = base + 256 * u; if (i < n) { if (MODE & 2) __builtin_nontemporal_store(v[u], y + i); else y[i] = v[u]; } }
}
// persistent grid-stride: G blocks, each loops
template <int U, int MODE>
__global__ __launch_bounds__(256) void copyP(const v4f* __restrict__ x, v4f* __restrict__ y, int64_t n) {
  for (int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += (int64_t)gridDim.x * 256 * U) {
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int64_t i = base + 256 * u; v[u] = i < n ? ((MODE & 1) ? __builtin_nontemporal_load(x + i) : x[i]) : v4f{0, 0, 0, 0}; }
#pragma unroll
    for (int u = 0; u < U; ++u) { const int64_t i = base + 256 * u; if (i < n) { if (MODE & 2) __builtin_nontemporal_store(v[u], y + i); else y[i] = v[u]; } }
  }
}
extern "C" int probe(int kind, int mode, const void* x, void* y, int64_t n4, int grid, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const v4f* xs = (const v4f*)x; v4f* ys = (v4f*)y;
#define L1(M) hipLaunchKernelGGL(copy1<M>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, xs, ys, n4)
#define LU(U, M) hipLaunchKernelGGL((copyU<U, M>), dim3((unsigned)((n4 + 256 * U - 1) / (256 * U))), dim3(256), 0, st, xs, ys, n4)
#define LP(U, M) hipLaunchKernelGGL((copyP<U, M>), dim3(grid), dim3(256), 0, st, xs, ys, n4)
  switch (kind * 10 + mode) {
    case 0: L1(0); break; case 1: L1(1); break; case 2: L1(2); break; case 3: L1(3); break;
    case 10: LU(4, 0); break; case 11: LU(4, 1); break; case 12: LU(4, 2); break; case 13: LU(4, 3); break;
    case 20: LU(8, 0); break; case 22: LU(8, 2); break; case 23: LU(8, 3); break;
    case 30: LP(4, 0); break; case 32: LP(4, 2); break; case 33: LP(4, 3); break;
    case 40: LP(8, 0); break; case 42: LP(8, 2); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
