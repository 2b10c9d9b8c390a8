// A synthetic attacker for tests/devtools/stress_attack.py-style probes: small workgroups issuing 16-bit MFMAs back to back, with a chosen
// amount of dynamic LDS (0 = fits beside ANY workgroup that leaves registers free) and optional extras (flags): 1 = LDS writes + reads +
// a workgroup barrier per iteration, 2 = a global load per iteration, 4 = packed-FP32 conversions of the loaded values, 8 = global stores.
#include <hip/hip_runtime.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using v2f = __attribute__((ext_vector_type(2))) float;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
__global__ __launch_bounds__(128, 2) void mfma_attacker_kernel(float* sink, const float* src, int iters, int flags, int lds_words) {
  extern __shared__ float smem[];
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x + i)); b[i] = (_Float16)(0.02f * (i + 1)); }
  f32x16 acc0 = {}, acc1 = {};
  float carry = 0.f;
  for (int it = 0; it < iters; ++it) {
    if ((flags & 2) && src) {
      const v2f g = *reinterpret_cast<const v2f*>(src + ((blockIdx.x * 128 + threadIdx.x + 977 * it) & 0xFFFFF) * 2);
      if (flags & 4) {
        const v2f s = g * v2f{0.5f, 0.25f};
        const f16x2 h = __builtin_convertvector(s, f16x2);
        const v2f r = s - __builtin_convertvector(h, v2f);
        a[0] = h.x; a[1] = h.y; b[0] = (_Float16)r.x; b[1] = (_Float16)r.y;
      } else { a[0] = (_Float16)g.x; b[0] = (_Float16)g.y; }
    }
    if ((flags & 1) && lds_words >= 256) {
      smem[(threadIdx.x * 2 + it) % lds_words] = carry + (float)a[0];
      __syncthreads();
      carry = smem[(threadIdx.x * 5 + 3 * it + 1) % lds_words];
      __syncthreads();
      b[2] = (_Float16)carry;
    }
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
    if ((flags & 8) && (it & 15) == 0) sink[1024 + ((blockIdx.x * 128 + threadIdx.x) & 0xFFFF)] = acc0[0];
  }
  float s = carry;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  if (s == 12345.678f) sink[0] = s;
}
extern "C" int mfma_attacker_launch(int blocks, int lds_bytes, int iters, int flags, float* d_sink, const float* d_src, void* stream) {
  hipLaunchKernelGGL(mfma_attacker_kernel, dim3(blocks), dim3(128), lds_bytes, static_cast<hipStream_t>(stream), d_sink, d_src, iters, flags, lds_bytes / 4);
  return (int)hipGetLastError();
}
