// Stand-alone attempt to reproduce profiles/r06_concurrency.txt outside the library (no torch, no libvasr):
//   victim   : workgroups of 512 threads doing packed-FP32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) or FP64 arithmetic on values
//              exchanged through LDS (the shape of the STFT kernel's butterflies), result = a checksum per thread
//   attacker : small workgroups (128 threads, 24 KB of LDS) issuing v_mfma_f32_32x32x16_f16 back to back
// Both are launched over and over on two streams; the victim's checksums are compared with the ones it produced on the idle device.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/pk_mfma_repro.hip -o /tmp/pk_mfma_repro && /tmp/pk_mfma_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using v2f = __attribute__((ext_vector_type(2))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MODE>   // 0: packed FP32, 1: plain FP32 (same arithmetic, one lane of the pair at a time), 2: FP64
__global__ __launch_bounds__(512, 4) void victim(float* out, int iters) {
  __shared__ float lds[512 * 2 + 64];
  const int tid = threadIdx.x;
  v2f z = {1.0f + 0.001f * tid, 0.5f - 0.0007f * tid};
  double zd = 1.0 + 0.001 * tid;
  const v2f w = {0.99991f, 0.01342f};
  for (int it = 0; it < iters; ++it) {
    lds[2 * tid] = z.x; lds[2 * tid + 1] = z.y;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int p = (tid * 7 + 3 * it + 1) & 511;
    v2f o = {lds[2 * p], lds[2 * p + 1]};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (MODE == 0) {
      // complex-style butterfly on the pair, as packed operations
      v2f t = z * w;                                              // v_pk_mul_f32
      t = __builtin_elementwise_fma(o, v2f{w.y, -w.x}, t);        // v_pk_fma_f32
      z = (z + o) * v2f{0.5f, 0.5f} + t * v2f{0.25f, 0.25f};      // v_pk_add_f32 / v_pk_fma_f32
    } else if (MODE == 1) {
      float tx = z.x * w.x, ty = z.y * w.y;
      tx = __builtin_fmaf(o.x, w.y, tx); ty = __builtin_fmaf(o.y, -w.x, ty);
      asm volatile("" : "+v"(tx), "+v"(ty));
      z.x = (z.x + o.x) * 0.5f + tx * 0.25f; z.y = (z.y + o.y) * 0.5f + ty * 0.25f;
    } else {
      zd = zd * 0.99991 + (double)o.x * 0.01342;
      zd = (zd + (double)o.y) * 0.5;
      z.x = (float)zd; z.y = (float)(zd * 0.5);
    }
  }
  out[(size_t)blockIdx.x * 512 + tid] = z.x + z.y;
}

__global__ __launch_bounds__(128, 2) void attacker(float* sink, int iters) {
  extern __shared__ unsigned char smem[];
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x + i)); b[i] = (_Float16)(0.02f * (i + 1)); }
  f32x16 acc0 = {}, acc1 = {};
  for (int it = 0; it < iters; ++it) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  if (s == 12345.678f) sink[0] = s + smem[threadIdx.x];
}

template <int MODE>
int run(const char* name, bool attack, int launches) {
  const int blocks = 2048, iters = 600;
  float *d_out, *d_sink;
  hipMalloc(&d_out, (size_t)blocks * 512 * 4); hipMalloc(&d_sink, 4);
  std::vector<float> want((size_t)blocks * 512), got(want.size());
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipLaunchKernelGGL(victim<MODE>, dim3(blocks), dim3(512), 0, sa, d_out, iters);
  hipStreamSynchronize(sa);
  hipMemcpy(want.data(), d_out, want.size() * 4, hipMemcpyDeviceToHost);
  int bad_launches = 0; long bad_values = 0;
  for (int l = 0; l < launches; ++l) {
    if (attack) for (int k = 0; k < 40; ++k) hipLaunchKernelGGL(attacker, dim3(2048), dim3(128), 24 * 1024, sb, d_sink, 600);
    hipLaunchKernelGGL(victim<MODE>, dim3(blocks), dim3(512), 0, sa, d_out, iters);
    hipStreamSynchronize(sa);
    hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost);
    long n = 0;
    for (size_t i = 0; i < got.size(); ++i) n += memcmp(&got[i], &want[i], 4) != 0;
    bad_launches += n != 0; bad_values += n;
  }
  hipStreamSynchronize(sb);
  printf("%-28s %-22s: %d launches, %d with a wrong value (%ld values)\n", name, attack ? "next to f16 MFMA kernel" : "idle device", launches, bad_launches, bad_values);
  hipFree(d_out); hipFree(d_sink); hipStreamDestroy(sa); hipStreamDestroy(sb);
  return bad_launches;
}

int main() {
  run<0>("victim: packed FP32", false, 200);
  run<0>("victim: packed FP32", true, 400);
  run<1>("victim: plain FP32", true, 400);
  run<2>("victim: FP64", false, 200);
  run<2>("victim: FP64", true, 400);
  return 0;
}
