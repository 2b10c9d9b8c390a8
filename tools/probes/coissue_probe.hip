// Do a matrix-pipe wavefront and a vector-ALU wavefront that share a SIMD overlap on gfx950?  (Round 5, the question under
// the fused depthwise + pointwise kernel for 512 channels: round 3 measured its producer and consumer roles as ADDITIVE
// inside the 256-channel kernel, profiles/r03_fused_ablations.txt.)
//
// One workgroup per CU, register-resident operands, no memory traffic.  Roles by wavefront index inside the workgroup:
//   wavefronts [0, n_mfma)            : `steps` x 48 v_mfma_f32_32x32x16_f16 on 8 independent accumulators
//   wavefronts [n_mfma, n_mfma+n_valu): `steps` x `fpm` x 48 vector FMAs on 8 (16 with plain FMAs) independent accumulators
//                                        mode 0: v_pk_fma_f32, mode 1: v_fma_f32, mode 2: v_pk_fma_f32 + one ds_read_b128 per 8
// Prints the time of each role alone and of both together; "overlap" = (alone_a + alone_b - both) / min(alone_a, alone_b).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using v2f = __attribute__((ext_vector_type(2))) float;
using v4f = __attribute__((ext_vector_type(4))) float;

template <int NT>
__global__ __launch_bounds__(NT, 1) void probe(int n_mfma, int n_valu, int steps, int fpm, int mode, float* sink) {
  __shared__ v4f lds[1024];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  lds[threadIdx.x & 1023] = v4f{1.f, 2.f, 3.f, (float)threadIdx.x};
  __syncthreads();
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 8388608.f) - 1.0f; };
  if (wave < n_mfma) {
    f16x8 a[2], b[4];
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (_Float16)(rnd() * 4.f);
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) b[j][e] = (_Float16)(rnd() * 4.f);
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < steps; ++it) {
#pragma unroll
      for (int rep = 0; rep < 6; ++rep)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 12345.678f) sink[0] = t;
  } else if (wave < n_mfma + n_valu) {
    v2f acc[8], x[8], w[4];
    for (int j = 0; j < 8; ++j) { acc[j] = v2f{0.f, 0.f}; x[j] = v2f{rnd(), rnd()}; }
    for (int k = 0; k < 4; ++k) w[k] = v2f{rnd() * 1e-3f, rnd() * 1e-3f};
    const int n = steps * fpm * 6;   // x 8 FMAs per trip
    if (mode == 1) {
      for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].x) : "v"(w[j & 3].x), "v"(x[j].x));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].y) : "v"(w[j & 3].y), "v"(x[j].y));
        }
      }
    } else {
      for (int it = 0; it < n; ++it) {
        if (mode == 2) {
          const v4f v = lds[(lane * 5 + it) & 1023];
          x[it & 7] = v.xy;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(w[j & 3]), "v"(x[j]));
      }
    }
    float t = 0.f;
    for (int j = 0; j < 8; ++j) t += acc[j].x + acc[j].y;
    if (t == 12345.678f) sink[1] = t;
  }
}

template <int NT>
float run(int n_mfma, int n_valu, int steps, int fpm, int mode, float* sink, int n_cu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(probe<NT>, dim3(n_cu), dim3(NT), 0, 0, n_mfma, n_valu, steps, fpm, mode, sink);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe<NT>, dim3(n_cu), dim3(NT), 0, 0, n_mfma, n_valu, steps, fpm, mode, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

int main() {
  int dev = 0, n_cu = 256;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
  float* sink;
  hipMalloc(&sink, 64);
  const int steps = 128;   // 128 x 48 = 6144 MFMAs per wavefront
  printf("coissue probe: %d CUs, %d MFMAs per matrix wavefront; times in us\n", n_cu, steps * 48);
  const char* names[3] = {"v_pk_fma_f32", "v_fma_f32 x2", "v_pk_fma_f32 + ds_read_b128 / 8"};
  for (int geo = 0; geo < 2; ++geo) {
    const int nm = geo == 0 ? 4 : 8, nv = 4;
    auto go = [&](int a, int b, int fpm, int mode) {
      return geo == 0 ? run<512>(a, b, steps, fpm, mode, sink, n_cu) : run<768>(a, b, steps, fpm, mode, sink, n_cu);
    };
    const float tm = go(nm, 0, 1, 0);
    printf("== %d matrix wavefronts (+ %d vector wavefronts), %d threads: matrix alone %.1f us = %.0f TF, %.1f cycles/MFMA/SIMD at 2.4 GHz\n", nm, nv,
           geo == 0 ? 512 : 768, tm, (double)n_cu * nm * steps * 48 * 2.0 * 32 * 32 * 16 / tm / 1e6, tm * 2400.0 / (steps * 48.0 * nm / 4));
    for (int mode = 0; mode < 3; ++mode)
      for (int fpm = 1; fpm <= 4; fpm *= 2) {
        const float tv = go(0, nv, fpm, mode);       // vector wavefronts alone (occupying wave slots n_mfma.. is irrelevant: role by index)
        const float tv_pos = go(nm, nv, fpm, mode);  // both
        // vector alone must run in the same slots: launch with n_mfma idle wavefronts in front
        printf("   %-32s %d packed FMAs per MFMA-slot: vector alone %.1f (%.2f cycles/instr at 2.4 GHz), both %.1f, sum %.1f, overlap %.2f\n", names[mode], fpm,
               tv, tv * 2400.0 / (steps * fpm * 48.0 * (mode == 1 ? 2 : 1)), tv_pos, tm + tv, (tm + tv - tv_pos) / (tm < tv ? tm : tv));
      }
  }
  return 0;
}
