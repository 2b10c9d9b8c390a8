// Pointwise (1x1) convolution as an fp32 MFMA GEMM with the BatchNorm / residual / ReLU epilogue,
// for gfx950.
//
// Replaces, per JasperBlock sub-block (reference nemo/collections/asr/parts/jasper.py):
//   MaskedConv1d(Cin, Cout, 1) (:113-132, :374-384)  ->  BatchNorm1d(eps=1e-3) eval (:392)
//   [-> + residual branch output (:438-439)]  ->  ReLU (:405 / mout :444)
// and the CTC head's Conv1d(1024, V+1, 1, bias=True) (jasper.py:249).
//
//   Y[b][m][t] = act( scale[m] * sum_k Wt[k][m] * Xm[b][k][t] + shift[m] (+ R[b][m][t]) )
//
// Exact-fp32 arithmetic on the matrix cores: v_mfma_f32_32x32x2_f32 accumulates as a k-ordered
// fmaf chain (no TF32/xf32 exists on gfx950), which keeps greedy argmax parity with the fp32
// reference.  Workgroup tile 128(M) x 128(T) x 32(K), 4 wavefronts in a 2x2 grid, each owning a
// 64x64 block = 2x2 MFMA tiles (64 accumulator VGPRs).  Weights are pre-packed K-major so both
// operand tiles land in LDS with 16-byte coalesced loads and are read back conflict-free
// (32 consecutive floats per half-wave).  Global loads of tile k+1 are issued before the MFMAs
// of tile k (register-staged prefetch).
//
// Work-group ids are remapped so that the M-tiles sharing one activation tile run on the same
// XCD (the 8 XCDs have private L2s; block b lands on XCD b % 8).
#include "vasr_internal.h"

namespace vasr {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDA = BM + 4;  // +4 floats: keeps 16-B alignment, breaks the 512-B row pitch for the stores
constexpr int LDB = BN + 4;

template <bool MASK>
__device__ __forceinline__ float4 load_b(const float* __restrict__ p, int t, int len) {
  float4 v = *reinterpret_cast<const float4*>(p);
  if (MASK) {
    if (t + 0 >= len) v.x = 0.f;
    if (t + 1 >= len) v.y = 0.f;
    if (t + 2 >= len) v.z = 0.f;
    if (t + 3 >= len) v.w = 0.f;
  }
  return v;
}

template <bool MASK, bool RES>
__global__ __launch_bounds__(256) void pw_gemm_kernel(PwArgs a, int tiles_m, int tiles_t, int n_blocks) {
  __shared__ __attribute__((aligned(16))) float As[BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];

  // ---- XCD-aware remap: consecutive logical ids -> same XCD ----
  int bid = blockIdx.x;
  {
    const int q = n_blocks / 8, r = n_blocks % 8, xcd = bid % 8, slot = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int mt = bid % tiles_m;
  const int nt = bid / tiles_m;
  const int b = nt / tiles_t;
  const int t0 = (nt % tiles_t) * BN;
  const int m0 = mt * BM;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int ld_row = tid >> 5, ld_col = (tid & 31) * 4;  // 8 rows x 128 cols per pass, 4 passes
  const int len = MASK ? a.lens[b] : 0;

  const float* __restrict__ wt = a.wt + m0;
  const float* __restrict__ xb = a.x + (int64_t)b * a.K * a.ldx + t0;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int kr = k0 + ld_row + 8 * p;
      ra[p] = *reinterpret_cast<const float4*>(wt + (int64_t)kr * a.M + ld_col);
      rb[p] = load_b<MASK>(xb + (int64_t)kr * a.ldx + ld_col, t0 + ld_col, len);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *reinterpret_cast<float4*>(&As[(ld_row + 8 * p) * LDA + ld_col]) = ra[p];
      *reinterpret_cast<float4*>(&Bs[(ld_row + 8 * p) * LDB + ld_col]) = rb[p];
    }
  };

  const int nk = a.K / BK;
  gload(0);
  sstore();
  __syncthreads();
  const int kh = lane >> 5, l31 = lane & 31;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK; ks += 2) {
      const float a0 = As[(ks + kh) * LDA + wm + l31];
      const float a1 = As[(ks + kh) * LDA + wm + 32 + l31];
      const float b0 = Bs[(ks + kh) * LDB + wn + l31];
      const float b1 = Bs[(ks + kh) * LDB + wn + 32 + l31];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      sstore();
      __syncthreads();
    }
  }

  // ---- epilogue: BN affine (+ residual) + ReLU, 128-B row segments per half-wave ----
  // C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      const float sc = a.scale[m], sh = a.shift[m];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = t0 + wn + j * 32 + l31;
        float v = fmaf(acc[i][j][r], sc, sh);
        if (RES) v += a.res[((int64_t)b * a.M + m) * a.ldr + t];
        if (a.relu) v = fmaxf(v, 0.f);
        if (t < a.frames && m < a.m_store) a.y[((int64_t)b * a.m_store + m) * a.ldy + t] = v;
      }
    }
  }
}

}  // namespace

void launch_pointwise(const PwArgs& a, hipStream_t st) {
  const int tiles_m = a.M / BM;
  const int tiles_t = (int)((a.ldx + BN - 1) / BN);
  const int n_blocks = tiles_m * tiles_t * a.batch;
  dim3 grid(n_blocks), block(256);
  const bool mask = a.lens != nullptr, res = a.res != nullptr;
  if (mask && res) hipLaunchKernelGGL((pw_gemm_kernel<true, true>), grid, block, 0, st, a, tiles_m, tiles_t, n_blocks);
  else if (mask) hipLaunchKernelGGL((pw_gemm_kernel<true, false>), grid, block, 0, st, a, tiles_m, tiles_t, n_blocks);
  else if (res) hipLaunchKernelGGL((pw_gemm_kernel<false, true>), grid, block, 0, st, a, tiles_m, tiles_t, n_blocks);
  else hipLaunchKernelGGL((pw_gemm_kernel<false, false>), grid, block, 0, st, a, tiles_m, tiles_t, n_blocks);
}

}  // namespace vasr
