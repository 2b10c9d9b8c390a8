// Pointwise (1x1) convolution GEMM for SMALL batches: the latency form of encoder_pw_split.hip's kF16x2 arithmetic
// (reference op: nemo/collections/asr/parts/jasper.py:113-132 MaskedConv1d, :374-392 conv + BN, :428-448 residual + ReLU).
//
// A batch-1 call of QuartzNet is a chain of ~150 dependent kernels, and the throughput kernel spends a GEMM's 8-13 us on
// it as K / 64 DEPENDENT chunk steps (rows requested two chunks ahead, converted, a workgroup barrier, four k-steps of
// MFMAs, next chunk): with 4 of 256 CUs' worth of work per layer every step is a round trip to L2 / HBM, not work.
// This kernel makes ONE trip: a workgroup (128 rows x 32 columns) requests ALL K rows of its 32 columns at once -- K x 128
// bytes as 16-byte loads, K / 32 of them per staging thread in flight --, converts them to the fp16 hi / lo B image of the
// WHOLE K range in LDS as they arrive (K = 1024: 128 KB) and runs the K / 16 k-steps as one MFMA chain per multiplying
// wavefront (one 32 x 32 tile each; weight fragments from L2 eight k-steps ahead, B fragments from LDS two k-steps
// ahead), 256 rows of K behind the staging.
//
// Results are BIT-IDENTICAL to pw_gemm_split_kernel<..., kF16x2> on the same operands: same scale from the producers'
// maxima, same conversion, same three products per k-step in the same order into the same single accumulator chain,
// k-steps ascending (the throughput kernel's leading zero chunk for odd chunk counts adds exact zeros), same epilogue
// arithmetic.  That is what lets run_encoder pick the kernel by batch size without changing an utterance's bits
// (tests/test_gpu_parity.py::test_results_do_not_depend_on_batch_size_or_tile_shape, row-independent mode).
#include <cstdlib>

#include "vasr_internal.h"
#include "vasr_device.h"

namespace vasr {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using v4f = __attribute__((ext_vector_type(4))) float;
using v2f = __attribute__((ext_vector_type(2))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

constexpr int LBN = 32;          // columns per workgroup
constexpr int LNW = 4;           // multiplying wavefronts = 32-row m-tiles: 128 rows per workgroup
constexpr int LNT = 128 * LNW;   // + as many staging wavefronts
constexpr int LAD = 8;           // weight fragments requested this many k-steps ahead
constexpr int LBD = 2;           // B fragments read from LDS this many k-steps ahead

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// KS = K / 16 k-steps (16, 32, 64).  Wavefronts 0-3 MULTIPLY (one 32-row m-tile each), wavefronts 4-7 STAGE: a staging thread
// owns PI = K / 256 items (8 consecutive k rows x four adjacent columns), requests them all at once and converts them in
// request order, one item = 256 rows of K = one PHASE; a barrier hands each phase's quarter or half of the B image to the
// multipliers, which run its 16 k-steps while the next items are still arriving.  The roles keep the two request streams
// on different wavefronts: s_waitcnt counts in order, and a multiplier that waited for its next weight fragments would
// otherwise wait for every row requested before them.
template <int KS, bool DUAL, bool RES>
__global__ __launch_bounds__(LNT, 1) void pw_gemm_latency_kernel(PwArgs a, int blocks_m, int tiles_t, int n_blocks) {
  constexpr int PI = KS / 16;          // items per staging thread = phases
  constexpr int SPH = 16;              // k-steps per phase
  extern __shared__ __attribute__((aligned(16))) uint4 Bl[];   // [plane][KS][2][LBN]
  auto bs = [&](int plane, int s, int kb, int n) -> uint4& { return Bl[((plane * KS + s) * 2 + kb) * LBN + n]; };

  int bid = blockIdx.x;
  {
    const int q = n_blocks / 8, r = n_blocks % 8, xcd = bid % 8, slot = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int mb = bid % blocks_m;
  const int nt = bid / blocks_m;
  const int b = nt / tiles_t;
  const int t0 = (nt % tiles_t) * LBN;
  const int m0 = mb * (32 * LNW);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool mul = wave < LNW;
  const int wm = (wave & (LNW - 1)) * 32;
  const int kh = lane >> 5, l31 = lane & 31;
  const bool masked1 = !DUAL && a.lens != nullptr;   // (a dual launch masks its second source only, as the throughput kernel)
  const int len = masked1 ? a.lens[b] : 0;
  const int len2 = DUAL ? a.lens2[b] : 0;
  const int K1 = DUAL ? a.K1 : a.K;

  // a time tile on which every input is zero has nothing to reduce (as in the throughput kernel)
  const int zf = a.zero_from ? max(a.zero_from[b], DUAL ? len2 : 0) : 0x7fffffff;
  const bool live = t0 < zf;

  if (!mul) {
    // =========================================== staging wavefronts ==================================================
    if (!live) return;
    unsigned amv[8], amv2[8];
    amax_request(a.amax_x.p, a.amax_x.stride, a.amax_x.n, b, lane, amv);
    if (DUAL) amax_request(a.amax_x2.p, a.amax_x2.stride, a.amax_x2.n, b, lane, amv2);
    // item i = k rows 8 g .. 8 g + 7 (g = g0 + 32 i) of the column quad c4: one 16-byte load per row, 8 lanes = a row's 128 bytes
    const int st = tid - 64 * LNW, c4 = st & 7, g0 = st >> 3;
    v4f xr[PI][8];
#pragma unroll
    for (int i = 0; i < PI; ++i) {
      const int k0 = 8 * (g0 + 32 * i);
      const bool second = DUAL && 256 * i >= K1;   // K1 % 256 == 0 (launch rule): uniform per i
      const float* __restrict__ src = second ? a.x2 + ((int64_t)b * (a.K - K1) + (k0 - K1)) * a.ldx2 + t0 + 4 * c4
                                             : a.x + ((int64_t)b * K1 + k0) * a.ldx + t0 + 4 * c4;
      const int64_t ld = second ? a.ldx2 : a.ldx;
#pragma unroll
      for (int e = 0; e < 8; ++e) xr[i][e] = *reinterpret_cast<const v4f*>(src + (int64_t)e * ld);
    }
    __builtin_amdgcn_sched_barrier(0);   // every request above is issued before anything below waits for one
    float xs, inv;
    {
      unsigned mx = amax_collect(a.amax_x.p, a.amax_x.stride, a.amax_x.n, b, lane, amv);
      if (DUAL) mx = max(mx, amax_collect(a.amax_x2.p, a.amax_x2.stride, a.amax_x2.n, b, lane, amv2));
      f16_scale(mx, &xs, &inv);
    }
#pragma unroll
    for (int i = 0; i < PI; ++i) {
      const int g = g0 + 32 * i;
      const bool second = DUAL && 256 * i >= K1;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int n = 4 * c4 + c;
        const bool keep = second ? (t0 + n < len2) : (!masked1 || t0 + n < len);   // MaskedConv1d (jasper.py:113-118)
        unsigned hh[4], ll[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float x0 = keep ? xr[i][2 * q][c] : 0.f, x1 = keep ? xr[i][2 * q + 1][c] : 0.f;
          const v2f v = {x0 * xs, x1 * xs};
          const f16x2 hv = __builtin_convertvector(v, f16x2);
          const v2f rr = v - __builtin_convertvector(hv, v2f);   // exact: the residual of a round-to-nearest conversion
          hh[q] = __builtin_bit_cast(unsigned, hv);
          ll[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(rr, f16x2));
        }
        bs(0, g >> 1, g & 1, n) = make_uint4(hh[0], hh[1], hh[2], hh[3]);
        bs(1, g >> 1, g & 1, n) = make_uint4(ll[0], ll[1], ll[2], ll[3]);
      }
      __syncthreads();   // phase i of the B image is complete
    }
    __syncthreads();     // (the multipliers' epilogue reuses the image)
    return;
  }

  // ============================================= multiplying wavefronts ==============================================
  const uint4* __restrict__ ap = reinterpret_cast<const uint4*>(a.wt) + ((int64_t)((m0 + wm) / 32) * KS) * 2 * 64 + lane;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // the epilogue's operands ride along with the first requests: BN scale / shift of this lane's rows (both epilogue forms
  // read the same eight float4), the residual pieces of the float4 form
  v4f scv[4], shv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    scv[q] = *reinterpret_cast<const v4f*>(a.scale + m0 + wm + 8 * q + 4 * kh);
    shv[q] = *reinterpret_cast<const v4f*>(a.shift + m0 + wm + 8 * q + 4 * kh);
  }
  const bool full = (t0 + LBN <= a.store_cols) && (m0 + 32 * LNW <= a.m_store);
  const bool vec = full && ((a.ldy | a.ldr) & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.res)) & 15) == 0;
  const int erow = lane / (LBN / 4), ec4 = lane % (LBN / 4);
  v4f rv[4];
  if (RES && vec) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      rv[q] = *reinterpret_cast<const v4f*>(a.res + ((int64_t)b * a.M + m0 + wm + 8 * q + erow) * a.ldr + t0 + 4 * ec4);
  }
  float out_scale = 1.f;
  if (live) {
    unsigned amv[8], amv2[8];
    amax_request(a.amax_x.p, a.amax_x.stride, a.amax_x.n, b, lane, amv);
    if (DUAL) amax_request(a.amax_x2.p, a.amax_x2.stride, a.amax_x2.n, b, lane, amv2);
    uint4 aw[LAD][2];
#pragma unroll
    for (int s = 0; s < LAD; ++s)
#pragma unroll
      for (int p = 0; p < 2; ++p) aw[s][p] = ap[((int64_t)s * 2 + p) * 64];
    {
      unsigned mx = amax_collect(a.amax_x.p, a.amax_x.stride, a.amax_x.n, b, lane, amv);
      if (DUAL) mx = max(mx, amax_collect(a.amax_x2.p, a.amax_x2.stride, a.amax_x2.n, b, lane, amv2));
      float xs, inv;
      f16_scale(mx, &xs, &inv);
      out_scale = inv * a.w_inv_scale;
    }
    uint4 bf[LBD + 1][2];
#pragma unroll
    for (int ph = 0; ph < PI; ++ph) {
      __syncthreads();   // phase ph of the B image is complete
      const int s0 = ph * SPH;
#pragma unroll
      for (int s = 0; s < LBD; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) bf[(s0 + s) % (LBD + 1)][p] = bs(p, s0 + s, kh, l31);
#pragma unroll
      for (int s = s0; s < s0 + SPH; ++s) {
        if (s + LBD < s0 + SPH) {
#pragma unroll
          for (int p = 0; p < 2; ++p) bf[(s + LBD) % (LBD + 1)][p] = bs(p, s + LBD, kh, l31);
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint4 w0 = aw[s % LAD][0], w1 = aw[s % LAD][1];
        const uint4 b0 = bf[s % (LBD + 1)][0], b1 = bf[s % (LBD + 1)][1];
        // cross terms, smallest first -- the order of pw_gemm_split_kernel
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w1), __builtin_bit_cast(f16x8, b0), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w0), __builtin_bit_cast(f16x8, b1), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w0), __builtin_bit_cast(f16x8, b0), acc, 0, 0, 0);
        if (s + LAD < KS) {   // the set this step has just released
#pragma unroll
          for (int p = 0; p < 2; ++p) aw[s % LAD][p] = ap[((int64_t)(s + LAD) * 2 + p) * 64];
        }
      }
    }
    __syncthreads();   // the epilogue reuses the LDS image
  }

  // ---- epilogue: BN affine (+ residual) + ReLU, as pw_gemm_split_kernel with TM = TN = 1 ----
  if (a.relu & 2) return;
  const int ylen = a.amax_y.p ? (a.lens_y ? a.lens_y[b] : a.frames) : 0;
  unsigned ymax = 0;
  auto track = [&](float v, int t) {
    const unsigned u = __float_as_uint(v) & 0x7fffffffu;
    ymax = (t < ylen && u > ymax) ? u : ymax;
  };
  const float relu_floor = (a.relu & 1) ? 0.f : -__builtin_inff();
  if (vec) {
    float* stage = reinterpret_cast<float*>(Bl) + wave * (2 * 8 * LBN);
    const int row = erow, c4 = ec4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float* buf = stage + (q & 1) * (8 * LBN);
      const int mq = m0 + wm + 8 * q;
      wave_fence();
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float v = acc[4 * q + rr] * out_scale;   // exact: a power of two
        buf[(4 * kh + rr) * LBN + l31] = fmaf(v, scv[q][rr], shv[q][rr]);
      }
      wave_fence();
      v4f v = *reinterpret_cast<const v4f*>(buf + row * LBN + 4 * c4);
      if (RES) v += rv[q];
      v = __builtin_elementwise_max(v, v4f{relu_floor, relu_floor, relu_floor, relu_floor});
      const int m = mq + row, t = t0 + 4 * c4;
      *reinterpret_cast<v4f*>(a.y + ((int64_t)b * a.m_store + m) * a.ldy + t) = v;
#pragma unroll
      for (int e = 0; e < 4; ++e) track(v[e], t + e);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mq = m0 + wm + 8 * q + 4 * kh;
      const v4f sc = scv[q], sh = shv[q];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int m = mq + rr, t = t0 + l31;
        float v = acc[4 * q + rr] * out_scale;
        v = fmaf(v, sc[rr], sh[rr]);
        if (RES) v += a.res[((int64_t)b * a.M + m) * a.ldr + t];
        if (a.relu & 1) v = fmaxf(v, 0.f);
        if (full || (t < a.store_cols && m < a.m_store)) {
          a.y[((int64_t)b * a.m_store + m) * a.ldy + t] = v;
          if (a.amax_y.p) track(v, t);
        }
      }
    }
  }
  if (a.amax_y.p) amax_publish(a.amax_y.p, a.amax_y.stride, b, (mb * tiles_t + nt % tiles_t) * LNW + wave, ymax, lane);
}

template <int KS, bool DUAL, bool RES>
int launch_lat_k(const PwArgs& a, hipStream_t st, int* amax_n) {
  const int blocks_m = a.M / (32 * LNW);
  const int tiles_t = (int)((a.ldx + LBN - 1) / LBN);
  const int n_blocks = blocks_m * tiles_t * a.batch;
  if (a.amax_y.p) {
    const int n = blocks_m * tiles_t * LNW;
    if (n > a.amax_y.stride) return (int)hipErrorInvalidValue;
    if (amax_n) *amax_n = n;
  }
  constexpr size_t lds = (size_t)2 * KS * 2 * LBN * sizeof(uint4);   // >= the epilogue's 4 x 2 KB for every KS
  auto kern = pw_gemm_latency_kernel<KS, DUAL, RES>;
  static std::atomic<uint64_t> lds_opted{0};   // per device (dyn_lds_opt_in)
  const hipError_t attr = dyn_lds_opt_in(reinterpret_cast<const void*>(kern), (int)lds, lds_opted);
  if (attr != hipSuccess) return (int)attr;
  VASR_LAUNCH(kern, dim3(n_blocks), dim3(LNT), lds, st, a, blocks_m, tiles_t, n_blocks);
  return 0;
}

template <int KS>
int launch_lat_s(const PwArgs& a, hipStream_t st, int* amax_n) {
  const bool dual = a.x2 != nullptr, res = a.res != nullptr;
  if (dual) return launch_lat_k<KS, true, false>(a, st, amax_n);   // (a dual launch has no residual tensor: launch_l)
  return res ? launch_lat_k<KS, false, true>(a, st, amax_n) : launch_lat_k<KS, false, false>(a, st, amax_n);
}

}  // namespace

// shapes the latency kernel covers: 128-row blocks, K = 256 / 512 / 1024 (a dual source splits K on the 256-row grid of the items)
bool pointwise_latency_supported(int M, int K, int K1) {
  return M % (32 * LNW) == 0 && (K == 256 || K == 512 || K == 1024) && (K1 == 0 || K1 % 256 == 0);
}

// the kF16x2 GEMM on the latency kernel; same arguments and results as launch_pointwise_split(a, 2, ...).  -1: shape not covered.
int launch_pointwise_latency(const PwArgs& a, hipStream_t st, int* amax_n) {
  if (!pointwise_latency_supported(a.M, a.K, a.x2 ? a.K1 : 0)) return -1;
  switch (a.K) {
    case 256: return launch_lat_s<16>(a, st, amax_n);
    case 512: return launch_lat_s<32>(a, st, amax_n);
    default: return launch_lat_s<64>(a, st, amax_n);
  }
}

}  // namespace vasr
