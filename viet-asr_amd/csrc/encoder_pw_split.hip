// Pointwise (1x1) convolution GEMM on the 16-bit matrix pipe with fp32-equivalent accuracy ("split operand" GEMMs).
//
// Same operation, epilogue and dual-source K reduction as encoder_pw.hip (reference
// nemo/collections/asr/parts/jasper.py:113-132, :374-392, :428-448).  Three arithmetics share the kernel (template ARITH):
//
//   kBf16x3  every fp32 operand split exactly into three bf16 terms  x = x_hi + x_mid + x_lo  (round-to-nearest each:
//            8 + 8 + 8 significant bits = the 24 of fp32), product from the six largest cross terms
//                a*b ~= a_hi b_hi + a_hi b_mid + a_mid b_hi + a_mid b_mid + a_hi b_lo + a_lo b_hi
//            on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; dropped terms < 2^-25 |a b|.
//   kF16x2   every operand scaled by a power of two (exact) and split into TWO fp16 terms  s x = x_hi + x_lo  (11 + 11
//            significant bits; the scale puts the utterance's largest |x| -- and the layer's largest |w| -- in
//            [2^14, 2^15), so that x_lo keeps all its bits down to |x| = 2^-17 max|x| and an absolute error of
//            2^-40 max|x| below that), product from the three largest cross terms  a_hi b_hi + a_hi b_lo + a_lo b_hi  on
//            v_mfma_f32_32x32x16_f16: HALF the matrix work of kBf16x3.  Per-product error <= 3 * 2^-22, but only three
//            fp32 accumulator roundings per 16-deep k-step instead of six: measured against fp64 the result is not
//            less accurate than kBf16x3's or the fp32-MFMA chain's (tests/test_gpu_parity.py::
//            test_split_gemms_are_as_accurate_as_fp32_mfma).  The accumulators carry s_x s_w times the result; the
//            epilogue multiplies by the (power-of-two, exact) inverse before the BN affine.  The per-utterance maxima
//            come from the kernel that produced the tensor (depthwise kernels and this kernel's own epilogue publish
//            max |y| per utterance with one atomic per wavefront: PwArgs::amax_*), so the scale of a row depends on
//            that row alone and results stay independent of the batch an utterance sits in.
//   kBf16x2  opt-in REDUCED precision: the two upper bf16 terms and three cross terms (16-bit significands).
//
//   * weights are split and packed at vasr_finalize() in A-fragment order [M/32][K/16][planes][64 lanes][8 x 16 bit]:
//     one 16-byte load per lane, plane and 16-deep k-step, straight from L2, one step ahead of use;
//   * activations are split ONCE per workgroup while they are staged into LDS, laid out
//     [plane][k-step][k-half][column][8 x 16 bit] so that every B fragment is one conflict-free ds_read_b128;
//   * throughput tile 512 x 128 (8 wavefronts, two 32-row m-tiles each, every wave owning all 128 columns, 128
//     accumulator registers, one workgroup per CU): a weight fragment is reused for 12-24 MFMAs;
//   * latency tile 64 x 32 (2 wavefronts) for small batches, where the throughput tile would leave most of the
//     256 CUs idle (B = 1: 4 workgroups per layer instead of 128).
#include <cstdlib>
#include <cstring>

#include "vasr_internal.h"
#include "vasr_device.h"
#include <type_traits>

namespace vasr {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using v4f = __attribute__((ext_vector_type(4))) float;
using v2f = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

enum { kBf16x3 = 0, kBf16x2 = 1, kF16x2 = 2 };

constexpr int BKC = 64;   // K rows per LDS buffer

constexpr int STEPS = BKC / 16;

__device__ __forceinline__ unsigned cvt2(float a, float b) {
  const v2f v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32, RNE
}

// 8 consecutive-k values of one column -> three 16-byte bf16 fragments
__device__ __forceinline__ void split3(const float (&x)[8], uint4& hi, uint4& mid, uint4& lo) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = x[2 * p], b = x[2 * p + 1];
    h[p] = cvt2(a, b);
    const float ra = a - __uint_as_float(h[p] << 16), rb = b - __uint_as_float(h[p] & 0xffff0000u);   // exact
    m[p] = cvt2(ra, rb);
    const float sa = ra - __uint_as_float(m[p] << 16), sb = rb - __uint_as_float(m[p] & 0xffff0000u); // exact
    l[p] = cvt2(sa, sb);
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  mid = make_uint4(m[0], m[1], m[2], m[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// 8 consecutive-k values of one column, scaled by the power of two s -> two 16-byte fp16 fragments:
// hi = rne16(s x), lo = rne16(s x - hi)  (v_pk_mul_f32, v_cvt_pk_f16_f32, 2 x v_cvt_f32_f16, v_pk_fma_f32,
// v_cvt_pk_f16_f32: six VALU instructions per pair)
__device__ __forceinline__ void split2h(const float (&x)[8], float s, uint4& hi, uint4& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const v2f v = {x[2 * p] * s, x[2 * p + 1] * s};
    const f16x2 hh = __builtin_convertvector(v, f16x2);
    const v2f r = v - __builtin_convertvector(hh, v2f);   // exact: the residual of a round-to-nearest conversion
    h[p] = __builtin_bit_cast(unsigned, hh);
    l[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}

template <int ARITH>
__device__ __forceinline__ f32x16 mma(uint4 a, uint4 b, f32x16 c) {
  if constexpr (ARITH == kF16x2)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// orders a wavefront's own LDS writes and reads for the compiler (the LDS pipeline itself keeps them in issue order)
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// NW wavefronts stacked along M, each TM m-tiles (32 rows) x all TN n-tiles (32 columns each):
// workgroup tile (32*TM*NW) x (32*TN).
template <int NW, int TM, int TN, int ARITH>
struct Geom {
  static constexpr int BM = 32 * TM * NW;
  static constexpr int BN = 32 * TN;
  static constexpr int NT = 64 * NW;                   // threads
  static constexpr int PL = ARITH == kBf16x3 ? 3 : 2;  // operand planes multiplied (and staged into LDS)
  static constexpr int PLW = ARITH == kF16x2 ? 2 : 3;  // planes in the weight pack (bf16x2 reads the bf16x3 pack)
  static constexpr int PATCHES = BKC / 8 * BN;         // staging patches (8 k-rows x 1 column) per chunk
  static constexpr int PPT = PATCHES / NT;             // patches per thread
  static constexpr size_t LDS_MAIN = (size_t)2 * PL * STEPS * 2 * BN * sizeof(uint4);
  static constexpr size_t LDS_EPI = (size_t)NW * 2 * 8 * BN * sizeof(float);   // epilogue transposition buffers
  static constexpr size_t LDS = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
  static_assert(PATCHES % NT == 0 && PPT >= 1, "staging patches must divide evenly over the threads");
};

template <int NW, int TM, int TN, bool MASK, bool RES, bool DUAL, int ARITH>
__global__ __launch_bounds__(64 * NW, 2) void pw_gemm_split_kernel(PwArgs a, int blocks_m, int tiles_t, int n_blocks) {
  using G = Geom<NW, TM, TN, ARITH>;
  constexpr int BM = G::BM, BN = G::BN, NT = G::NT, PPT = G::PPT, PL = G::PL, PLW = G::PLW;
  extern __shared__ __attribute__((aligned(16))) uint4 Bs[];   // [2][PL][STEPS][2][BN]
  auto bs = [&](int buf, int plane, int s, int kb, int n) -> uint4& {
    return Bs[(((buf * PL + plane) * STEPS + s) * 2 + kb) * BN + n];
  };

  int bid = blockIdx.x;
  {
    const int q = n_blocks / 8, r = n_blocks % 8, xcd = bid % 8, slot = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int mb = bid % blocks_m;
  const int nt = bid / blocks_m;
  const int b = nt / tiles_t;
  const int t0 = (nt % tiles_t) * BN;
  const int m0 = mb * BM;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave * 32 * TM;
  const int kh = lane >> 5, l31 = lane & 31;
  const int len = MASK ? a.lens[b] : 0;
  const int len2 = DUAL ? a.lens2[b] : 0;

  // kF16x2: one power-of-two scale per utterance from the maxima the producers of x (and x2) published -- requested
  // below together with the first rows and weight fragments, so that the reduction's trip to L2 runs inside the rows'
  // HBM round trip instead of in front of it
  float xs = 1.f, out_scale = 1.f;

  const int K1 = DUAL ? a.K1 : a.K;
  const float* __restrict__ xb = a.x + (int64_t)b * K1 * a.ldx + t0;
  const float* __restrict__ xb2 = DUAL ? a.x2 + (int64_t)b * (a.K - K1) * a.ldx2 + t0 : nullptr;
  // A fragments [M/32][K/16][PLW][64] uint4
  const int ksteps = a.K / 16;
  const uint4* __restrict__ ap = reinterpret_cast<const uint4*>(a.wt) + ((int64_t)((m0 + wm) / 32) * ksteps) * PLW * 64 + lane;
  const int64_t a_tile = (int64_t)ksteps * PLW * 64;   // uint4 stride between m-tiles

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- activation staging ----
  // A thread owns PPT patches of a chunk: patch = 8 consecutive k rows x one column (one 16-byte B-fragment slot per
  // plane).  With PPT even it loads them as float2 (two adjacent columns per row: 512 contiguous bytes per wavefront and
  // row, half the load instructions of the dword form).  The rows of chunk c + 1 are CONVERTED during chunk c, a
  // quarter per k-step, out of registers that were requested during chunk c - 1: an HBM round trip has a whole chunk
  // (~5 us) to complete, and the conversion's VALU work is spread under the MFMAs instead of sitting behind the last
  // one (measured on the one-chunk-ahead form: staging cost 7.5 of 42 us of a 512-channel layer's main loop).
  constexpr bool PAIRS = PPT % 2 == 0;
  auto patch_of = [&](int p, int& n, int& g) {
    if constexpr (PAIRS) {
      const int pi = tid + (p >> 1) * NT;
      n = 2 * (pi % (BN / 2)) + (p & 1);
      g = pi / (BN / 2);
    } else {
      const int idx = tid + p * NT;
      n = idx % BN;
      g = idx / BN;
    }
  };
  // requests chunk k0 (clamped to the last chunk: past the end the last chunk is simply requested again, so that the
  // number of loads in flight is the same on every path -- s_waitcnt vmcnt counts, it does not name)
  // two stage register sets: rows are requested two chunks ahead and converted a quarter per k-step (one set -- requested at the
  // start of the previous chunk, converted during its last k-step -- saved 16 registers and measured slower: DESIGN_HISTORY.md)
  constexpr int SSETS = 2;
  float rr[SSETS][PPT][8];   // one set being converted, one being filled (indexed by compile-time constants only)
  auto gload = [&](int k0, auto set_tag) {
    auto& r = rr[decltype(set_tag)::value % SSETS];
    k0 = k0 < a.K - BKC ? k0 : a.K - BKC;
    const bool second = DUAL && k0 >= K1;
    const float* __restrict__ base = second ? xb2 + (int64_t)(k0 - K1) * a.ldx2 : xb + (int64_t)k0 * a.ldx;
    const int64_t ld = second ? a.ldx2 : a.ldx;
    if constexpr (PAIRS) {
#pragma unroll
      for (int q = 0; q < PPT / 2; ++q) {
        int n, g;
        patch_of(2 * q, n, g);
        const float* __restrict__ src = base + (int64_t)(8 * g) * ld + n;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const v2f v = *reinterpret_cast<const v2f*>(src + (int64_t)e * ld);
          r[2 * q][e] = v.x;
          r[2 * q + 1][e] = v.y;
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < PPT; ++p) {
        int n, g;
        patch_of(p, n, g);
        const float* __restrict__ src = base + (int64_t)(8 * g) * ld + n;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[p][e] = src[(int64_t)e * ld];
      }
    }
  };
  // converts half h (k rows 4 h .. 4 h + 3) of patch p of chunk k0 and stores it into LDS buffer `buf`
  auto sstore_half = [&](int buf, int k0, auto set_tag, int p, int h) {
    const auto& r = rr[decltype(set_tag)::value % SSETS];
    const bool second = DUAL && k0 >= K1;
    const int ml = second ? len2 : len;
    const bool masked = second || MASK;
    int n, g;
    patch_of(p, n, g);
    const bool keep = !masked || (t0 + n < ml);   // MaskedConv1d: x.masked_fill(t >= lens, 0) (jasper.py:113-118)
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = keep ? r[p][4 * h + e] : 0.f;
    if constexpr (ARITH == kF16x2) {
      unsigned hh[2], ll[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const v2f v = {x[2 * q] * xs, x[2 * q + 1] * xs};
        const f16x2 hv = __builtin_convertvector(v, f16x2);
        const v2f rr = v - __builtin_convertvector(hv, v2f);   // exact: the residual of a round-to-nearest conversion
        hh[q] = __builtin_bit_cast(unsigned, hv);
        ll[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(rr, f16x2));
      }
      reinterpret_cast<uint2*>(&bs(buf, 0, g >> 1, g & 1, n))[h] = make_uint2(hh[0], hh[1]);
      reinterpret_cast<uint2*>(&bs(buf, 1, g >> 1, g & 1, n))[h] = make_uint2(ll[0], ll[1]);
    } else {
      unsigned hh[2], mm[2], ll[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float u = x[2 * q], v = x[2 * q + 1];
        hh[q] = cvt2(u, v);
        const float ru = u - __uint_as_float(hh[q] << 16), rv = v - __uint_as_float(hh[q] & 0xffff0000u);   // exact
        mm[q] = cvt2(ru, rv);
        const float su = ru - __uint_as_float(mm[q] << 16), sv = rv - __uint_as_float(mm[q] & 0xffff0000u); // exact
        ll[q] = cvt2(su, sv);
      }
      reinterpret_cast<uint2*>(&bs(buf, 0, g >> 1, g & 1, n))[h] = make_uint2(hh[0], hh[1]);
      reinterpret_cast<uint2*>(&bs(buf, 1, g >> 1, g & 1, n))[h] = make_uint2(mm[0], mm[1]);
      if constexpr (PL == 3) reinterpret_cast<uint2*>(&bs(buf, 2, g >> 1, g & 1, n))[h] = make_uint2(ll[0], ll[1]);
    }
  };

  // weights: the next k-step's fragments are in flight while the current ones are multiplied
  // two fragment sets in rotation: `af` is multiplied while `an` is filled (four sets, three k-steps ahead, measured the same)
  uint4 aw[2][TM][PL];
  uint4 (&af)[TM][PL] = aw[0];
  uint4 (&an)[TM][PL] = aw[1];
  auto aload = [&](int s, uint4 (&dst)[TM][PL]) {
    const int sc = s < 0 ? 0 : (s < ksteps ? s : ksteps - 1);   // harmless re-reads before the start (zero chunk) / past the end
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int p = 0; p < PL; ++p) dst[i][p] = ap[i * a_tile + ((int64_t)sc * PLW + p) * 64];
  };

  // A time tile on which every input is zero (past the utterance's length in a ragged batch) has nothing to reduce:
  // its outputs are shift (+ residual) through the ReLU, which is what the epilogue makes of zero accumulators.
  const int zf = a.zero_from ? max(a.zero_from[b], DUAL ? len2 : 0) : 0x7fffffff;
  const int nchunks = t0 >= zf ? 0 : a.K / BKC;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // Chunks are processed in PAIRS (the two stage register sets swap roles with every chunk, and the register names
  // must be static).  An odd count is made even by a leading chunk "-1" that multiplies zeros: LDS buffer 1 is cleared
  // instead of converted, the weight index clamps to step 0 (0 * w adds exact zeros), and chunk 0's rows are converted
  // during it like any other.  Only the K = 64 layer of the first block (one chunk) and odd dual-source sums pay for it.
  // (A separate tail for the odd chunk was tried first: a second conditional copy of the chunk body next to the
  // final-pair block makes the register allocator spill ~160 registers around the merge of the 128 accumulators.)
  const int odd = nchunks & 1;
  unsigned amv[8], amv2[8];
  if (nchunks) {
    if constexpr (ARITH == kF16x2) {
      amax_request(a.amax_x.p, a.amax_x.stride, a.amax_x.n, b, lane, amv);
      if (DUAL) amax_request(a.amax_x2.p, a.amax_x2.stride, a.amax_x2.n, b, lane, amv2);
    }
    gload(0, S0{});
    aload(0, af);
    if constexpr (ARITH == kF16x2) {
      // every wavefront reduces the producers' per-wavefront maxima of its utterance itself (a few KB from L2, no barrier);
      // a tile without chunks multiplies nothing and needs no scale
      unsigned mx = amax_collect(a.amax_x.p, a.amax_x.stride, a.amax_x.n, b, lane, amv);
      if (DUAL) mx = max(mx, amax_collect(a.amax_x2.p, a.amax_x2.stride, a.amax_x2.n, b, lane, amv2));
      float inv;
      f16_scale(mx, &xs, &inv);
      out_scale = inv * a.w_inv_scale;
    }
    if (odd) {
#pragma unroll
      for (int p = 0; p < PPT; ++p) {
        int n, g;
        patch_of(p, n, g);
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) bs(1, pl, g >> 1, g & 1, n) = make_uint4(0u, 0u, 0u, 0u);
      }
    } else {
#pragma unroll
      for (int p = 0; p < PPT; ++p) {
        sstore_half(0, 0, S0{}, p, 0);
        sstore_half(0, 0, S0{}, p, 1);
      }
      gload(BKC, S0{});   // chunk 1: converted during chunk 0
    }
  }
  __syncthreads();

  // One K chunk c.  `cur` holds the rows of chunk c + 1 (requested one chunk ago), `nxt` receives those of chunk c + 2.
  // LAST = false: request, convert a share of `cur` per k-step into the idle LDS buffer -- unconditionally, no branch
  // anywhere in this body (with one the compiler merges the paths' load counters and waits for everything, vmcnt(0),
  // before the first MFMA of every chunk).  LAST = true is the final chunk, which stages nothing.
  auto run_chunk = [&](const int c, auto cur, auto nxt, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    constexpr int HALVES = 2 * PPT;
    const int cn = c + 1;
    // activation fragments: the n-tile being multiplied and the next one being read; the rotation runs across the
    // k-steps of the chunk, so that a step's first fragments are already in flight when the step starts
    uint4 bf[2][PL];
#pragma unroll
    for (int p = 0; p < PL; ++p) bf[0][p] = bs(c & 1, p, 0, kh, l31);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      aload(c * STEPS + s + 1, an);
      if (!LAST && s == 0) gload((c + SSETS) * BKC, nxt);
      // Pins the loads at the top of the step.  Left alone, the scheduler sinks them towards their first use to save
      // registers: the weight prefetch then runs ~8 MFMAs ahead instead of a whole step.
      __builtin_amdgcn_sched_barrier(0);
      uint4 (&cw)[TM][PL] = af;   // this k-step's weight fragments
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int g = s * TN + j;
        const int cur_f = g & 1, nxt_f = cur_f ^ 1;
        if (j + 1 < TN) {
#pragma unroll
          for (int p = 0; p < PL; ++p) bf[nxt_f][p] = bs(c & 1, p, s, kh, (j + 1) * 32 + l31);
        } else if (s + 1 < STEPS) {
#pragma unroll
          for (int p = 0; p < PL; ++p) bf[nxt_f][p] = bs(c & 1, p, s + 1, kh, l31);
        }
        // cross terms, smallest first; the m-tiles alternate so that consecutive MFMAs never chain on one accumulator
        if constexpr (ARITH == kBf16x3) {
#pragma unroll
          for (int i = 0; i < TM; ++i) acc[i][j] = mma<ARITH>(cw[i][2], bf[cur_f][0], acc[i][j]);   // lo  * hi
#pragma unroll
          for (int i = 0; i < TM; ++i) acc[i][j] = mma<ARITH>(cw[i][0], bf[cur_f][2], acc[i][j]);   // hi  * lo
#pragma unroll
          for (int i = 0; i < TM; ++i) acc[i][j] = mma<ARITH>(cw[i][1], bf[cur_f][1], acc[i][j]);   // mid * mid
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mma<ARITH>(cw[i][1], bf[cur_f][0], acc[i][j]);   // mid (lo) * hi
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mma<ARITH>(cw[i][0], bf[cur_f][1], acc[i][j]);   // hi  * mid (lo)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mma<ARITH>(cw[i][0], bf[cur_f][0], acc[i][j]);   // hi  * hi
        // this k-step's share of the next chunk's conversion, after the step's first n-tile: the scheduler spreads
        // it under the MFMAs that follow
        if (!LAST && j == 0) {
          const int h0 = s * HALVES / STEPS, h1 = (s + 1) * HALVES / STEPS;
#pragma unroll
          for (int hv = h0; hv < h1; ++hv) sstore_half(cn & 1, cn * BKC, cur, hv >> 1, hv & 1);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < PL; ++p) af[i][p] = an[i][p];
    }
    __syncthreads();   // after the last chunk: the epilogue reuses the LDS buffers
  };
  {
    int c = -odd;
    for (; c + 2 < nchunks; c += 2) {
      run_chunk(c, S0{}, S1{}, std::false_type{});
      run_chunk(c + 1, S1{}, S0{}, std::false_type{});
    }
    if (nchunks) {   // the final pair: c + 2 == nchunks
      run_chunk(c, S0{}, S1{}, std::false_type{});
      run_chunk(c + 1, S1{}, S0{}, std::true_type{});
    }
  }

  // ---- epilogue (as encoder_pw.hip): BN affine (+ residual) + ReLU, 128-byte row segments per half-wave ----
  // max |y| over the utterance's VALID output frames, for the split of the next kF16x2 consumer of y
  const int ylen = a.amax_y.p ? (a.lens_y ? a.lens_y[b] : a.frames) : 0;
  unsigned ymax = 0;
  auto track = [&](float v, int t) {
    const unsigned u = __float_as_uint(v) & 0x7fffffffu;
    ymax = (t < ylen && u > ymax) ? u : ymax;
  };
  const float relu_floor = (a.relu & 1) ? 0.f : -__builtin_inff();   // max(v, floor): ReLU or nothing, without a branch
  const bool full = (t0 + BN <= a.store_cols) && (m0 + BM <= a.m_store);
  const bool vec = full && ((a.ldy | a.ldr) & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.res)) & 15) == 0;
  if (vec) {
    // Interior tiles: the accumulators (one column, 16 scattered rows per lane) go through LDS, 8 rows x BN columns
    // per pass in a wavefront-private buffer, and come back as float4 row pieces, so that a store instruction writes
    // whole row segments (2 x 512 B for BN = 128) instead of 2 x 128 B dwords and the residual is read the same way.
    // The 67 MB output of a 512-channel layer is otherwise store-issue bound (4.8 TB/s against ~7.5 TB/s for a fill).
    float* stage = reinterpret_cast<float*>(Bs) + wave * (2 * 8 * BN);
    constexpr int F4 = 8 * BN / 4 / 64;   // float4 pieces per lane and pass
    // Every pass's BN scale / shift is requested BEFORE the first store.  Stores count in vmcnt like loads and the counter
    // retires in order: a load issued between two passes made its consumer wait -- vmcnt(0) -- for every store of the
    // passes before it, i.e. the 4 TM passes ran as 4 TM store round trips in series (round 2's "12 us store-bound
    // epilogue" of a 512-channel layer).
    v4f scv[TM][4], shv[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        scv[i][q] = *reinterpret_cast<const v4f*>(a.scale + m0 + wm + i * 32 + 8 * q + 4 * kh);
        shv[i][q] = *reinterpret_cast<const v4f*>(a.shift + m0 + wm + i * 32 + 8 * q + 4 * kh);
      }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float* buf = stage + ((i * 4 + q) & 1) * (8 * BN);
        const int mq = m0 + wm + i * 32 + 8 * q;
        const v4f sc = scv[i][q], sh = shv[i][q];
        wave_fence();   // LDS executes one wavefront's accesses in order: pass p+2's writes follow pass p's reads
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float v = acc[i][j][4 * q + rr];
            if constexpr (ARITH == kF16x2) v *= out_scale;   // exact: a power of two
            buf[(4 * kh + rr) * BN + 32 * j + l31] = fmaf(v, sc[rr], sh[rr]);
          }
        wave_fence();
        // The pass is straight-line code: all of its row pieces (and residual pieces) are requested together, the ReLU is a
        // maximum with a uniform floor and the maxima are tracked unconditionally (ylen = 0 without a table).  With a
        // uniform BRANCH per piece (relu / table / store kind) every piece was its own basic block -- LDS read, wait,
        // arithmetic, store, branch -- and the pass paid one LDS round trip per piece in series.
        v4f pv[F4], rv[F4];
#pragma unroll
        for (int k = 0; k < F4; ++k) {
          const int f = lane + 64 * k, row = f / (BN / 4), c4 = f % (BN / 4);
          pv[k] = *reinterpret_cast<const v4f*>(buf + row * BN + 4 * c4);
          if (RES) rv[k] = *reinterpret_cast<const v4f*>(a.res + ((int64_t)b * a.M + mq + row) * a.ldr + t0 + 4 * c4);
        }
#pragma unroll
        for (int k = 0; k < F4; ++k) {
          const int f = lane + 64 * k, row = f / (BN / 4), c4 = f % (BN / 4);
          const int m = mq + row, t = t0 + 4 * c4;
          v4f v = pv[k];
          if (RES) v += rv[k];
          v = __builtin_elementwise_max(v, v4f{relu_floor, relu_floor, relu_floor, relu_floor});
          v4f* dstp = reinterpret_cast<v4f*>(a.y + ((int64_t)b * a.m_store + m) * a.ldy + t);
          *dstp = v;   // (non-temporal stores for outputs beyond the Infinity Cache: measured, no gain)
#pragma unroll
          for (int e = 0; e < 4; ++e) track(v[e], t + e);
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int mq = m0 + wm + i * 32 + 8 * q + 4 * kh;
        const v4f sc = *reinterpret_cast<const v4f*>(a.scale + mq);
        const v4f sh = *reinterpret_cast<const v4f*>(a.shift + mq);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int m = mq + rr;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int t = t0 + j * 32 + l31;
            float v = acc[i][j][4 * q + rr];
            if constexpr (ARITH == kF16x2) v *= out_scale;
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
    }
  }
  if (a.amax_y.p) amax_publish(a.amax_y.p, a.amax_y.stride, b, (mb * tiles_t + nt % tiles_t) * NW + wave, ymax, lane);
}

template <int NW, int TM, int TN, bool MASK, bool RES, bool DUAL, int ARITH>
int launch_k(const PwArgs& a, hipStream_t st, int* amax_n) {
  using G = Geom<NW, TM, TN, ARITH>;
  const int blocks_m = a.M / G::BM;
  const int tiles_t = (int)((a.ldx + G::BN - 1) / G::BN);
  const int n_blocks = blocks_m * tiles_t * a.batch;
  if (a.amax_y.p) {
    const int n = blocks_m * tiles_t * NW;
    if (n > a.amax_y.stride) return (int)hipErrorInvalidValue;
    if (amax_n) *amax_n = n;
  }
  auto kern = pw_gemm_split_kernel<NW, TM, TN, MASK, RES, DUAL, ARITH>;
  static std::atomic<uint64_t> lds_opted{0};   // per device (dyn_lds_opt_in)
  const hipError_t attr = dyn_lds_opt_in(reinterpret_cast<const void*>(kern), (int)G::LDS, lds_opted);
  if (attr != hipSuccess) return (int)attr;
  VASR_LAUNCH(kern, dim3(n_blocks), dim3(G::NT), G::LDS, st, a, blocks_m, tiles_t, n_blocks);
  return 0;
}

template <int NW, int TM, int TN, int ARITH>
int launch_l(const PwArgs& a, hipStream_t st, int* amax_n) {
  const bool mask = a.lens != nullptr, res = a.res != nullptr, dual = a.x2 != nullptr;
  if (dual) return launch_k<NW, TM, TN, false, false, true, ARITH>(a, st, amax_n);
  if (mask && res) return launch_k<NW, TM, TN, true, true, false, ARITH>(a, st, amax_n);
  if (mask) return launch_k<NW, TM, TN, true, false, false, ARITH>(a, st, amax_n);
  if (res) return launch_k<NW, TM, TN, false, true, false, ARITH>(a, st, amax_n);
  return launch_k<NW, TM, TN, false, false, false, ARITH>(a, st, amax_n);
}

template <int NW, int TM, int TN>
int launch_t(const PwArgs& a, int arith, hipStream_t st, int* amax_n) {
  switch (arith) {
    case kF16x2: return launch_l<NW, TM, TN, kF16x2>(a, st, amax_n);
    case kBf16x2: return launch_l<NW, TM, TN, kBf16x2>(a, st, amax_n);
    default: return launch_l<NW, TM, TN, kBf16x3>(a, st, amax_n);
  }
}

inline unsigned short bf16_rne(float x, float* back) {
  unsigned u = __builtin_bit_cast(unsigned, x);
  u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
  *back = __builtin_bit_cast(float, u);
  return (unsigned short)(u >> 16);
}

}  // namespace

bool pointwise_split_supported(int M, int K, int K1) {
  return M % 64 == 0 && K % BKC == 0 && (K1 == 0 || K1 % BKC == 0);
}

// arith: 0 = 3 x bf16 (six products), 1 = 2 x bf16 (three products, reduced), 2 = 2 x fp16 scaled (three products; needs
// a.amax_x (and a.amax_x2 for a dual source), a.w_inv_scale and the fp16 weight pack).  Returns 0 or a hipError_t.
// the smallest tile (64 x 32 on two wavefronts) uses the most slots: (M / 64) * (ld / 32) * 2
int pointwise_amax_slots(int M, int64_t ld) { return (int)((int64_t)(M / 64) * ((ld + 31) / 32) * 2); }

int launch_pointwise_split(const PwArgs& args, int arith, hipStream_t st, int* amax_n) {
  const int force = dev_switches().pw3_tile;   // devtools build: 1..5 pins a tile shape (tests of the alternate tiles)
  const PwArgs& a = args;
  // the largest tile that divides M and still gives (almost) every one of the 256 CUs a workgroup:
  // 512x128, 256x128, 128x64, 64x32 (the CTC head, 29 or 91 rows padded to 128, runs 128x64 tiles)
  auto blocks = [&](int bm, int bn) { return (int64_t)(a.M / bm) * ((a.ldx + bn - 1) / bn) * a.batch; };
  const int rows[6] = {0, 512, 256, 128, 64, 256};
  int tile = 4;
  if (a.M % 512 == 0 && blocks(512, 128) >= 192) tile = 1;
  else if (a.M % 256 == 0 && blocks(256, 128) >= 192) tile = 2;
  else if (a.M % 128 == 0 && blocks(128, 64) >= 192) tile = 3;
  // 256-channel layers: 256 x 64 tiles put two or three workgroups on a CU (49 KB of LDS each), which hides more of
  // one workgroup's prologue / epilogue behind another's main loop: 30.0 -> 28.7 us at K = 256, 49.1 -> 48.1 at K = 512
  if (tile == 2 && a.M == 256 && blocks(256, 64) >= 384) tile = 5;
  // (for the 512-channel layers both 256 x 64 and 512 x 64 measured slower than 512 x 128: 88-91 / 94-97 vs 86 us)
  // ... when the 512 x 128 workgroups fill whole rounds of the chip (one per CU).  T' = 516 (a 10.3 s clip) is five
  // 128-column tiles per utterance: 320 workgroups = two rounds, the second a quarter full.  256 x 64 tiles (two per CU,
  // four times as many) quantise four times finer: measured 4.81 vs 5.68 ms of GEMM per step at 10.3 s, 4.84 vs 5.75 at
  // 12 s, 5.79 vs 6.16 at 15 s -- and 7.95 vs 7.02 at 20 s, where 512 x 128 fills its two rounds exactly.
  if (tile == 1) {
    static const int n_cu = [] {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
      return n;
    }();
    // (units a concurrent kernel of the caller's holds -- the overlapped beam search -- do not take workgroups)
    const int cus = a.busy_cus > 0 && a.busy_cus < n_cu - 32 ? n_cu - a.busy_cus : n_cu;
    const int64_t n1 = blocks(512, 128), rounds = (n1 + cus - 1) / cus;
    if ((double)n1 < 0.85 * (double)(rounds * cus)) tile = 5;
  }
  if (force >= 1 && force <= 5 && a.M % rows[force] == 0) tile = force;
  // Small batches (the 64 x 32 tile's territory: <= 5 utterances of 10 s at 512 channels): the whole K range in ONE trip to
  // memory instead of K / 64 dependent chunk steps -- encoder_pw_lat.hip, same bits.  (Devtools: VASR_PW_LAT=0 keeps the chunked
  // kernel.)
  if (arith == kF16x2 && !force && dev_switches().pw_lat > 0 && tile == 4 && pointwise_latency_supported(a.M, a.K, a.x2 ? a.K1 : 0)) {
    const int e = launch_pointwise_latency(a, st, amax_n);
    if (e >= 0) return e;
  }
  // (tiles measured and dropped, DESIGN_HISTORY.md section 4: 256 x 128 on four wavefronts with two workgroups per CU, 512 x 64,
  // 1024 x 64 -- the tile a CTC head folded into the last GEMM would need --, 128 x 32 for the head, 32 x 32 on one wavefront)
  switch (tile) {
    case 1: return launch_t<8, 2, 4>(a, arith, st, amax_n);
    case 2: return launch_t<8, 1, 4>(a, arith, st, amax_n);
    case 5: return launch_t<8, 1, 2>(a, arith, st, amax_n);
    case 3: return launch_t<4, 1, 2>(a, arith, st, amax_n);
    default: return launch_t<2, 1, 1>(a, arith, st, amax_n);
  }
}

// [cout][cin] row-major fp32 -> [m_pad/32][cin/16][3 planes][64 lanes][8] bf16 bit patterns:
//   lane (l31, kh), element e  <-  plane_p( W[mt*32 + l31][s*16 + 8*kh + e] )
void pack_pointwise_weights_bf16x3(const float* w, int cout, int cin, int m_pad, unsigned short* out) {
  const int ksteps = cin / 16;
  for (int mt = 0; mt < m_pad / 32; ++mt)
    for (int s = 0; s < ksteps; ++s)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int m = mt * 32 + (lane & 31), k = s * 16 + 8 * (lane >> 5) + e;
          const float x = m < cout ? w[(size_t)m * cin + k] : 0.f;
          float hf, mf, lf;
          const unsigned short h = bf16_rne(x, &hf);
          const unsigned short mi = bf16_rne(x - hf, &mf);
          const unsigned short lo = bf16_rne((x - hf) - mf, &lf);
          const size_t base = (((size_t)mt * ksteps + s) * 3) * 64 * 8 + (size_t)lane * 8 + e;
          out[base] = h;
          out[base + 64 * 8] = mi;
          out[base + 2 * 64 * 8] = lo;
        }
}

// Same fragment order with two fp16 planes [m_pad/32][cin/16][2][64][8]: hi = rne16(s w), lo = rne16(s w - hi), s the
// power of two that puts the layer's largest |w| into [2^14, 2^15).  Returns 1/s for the epilogue.
float pack_pointwise_weights_f16x2(const float* w, int cout, int cin, int m_pad, unsigned short* out) {
  float mx = 0.f;
  for (size_t i = 0; i < (size_t)cout * cin; ++i) mx = fabsf(w[i]) > mx ? fabsf(w[i]) : mx;
  int e = (int)(__builtin_bit_cast(unsigned, mx) >> 23);
  e = e < 16 ? 16 : (e > 254 ? 254 : e);
  const float s = __builtin_bit_cast(float, (unsigned)(268 - e) << 23), inv = __builtin_bit_cast(float, (unsigned)(e - 14) << 23);
  const int ksteps = cin / 16;
  auto bits = [](_Float16 h) { return __builtin_bit_cast(unsigned short, h); };
  for (int mt = 0; mt < m_pad / 32; ++mt)
    for (int st = 0; st < ksteps; ++st)
      for (int lane = 0; lane < 64; ++lane)
        for (int el = 0; el < 8; ++el) {
          const int m = mt * 32 + (lane & 31), k = st * 16 + 8 * (lane >> 5) + el;
          const float x = (m < cout ? w[(size_t)m * cin + k] : 0.f) * s;
          const _Float16 h = (_Float16)x;              // round to nearest even
          const _Float16 l = (_Float16)(x - (float)h);
          const size_t base = (((size_t)mt * ksteps + st) * 2) * 64 * 8 + (size_t)lane * 8 + el;
          out[base] = bits(h);
          out[base + 64 * 8] = bits(l);
        }
  return inv;
}

}  // namespace vasr

// ---- sustained-MFMA reference point for the roofline (bench.py: box.measured_mfma_tflops) ---------------------------
// The GEMM's OWN instruction stream with everything but the MFMAs removed, for the arithmetic it is quoted against
// (round 6: like for like -- rounds 4-5 normalised the f16x2 kernel by the bf16x3 stream, six bf16 products on three planes,
// although the rate a dense MFMA stream sustains under the power limit depends on the instruction and on the operand bits):
//   kF16x2   per k-step and (m-tile, n-tile) pair THREE v_mfma_f32_32x32x16_f16: lo*hi, hi*lo, hi*hi on the fp16 hi / lo
//            planes of scaled operands -- 24 per k-step on the 2 x 4 accumulator tile, n-tile by n-tile, the m-tiles
//            alternating inside every product, exactly the order of run_chunk above;
//   kBf16x3  SIX v_mfma_f32_32x32x16_bf16 on the hi / mid / lo bf16 planes -- 48 per k-step.
// 8 wavefronts per workgroup (2 per SIMD), one workgroup per CU, operands resident in registers and distributed like the
// real ones: activations = rectified (half of them zero, as after the ReLU of the producing layer) bell-shaped values
// scaled so that the largest lies in [2^14, 2^15) (the per-utterance scale of the f16 split), weights = bell-shaped values
// of mixed sign scaled likewise.  What this sustains is the ceiling the GEMM's main loop can approach on THIS box; the
// nominal 2.5 PFLOP/s assumes 2.4 GHz, which the part does not hold under dense MFMA load.
namespace vasr {
namespace {
template <int ARITH>
__global__ __launch_bounds__(512) void mfma_sustained_kernel(int steps, float* __restrict__ sink) {
  constexpr int PL = ARITH == kF16x2 ? 2 : 3;
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  auto bell = [](unsigned& s) {                                  // sum of four uniforms: [-2, 2), sigma ~0.58
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s = s * 1664525u + 1013904223u; t += (float)(s >> 8) * (1.0f / 16777216.f); }
    return t - 2.0f;
  };
  const float top = ARITH == kF16x2 ? 16000.f : 4.f;             // f16x2: the scaled operands' largest value is ~2^14
  uint4 af[2][PL], bf[4][PL];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = bell(s) * top;            // weights: mixed sign
    if constexpr (ARITH == kF16x2) split2h(x, 1.0f, af[i][0], af[i][1]);
    else split3(x, af[i][0], af[i][1], af[i][2]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = fmaxf(bell(s), 0.f) * top;   // activations: after a ReLU
    if constexpr (ARITH == kF16x2) split2h(x, 1.0f, bf[j][0], bf[j][1]);
    else split3(x, bf[j][0], bf[j][1], bf[j][2]);
  }
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < steps; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (ARITH == kBf16x3) {
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = mma<ARITH>(af[i][2], bf[j][0], acc[i][j]);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = mma<ARITH>(af[i][0], bf[j][2], acc[i][j]);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = mma<ARITH>(af[i][1], bf[j][1], acc[i][j]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = mma<ARITH>(af[i][1], bf[j][0], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = mma<ARITH>(af[i][0], bf[j][1], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = mma<ARITH>(af[i][0], bf[j][0], acc[i][j]);
    }
    // (accumulators that run to infinity would change the bits the pipe toggles: fold them back now and then)
    if ((it & 255) == 255) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] *= 0x1p-40f;
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[i][j][r];
  if (t == 12345.678f) sink[0] = t;   // keeps the accumulators alive
}
}  // namespace

// launches one workgroup per CU (n_cu of them) of the stream of `gemm_mode` (vasr.h vasr_set_gemm_mode: 1 = bf16x3,
// 3 = f16x2); returns the 16-bit flops issued, 0 for a mode that has no 16-bit stream
double launch_mfma_sustained(int gemm_mode, int n_cu, int steps, float* sink, hipStream_t st) {
  if (gemm_mode == 3) {
    hipLaunchKernelGGL(mfma_sustained_kernel<kF16x2>, dim3(n_cu), dim3(512), 0, st, steps, sink);
    return (double)n_cu * 8 * steps * 24 * (2.0 * 32 * 32 * 16);
  }
  if (gemm_mode == 1) {
    hipLaunchKernelGGL(mfma_sustained_kernel<kBf16x3>, dim3(n_cu), dim3(512), 0, st, steps, sink);
    return (double)n_cu * 8 * steps * 48 * (2.0 * 32 * 32 * 16);
  }
  return 0.0;
}
}  // namespace vasr
