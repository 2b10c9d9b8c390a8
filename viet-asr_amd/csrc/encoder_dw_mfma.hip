// Depthwise masked 1-D convolution on the MATRIX pipe (gfx950): the Toeplitz form.
//
// Same contract as encoder_dw.hip (reference nemo/collections/asr/parts/jasper.py:113-132 mask + conv, :360-373 layer
// order): masked input, output zeroed past lens_out, every column < ldy written.  The packed-FMA kernels there are
// VALU-issue bound from K = 51 up (600 v_pk_fma_f32 per wavefront-pass at K = 75: 23 us of pure issue per 512-channel
// layer against 17-20 us of copy-speed traffic).  Here the taps of ONE channel are laid out as a banded matrix
//
//     A[m][j] = w[(j - m - OFF) / DIL]     (0 where that is not a tap)        m = 0..31 output offset inside a window
//
// and 32 windows of 32 consecutive outputs -- 16 windows = 512 frames of each of TWO utterances -- are its right-hand
// sides: B[j][n] = x_n[t_n - PADL + j], eight consecutive samples per lane and k-step, so every B fragment is one
// aligned 16-byte LDS read.  One v_mfma_f32_32x32x16 then produces 32 x 32 outputs from a 16-sample slice of the windows:
// a K = 75 kernel spans 31 + 74 + OFF + 1 = 109 window samples = 7 k-steps.  Arithmetic as in the fp16-split GEMM
// (encoder_pw_split.hip, kF16x2): operands scaled by a power of two (taps per channel, samples per utterance -- from the
// maxima the producing GEMM published) and split into two fp16 terms, three products per k-step, fp32 accumulation:
// 21 MFMAs for 1024 outputs x 75 taps instead of 600 packed FMAs, 5 us of matrix time per 512-channel layer.  What is
// left is the HBM traffic: each sample is read once (16-byte coalesced loads), converted once, and every store
// instruction writes 1 KB of one row (the accumulators are transposed through LDS).
//
//   grid (C / 4, ceil(pairs / kPairsPerWave), ceil(ldy / 512)), block 256 = 4 wavefronts = 4 channels; a wavefront builds
//   its channel's A fragments once (from a [hi | lo] fp16 tap table packed at vasr_finalize()) and walks kPairsPerWave
//   utterance pairs with them, the next pair's rows in flight while the current pair is multiplied.
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

#ifndef VASR_TZ_ABLATE
#define VASR_TZ_ABLATE 0   // dev-only timing ablations (results are wrong): 1 no MFMAs, 2 no global stores, 4 no conversion /
#endif                     // LDS staging, 8 no row loads, 16 no transposition epilogue at all, 32 no maxima publishing
constexpr int kTile = 512;          // output frames per utterance and task: 16 windows of 32
// Utterance pairs one wavefront walks with its channel's A fragments.  8: a 64-utterance batch gives 2048 wavefronts per
// 512-channel layer, all resident at once (8 per CU), each streaming its pairs with the next TWO pairs' rows in flight --
// with 2 (and one pair ahead) the kernel took 30 us per layer whatever K: every wavefront paid the fragment set-up and an
// exposed HBM round trip per pair.
constexpr int kPairsPerWave = 8;   // upper bound; the launch passes the count in use (VASR_DW_PPW, default below)

template <int K, int DIL>
struct TzGeom {
  static constexpr int PAD = DIL > 1 ? (DIL * K) / 2 - 1 : K / 2;   // get_same_padding (jasper.py:60-65)
  static constexpr int PADL = (PAD + 7) & ~7;                       // window origin: 8-sample aligned
  static constexpr int OFF = PADL - PAD;
  static constexpr int SPAN = 31 + DIL * (K - 1) + OFF + 1;         // window samples one 32-output window touches
  static constexpr int NS = (SPAN + 15) / 16;                       // k-steps
  static constexpr int TSZ = (16 * NS + 31 + 3) & ~3;               // tap-table entries: entry i = dense tap i - (31 + OFF)
  static constexpr int NBLK = 15 + (NS + 1) / 2;                    // 32-sample blocks of one staged row
  static constexpr int ROWS = 32 * NBLK;                            // samples staged per utterance
  static constexpr int NLD = (ROWS / 4 + 63) / 64;                  // float4 loads per lane and utterance
  static constexpr int UROW = (80 * NBLK + 255) & ~255;             // bytes of one plane of one utterance (80 B per block)
  static constexpr int LDS_DATA = 2 * 2 * UROW;                     // [utterance][plane]
  static constexpr int LDS_TAB = 4 * TSZ;
  static constexpr int ORS = 144;                                   // output transposition: bytes per 32-float row
  static constexpr int LDS_OUT = 32 * ORS;
  static constexpr int LDS_SCL = 16 * kPairsPerWave;                // per pair: (scale, 1 / scale) of both utterances
  static constexpr int LDS_WAVE = LDS_DATA + LDS_TAB + LDS_OUT + LDS_SCL;
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ f32x16 mma(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int K, int DIL>
__global__ __launch_bounds__(256) void dw_toeplitz_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const unsigned* __restrict__ taps,     // [C][TSZ] (hi | lo << 16)
                                                          const float* __restrict__ tap_inv,     // [C] 1 / tap scale
                                                          const int32_t* __restrict__ lens_in,
                                                          const int32_t* __restrict__ lens_out,
                                                          const unsigned* __restrict__ amax_x, int amax_x_stride,
                                                          int amax_x_n, int channels, int batch,
                                                          float* __restrict__ y, int64_t ldy,
                                                          unsigned* __restrict__ amax_y, int amax_y_stride, int ppw) {
  using G = TzGeom<K, DIL>;
  constexpr int NS = G::NS, NLD = G::NLD;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* base = lds_raw + wave * G::LDS_WAVE;
  unsigned char* dat = base;                                               // [u][plane][UROW]
  unsigned* tab = reinterpret_cast<unsigned*>(base + G::LDS_DATA);
  unsigned char* outb = base + G::LDS_DATA + G::LDS_TAB;
  float4* scl = reinterpret_cast<float4*>(base + G::LDS_DATA + G::LDS_TAB + G::LDS_OUT);
  const int c = blockIdx.x * 4 + wave;
  const int t_tile = blockIdx.z * kTile;
  const int l31 = lane & 31, kh = lane >> 5;
  const int n_pairs = (batch + 1) / 2;
  const int p_lo = blockIdx.y * ppw;
  const int p_hi = min(p_lo + ppw, n_pairs);

  uint4 ah[NS], al[NS];   // A fragments of this channel (built below, after the first rows have been requested)
  const float w_inv = tap_inv[c];

  // ---- staging of one utterance pair: both rows, branch-free, all loads in flight together; the producers' maxima of
  //      the two utterances ride along (one word per lane and 64 slots) ----
  // The kernel is instruction-issue bound (round-2 ablations: ~500 instructions per pair at ~4.5 cycles each), so every
  // mask and address computation that hardware can do is handed to it: rows are read through raw buffer descriptors
  // whose range is the utterance's length -- MaskedConv1d's x.masked_fill(t >= lens, 0) (jasper.py:113-118) and the
  // conv's zero padding left of frame 0 (a negative offset is a huge unsigned one) both come back as zeros, per dword.
  struct Stage { v4f r0[NLD], r1[NLD]; };
  const int64_t row_stride = (int64_t)channels * ldx;                 // floats between utterances of one channel
  const float* xrow = x + ((int64_t)(2 * p_lo) * channels + c) * ldx;   // row of the first utterance of pair p_lo
  const int64_t yrow_stride = (int64_t)channels * ldy;
  float* yrow0 = y + ((int64_t)(2 * p_lo) * channels + c) * ldy;
  auto gload = [&](int p, Stage& sg) {
    const int b0 = 2 * p, b1 = b0 + 1 < batch ? b0 + 1 : b0;
    const float* xr0 = xrow + (int64_t)(b0 - 2 * p_lo) * row_stride;
    const float* xr1 = xrow + (int64_t)(b1 - 2 * p_lo) * row_stride;
    // (lengths through the scalar cache: the descriptor words must be wave-uniform for the compiler, or every load
    // becomes a waterfall loop)
    const auto d0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xr0), 0, 4 * min(lens_in[b0], (int)ldx), 0x00020000);
    const auto d1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xr1), 0, 4 * min(lens_in[b1], (int)ldx), 0x00020000);
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int off = 4 * (t_tile - G::PADL + 4 * (lane + 64 * j));
      if (!(VASR_TZ_ABLATE & 8)) {
        sg.r0[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(d0, off, 0, 0));
        sg.r1[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(d1, off, 0, 0));
      } else {
        sg.r0[j] = v4f{(float)off, 1.f, 2.f, 3.f};
        sg.r1[j] = v4f{(float)off, 3.f, 2.f, 1.f};
      }
    }
  };
  // scale by the utterance's power of two, split into fp16 hi / lo, 8 + 8 bytes into the two planes
  auto sstore = [&](const v4f (&sv)[NLD], int u, float sx) {
    unsigned char* ph = dat + (u * 2 + 0) * G::UROW;
    unsigned char* pl = dat + (u * 2 + 1) * G::UROW;
    // Branch-free: a lane past the staged row (last slab only) writes into the 16 spare bytes behind it.  With a
    // lane-dependent branch here the compiler cannot tell, where the paths merge, whether the skipped lanes' loads have
    // been waited for, and drains the vector-memory counter (vmcnt(0)) before the next prefetch overwrites the stage
    // registers -- which also waits for the OTHER stage's rows and for the previous pair's stores: no prefetch at all.
    static_assert(G::UROW - 80 * G::NBLK >= 16, "no spare bytes behind the staged row");
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int tau = 4 * (lane + 64 * j);
      {
        const v4f v = sv[j];
        const v2f a = {v.x * sx, v.y * sx}, b = {v.z * sx, v.w * sx};
        const f16x2 ha = __builtin_convertvector(a, f16x2), hb = __builtin_convertvector(b, f16x2);
        const f16x2 la = __builtin_convertvector(a - __builtin_convertvector(ha, v2f), f16x2);
        const f16x2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, v2f), f16x2);
        const int off = tau < G::ROWS ? 80 * (tau >> 5) + 2 * (tau & 31) : 80 * G::NBLK + 8 * (lane & 1);
        *reinterpret_cast<uint2*>(ph + off) = make_uint2(__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb));
        *reinterpret_cast<uint2*>(pl + off) = make_uint2(__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb));
      }
    }
  };

  // B fragments: lane (n' = l31, kh): utterance u = n' / 16, window n = n' % 16 -> samples 32 n + 16 s + 8 kh + e
  const unsigned char* bb = dat + (l31 >> 4) * 2 * G::UROW + 80 * (l31 & 15) + 16 * kh;
  unsigned char* orow = outb + l31 * G::ORS + 16 * kh;   // transposition: lane writes row n', floats 8 q + 4 kh .. + 3

  // One pair: convert the staged rows, refill the stage with the pair two ahead, multiply, store.
  auto do_pair = [&](int p, Stage& sg) {
    const int b0 = 2 * p;
    const bool twin = b0 + 1 < batch;
    const int b1 = twin ? b0 + 1 : b0;
    const float4 sc = scl[p - p_lo];   // (scale, 1 / scale) of both utterances: LDS broadcast, no vector-memory wait
    const float sx0 = sc.x, ix0 = sc.y, sx1 = sc.z, ix1 = sc.w;
    if (!(VASR_TZ_ABLATE & 4)) {
      sstore(sg.r0, 0, sx0);
      sstore(sg.r1, 1, sx1);
    } else {
      asm volatile("" :: "v"(sg.r0[0]), "v"(sg.r1[NLD - 1]));
    }
    // The pair two ahead, in flight while this pair and the next are multiplied and stored.  UNCONDITIONAL (past the end
    // the last pair is requested again): s_waitcnt vmcnt counts outstanding operations, so the compiler can only leave
    // the other stage's rows in flight if it knows how many younger loads there are on every path.
    gload(min(p + 2, p_hi - 1), sg);
    wave_sync();

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int off = 80 * (s >> 1) + 32 * (s & 1);
      const uint4 bh = *reinterpret_cast<const uint4*>(bb + off);
      const uint4 bl = *reinterpret_cast<const uint4*>(bb + G::UROW + off);
      if (!(VASR_TZ_ABLATE & 1)) {
        acc = mma(al[s], bh, acc);
        acc = mma(ah[s], bl, acc);
        acc = mma(ah[s], bh, acc);
      } else {
        acc[s & 15] += __uint_as_float(bh.x ^ bl.y ^ al[s].x ^ ah[s].y);
      }
    }

    // ---- epilogue: unscale, transpose through LDS, zero past lens_out, 1 KB of one row per store instruction ----
    const float os = w_inv * (l31 < 16 ? ix0 : ix1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v4f o = {acc[4 * q] * os, acc[4 * q + 1] * os, acc[4 * q + 2] * os, acc[4 * q + 3] * os};
      *reinterpret_cast<v4f*>(orow + 32 * q) = o;
    }
    wave_sync();
    // Read back as rows: lanes 0-31 take utterance 0 (windows 0-15), lanes 32-63 utterance 1 -- 2 x 512 contiguous
    // bytes per store instruction, and ONE row-wise reduction yields both utterances' maxima (rows 0-1 / rows 2-3).
    const int u = lane >> 5;
    const int lo0 = lens_out[b0], lo1 = lens_out[b1];   // scalar loads
    const int lo_u = u ? lo1 : lo0;
    float* y0 = yrow0 + (int64_t)(b0 - 2 * p_lo) * yrow_stride;
    float* y1 = yrow0 + (int64_t)(b1 - 2 * p_lo) * yrow_stride;
    float* yrow = u ? y1 : y0;
    const bool live = u == 0 || twin;
    float mx = 0.f;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      // float4 index (lane & 31) + 32 ps inside the utterance's 16 x 8 float4 -> row (window) of the block, column 4 (idx % 8)
      const int idx = (lane & 31) + 32 * ps, row = idx >> 3;
      v4f v = *reinterpret_cast<const v4f*>(outb + (16 * u + row) * G::ORS + 16 * (idx & 7));
      const int t = t_tile + 32 * row + 4 * (idx & 7);
      const int nv = lo_u - t;
      v.x = nv > 0 ? v.x : 0.f;
      v.y = nv > 1 ? v.y : 0.f;
      v.z = nv > 2 ? v.z : 0.f;
      v.w = nv > 3 ? v.w : 0.f;
      if (t < ldy && live && (!(VASR_TZ_ABLATE & 2) || v.x == 12345.678f)) {
        *reinterpret_cast<v4f*>(yrow + t) = v;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      }
    }
    if (amax_y && !(VASR_TZ_ABLATE & 32)) {
      unsigned m = __float_as_uint(mx);   // |x| bit patterns order like unsigned integers
#define VASR_DPP(xx, ctrl) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(xx), (ctrl), 0xF, 0xF, false))
      m = max(m, VASR_DPP(m, 0xB1));
      m = max(m, VASR_DPP(m, 0x4E));
      m = max(m, VASR_DPP(m, 0x141));
      m = max(m, VASR_DPP(m, 0x140));
#undef VASR_DPP
      const unsigned m0 = max((unsigned)__builtin_amdgcn_readlane((int)m, 0), (unsigned)__builtin_amdgcn_readlane((int)m, 16));
      const unsigned m1 = max((unsigned)__builtin_amdgcn_readlane((int)m, 32), (unsigned)__builtin_amdgcn_readlane((int)m, 48));
      const int slot = c * gridDim.z + blockIdx.z;
      if (lane == 0) {
        amax_y[(int64_t)b0 * amax_y_stride + slot] = m0;
        if (twin) amax_y[(int64_t)b1 * amax_y_stride + slot] = m1;
      }
    }
    wave_sync();   // the next pair's staging overwrites the rows, its epilogue the transposition buffer
  };

  if (p_lo >= p_hi) return;
  Stage sa, sb;
  gload(p_lo, sa);
  gload(min(p_lo + 1, p_hi - 1), sb);

  // tap table of this channel: requested now, so that it shares the flight of the rows and of the maxima below
  constexpr int NTL = (G::TSZ + 63) / 64;
  unsigned tapv[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j) tapv[j] = lane + 64 * j < G::TSZ ? taps[(int64_t)c * G::TSZ + lane + 64 * j] : 0u;

  // ---- scales of every utterance this wavefront will touch, once: the reduction of the producers' slots is a loop of
  //      dependent loads (a wait for ALL outstanding vector-memory operations -- inside the pair loop it would also wait
  //      for the rows just requested two pairs ahead); here it overlaps the first rows' flight ----
  for (int p = p_lo; p < p_hi; ++p) {
    const int b0 = 2 * p, b1 = b0 + 1 < batch ? b0 + 1 : b0;
    float4 sc;
    f16_scale(amax_read(amax_x, amax_x_stride, amax_x_n, b0, lane), &sc.x, &sc.y);
    f16_scale(amax_read(amax_x, amax_x_stride, amax_x_n, b1, lane), &sc.z, &sc.w);
    if (lane == 0) scl[p - p_lo] = sc;
  }

  // ---- A fragments of this channel: lane (m = l31, kh), step s, element e <- table[31 - m + 16 s + 8 kh + e] ----
#pragma unroll
  for (int j = 0; j < NTL; ++j)
    if (lane + 64 * j < G::TSZ) tab[lane + 64 * j] = tapv[j];
  wave_sync();
  {
    const unsigned* tp = tab + 31 - l31 + 8 * kh;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      unsigned v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tp[16 * s + e];
      // (hi | lo << 16) pairs -> four dwords of hi halves, four of lo halves (v_perm_b32 each)
      ah[s] = make_uint4(__builtin_amdgcn_perm(v[1], v[0], 0x05040100u), __builtin_amdgcn_perm(v[3], v[2], 0x05040100u),
                         __builtin_amdgcn_perm(v[5], v[4], 0x05040100u), __builtin_amdgcn_perm(v[7], v[6], 0x05040100u));
      al[s] = make_uint4(__builtin_amdgcn_perm(v[1], v[0], 0x07060302u), __builtin_amdgcn_perm(v[3], v[2], 0x07060302u),
                         __builtin_amdgcn_perm(v[5], v[4], 0x07060302u), __builtin_amdgcn_perm(v[7], v[6], 0x07060302u));
    }
  }

  // two pairs per trip, both always executed (same reason as above); with an odd count the last pair is simply computed
  // and stored twice
  for (int p = p_lo; p < p_hi; p += 2) {
    do_pair(p, sa);
    do_pair(min(p + 1, p_hi - 1), sb);
  }
}

template <int K, int DIL>
int launch_tz(const float* x, int64_t ldx, const unsigned* taps, const float* tap_inv, const int32_t* li, const int32_t* lo,
              AmaxTab amax_x, int batch, int channels, float* y, int64_t ldy, AmaxTab* amax_y, hipStream_t st) {
  using G = TzGeom<K, DIL>;
  auto kern = dw_toeplitz_kernel<K, DIL>;
  constexpr int lds = 4 * G::LDS_WAVE;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (attr != hipSuccess) return (int)attr;
  const int n_pairs = (batch + 1) / 2;
  static const int ppw_env = getenv("VASR_DW_PPW") ? atoi(getenv("VASR_DW_PPW")) : 0;
  const int ppw = ppw_env >= 1 && ppw_env <= kPairsPerWave ? ppw_env : 8;
  dim3 grid(channels / 4, (n_pairs + ppw - 1) / ppw, (unsigned)((ldy + kTile - 1) / kTile));
  if (amax_y) {
    amax_y->n = channels * grid.z;
    if (amax_y->n > amax_y->stride) return -1;
  }
  VASR_LAUNCH(kern, grid, dim3(256), lds, st, x, ldx, taps, tap_inv, li, lo, amax_x.p, amax_x.stride, amax_x.n, channels, batch,
              y, ldy, amax_y ? amax_y->p : nullptr, amax_y ? amax_y->stride : 0, ppw);
  return 0;
}

template <int K, int DIL>
int table_size() { return TzGeom<K, DIL>::TSZ; }

}  // namespace

// (kernel, dilation) pairs with a Toeplitz instantiation; 0 = none
int depthwise_mfma_table_size(int kernel, int dilation) {
  if (dilation == 2 && kernel == 87) return table_size<87, 2>();
  if (dilation != 1) return 0;
  switch (kernel) {
    case 33: return table_size<33, 1>();
    case 39: return table_size<39, 1>();
    case 51: return table_size<51, 1>();
    case 63: return table_size<63, 1>();
    case 75: return table_size<75, 1>();
    default: return 0;
  }
}

// Host: one channel's taps w[K] -> table[tsz] of (hi | lo << 16) fp16 pairs of the scaled DENSE tap sequence
// (entry i = dense tap i - (31 + OFF), dense tap k * DIL = w[k]); returns 1 / scale.
float pack_depthwise_taps_f16x2(const float* w, int kernel, int dilation, int tsz, unsigned* table) {
  const int pad = dilation > 1 ? (dilation * kernel) / 2 - 1 : kernel / 2;
  const int off = ((pad + 7) & ~7) - pad;
  float mx = 0.f;
  for (int k = 0; k < kernel; ++k) mx = fabsf(w[k]) > mx ? fabsf(w[k]) : mx;
  int e = (int)(__builtin_bit_cast(unsigned, mx) >> 23);
  e = e < 16 ? 16 : (e > 254 ? 254 : e);
  const float s = __builtin_bit_cast(float, (unsigned)(268 - e) << 23), inv = __builtin_bit_cast(float, (unsigned)(e - 14) << 23);
  for (int i = 0; i < tsz; ++i) {
    const int d = i - (31 + off);
    float v = 0.f;
    if (d >= 0 && d % dilation == 0 && d / dilation < kernel) v = w[d / dilation] * s;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    table[i] = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
  }
  return inv;
}

// Returns 0, a hipError_t, or -1 when the shape has no Toeplitz instantiation (caller falls back to encoder_dw.hip).
int launch_depthwise_mfma(const float* x, int64_t ldx, const unsigned* taps, const float* tap_inv, const int32_t* lens_in,
                          const int32_t* lens_out, AmaxTab amax_x, int batch, int channels, int kernel, int dilation,
                          float* y, int64_t ldy, AmaxTab* amax_y, hipStream_t st) {
  const bool aligned = channels % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= 4 &&
                       (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
  if (!aligned || !taps || !amax_x.p || amax_x.n <= 0) return -1;
#define TZ(KK, DD) return launch_tz<KK, DD>(x, ldx, taps, tap_inv, lens_in, lens_out, amax_x, batch, channels, y, ldy, amax_y, st)
  if (dilation == 2 && kernel == 87) TZ(87, 2);
  if (dilation == 1) {
    switch (kernel) {
      case 33: TZ(33, 1);
      case 39: TZ(39, 1);
      case 51: TZ(51, 1);
      case 63: TZ(63, 1);
      case 75: TZ(75, 1);
      default: break;
    }
  }
#undef TZ
  return -1;
}

}  // namespace vasr
