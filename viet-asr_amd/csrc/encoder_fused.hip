// Fused depthwise + pointwise sub-block of a 256-channel JasperBlock (gfx950): ONE kernel for
//
//     MaskedConv1d(C, C, K, groups = C)  ->  MaskedConv1d(C, C, 1)  ->  BatchNorm (eval)  [-> + BN(res 1x1 conv)]  ->  ReLU
//
// (reference nemo/collections/asr/parts/jasper.py:360-373 `_get_conv_bn_layer`, masks :108-132, block forward :408-448) --
// the "fused depthwise + pointwise + BN + ReLU + residual" kernel north_star names, for the shapes where it pays: the
// C = 256 blocks of QuartzNet (K = 33 / 39; 30 of the 75 separable sub-blocks of 15x5).  The depthwise output never goes
// to HBM: 33.5 MB written + 33.5 MB read again per sub-block at 64 x 10 s, one launch and its dispatch gap.
//
// Structure (numbers for the 128-frame tile; the 64-frame form of round 4 -- template parameter BN, FTile -- halves the
// columns of every role: 8 frames per producer lane, 2 x 2 MFMA tiles per consumer, 110 KB of LDS; it is the form for
// batches whose 128-frame tiles would leave the chip half empty, vasr_api.cpp run_encoder):
// a workgroup owns ALL 256 input and output channels of a 128-frame time tile and is split by ROLE
// (wave specialisation -- one kernel, two instruction streams, no compiler-interleaved software pipeline):
//
//   wavefronts 4..7  PRODUCERS, vector ALU.  Per 64-channel chunk each takes 16 channels = 8 channel pairs.  The masked
//     input window (128 + K - 1 frames, 16-byte aligned origin) of its 16 rows goes through a wavefront-private LDS
//     window with the two channels of a pair interleaved, so that every v_pk_fma_f32 computes the same (frame, tap) of
//     both channels; lane = (pair, 16 consecutive frames): a sliding register window, 16 x K packed FMAs, taps of the
//     pair from an LDS table as 64-bit operands.  The result is scaled by the utterance's power of two, split into fp16
//     hi / lo (encoder_pw_split.hip kF16x2) and written straight into the GEMM's B-fragment image in LDS.
//   wavefronts 0..3  CONSUMERS, matrix pipe.  64 output rows x 128 columns each (2 x 4 tiles of 32x32x16, 128
//     accumulators): per 16-deep k-step 4 weight fragments from L2 (the f16x2 pack of vasr_finalize, two k-steps ahead),
//     8 B fragments from LDS, 24 MFMAs.  No staging, no conversion, no address arithmetic in this stream.
//   One LDS-only barrier per chunk hands a finished B image over (double buffered): 4 barriers per tile (8 with a
//   residual source).  The residual branch of the block's last sub-block is a second K range [256, 512) as in the
//   unfused kernel (pack_fused_residual): its chunks are produced by plain conversion of the masked block input.
//
// The fp16 split needs the utterance's scale BEFORE the depthwise output exists, so it comes from a bound instead of
// the measured maximum: max |x| of the utterance (published by x's producer, AmaxTab) times the layer's largest
// sum_k |w[c][k]| (FusedArgs::dw_l1, a constant from vasr_finalize).  The bound is 2^2 .. 2^4 loose, i.e. the split
// carries 18-20 bits where the measured maximum would give 22: still below an fp32 dot product's own rounding
// (tools/split_scale_study.py; tests hold the fused path to the same tolerance as every other arithmetic).
//
// LDS (159 744 of 163 840 bytes, one workgroup per CU by design): B images 2 x 32 KB | producer windows 4 x 13 KB
// (reused by the epilogue's transposition) | tap table [128 pairs][K + 1][2] f32.
#include <cstdlib>

#include "vasr_internal.h"
#include "vasr_device.h"
#include <type_traits>

namespace vasr {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using v4f = __attribute__((ext_vector_type(4))) float;
using v2f = __attribute__((ext_vector_type(2))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

constexpr int FC = 256;     // channels in = out
constexpr int FCH = 64;     // channels per chunk (4 k-steps of 16)
constexpr int FNT = 512;    // threads: 4 consumer + 4 producer wavefronts

// Tile width BN (columns = frames per workgroup): 128 where such tiles fill the chip, 64 for the batches in between (24-47
// utterances of 10 s: 192-376 tiles of 64 frames against 96-188 of 128).  Everything below that depends on it:
template <int BN>
struct FTile {
  static_assert(BN == 128 || BN == 64, "tile width");
  static constexpr int FPS = BN / 8;                       // frames per producer lane (8 segments per channel pair): 16 / 8
  static constexpr int NT = BN / 32;                       // 32-column MFMA tiles of a consumer wavefront: 4 / 2
  static constexpr int KHS = BN == 128 ? 3 : 2;            // bit of the k-half in the B image's column swizzle
  static constexpr int kBimgBytes = 2 * 2 * 4 * 2 * BN * 16;   // [buf][plane][k-step][k-half][column] x 16 B = 65536 / 32768
  static constexpr int kWPitch = BN == 128 ? 1664 : 1152;  // bytes per channel pair of a producer window (= 128 mod 256)
  static constexpr int kWWave = 8 * kWPitch;               // 13312 / 9216
  static constexpr int kWinBytes = 4 * kWWave;             // 53248 / 36864
  static constexpr int kSegBytes = FPS * 8 + 16;           // a lane's segment of its pair's window row incl. its pad: 144 / 80
};

template <int K, int BN>
struct FGeom {
  using TL = FTile<BN>;
  static constexpr int FPS = TL::FPS;
  static constexpr int PAD = K / 2;                              // get_same_padding, stride 1, dilation 1 (jasper.py:60-65)
  static constexpr int PADL = (PAD + 3) & ~3;                    // window origin t0 - PADL: 16-byte aligned in x
  static constexpr int OFF = PADL - PAD;
  static constexpr int W = (PADL + BN + (K - 1 - PAD) + 3) & ~3;    // window frames: 160 (K = 33), 168 (K = 39); 96 / 104 at BN = 64
  static constexpr int W4 = W / 4;
  static constexpr int NLD = (8 * W4 + 63) / 64;                 // (pair, float4-column) items per lane: 5 / 6 (3 / 4)
  static constexpr int NF = OFF + FPS - 1 + K;                   // frames a lane reads from its region's start
  static constexpr int NR = (NF + 1) / 2;                        // ds_read_b128 per lane (two interleaved frames each)
  static constexpr int KP = (K + 2) & ~1;                        // taps per pair in the LDS table (zero padded, even)
  static constexpr int TB = 8;                                   // taps per block of the sliding window
  static constexpr int NB = (K + TB - 1) / TB;
  static constexpr int kTapBytes = (FC / 2) * KP * 8;
  static constexpr size_t LDS = (size_t)TL::kBimgBytes + TL::kWinBytes + kTapBytes;
  static_assert(W * 8 + ((W + FPS - 1) / FPS) * 16 <= TL::kWPitch, "window row does not fit its pitch");
  static_assert(FPS * 7 + 2 * NR <= W, "a lane's reads leave the window");
  static_assert(NLD >= FPS / 4, "the residual rows of a lane use the staging registers of the window items");
  static_assert(LDS <= 163840, "LDS budget");
  // last b128 unit (exclusive) a tap block needs: frames < OFF + min(K, TB (blk + 1)) + FPS - 1
  static constexpr int qend(int blk) {
    const int k1 = TB * (blk + 1) < K ? TB * (blk + 1) : K;
    return (OFF + k1 + FPS - 1 + 1) / 2;
  }
};

struct FusedArgs {
  const float* x;            // [B][256][ldx] depthwise input
  int64_t ldx;
  const int32_t* lens_in;    // depthwise input mask
  const int32_t* lens_out;   // depthwise output (= pointwise input) mask
  const float* taps;         // [128 pairs][KP][2] depthwise taps, channel pairs interleaved
  float dw_l1;               // max_c sum_k |w[c][k]|
  AmaxTab amax_x;
  const uint4* wt;           // f16x2 A-fragment pack [256/32][Ktot/16][2][64] uint4
  float w_inv_scale;
  const float* scale;        // [256] BN affine (1 / h1 + h2 when the residual is folded into the weights)
  const float* shift;
  float* y;                  // [B][256][ldy]
  int64_t ldy;
  int32_t frames;            // valid columns
  int32_t relu;
  AmaxTab amax_y;
  const int32_t* lens_y;
  // residual source (DUAL): K range [256, 512) = masked block input
  const float* x2;
  int64_t ldx2;
  const int32_t* lens2;
  AmaxTab amax_x2;
  int32_t batch;
};

// LDS-only workgroup barrier: __syncthreads() would also drain the vector-memory counter, i.e. make the producers wait
// for the rows they have just requested for the NEXT chunk
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// B image: element (plane, k-step ks, k-half kh, column n) is one 16-byte slot (8 consecutive channels of one frame).
// The column index is swizzled inside its aligned group of 16 -- low4 ^= (n >> 4) | (kh << KHS) -- so that the producers'
// 4-byte writes (8 lanes of one pair sit 16 (8) columns apart: the same bank(s) unswizzled) and the consumers' 16-byte
// reads are both conflict free.  (BN = 64: the 8 lanes of a pair give low4 ^ j = {0, 8, 1, 9, 2, 10, 3, 11}, the other
// k-half the same set ^ 4.)
template <int BN>
__device__ __forceinline__ int bimg_slot(int n, int kh) { return (n & ~15) | ((n & 15) ^ ((n >> 4) | (kh << FTile<BN>::KHS))); }
template <int BN>
__device__ __forceinline__ constexpr int bimg_off(int buf, int plane, int ks) { return (((buf * 2 + plane) * 4 + ks) * 2) * BN * 16; }

template <int K, bool DUAL, int BN>
__global__ __launch_bounds__(FNT, 1) void dwpw_fused_kernel(FusedArgs a, int tiles_t, int n_blocks) {
  using G = FGeom<K, BN>;
  using TL = FTile<BN>;
  constexpr int FPS = TL::FPS, NT = TL::NT, kWPitch = TL::kWPitch, kWWave = TL::kWWave;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* bimg = smem;
  unsigned char* wins = smem + TL::kBimgBytes;
  unsigned char* tapl = smem + TL::kBimgBytes + TL::kWinBytes;

  int bid = blockIdx.x;
  {   // XCD-aware order: consecutive tiles of an utterance on one XCD (its L2 then serves the neighbouring windows' halo)
    const int q = n_blocks / 8, r = n_blocks % 8, xcd = bid % 8, slot = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int b = bid / tiles_t;
  const int tile = bid % tiles_t;
  const int t0 = tile * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // role index: < 4 consumes
  const int len_in = a.lens_in[b], len_out = a.lens_out[b];
  const int len2 = DUAL ? a.lens2[b] : 0;
  constexpr int NCH = (DUAL ? 2 : 1) * (FC / FCH);

  // every source is zero from zf on: such a tile multiplies nothing, its outputs are relu(shift)
  const int zf = max(len_out, DUAL ? len2 : 0);
  const bool live = t0 < zf;

  // ---- the utterance's fp16 scale, from the bound max|x| * dw_l1 (and max|x2|) ----
  float xs = 1.f, out_scale = 1.f;
  unsigned amv[8], amv2[8];
  if (live) {
    amax_request(a.amax_x.p, a.amax_x.stride, a.amax_x.n, b, lane, amv);
    if (DUAL) amax_request(a.amax_x2.p, a.amax_x2.stride, a.amax_x2.n, b, lane, amv2);
  }
  // tap table -> LDS, all wavefronts (read by the producers after the first barrier).  Called by each role AFTER it has
  // requested its first operands (rows from HBM, weight fragments from L2): the table's trip to L2 then runs inside
  // those round trips instead of in front of them (1.6 us of a 31 us kernel when it came first).
  constexpr int kTapV4 = G::kTapBytes / 16, kTapIt = (kTapV4 + FNT - 1) / FNT;
  auto tap_copy = [&]() {
    v4f tv[kTapIt];
#pragma unroll
    for (int i = 0; i < kTapIt; ++i) {
      const int idx = tid + i * FNT;
      tv[i] = reinterpret_cast<const v4f*>(a.taps)[idx < kTapV4 ? idx : kTapV4 - 1];
    }
#pragma unroll
    for (int i = 0; i < kTapIt; ++i) {
      const int idx = tid + i * FNT;
      if (idx < kTapV4) reinterpret_cast<v4f*>(tapl)[idx] = tv[i];
    }
  };

  if (wave < 4) {
    // =========================================== consumers =========================================================
    const int kh = lane >> 5, l31 = lane & 31;
    const int wm = wave * 64;
    constexpr int ksteps = NCH * 4;
    const uint4* __restrict__ ap = a.wt + ((int64_t)(wm / 32) * ksteps) * 2 * 64 + lane;
    const int64_t a_tile = (int64_t)ksteps * 2 * 64;
    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (live) {
      // B-fragment addresses of the n-tiles (swizzled column, k-half): constant over the whole tile
      int boff[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) boff[j] = (kh * BN + bimg_slot<BN>(32 * j + l31, kh)) * 16;
      uint4 aw[4][2][2];   // weight fragments, set = k-step % 4, requested three k-steps ahead: [set][m-tile][plane]
      auto aload = [&](int s, uint4 (&dst)[2][2]) {
        const int sc = s < ksteps ? s : ksteps - 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int p = 0; p < 2; ++p) dst[i][p] = ap[i * a_tile + ((int64_t)sc * 2 + p) * 64];
      };
      aload(0, aw[0]);
      aload(1, aw[1]);
      aload(2, aw[2]);
      tap_copy();
      {
        unsigned mx = amax_collect(a.amax_x.p, a.amax_x.stride, a.amax_x.n, b, lane, amv);
        mx = __float_as_uint(__uint_as_float(mx) * a.dw_l1);
        if (DUAL) mx = max(mx, amax_collect(a.amax_x2.p, a.amax_x2.stride, a.amax_x2.n, b, lane, amv2));
        float inv;
        f16_scale(mx, &xs, &inv);
        out_scale = inv * a.w_inv_scale;
      }
      lds_barrier();   // tap table complete (consumers only pass through)
#pragma unroll 1
      for (int c = 0; c < NCH; c += 2) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {       // two chunks per trip: the LDS buffer index is a compile-time constant
          lds_barrier();                        // B image of chunk c + cc is complete
          // B fragments: three sets in rotation, requested TWO n-tiles (12 MFMAs) ahead -- this wavefront is alone on its
          // SIMD's matrix pipe, nobody covers an LDS round trip it waits for
          uint4 bf[3][2];
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            bf[0][p] = *reinterpret_cast<const uint4*>(bimg + bimg_off<BN>(cc, p, 0) + boff[0]);
            bf[1][p] = *reinterpret_cast<const uint4*>(bimg + bimg_off<BN>(cc, p, 1 / NT) + boff[1 % NT]);
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int gs = (c + cc) * 4 + s;
            aload(gs + 3, aw[(s + 3) % 4]);   // the set k-step s - 1 has just released
            __builtin_amdgcn_sched_barrier(0);
            uint4 (&cw)[2][2] = aw[s];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              const int g = s * NT + j, cur_f = g % 3, nxt_f = (g + 2) % 3;
              if (g + 2 < 4 * NT) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
                  bf[nxt_f][p] = *reinterpret_cast<const uint4*>(bimg + bimg_off<BN>(cc, p, (g + 2) / NT) + boff[(g + 2) % NT]);
              }
              __builtin_amdgcn_sched_barrier(0);   // (the scheduler would sink the reads to two MFMAs before their use)
#pragma unroll
              for (int i = 0; i < 2; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cw[i][1]), __builtin_bit_cast(f16x8, bf[cur_f][0]), acc[i][j], 0, 0, 0);
#pragma unroll
              for (int i = 0; i < 2; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cw[i][0]), __builtin_bit_cast(f16x8, bf[cur_f][1]), acc[i][j], 0, 0, 0);
#pragma unroll
              for (int i = 0; i < 2; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cw[i][0]), __builtin_bit_cast(f16x8, bf[cur_f][0]), acc[i][j], 0, 0, 0);
            }
          }
        }
      }
    }
    // ---- epilogue: BN affine + ReLU, rows transposed through the (now idle) producer windows into float4 stores ----
    const int ylen = a.amax_y.p ? (a.lens_y ? a.lens_y[b] : a.frames) : 0;
    unsigned ymax = 0;
    const float relu_floor = (a.relu & 1) ? 0.f : -__builtin_inff();
    float* stage = reinterpret_cast<float*>(wins + wave * kWWave);   // 2 x 8 rows x BN columns = 8 KB of 13 KB (4 of 9)
    // every pass's BN scale / shift BEFORE the first store: stores count in vmcnt like loads, so a load issued between
    // two passes makes its consumer wait (vmcnt(0)) for every store before it -- eight store round trips in series
    v4f scv[2][4], shv[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        scv[i][q] = *reinterpret_cast<const v4f*>(a.scale + wm + i * 32 + 8 * q + 4 * kh);
        shv[i][q] = *reinterpret_cast<const v4f*>(a.shift + wm + i * 32 + 8 * q + 4 * kh);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float* buf = stage + ((i * 4 + q) & 1) * (8 * BN);
        const int mq = wm + i * 32 + 8 * q;
        const v4f sc = scv[i][q], sh = shv[i][q];
        wave_fence();
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            buf[(4 * kh + rr) * BN + 32 * j + l31] = fmaf(acc[i][j][4 * q + rr] * out_scale, sc[rr], sh[rr]);
        wave_fence();
        // straight-line pass (no uniform branch per piece: see encoder_pw_split.hip's epilogue): all row pieces requested
        // together, ReLU as a maximum with a uniform floor
        v4f pv[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          const int f = lane + 64 * k, row = f / (BN / 4), c4 = f % (BN / 4);
          pv[k] = *reinterpret_cast<const v4f*>(buf + row * BN + 4 * c4);
        }
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          const int f = lane + 64 * k, row = f / (BN / 4), c4 = f % (BN / 4);
          const int m = mq + row, t = t0 + 4 * c4;
          const v4f v = __builtin_elementwise_max(pv[k], v4f{relu_floor, relu_floor, relu_floor, relu_floor});
          *reinterpret_cast<v4f*>(a.y + ((int64_t)b * FC + m) * a.ldy + t) = v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned u = abs_bits(v[e]);
            ymax = (t + e < ylen && u > ymax) ? u : ymax;
          }
        }
      }
    }
    if (a.amax_y.p) amax_publish(a.amax_y.p, a.amax_y.stride, b, tile * 4 + wave, ymax, lane);
    return;
  }

  // ============================================= producers ===========================================================
  if (!live) return;
  const int pw = wave - 4;                    // k-step of the chunk this wavefront produces
  const int p = lane >> 3, s = lane & 7;      // channel pair, segment of FPS frames
  unsigned char* win = wins + pw * kWWave;
  // the rows of a chunk as (pair, float4 column) items: item = lane + 64 j
  auto item = [&](int j, int& sp, int& q) { const int idx = lane + 64 * j; sp = idx / G::W4; q = idx - sp * G::W4; };
  v4f st[G::NLD][2];
  auto gload_dw = [&](int c) {
    const float* xb = a.x + ((int64_t)b * FC + c * FCH + pw * 16) * a.ldx;
#pragma unroll
    for (int j = 0; j < G::NLD; ++j) {
      int sp, q;
      item(j, sp, q);
      sp = sp < 8 ? sp : 7;
      int t = t0 - G::PADL + 4 * q;
      t = t < 0 ? 0 : (t > (int)a.ldx - 4 ? (int)a.ldx - 4 : t);
      const float* r0 = xb + (int64_t)(2 * sp) * a.ldx + t;
      st[j][0] = *reinterpret_cast<const v4f*>(r0);
      st[j][1] = *reinterpret_cast<const v4f*>(r0 + a.ldx);
    }
  };
  // residual chunks: 2 rows x FPS frames per lane, no halo (st[0 .. FPS / 4) hold them)
  auto gload_x2 = [&](int c) {
    const float* xb = a.x2 + ((int64_t)b * FC + (c - FC / FCH) * FCH + pw * 16 + 2 * p) * a.ldx2 + t0 + FPS * s;
#pragma unroll
    for (int j = 0; j < FPS / 4; ++j) {
      st[j][0] = *reinterpret_cast<const v4f*>(xb + 4 * j);
      st[j][1] = *reinterpret_cast<const v4f*>(xb + a.ldx2 + 4 * j);
    }
  };
  gload_dw(0);
  tap_copy();
  {
    unsigned mx = amax_collect(a.amax_x.p, a.amax_x.stride, a.amax_x.n, b, lane, amv);
    mx = __float_as_uint(__uint_as_float(mx) * a.dw_l1);
    if (DUAL) mx = max(mx, amax_collect(a.amax_x2.p, a.amax_x2.stride, a.amax_x2.n, b, lane, amv2));
    float inv;
    f16_scale(mx, &xs, &inv);
  }
  lds_barrier();   // tap table complete

  const int kh = p >> 2;
  // address of this lane's 4 bytes in the B image, for frame j of its segment: column n = FPS s + j (bimg_slot: the bits
  // of j sit below FPS, so low4 = j ^ const and the aligned group is the segment's)
  const int wbase = (kh * BN) * 16 + 4 * (p & 3);
  const int nsw = (FPS * s) & ~15;
  const int sx = (((FPS * s) & 15) ^ (((FPS * s) >> 4) | (kh << TL::KHS))) << 4;
  // converts the FPS (pair, frame) results of this lane and writes them into B image `buf`, k-step pw
  auto emit = [&](int buf, const v2f (&d)[FPS], int nvalid) {
#pragma unroll
    for (int j = 0; j < FPS; ++j) {
      const float sj = j < nvalid ? xs : 0.f;              // MaskedConv1d: the pointwise conv sees zeros past lens_out
      const v2f v = {d[j].x * sj, d[j].y * sj};
      const f16x2 hv = __builtin_convertvector(v, f16x2);
      const v2f r = v - __builtin_convertvector(hv, v2f);   // exact: the residual of a round-to-nearest conversion
      const f16x2 lv = __builtin_convertvector(r, f16x2);
      const int off = wbase + ((nsw << 4) | ((j << 4) ^ sx));
      *reinterpret_cast<unsigned*>(bimg + bimg_off<BN>(buf, 0, 0) + pw * (2 * BN * 16) + off) = __builtin_bit_cast(unsigned, hv);
      *reinterpret_cast<unsigned*>(bimg + bimg_off<BN>(buf, 1, 0) + pw * (2 * BN * 16) + off) = __builtin_bit_cast(unsigned, lv);
    }
  };

  const unsigned char* rd = win + p * kWPitch + TL::kSegBytes * s;   // this lane's window region
  const int nvalid = len_out - (t0 + FPS * s);
#pragma unroll 1
  for (int c = 0; c < FC / FCH; c += 2) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int ch = c + cc;
      // ---- masked rows of chunk ch -> interleaved pairs in the wavefront's LDS window ----
#pragma unroll
      for (int j = 0; j < G::NLD; ++j) {
        int sp, q;
        item(j, sp, q);
        const int t = t0 - G::PADL + 4 * q;
        v4f u = st[j][0], w = st[j][1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = (unsigned)(t + e) < (unsigned)len_in;   // t < 0: conv zero padding; t >= lens: masked_fill
          u[e] = ok ? u[e] : 0.f;
          w[e] = ok ? w[e] : 0.f;
        }
        if (sp < 8) {
          unsigned char* dst = win + sp * kWPitch + (4 * q) * 8 + ((4 * q) / FPS) * 16;   // 16 bytes of pad per FPS frames
          *reinterpret_cast<v4f*>(dst) = v4f{u.x, w.x, u.y, w.y};
          *reinterpret_cast<v4f*>(dst + 16) = v4f{u.z, w.z, u.w, w.w};
        }
      }
      wave_fence();
      // rows of the next chunk (or of the first residual chunk): in flight during this chunk's FMAs
      if (ch + 1 < FC / FCH) gload_dw(ch + 1);
      else if (DUAL) gload_x2(FC / FCH);
      // ---- depthwise: acc[j] = sum_k w[k] * x[OFF + j + k], both channels of the pair per packed FMA ----
      const unsigned char* tp = tapl + ((ch * FCH + pw * 16) / 2 + p) * (G::KP * 8);
      v2f xw[2 * G::NR];
      auto load_units = [&](int qa, int qb) {
#pragma unroll
        for (int qq = qa; qq < qb; ++qq) {
          const v4f v = *reinterpret_cast<const v4f*>(rd + qq * 16 + (qq / (FPS / 2)) * 16);   // frames 2 qq, 2 qq + 1
          xw[2 * qq] = v.xy;
          xw[2 * qq + 1] = v.zw;
        }
      };
      v2f acc[FPS];
#pragma unroll
      for (int j = 0; j < FPS; ++j) acc[j] = v2f{0.f, 0.f};
      load_units(0, G::qend(0));
      v2f wn[G::TB];   // taps of the NEXT block: their LDS round trip runs under this block's FMAs
      auto load_taps = [&](int blk) {
#pragma unroll
        for (int k2 = 0; k2 < G::TB / 2; ++k2) {
          if (G::TB * blk + 2 * k2 < G::KP) {
            const v4f v = *reinterpret_cast<const v4f*>(tp + (G::TB * blk + 2 * k2) * 8);
            wn[2 * k2] = v.xy;
            wn[2 * k2 + 1] = v.zw;
          }
        }
      };
      load_taps(0);
#pragma unroll
      for (int blk = 0; blk < G::NB; ++blk) {
        v2f wt[G::TB];
#pragma unroll
        for (int k = 0; k < G::TB; ++k) wt[k] = wn[k];
        if (blk + 1 < G::NB) {
          load_taps(blk + 1);
          load_units(G::qend(blk), G::qend(blk + 1));
        }
#pragma unroll
        for (int k = G::TB * blk; k < G::TB * (blk + 1) && k < K; ++k) {
#pragma unroll
          for (int j = 0; j < FPS; ++j) acc[j] = __builtin_elementwise_fma(wt[k - G::TB * blk], xw[G::OFF + j + k], acc[j]);
          // one tap at a time over all FPS accumulators: left alone, the scheduler turns the loops inside out (one output
          // at a time, its taps as a dependent chain with a wait state between links) to save registers
#pragma unroll
          for (int j = 0; j < FPS; ++j) asm volatile("" : "+v"(acc[j]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      emit(cc, acc, nvalid);
      lds_barrier();   // hands B image `cc` (chunk ch) to the consumers; they have finished chunk ch - 1
    }
  }
  if constexpr (DUAL) {
    const int nv2 = len2 - (t0 + FPS * s);
#pragma unroll 1
    for (int c = FC / FCH; c < NCH; c += 2) {
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int ch = c + cc;
        v2f d[FPS];
#pragma unroll
        for (int j = 0; j < FPS / 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) d[4 * j + e] = v2f{st[j][0][e], st[j][1][e]};
        if (ch + 1 < NCH) gload_x2(ch + 1);
        emit(cc, d, nv2);   // the block input is masked with ITS lens (jasper.py:428-436), zeros past it
        lds_barrier();
      }
    }
  }
}

template <int K, bool DUAL, int BN>
int launch_fused_t(const FusedArgs& a, hipStream_t st, int* amax_n) {
  using G = FGeom<K, BN>;
  const int tiles_t = (int)(a.ldy / BN);
  const int n_blocks = tiles_t * a.batch;
  if (a.amax_y.p) {
    if (tiles_t * 4 > a.amax_y.stride) return (int)hipErrorInvalidValue;
    if (amax_n) *amax_n = tiles_t * 4;
  }
  auto kern = dwpw_fused_kernel<K, DUAL, BN>;
  static std::atomic<uint64_t> lds_opted{0};   // per device (dyn_lds_opt_in)
  const hipError_t attr = dyn_lds_opt_in(reinterpret_cast<const void*>(kern), (int)G::LDS, lds_opted);
  if (attr != hipSuccess) return (int)attr;
  VASR_LAUNCH(kern, dim3(n_blocks), dim3(FNT), G::LDS, st, a, tiles_t, n_blocks);
  return 0;
}

}  // namespace

bool fused_dwpw_supported(int channels, int cout, int kernel, int stride, int dilation) {
  return channels == FC && cout == FC && stride == 1 && dilation == 1 && (kernel == 33 || kernel == 39);
}
int fused_dwpw_taps_per_pair(int kernel) { return (kernel + 2) & ~1; }

// host: depthwise weights [C][K] -> [C / 2][KP][2] (channel pairs interleaved, zero padded); returns max_c sum_k |w|
float pack_fused_taps(const float* w, int channels, int kernel, float* out) {
  const int kp = fused_dwpw_taps_per_pair(kernel);
  float l1 = 0.f;
  for (int c = 0; c < channels; ++c) {
    double s = 0;
    for (int k = 0; k < kernel; ++k) s += std::fabs((double)w[(size_t)c * kernel + k]);
    l1 = (float)s > l1 ? (float)s : l1;
  }
  for (int pr = 0; pr < channels / 2; ++pr)
    for (int k = 0; k < kp; ++k)
      for (int h = 0; h < 2; ++h)
        out[((size_t)pr * kp + k) * 2 + h] = k < kernel ? w[(size_t)(2 * pr + h) * kernel + k] : 0.f;
  // one ulp up: the bound must not fall below the true sum through the float rounding of this very sum
  return __builtin_nextafterf(l1, 3.0e38f);
}

int launch_fused_dwpw(const FusedLaunch& f, hipStream_t st, int* amax_n) {
  FusedArgs a{};
  a.x = f.x; a.ldx = f.ldx; a.lens_in = f.lens_in; a.lens_out = f.lens_out; a.taps = f.taps; a.dw_l1 = f.dw_l1;
  a.amax_x = f.amax_x; a.wt = reinterpret_cast<const uint4*>(f.wt); a.w_inv_scale = f.w_inv_scale; a.scale = f.scale;
  a.shift = f.shift; a.y = f.y; a.ldy = f.ldy; a.frames = f.frames; a.relu = f.relu; a.amax_y = f.amax_y;
  a.lens_y = f.lens_y; a.x2 = f.x2; a.ldx2 = f.ldx2; a.lens2 = f.lens2; a.amax_x2 = f.amax_x2; a.batch = f.batch;
  const bool dual = f.x2 != nullptr;
  if (f.tile_cols == 64) {
    if (f.kernel == 33) return dual ? launch_fused_t<33, true, 64>(a, st, amax_n) : launch_fused_t<33, false, 64>(a, st, amax_n);
    if (f.kernel == 39) return dual ? launch_fused_t<39, true, 64>(a, st, amax_n) : launch_fused_t<39, false, 64>(a, st, amax_n);
    return -1;
  }
  if (f.kernel == 33) return dual ? launch_fused_t<33, true, 128>(a, st, amax_n) : launch_fused_t<33, false, 128>(a, st, amax_n);
  if (f.kernel == 39) return dual ? launch_fused_t<39, true, 128>(a, st, amax_n) : launch_fused_t<39, false, 128>(a, st, amax_n);
  return -1;
}

}  // namespace vasr
