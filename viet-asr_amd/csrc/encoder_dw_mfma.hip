// Depthwise masked 1-D convolution on the MATRIX pipe (gfx950): the Toeplitz form.
//
// Same contract as encoder_dw.hip (reference nemo/collections/asr/parts/jasper.py:113-132 mask + conv, :360-373 layer
// order): masked input, output zeroed past lens_out, every column < ldy written.  The packed-FMA kernels there are
// VALU-issue bound from K = 51 up (600 v_pk_fma_f32 per wavefront-pass at K = 75: 23 us of pure issue per 512-channel
// layer against 17-20 us of copy-speed traffic).  Here the taps of ONE channel are laid out as a banded matrix
//
//     A[m][j] = w[(j - m - OFF) / DIL]     (0 where that is not a tap)        m = 0..15 output offset inside a window
//
// and 16 windows of 16 consecutive outputs -- 256 frames of one utterance -- are its right-hand sides:
// B[j][n] = x[t0 + 16 n - PADL + j], eight consecutive samples per lane and k-step, so every B fragment is one aligned
// 16-byte LDS read of a plainly linear fp16 row (the lane groups of ds_read_b128 land on distinct 16-byte slots without
// any padding).  One v_mfma_f32_16x16x32 then produces 16 x 16 outputs from a 32-sample slice of the windows: a K = 75
// kernel spans 15 + 74 + OFF + 1 = 93 window samples = 3 k-steps.  Arithmetic as in the fp16-split GEMM
// (encoder_pw_split.hip, kF16x2): operands scaled by a power of two (taps per channel, samples per utterance -- from the
// maxima the producing GEMM published) and split into two fp16 terms, three products per k-step, fp32 accumulation:
// 9 MFMAs of 16 cycles for 256 outputs x 75 taps instead of 150 packed FMAs of 4-8 cycles.
//
// Round 2 first built this on v_mfma_f32_32x32x16 (32 windows of 32 outputs, two utterances per task): 7 k-steps at
// K = 75 = 112 VGPRs of A fragments, 16 accumulators per lane that had to be transposed through LDS before a store could
// write rows, ~500 instructions per 1024 outputs at under two wavefronts per SIMD -- 30-36 us per 512-channel layer,
// slower than the packed-FMA kernels.  The 16 x 16 x 32 shape fixes all three: 24 VGPRs of A
// fragments, and its D layout (lane = window n + 16 g, registers = outputs 4 g .. 4 g + 3) already holds four
// CONSECUTIVE frames per lane, so one lane rotation (4 ds_bpermute_b32, no LDS storage, no barrier) makes every store
// instruction write 1 KB of one row.
//
//   grid: 1-D over (C / 4) x ceil(B / utterances per wave) x ceil(ldy / 512) (XCD-aware order, see the kernel), block 256 =
//   4 wavefronts = 4 channels; a wavefront builds
//   its channel's A fragments once (from a [hi | lo] fp16 tap table packed at vasr_finalize()) and walks up to eight
//   utterances with them, 512 frames (two MFMA groups sharing one staged row) per utterance, the next utterance's row in
//   flight while the current one is multiplied.  The walk is fully unrolled: what depends on the utterance (row
//   descriptors, lengths, scales) is fetched once into scalar registers, and the eight output maxima are reduced
//   across the lanes together at the end.
//
// Measured (MI355X, B = 64 x 501 frames, 512 channels, in the pipeline): 22-23 us per layer whatever K (134 MB -> 5.8-6.0
// TB/s = 0.72-0.75 of the 8 TB/s peak; a copy_ of the same bytes 19.7 us), against 25.6 / 28.4 / 31.0 / 35.5 us of the
// packed-FMA kernels at K = 51 / 63 / 75 / 87 x 2.  Compiled-out ablations: rows and stores only 18.1 us, everything but
// rows and stores 16.0 us -- both streams are long and they overlap imperfectly; the MFMAs themselves cost 0.4 us.
#include <cstdlib>

#include "vasr_internal.h"
#include "vasr_device.h"
#include <type_traits>

namespace vasr {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using v4f = __attribute__((ext_vector_type(4))) float;
using v2f = __attribute__((ext_vector_type(2))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using u32x4 = __attribute__((vector_size(16))) unsigned int;

constexpr int kTile = 512;          // output frames per task: 2 groups of 16 windows x 16 outputs
constexpr int kGroup = 256;
constexpr int kUttPerWave = 8;      // upper bound; the launch passes the count in use
// rows in flight per wavefront beyond the one being multiplied: 1 and 2 measure the same at 512 channels (23.2 us at K = 75),
// 1 is 5-10 % faster at 256; 3 and 4 are 25-30 % SLOWER (31 us)
constexpr int kRowStages = 1;

template <int K, int DIL>
struct TzGeom {
  static constexpr int PAD = DIL > 1 ? (DIL * K) / 2 - 1 : K / 2;   // get_same_padding (jasper.py:60-65)
  static constexpr int PADL = (PAD + 3) & ~3;                       // staged row origin: 16-byte aligned in x
  static constexpr int OFF = PADL - PAD;
  static constexpr int SPAN = 15 + DIL * (K - 1) + OFF + 1;         // window samples one 16-output window touches
  static constexpr int NS = (SPAN + 31) / 32;                       // k-steps
  static constexpr int STAGES = kRowStages;                         // register stages of NLD float4 each
  static constexpr int TSZ = 32 * NS + 16;                          // tap-table entries: entry i = dense tap i - (15 + OFF)
  static constexpr int ROWS = kTile - 16 + 32 * NS;                 // samples staged per task
  static constexpr int NLD = (ROWS / 4 + 63) / 64;                  // float4 loads per lane and task
  static constexpr int PLANE = 2 * 256 * NLD;                       // bytes of one fp16 plane (every lane stores, no branch)
  static constexpr int LDS_DATA = 2 * PLANE;                        // [hi | lo]
  static constexpr int LDS_TAB = 4 * TSZ;
  static constexpr int LDS_WAVE = LDS_DATA + LDS_TAB;
  static_assert(ROWS % 4 == 0 && LDS_WAVE % 16 == 0, "staging granularity");
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ f32x4 mma(uint4 a, uint4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float lane_pull(int src4, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(v)));
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
                                                          unsigned* __restrict__ amax_y, int amax_y_stride, int upw_nt,
                                                          int tiles, int n_rest) {
  using G = TzGeom<K, DIL>;
  // 1-D grid -> (channel group, utterance group, time tile).  Workgroup i runs on XCD i % 8 (each with an L2 of its own).
  // A row longer than one 512-frame tile is cut into time tiles whose staged rows OVERLAP by the window overhang (80 of 592
  // samples): dealt to the XCDs round-robin -- rounds 2-5: grid (C / 4, B / upw, tiles), a row's neighbouring tiles 8 192
  // workgroups apart -- that overhang came from HBM a second time (profiles/r06_c5_pmc_traffic_summary.json, the 512 x 30 s
  // shard: FETCH_SIZE 1.10-1.18 x the tensor, WRITE_SIZE 1.00 x).  Here the tiles of one (channel group, utterance group) get
  // consecutive slots of ONE XCD, so that the neighbour's overhang is in that L2 when it is asked for.  With a single tile
  // per row (the 10 s headline) this is the old order: channel group fastest.
  const int bid = (int)blockIdx.x;
  const int rest = ((bid >> 3) / tiles) * 8 + (bid & 7);      // (channel group, utterance group) index, channel group fastest
  if (rest >= n_rest) return;                                 // (the grid is padded to whole rounds of the 8 XCDs)
  const int ncg = channels >> 2;
  const int bx = rest % ncg, by = rest / ncg, bz = (bid >> 3) % tiles;
  const int upw = upw_nt;                  // utterances one wavefront walks
  constexpr int NS = G::NS, NLD = G::NLD, kStages = G::STAGES;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* base = lds_raw + wave * G::LDS_WAVE;
  unsigned char* dat = base;                                               // [plane][PLANE]
  unsigned* tab = reinterpret_cast<unsigned*>(base + G::LDS_DATA);
  const int c = bx * 4 + wave;
  const int t_tile = bz * kTile;
  const int n16 = lane & 15, kg = lane >> 4;
  const int b_lo = by * upw;
  const int b_hi = min(b_lo + upw, batch);
  if (b_lo >= b_hi) return;
  const bool two = t_tile + kGroup < ldy;   // the second group has columns to write (wave-uniform)

  uint4 ah[NS], al[NS];   // A fragments of this channel (built below, after the first rows have been requested)
  const float w_inv = tap_inv[c];

  // ---- staging of one task's row: branch-free, all loads in flight together ----
  // Every mask and address computation that hardware can do is handed to it: rows are read through raw buffer
  // descriptors whose range is the utterance's length -- MaskedConv1d's x.masked_fill(t >= lens, 0) (jasper.py:113-118)
  // and the conv's zero padding left of frame 0 (a negative offset is a huge unsigned one) both come back as zeros, per
  // dword; so do the lanes past the staged row (offset 2^31).  The offsets depend on the tile only: computed once.
  // A last tile with few columns (a row pitch off the 512-frame grid: 10.3 s clips have 640) needs only the head of the
  // staged row -- its columns plus the window overhang: the 256-sample slabs behind that are neither requested (offset
  // 2^31 again) nor converted; they are cleared once, so that the windows nobody stores multiply zeros.
  struct Stage { v4f r[NLD]; };
  const int n_ld = min(NLD, (min((int)ldy - t_tile, kTile) - 16 + 32 * NS + 255) / 256);   // wave-uniform, 1 .. NLD
  int voff[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int tau = 4 * (lane + 64 * j);
    voff[j] = tau < G::ROWS && j < n_ld ? 4 * (t_tile - G::PADL + tau) : (int)0x80000000u;
    if (j >= n_ld) {
      *reinterpret_cast<uint2*>(dat + 8 * lane + 512 * j) = make_uint2(0u, 0u);
      *reinterpret_cast<uint2*>(dat + G::PLANE + 8 * lane + 512 * j) = make_uint2(0u, 0u);
    }
  }
  // Everything a task needs that depends on its utterance alone is a wave-uniform scalar, fetched ONCE up front; the task
  // sequence below is fully unrolled, so that these live in SGPRs under compile-time indices.  (As a loop over the
  // utterance index the kernel spent 82 scalar instructions per task -- 27 % of its issue cycles, SQ_INSTS_SALU -- on
  // 64-bit row addresses, length loads and their waits, descriptor words and clamps.)
  const int n_utt = b_hi - b_lo;   // 1 .. kUttPerWave
  const int64_t row_stride = (int64_t)channels * ldx;                 // floats between utterances of one channel
  const float* xrow0 = x + ((int64_t)b_lo * channels + c) * ldx;
  const int64_t yrow_stride = (int64_t)channels * ldy;
  float* yrow0 = y + ((int64_t)b_lo * channels + c) * ldy;
  int rng[kUttPerWave], lout[kUttPerWave];
#pragma unroll
  for (int u = 0; u < kUttPerWave; ++u) {
    const int b = min(b_lo + u, batch - 1);
    // past the wavefront's last utterance: an empty range -- the unconditional prefetches there return zeros without a
    // memory access
    rng[u] = u < n_utt ? 4 * min(lens_in[b], (int)ldx) : 0;
    lout[u] = lens_out[b];
  }
  auto gload = [&](int u, Stage& sg) {
    // (descriptor words are wave-uniform scalars, or every load becomes a waterfall loop)
    const auto d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xrow0 + u * row_stride), 0, rng[u], 0x00020000);
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      sg.r[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(d, voff[j], 0, 0));
    }
  };
  // scale by the utterance's power of two, split into fp16 hi / lo, 8 + 8 bytes into the two planes: sample tau at byte
  // 2 tau.  EVERY lane stores (the planes hold 256 NLD samples): with a lane-dependent branch here the compiler cannot
  // tell, where the paths merge, whether the skipped lanes' loads have been waited for, and drains the vector-memory
  // counter (vmcnt(0)) before the next prefetch overwrites the stage registers -- no prefetch at all.
  auto sstore = [&](const Stage& sg, float sx) {
    unsigned char* ph = dat + 8 * lane;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      if (j >= n_ld) continue;   // wave-uniform; VALU and LDS only, the loads themselves stay unconditional
      const v4f v = sg.r[j];
      const v2f a = {v.x * sx, v.y * sx}, b = {v.z * sx, v.w * sx};
      const f16x2 ha = __builtin_convertvector(a, f16x2), hb = __builtin_convertvector(b, f16x2);
      const f16x2 la = __builtin_convertvector(a - __builtin_convertvector(ha, v2f), f16x2);
      const f16x2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, v2f), f16x2);
      *reinterpret_cast<uint2*>(ph + 512 * j) = make_uint2(__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb));
      *reinterpret_cast<uint2*>(ph + G::PLANE + 512 * j) = make_uint2(__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb));
    }
  };

  // B fragments: lane (n, kg), group q, step s -> samples 256 q + 16 n + 32 s + 8 kg + e of the staged row
  const unsigned char* bb = dat + 32 * n16 + 16 * kg;
  // D fragment: lane n + 16 g holds frames 16 n + 4 g .. + 3; lane L pulls the registers of lane (L >> 2) + 16 (L & 3) and
  // then holds frames 4 L .. 4 L + 3
  const int pull = 4 * ((lane >> 2) + 16 * (lane & 3));
  const int slot = c * tiles + bz;

  float sxu[kUttPerWave], ixu[kUttPerWave];   // (scale, 1 / scale) per utterance, set below
  unsigned mxu[kUttPerWave];                  // per-lane maxima of the outputs, per utterance
#pragma unroll
  for (int u = 0; u < kUttPerWave; ++u) mxu[u] = 0u;
  // ---- the eight maxima across the wavefront in ONE butterfly (29 instructions instead of 8 x 16): each exchange step
  //      also halves the number of registers -- a lane keeps the utterances whose index agrees with its lane bit and
  //      hands the others to its partner -- until lane L holds utterance L & 7; one store writes all of them ----
  auto publish = [&]() {
    if (!amax_y) return;
#define VASR_DPP(xx, ctrl) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(xx), (ctrl), 0xF, 0xF, false))
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    unsigned r[4], q[2];
#pragma unroll
    for (int k = 0; k < 4; ++k)   // lanes L, L ^ 1
      r[k] = max(b0 ? mxu[2 * k + 1] : mxu[2 * k], VASR_DPP(b0 ? mxu[2 * k] : mxu[2 * k + 1], 0xB1));
#pragma unroll
    for (int k = 0; k < 2; ++k)   // lanes L, L ^ 2
      q[k] = max(b1 ? r[2 * k + 1] : r[2 * k], VASR_DPP(b1 ? r[2 * k] : r[2 * k + 1], 0x4E));
    // lanes L, L ^ 4: a shift by four lanes in BOTH directions -- whatever arrives from four lanes away (inside the row of
    // 16; nothing, i.e. zero, from outside) is a partial maximum of the very utterance the lane keeps
    const unsigned send = b2 ? q[0] : q[1];
    unsigned m = max(b2 ? q[1] : q[0], max(VASR_DPP(send, 0x104), VASR_DPP(send, 0x114)));   // row_shl:4, row_shr:4
    m = max(m, VASR_DPP(m, 0x128));   // row_ror:8: lanes L, L ^ 8
#undef VASR_DPP
    {
      const auto sw = __builtin_amdgcn_permlane16_swap(m, m, false, false);   // rows 0 <-> 1, 2 <-> 3
      m = max((unsigned)sw[0], (unsigned)sw[1]);
    }
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(m, m, false, false);   // halves
      m = max((unsigned)sw[0], (unsigned)sw[1]);
    }
    // lane u < n_utt -> row b_lo + u of the table; every other lane's offset lies outside the descriptor's range
    const auto da = __builtin_amdgcn_make_buffer_rsrc(amax_y + (int64_t)b_lo * amax_y_stride, 0, 4 * n_utt * amax_y_stride, 0x00020000);
    const int off = lane < kUttPerWave ? 4 * (lane * amax_y_stride + slot) : (int)0x80000000u;
    __builtin_amdgcn_raw_buffer_store_b32(m, da, off, 0, 0);
  };
  auto do_task = [&](auto u_tag, Stage& sg) {
    constexpr int u = decltype(u_tag)::value;
    sstore(sg, sxu[u]);
    // The utterance kStages ahead, in flight while this one and the next ones are multiplied and stored: issued whether
    // or not it exists (see rng[]), so that on the straight-line path the compiler knows how many vector-memory
    // operations are younger than the rows it waits for (s_waitcnt vmcnt counts, it does not name).
    if constexpr (u + kStages < kUttPerWave) gload(u + kStages, sg);
    wave_sync();

    // All B fragments of the task are requested before the first multiply (one exposed LDS round trip per task instead of
    // one per k-step), and the two groups' accumulation chains alternate on the matrix pipe.
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    constexpr bool kAllB = NS <= 3;   // 16 NS registers of B fragments for both groups
    auto load_b = [&](int q, uint4 (&bh)[NS], uint4 (&bl)[NS]) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        bh[s] = *reinterpret_cast<const uint4*>(bb + 2 * kGroup * q + 64 * s);
        bl[s] = *reinterpret_cast<const uint4*>(bb + G::PLANE + 2 * kGroup * q + 64 * s);
      }
    };
    auto step = [&](f32x4 acc, int s, const uint4 (&bh)[NS], const uint4 (&bl)[NS]) {
      acc = mma(al[s], bh[s], acc);
      acc = mma(ah[s], bl[s], acc);
      acc = mma(ah[s], bh[s], acc);
      return acc;
    };
    {
      uint4 b0h[NS], b0l[NS], b1h[NS], b1l[NS];
      load_b(0, b0h, b0l);
      if (kAllB && two) {
        load_b(1, b1h, b1l);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          acc0 = step(acc0, s, b0h, b0l);
          acc1 = step(acc1, s, b1h, b1l);
        }
      } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) acc0 = step(acc0, s, b0h, b0l);
        if (two) {
          load_b(1, b1h, b1l);
#pragma unroll
          for (int s = 0; s < NS; ++s) acc1 = step(acc1, s, b1h, b1l);
        }
      }
    }

    // ---- epilogue: unscale, rotate lanes, zero past lens_out, 1 KB of one row per store instruction ----
    const float os = w_inv * ixu[u];
    const int lo = lout[u];
    float* yr = yrow0 + u * yrow_stride;
    // columns >= ldy are dropped by the descriptor's range, like the whole second group of a tile that has none
    const auto dy = __builtin_amdgcn_make_buffer_rsrc(yr, 0, 4 * (int)ldy, 0x00020000);
    v4f o0 = {acc0[0] * os, acc0[1] * os, acc0[2] * os, acc0[3] * os};
    v4f o1 = {acc1[0] * os, acc1[1] * os, acc1[2] * os, acc1[3] * os};
    // all eight pulls in flight together
    o0.x = lane_pull(pull, o0.x); o0.y = lane_pull(pull, o0.y); o0.z = lane_pull(pull, o0.z); o0.w = lane_pull(pull, o0.w);
    o1.x = lane_pull(pull, o1.x); o1.y = lane_pull(pull, o1.y); o1.z = lane_pull(pull, o1.z); o1.w = lane_pull(pull, o1.w);
    float mx = 0.f;
    auto finish = [&](v4f o, int q) {
      const int tq = t_tile + kGroup * q;
      if (lo < tq + kGroup) {   // wave-uniform: only the group that straddles the utterance's end masks per element
        const int nv = lo - tq - 4 * lane;
        o.x = nv > 0 ? o.x : 0.f;
        o.y = nv > 1 ? o.y : 0.f;
        o.z = nv > 2 ? o.z : 0.f;
        o.w = nv > 3 ? o.w : 0.f;
      }
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
      // (non-temporal stores for outputs beyond the Infinity Cache were measured in rounds 1-3: no gain in the pipeline)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), dy, 4 * (tq + 4 * lane), 0, 0);
    };
    finish(o0, 0);
    finish(o1, 1);
    mxu[u] = __float_as_uint(mx);   // |x| bit patterns order like unsigned integers; reduced across the lanes at the end
    wave_sync();   // the next task's staging overwrites the row
  };

  Stage stg[kStages];
#pragma unroll
  for (int i = 0; i < kStages; ++i) gload(i, stg[i]);

  // tap table of this channel: requested now, so that it shares the flight of the rows and of the maxima below
  constexpr int NTL = (G::TSZ + 63) / 64;
  unsigned tapv[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j) tapv[j] = lane + 64 * j < G::TSZ ? taps[(int64_t)c * G::TSZ + lane + 64 * j] : 0u;

  // ---- scales of every utterance this wavefront will touch, once, ALL AT ONCE: the reduction of a producer's slots is a
  //      round trip to L2 per 64 slots, and a wavefront that took its utterances one after the other would sit through
  //      8-30 of them (5-15 us on a 25 us kernel) before its first multiply.  Unrolled over the utterances, the loads of
  //      one trip are in flight together, next to the first rows and the tap table. ----
  {
    unsigned m[kUttPerWave];
    const unsigned* ap[kUttPerWave];
#pragma unroll
    for (int u = 0; u < kUttPerWave; ++u) {
      m[u] = 0u;
      ap[u] = amax_x + (int64_t)min(b_lo + u, b_hi - 1) * amax_x_stride;
    }
    for (int i0 = 0; i0 < amax_x_n; i0 += 64) {
      const int i = min(i0 + lane, amax_x_n - 1);   // clamped, not predicated: a slot read twice does not change a maximum
#pragma unroll
      for (int u = 0; u < kUttPerWave; ++u) m[u] = max(m[u], ap[u][i]);
    }
#pragma unroll
    for (int u = 0; u < kUttPerWave; ++u) f16_scale(wave_max_u32(m[u]), &sxu[u], &ixu[u]);
  }

  // ---- A fragments of this channel: lane (m = n16, kg), step s, element e <- table[15 - m + 32 s + 8 kg + e] ----
#pragma unroll
  for (int j = 0; j < NTL; ++j)
    if (lane + 64 * j < G::TSZ) tab[lane + 64 * j] = tapv[j];
  wave_sync();
  {
    const unsigned* tp = tab + 15 - n16 + 8 * kg;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      unsigned v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tp[32 * s + e];
      // (hi | lo << 16) pairs -> four dwords of hi halves, four of lo halves (v_perm_b32 each)
      ah[s] = make_uint4(__builtin_amdgcn_perm(v[1], v[0], 0x05040100u), __builtin_amdgcn_perm(v[3], v[2], 0x05040100u),
                         __builtin_amdgcn_perm(v[5], v[4], 0x05040100u), __builtin_amdgcn_perm(v[7], v[6], 0x05040100u));
      al[s] = make_uint4(__builtin_amdgcn_perm(v[1], v[0], 0x07060302u), __builtin_amdgcn_perm(v[3], v[2], 0x07060302u),
                         __builtin_amdgcn_perm(v[5], v[4], 0x07060302u), __builtin_amdgcn_perm(v[7], v[6], 0x07060302u));
    }
  }

  // the wavefront's utterances, unrolled; one that has fewer leaves early (never rejoining: the counts stay exact)
  auto run = [&](auto u_tag, auto& self) -> void {
    constexpr int u = decltype(u_tag)::value;
    if constexpr (u < kUttPerWave) {
      if (u >= n_utt) return publish();
      do_task(u_tag, stg[u % kStages]);
      self(std::integral_constant<int, u + 1>{}, self);
    } else {
      publish();
    }
  };
  run(std::integral_constant<int, 0>{}, run);
}

template <int K, int DIL>
int launch_tz(const float* x, int64_t ldx, const unsigned* taps, const float* tap_inv, const int32_t* li, const int32_t* lo,
              AmaxTab amax_x, int batch, int channels, float* y, int64_t ldy, AmaxTab* amax_y, hipStream_t st) {
  using G = TzGeom<K, DIL>;
  auto kern = dw_toeplitz_kernel<K, DIL>;
  constexpr int lds = 4 * G::LDS_WAVE;
  const unsigned tiles = (unsigned)((ldy + kTile - 1) / kTile);
  // utterances one wavefront walks with its channel's A fragments: as many as leave >= 4096 wavefronts (4 per SIMD)
  const int upw_env = dev_switches().dw_upw;   // devtools build: pins it
  int upw = kUttPerWave;
  if (upw_env >= 1 && upw_env <= kUttPerWave) upw = upw_env;
  else
    while (upw > 2 && (int64_t)channels * ((batch + upw - 1) / upw) * tiles < 4096) upw /= 2;
  const int n_rest = (channels / 4) * ((batch + upw - 1) / upw);
  dim3 grid((unsigned)(((n_rest + 7) / 8) * 8) * tiles);      // 1-D: the kernel deals it to (channel group, utterance group, tile)
  if (amax_y) {
    amax_y->n = channels * (int)tiles;
    if (amax_y->n > amax_y->stride) return -1;
  }
  VASR_LAUNCH(kern, grid, dim3(256), lds, st, x, ldx, taps, tap_inv, li, lo, amax_x.p, amax_x.stride, amax_x.n, channels, batch,
              y, ldy, amax_y ? amax_y->p : nullptr, amax_y ? amax_y->stride : 0,
              upw, (int)tiles, n_rest);
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
// (entry i = dense tap i - (15 + OFF), dense tap k * DIL = w[k]); returns 1 / scale.
float pack_depthwise_taps_f16x2(const float* w, int kernel, int dilation, int tsz, unsigned* table) {
  const int pad = dilation > 1 ? (dilation * kernel) / 2 - 1 : kernel / 2;
  const int off = ((pad + 3) & ~3) - pad;
  float mx = 0.f;
  for (int k = 0; k < kernel; ++k) mx = fabsf(w[k]) > mx ? fabsf(w[k]) : mx;
  int e = (int)(__builtin_bit_cast(unsigned, mx) >> 23);
  e = e < 16 ? 16 : (e > 254 ? 254 : e);
  const float s = __builtin_bit_cast(float, (unsigned)(268 - e) << 23), inv = __builtin_bit_cast(float, (unsigned)(e - 14) << 23);
  for (int i = 0; i < tsz; ++i) {
    const int d = i - (15 + off);
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
