// Depthwise masked 1-D convolution for gfx950 + the length chain + re-padding helper.
//
// Replaces the groups == channels MaskedConv1d of every separable JasperBlock
// (reference nemo/collections/asr/parts/jasper.py:113-132 mask + conv, :360-373 layer order).
//
// HBM-bound by design (2*K flops per 8 bytes); from K = 51 up the packed-fp32 FMA issue rate is the second bound.
// Two kernels, same contract (masked input, output zeroed past lens_out so the following GEMM needs no mask):
//   * dw_pair_kernel (default, further down): one wavefront = one channel x one 512-frame tile of TWO utterances;
//     the halves of every v_pk_fma_f32 are the two utterances, 8 consecutive frames per lane.
//   * dw_conv_kernel (VASR_DW_PAIR=0): one wavefront = one (utterance, channel) row segment of 512 outputs; the
//     masked input window is staged once into LDS with 16-byte coalesced loads; every lane pulls a register window of
//     K+4 samples with ds_read_b128 and produces two groups of 4 consecutive outputs, pairing time-adjacent outputs
//     in the packed FMAs.
// In both, each input sample is read from HBM exactly once, LDS reads are conflict free, stores are 16 bytes per
// lane, and the K taps of the row's channel are wave-uniform and travel through SGPRs.
#include <cstdlib>

#include "vasr_internal.h"
#include "len_chain.h"
#include "vasr_device.h"

namespace vasr {

namespace {

constexpr int kTile = 512;  // outputs per wavefront (2 groups x 64 lanes x 4)
using v4f = __attribute__((ext_vector_type(4))) float;  // native vector: one ds_read_b128 / global dwordx4

template <int K, int DIL = 1>
struct DwGeom {
  static constexpr int PAD = DIL > 1 ? (DIL * K) / 2 - 1 : K / 2;  // get_same_padding (jasper.py:60-65)
  static constexpr int PADL = (PAD + 3) & ~3;
  static constexpr int OFF = PADL - PAD;
  static constexpr int NQ = (OFF + K + 3 + 3) / 4;       // float4 reads per lane per group
  static constexpr int WIN = 256 + 252 + 4 * NQ;         // floats of LDS per wavefront
};

using v2f = __attribute__((ext_vector_type(2))) float;

// Each wavefront owns a private LDS window, so no workgroup barrier is needed: the LDS pipeline
// executes one wavefront's ds_write / ds_read in issue order; this only pins the compiler.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One sub-group of 4 consecutive outputs, taps [K0, K1), on packed-fp32 FMAs.
// v_pk_fma_f32 wants its two lanes in one even-aligned register pair, but output pair (r, r+1) at
// tap k needs x[r+k], x[r+k+1], which is an aligned pair only when r+k is even.  So taps whose
// window offset e = OFF+k is even accumulate output pairs (0,1),(2,3); odd-e taps accumulate the
// shifted pairs (-1,0),(1,2),(3,4) (the two outer halves are discarded): 2.5 packed FMAs per tap
// per 4 outputs instead of 4 scalar ones, with no register shuffles.
struct DwAcc { v2f e01, e23, om0, o12, o34; };

template <int K0, int K1, int OFF, int DIL, class W>
__device__ __forceinline__ void dw_taps(const v4f* __restrict__ win, const W& wc, DwAcc& a) {
  constexpr int Q0 = (OFF + DIL * K0 - 1 < 0 ? 0 : OFF + DIL * K0 - 1) / 4;   // first / last float4 of the window
  constexpr int Q1 = (OFF + DIL * (K1 - 1) + 4) / 4;
  v2f xw[2 * (Q1 - Q0 + 1)];
#pragma unroll
  for (int q = Q0; q <= Q1; ++q) {
    const v4f v = win[q];
    xw[2 * (q - Q0)] = v.xy;
    xw[2 * (q - Q0) + 1] = v.zw;
  }
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    const float wk = wc[k - K0];
    const v2f w2 = {wk, wk};
    const int e = OFF + DIL * k - 4 * Q0;
    if ((OFF + DIL * k) % 2 == 0) {
      a.e01 = __builtin_elementwise_fma(w2, xw[e / 2], a.e01);
      a.e23 = __builtin_elementwise_fma(w2, xw[e / 2 + 1], a.e23);
    } else {
      a.om0 = __builtin_elementwise_fma(w2, xw[(e - 1) / 2], a.om0);
      a.o12 = __builtin_elementwise_fma(w2, xw[(e + 1) / 2], a.o12);
      a.o34 = __builtin_elementwise_fma(w2, xw[(e + 3) / 2], a.o34);
    }
  }
}

// grid (C/(4*ROWS), B, ceil(ldy/512)), block 256 = 4 wavefronts; each wavefront walks ROWS channels
// of one utterance, prefetching the next row's window into registers while it computes the current.
template <int K, int ROWS, int DIL = 1>
__global__ __launch_bounds__(256) void dw_conv_kernel(const float* __restrict__ x, int64_t ldx,
                                                      const float* __restrict__ w,
                                                      const int32_t* __restrict__ lens_in,
                                                      const int32_t* __restrict__ lens_out, int channels,
                                                      float* __restrict__ y, int64_t ldy, unsigned* __restrict__ amax,
                                                      int amax_stride) {
  using G = DwGeom<K, DIL>;
  constexpr int NQE = (G::OFF + DIL * (K - 1) + 4 + 1 + 3) / 4;  // +1: the odd-tap pairing reads one sample further
  constexpr int NLD = ((256 + 252 + 4 * NQE) / 4 + 63) / 64;  // staging float4s per lane (3)
  constexpr int WIN4 = NLD * 64;  // float4s of LDS per wavefront, rounded up so staging needs no predicate
  __shared__ v4f lds4[4 * WIN4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = (blockIdx.x * 4 + wave) * ROWS;
  const int b = blockIdx.y;
  const int t_start = blockIdx.z * kTile;
  const int len_in = lens_in[b];
  const int len_out = lens_out[b];
  v4f* win4 = lds4 + wave * WIN4;

  // masked window of one row: LDS float4 index i4 <-> frames t_start - PADL + 4*i4 .. +3.
  // Branch-free (clamped address + selects) so that all staging loads are in flight together.
  v4f stage[NLD];
  auto gload = [&](int c) {
    const float* xr = x + ((int64_t)b * channels + c) * ldx;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int t = t_start - G::PADL + 4 * (lane + 64 * j);
      int tc = t < 0 ? 0 : t;
      tc = tc > (int)ldx - 4 ? (int)ldx - 4 : tc;
      stage[j] = *reinterpret_cast<const v4f*>(xr + tc);
    }
  };
  // MaskedConv1d: x.masked_fill(t >= lens, 0) (jasper.py:113-118); t < 0 is the conv zero padding
  auto masked = [&](int j) {
    const int t = t_start - G::PADL + 4 * (lane + 64 * j);
    v4f v = stage[j];
    v.x = (t >= 0 && t + 0 < len_in) ? v.x : 0.f;
    v.y = (t >= 0 && t + 1 < len_in) ? v.y : 0.f;
    v.z = (t >= 0 && t + 2 < len_in) ? v.z : 0.f;
    v.w = (t >= 0 && t + 3 < len_in) ? v.w : 0.f;
    return v;
  };

  unsigned row_max = 0;
  gload(c0);
#pragma unroll 1
  for (int r = 0; r < ROWS; ++r) {
    const int c = c0 + r;
    // taps of this row's channel: wave-uniform -> scalar loads.  For the short kernels they are pinned
    // here so that their latency overlaps the staging loads already in flight instead of following
    // the LDS round trip (for K > 44 pinning all taps at once would spill SGPRs).
    const float* wg = w + (int64_t)c * K;
    float wc[K <= 44 ? K : 1];
    if constexpr (K <= 44) {
#pragma unroll
      for (int k = 0; k < K; ++k) wc[k] = wg[k];
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("" : "+s"(wc[k]));
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) win4[lane + 64 * j] = masked(j);
    if (r + 1 < ROWS) gload(c + 1);   // in flight while this row is computed
    wave_sync();
    float* yr = y + ((int64_t)b * channels + c) * ldy;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      DwAcc acc{{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
      if constexpr (K <= 44) dw_taps<0, K, G::OFF, DIL>(win4 + g * 64 + lane, wc, acc);
      else dw_taps<0, K, G::OFF, DIL>(win4 + g * 64 + lane, wg, acc);
      v4f o;
      o.x = acc.e01.x + acc.om0.y;
      o.y = acc.e01.y + acc.o12.x;
      o.z = acc.e23.x + acc.o12.y;
      o.w = acc.e23.y + acc.o34.x;
      const int t = t_start + g * 256 + lane * 4;
      if (t < ldy) {
        // the following 1x1 MaskedConv1d masks with lens_out: zero here so the GEMM needs no predicate
        if (t + 0 >= len_out) o.x = 0.f;
        if (t + 1 >= len_out) o.y = 0.f;
        if (t + 2 >= len_out) o.z = 0.f;
        if (t + 3 >= len_out) o.w = 0.f;
        *reinterpret_cast<v4f*>(yr + t) = o;
        if (amax) row_max = max(row_max, max(max(abs_bits(o.x), abs_bits(o.y)), max(abs_bits(o.z), abs_bits(o.w))));
      }
    }
    wave_sync();
  }
  if (amax) amax_publish(amax, amax_stride, b, (blockIdx.x * 4 + wave) * gridDim.z + blockIdx.z, row_max, lane);
}

// ---- utterance-pair variant ---------------------------------------------------------------------------------------
// The kernel above is VALU-issue bound from K = 51 up (2.5 packed FMAs per tap and 4 outputs, because time-adjacent
// output pairs only line up with even-aligned register pairs for every other tap).  Here the two halves of every
// packed-fp32 operation are the SAME (channel, frame) of two different utterances b0 = 2*blockIdx.y and b0 + 1:
//   * the LDS window holds pairs (x[b0][c][t], x[b0+1][c][t]); a pair is an aligned register pair for every tap, so
//     a tap costs exactly one v_pk_fma_f32 per output pair -- 2.0 per tap and 4 outputs, nothing discarded;
//   * both utterances share the channel's taps: K wave-uniform scalars, broadcast to both halves by op_sel;
//   * every lane slides over 8 consecutive frames (K + 7 window pairs from LDS instead of 2 x (K + 3) samples), which
//     also halves the LDS read traffic per output;
//   * a lane's window starts 64 bytes (4 quads of 16 B) after its neighbour's, a 4-way bank conflict for
//     ds_read_b128 (16 slots of 16 B, serviced in four 16-lane groups whose lane numbers cover every residue mod 16).
//     The window is therefore stored with one pad quad after every 4 (quad L at L + L/4): lane l starts at slot 5 l,
//     an odd stride, so the 16 lanes of a group always hit 16 different slots, and because 4 l is a multiple of 4
//     the pad seen at window offset q is q/4 for every lane -- all reads stay immediate offsets from one base.
// Odd batch tail: the last utterance is paired with itself and stored once.
// SUB (1, 2 or 4): utterance PAIRS per wavefront.  SUB = 1 is the tile described above (64 lanes x 8 frames = 512 frames of
// one pair).  A row pitch off the 512-frame grid leaves a short tail tile -- T' = 516 is 640 columns: one full tile and
// 128 columns -- which the SUB = 1 form computes at full price (depthwise 1.75 -> 2.75 ms per step at 10.3 s clips).  The
// tail launch instead gives each pair 64 / SUB lanes (SUB = 4: 16 lanes x 8 frames = 128 frames, SUB = 2: 256) and a
// window of its own in the wavefront's LDS region: same channel, so the taps stay wave-uniform scalars, and the FMA phase
// is instruction for instruction the one of the full tile -- a quarter (half) of the wavefront-passes for the tail.
template <int K, int DIL = 1, int SUB = 1>
struct PairGeom {
  static constexpr int LP = 64 / SUB;                               // lanes per pair
  static constexpr int PAD = DIL > 1 ? (DIL * K) / 2 - 1 : K / 2;   // get_same_padding (jasper.py:60-65)
  static constexpr int PADL = (PAD + 3) & ~3;
  static constexpr int OFF = PADL - PAD;
  static constexpr int NP = OFF + DIL * (K - 1) + 8;                // pairs in one lane's register window
  static constexpr int NQ = (NP + 1) / 2;                           // quads = 2 pairs = one ds_read_b128
  static constexpr int QNEED = 4 * (LP - 1) + NQ;                   // logical quads the last lane's window reaches
  static constexpr int NLD = (QNEED + 2 * LP - 1) / (2 * LP);       // staging float4 per lane and utterance
  static constexpr int LAST = (QNEED - 2 * LP * (NLD - 1) + 1) / 2; // lanes (of a pair's LP) that take part in the last slab
  static constexpr int QUADS = 2 * LP * (NLD - 1) + 2 * (LAST < LP ? LAST : LP);   // logical quads staged per pair
  static constexpr int PHYS1 = QUADS + QUADS / 4 + 1;               // + one pad quad per 4
  // SUB > 1: the pairs' regions start a multiple of 16 quads apart, so that the 16 lanes of a ds_read_b128 service group
  // -- which then straddle two pairs -- still hit 16 different 16-byte slots (5 ll mod 16 is a permutation)
  static constexpr int PHYS_SUB = SUB == 1 ? PHYS1 : (PHYS1 + 15) / 16 * 16;
  static constexpr int PHYS = SUB * PHYS_SUB;
  static constexpr int JSTRIDE = 2 * LP + LP / 2;                   // physical quads between a lane's staging slabs
  static constexpr int TB = 8;                                      // taps per software-pipeline block
  static constexpr int NB = (K + TB - 1) / TB;
  // one past the last quad that taps [0, min(K, TB*(blk+1))) touch
  static constexpr int qend(int blk) {
    const int klast = (K < TB * (blk + 1) ? K : TB * (blk + 1)) - 1;
    return (OFF + DIL * klast + 7) / 2 + 1;
  }
};

// grid (C/4, ceil(B/2), ceil(ldy/512)), block 256 = 4 wavefronts, one (channel, utterance pair, 512-frame tile) each.
// (Walking several pairs per wavefront with the next pair's rows prefetched measured 5-15% slower than simply keeping
// 6-7 short-lived wavefronts per SIMD resident, so there is no row loop.)
// t_base: first frame of this launch's tile 0; tile0 / tiles_total: this launch's tiles in the layer's numbering (slots of
// the maxima table are c * tiles_total + tile).
// One workgroup's work: `group` = index of its SUB pairs, `tile` = time tile (512 frames each; a SUB > 1 tile is the short one
// that starts at 512 * tile), slots of the maxima table are c * tiles_total + tile.
template <int K, int DIL, int SUB>
__device__ __forceinline__ void dw_pair_body(v4f* __restrict__ lds4, const float* __restrict__ x, int64_t ldx,
                                             const float* __restrict__ w, const int32_t* __restrict__ lens_in,
                                             const int32_t* __restrict__ lens_out, int channels, int batch,
                                             float* __restrict__ y, int64_t ldy, unsigned* __restrict__ amax,
                                             int amax_stride, int group, int tile, int tiles_total) {
  using G = PairGeom<K, DIL, SUB>;
  constexpr int NLD = G::NLD, LP = G::LP;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // (Mapping workgroups to XCDs so that XCD k gets the utterances the GEMM kernels give it was measured: depthwise
  // +5 %, GEMM -0.3 % -- nothing useful survives the kernel boundary in an XCD's L2.)
  const int c = blockIdx.x * 4 + wave;
  const int sub = SUB == 1 ? 0 : lane / LP, ll = SUB == 1 ? lane : lane % LP;   // pair of this lane, lane inside the pair
  const int n_pairs = (batch + 1) / 2;
  const int pp = group * SUB + sub;
  const bool live = SUB == 1 || pp < n_pairs;                       // SUB > 1: the last wavefront may hold fewer pairs
  const int b0 = 2 * (live ? pp : n_pairs - 1);                     // (idle lanes compute the last pair again and store nothing)
  const bool twin = b0 + 1 < batch;
  const int b1 = twin ? b0 + 1 : b0;
  const int t_start = tile * kTile;
  v4f* win = lds4 + wave * G::PHYS + sub * G::PHYS_SUB;

  // ---- staging: both rows, branch-free (clamped address + selects), all loads in flight together ----
  // The last slab only reaches as far as the pair's last lane's window: lanes past it neither load nor store.
  v4f s0[NLD], s1[NLD];
  const bool in_last = G::LAST >= LP || ll < G::LAST;
  {
    const float* xr0 = x + ((int64_t)b0 * channels + c) * ldx;
    const float* xr1 = x + ((int64_t)b1 * channels + c) * ldx;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int t = t_start - G::PADL + 4 * (ll + LP * j);
      int tc = t < 0 ? 0 : t;
      tc = tc > (int)ldx - 4 ? (int)ldx - 4 : tc;
      if (j + 1 < NLD || in_last) {
        s0[j] = *reinterpret_cast<const v4f*>(xr0 + tc);
        s1[j] = *reinterpret_cast<const v4f*>(xr1 + tc);
      }
    }
  }
  const float* wg = w + (int64_t)c * K;
  // all taps resident in SGPRs; longer kernels stream them block by block
  constexpr bool PIN = K <= 44;
  float wc[PIN ? K : 1];
  if constexpr (PIN) {
#pragma unroll
    for (int k = 0; k < K; ++k) wc[k] = wg[k];
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("" : "+s"(wc[k]));
  }
  // MaskedConv1d: x.masked_fill(t >= lens, 0) (jasper.py:113-118); t < 0 is the conv zero padding.  t and PADL are
  // multiples of 4, so a float4 is either entirely left of frame 0 or not at all.
  auto masked = [&](v4f v, int t, int len) {
    const int n = t < 0 ? 0 : len - t;
    v.x = n > 0 ? v.x : 0.f;
    v.y = n > 1 ? v.y : 0.f;
    v.z = n > 2 ? v.z : 0.f;
    v.w = n > 3 ? v.w : 0.f;
    return v;
  };
  v4f* wr = win + 2 * ll + (ll >> 1);   // logical quads 2*(ll + LP j), +1 -> physical (+ JSTRIDE j), +1
  // sliding window: logical quad 4*ll + q lives at 4*ll + q + ((4*ll + q) >> 2) = 5*ll + q + (q >> 2)
  const v4f* rb = win + 5 * ll;

  const int len_in0 = lens_in[b0], len_in1 = lens_in[b1];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int t = t_start - G::PADL + 4 * (ll + LP * j);
    const v4f a = masked(s0[j], t, len_in0), bq = masked(s1[j], t, len_in1);
    const v4f q0 = {a.x, bq.x, a.y, bq.y}, q1 = {a.z, bq.z, a.w, bq.w};
    if (j + 1 < NLD || in_last) {
      wr[G::JSTRIDE * j] = q0;
      wr[G::JSTRIDE * j + 1] = q1;
    }
  }
  wave_sync();

  v2f xw[2 * G::NQ];
  auto load_quads = [&](int qa, int qb) {
#pragma unroll
    for (int q = qa; q < qb; ++q) {
      const v4f v = rb[q + (q >> 2)];
      xw[2 * q] = v.xy;
      xw[2 * q + 1] = v.zw;
    }
  };
  v2f acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = v2f{0.f, 0.f};

  load_quads(0, G::qend(0));
  float wn[PIN ? 1 : G::TB];
  if constexpr (!PIN) {
#pragma unroll
    for (int k = 0; k < G::TB; ++k) wn[k] = wg[k < K ? k : K - 1];
  }
#pragma unroll
  for (int blk = 0; blk < G::NB; ++blk) {
    float wb[PIN ? 1 : G::TB];
    if constexpr (!PIN) {
#pragma unroll
      for (int k = 0; k < G::TB; ++k) wb[k] = wn[k];
      if (blk + 1 < G::NB) {
        int off = G::TB * (blk + 1);
        asm volatile("" : "+s"(off));   // opaque: keeps the compiler from hoisting every block's taps to the top
#pragma unroll
        for (int k = 0; k < G::TB; ++k) wn[k] = wg[off + (G::TB * (blk + 1) + k < K ? k : K - 1 - G::TB * (blk + 1))];
      }
    }
    if (blk + 1 < G::NB) load_quads(G::qend(blk), G::qend(blk + 1));
#pragma unroll
    for (int k = G::TB * blk; k < G::TB * (blk + 1) && k < K; ++k) {
      float wk;
      if constexpr (PIN) wk = wc[k];
      else wk = wb[k - G::TB * blk];
      const v2f w2 = {wk, wk};
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] = __builtin_elementwise_fma(w2, xw[G::OFF + DIL * k + r], acc[r]);
    }
    // anchor: without it LLVM sinks the whole FMA chain into the conditional store block below, which serialises
    // "load the entire window" -> "all FMAs" and keeps every window register live
#pragma unroll
    for (int r = 0; r < 8; ++r) asm volatile("" : "+v"(acc[r]));
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- store: the following 1x1 MaskedConv1d masks with lens_out, so frames past it are written as zeros ----
  const int t = t_start + 8 * ll;
  const int n0 = lens_out[b0] - t, n1 = lens_out[b1] - t;
  float* y0 = y + ((int64_t)b0 * channels + c) * ldy + t;
  float* y1 = y + ((int64_t)b1 * channels + c) * ldy + t;
  unsigned m0 = 0, m1 = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (t + 4 * h < ldy && live) {
      v4f o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o0[e] = n0 > 4 * h + e ? acc[4 * h + e].x : 0.f;
        o1[e] = n1 > 4 * h + e ? acc[4 * h + e].y : 0.f;
      }
      *reinterpret_cast<v4f*>(y0 + 4 * h) = o0;
      if (twin) *reinterpret_cast<v4f*>(y1 + 4 * h) = o1;
      if (amax) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          m0 = max(m0, abs_bits(o0[e]));
          m1 = max(m1, abs_bits(o1[e]));
        }
      }
    }
  }
  if (amax) {   // masked outputs only (zeros past lens_out): the maximum over the utterance's valid frames
    const int slot = c * tiles_total + tile;
    if constexpr (SUB == 1) {
      amax_publish(amax, amax_stride, b0, slot, m0, lane);
      if (twin) amax_publish(amax, amax_stride, b1, slot, m1, lane);
    } else {
      // one reduction inside each 16-lane row (DPP), the rows of a pair combined through SGPRs; lane 0 stores every pair's
#define VASR_DPP(xx, ctrl) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(xx), (ctrl), 0xF, 0xF, false))
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        unsigned m = u ? m1 : m0;
        m = max(m, VASR_DPP(m, 0xB1));
        m = max(m, VASR_DPP(m, 0x4E));
        m = max(m, VASR_DPP(m, 0x141));
        m = max(m, VASR_DPP(m, 0x140));
        unsigned r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = (unsigned)__builtin_amdgcn_readlane((int)m, 16 * k);
#pragma unroll
        for (int sp = 0; sp < SUB; ++sp) {
          const unsigned ms = SUB == 4 ? r[sp] : max(r[2 * sp], r[2 * sp + 1]);
          const int bq0 = 2 * (group * SUB + sp) + u;
          if (lane == 0 && bq0 < batch) amax[(int64_t)bq0 * amax_stride + slot] = ms;
        }
      }
#undef VASR_DPP
    }
  }
}

// grid (C/4, pairs, tiles): tiles below nt_main are full 512-frame tiles (one pair per wavefront); the last one, when
// SUBT > 1, is the short tail (SUBT pairs per wavefront: only the first ceil(pairs / SUBT) workgroups of the row have work).
// ONE launch for both: as a launch of its own the tail cost almost a full tile (few, latency-bound wavefronts plus a
// dispatch gap: 0.88 of 1.05 ms per step at 10.3 s clips); inside the main launch its wavefronts run beside the others.
template <int K, int DIL, int SUBT>
__global__ __launch_bounds__(256) void dw_pair_kernel(const float* __restrict__ x, int64_t ldx,
                                                      const float* __restrict__ w,
                                                      const int32_t* __restrict__ lens_in,
                                                      const int32_t* __restrict__ lens_out, int channels, int batch,
                                                      float* __restrict__ y, int64_t ldy, unsigned* __restrict__ amax,
                                                      int amax_stride, int nt_main) {
  constexpr int PHYS = PairGeom<K, DIL, 1>::PHYS > PairGeom<K, DIL, SUBT>::PHYS ? PairGeom<K, DIL, 1>::PHYS : PairGeom<K, DIL, SUBT>::PHYS;
  __shared__ v4f lds4[4 * PHYS];
  const int tile = blockIdx.z, tiles_total = gridDim.z;
  if (SUBT == 1 || tile < nt_main) {
    dw_pair_body<K, DIL, 1>(lds4, x, ldx, w, lens_in, lens_out, channels, batch, y, ldy, amax, amax_stride, blockIdx.y, tile,
                            tiles_total);
  } else {
    if ((int)blockIdx.y * SUBT >= (batch + 1) / 2) return;
    dw_pair_body<K, DIL, SUBT>(lds4, x, ldx, w, lens_in, lens_out, channels, batch, y, ldy, amax, amax_stride, blockIdx.y,
                               tile, tiles_total);
  }
}

template <int K, int DIL>
void launch_dw_pair(const float* x, int64_t ldx, const float* w, const int32_t* li, const int32_t* lo, int batch,
                    int channels, float* y, int64_t ldy, hipStream_t st, AmaxTab* amax) {
  const int n_pairs = (batch + 1) / 2;
  // full 512-frame tiles, then a tail of 128 or 256 columns (the pitch is a multiple of 128; 384 runs as a full tile)
  int nt_main = (int)(ldy / kTile);
  int rest = (int)(ldy - (int64_t)nt_main * kTile);
  if (rest > 256) { nt_main += 1; rest = 0; }
  const int tiles_total = nt_main + (rest ? 1 : 0);
  if (amax) amax->n = channels * tiles_total;
  unsigned* ap = amax ? amax->p : nullptr;
  const int as = amax ? amax->stride : 0;
  dim3 grid(channels / 4, n_pairs, tiles_total);
  if (rest == 128)
    VASR_LAUNCH((dw_pair_kernel<K, DIL, 4>), grid, dim3(256), 0, st, x, ldx, w, li, lo, channels, batch, y, ldy, ap, as, nt_main);
  else if (rest == 256)
    VASR_LAUNCH((dw_pair_kernel<K, DIL, 2>), grid, dim3(256), 0, st, x, ldx, w, li, lo, channels, batch, y, ldy, ap, as, nt_main);
  else
    VASR_LAUNCH((dw_pair_kernel<K, DIL, 1>), grid, dim3(256), 0, st, x, ldx, w, li, lo, channels, batch, y, ldy, ap, as, nt_main);
}

// Any kernel / stride / dilation / row pitch: one thread per output, taps straight from L1/L2.
// Used for the stride-2 prologue block (64 channels) and the dilated K=87 block.
__global__ __launch_bounds__(256) void dw_conv_generic_kernel(const float* __restrict__ x, int64_t ldx,
                                                              int frames_in, const float* __restrict__ w,
                                                              const int32_t* __restrict__ lens_in,
                                                              const int32_t* __restrict__ lens_out,
                                                              int channels, int K, int stride, int dil, int pad,
                                                              float* __restrict__ y, int64_t ldy,
                                                              unsigned* __restrict__ amax, int amax_stride) {
  constexpr int kSpan = 2048;
  __shared__ float xs[kSpan];
  const int c = blockIdx.y, b = blockIdx.z;
  const int t0 = blockIdx.x * blockDim.x;
  const int t = t0 + threadIdx.x;
  int len_in = lens_in[b];
  if (len_in > frames_in) len_in = frames_in;
  const int64_t row = (int64_t)b * channels + c;
  const float* xr = x + row * ldx;
  const float* wc = w + (int64_t)c * K;   // block-uniform: scalar loads
  // the masked input span of the block's 256 outputs goes through LDS once (the stride-2 prologue block read every
  // sample ~16 times from L1); a masked sample is a zero term of the same fmaf chain
  const int span = 255 * stride + dil * (K - 1) + 1, s_base = t0 * stride - pad;
  const bool staged = span <= kSpan;
  if (staged) {
    for (int j = threadIdx.x; j < span; j += blockDim.x) {
      const int s = s_base + j;
      xs[j] = (s >= 0 && s < len_in) ? xr[s] : 0.f;
    }
    __syncthreads();
  }
  float acc = 0.f;
  if (t < ldy && t < lens_out[b]) {
    if (staged) {
      const float* xt = xs + threadIdx.x * stride;
      for (int k = 0; k < K; ++k) acc = fmaf(wc[k], xt[k * dil], acc);
    } else {
      const int s0 = t * stride - pad;
      for (int k = 0; k < K; ++k) {
        const int s = s0 + k * dil;
        if (s >= 0 && s < len_in) acc = fmaf(wc[k], xr[s], acc);
      }
    }
  }
  if (t < ldy) y[row * ldy + t] = acc;
  if (amax)   // whole wavefronts reach this point
    amax_publish(amax, amax_stride, b, (c * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6), abs_bits(acc), threadIdx.x & 63);
}

// MaskedConv1d.get_seq_len chain (len_chain.h), one thread per utterance.
__global__ void len_chain_kernel(const int64_t* __restrict__ seq, int batch, const LenStep* __restrict__ steps,
                                 int n_steps, int32_t* __restrict__ lens_tab, float* __restrict__ enc_len,
                                 const int64_t* __restrict__ wav_len, int hop, int frames_cap) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  len_chain_body(b, b < batch ? seq[b] : 0, batch, steps, n_steps, lens_tab, enc_len, wav_len, hop, frames_cap);
}

// [rows][frames] (pitch src_ld) -> [rows][dst_ld], zero filled past `frames`
__global__ __launch_bounds__(256) void repad_kernel(const float* __restrict__ src, int64_t src_ld, int frames,
                                                    float* __restrict__ dst, int64_t dst_ld) {
  const int64_t r = blockIdx.y;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < dst_ld; t += gridDim.x * blockDim.x)
    dst[r * dst_ld + t] = t < frames ? src[r * src_ld + t] : 0.f;
}

template <int K>
void launch_dw_t(const float* x, int64_t ldx, const float* w, const int32_t* li, const int32_t* lo, int batch,
                 int channels, float* y, int64_t ldy, hipStream_t st, AmaxTab* amax) {
  const unsigned tiles = (unsigned)((ldy + kTile - 1) / kTile);
  unsigned* ap = amax ? amax->p : nullptr;
  const int as = amax ? amax->stride : 0;
  // (one channel per wavefront; 2 / 4 / 8 channels per wavefront were measured slower and are no longer instantiated)
  dim3 grid(channels / 4, batch, tiles);
  if (amax) amax->n = grid.x * 4 * tiles;
  VASR_LAUNCH((dw_conv_kernel<K, 1>), grid, dim3(256), 0, st, x, ldx, w, li, lo, channels, y, ldy, ap, as);
}

}  // namespace

int depthwise_amax_slots(int channels, int64_t ldy) {
  // the generic kernel (any shape) uses channels * ceil(ldy / 256) * 4 slots, the tiled ones channels * ceil(ldy / 512)
  return channels * (int)((ldy + 255) / 256) * 4;
}

static int launch_depthwise_impl(const float* x, int64_t ldx, int frames_in, const float* w, const int32_t* lens_in,
                                 const int32_t* lens_out, int batch, int channels, int kernel, int stride, int dilation,
                                 int pad, float* y, int64_t ldy, hipStream_t st, AmaxTab* amax);

int launch_depthwise(const float* x, int64_t ldx, int frames_in, const float* w, const int32_t* lens_in,
                     const int32_t* lens_out, int batch, int channels, int kernel, int stride, int dilation,
                     int pad, float* y, int64_t ldy, hipStream_t st, AmaxTab* amax) {
  if (amax && amax->stride < depthwise_amax_slots(channels, ldy)) return -1;   // table too small for this shape
  return launch_depthwise_impl(x, ldx, frames_in, w, lens_in, lens_out, batch, channels, kernel, stride, dilation, pad, y,
                               ldy, st, amax);
}

static int launch_depthwise_impl(const float* x, int64_t ldx, int frames_in, const float* w, const int32_t* lens_in,
                                 const int32_t* lens_out, int batch, int channels, int kernel, int stride, int dilation,
                                 int pad, float* y, int64_t ldy, hipStream_t st, AmaxTab* amax) {
  const bool aligned = channels % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 &&
                       (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
  const bool pair = dev_switches().dw_pair;   // devtools build: VASR_DW_PAIR=0 sends every shape to the generic kernel
  if (pair && aligned && stride == 1) {
    if (dilation == 2 && kernel == 87 && pad == 86)
      { launch_dw_pair<87, 2>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st, amax); return 0; }
    if (dilation == 1 && pad == kernel / 2) {
      switch (kernel) {
        case 33: { launch_dw_pair<33, 1>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st, amax); return 0; }
        case 39: { launch_dw_pair<39, 1>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st, amax); return 0; }
        case 51: { launch_dw_pair<51, 1>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st, amax); return 0; }
        case 63: { launch_dw_pair<63, 1>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st, amax); return 0; }
        case 75: { launch_dw_pair<75, 1>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st, amax); return 0; }
        default: break;
      }
    }
  }
  if (aligned && stride == 1 && dilation == 2 && kernel == 87 && pad == 86) {
    dim3 grid(channels / 4, batch, (unsigned)((ldy + kTile - 1) / kTile));
    if (amax) amax->n = grid.x * 4 * grid.z;
    VASR_LAUNCH((dw_conv_kernel<87, 1, 2>), grid, dim3(256), 0, st, x, ldx, w, lens_in, lens_out, channels,
                       y, ldy, amax ? amax->p : nullptr, amax ? amax->stride : 0);
    return 0;
  }
  const bool fast = aligned && stride == 1 && dilation == 1 && pad == kernel / 2;
  if (fast) {
    switch (kernel) {
      case 33: { launch_dw_t<33>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st, amax); return 0; }
      case 39: { launch_dw_t<39>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st, amax); return 0; }
      case 51: { launch_dw_t<51>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st, amax); return 0; }
      case 63: { launch_dw_t<63>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st, amax); return 0; }
      case 75: { launch_dw_t<75>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st, amax); return 0; }
      default: break;
    }
  }
  dim3 grid((unsigned)((ldy + 255) / 256), channels, batch);
  if (amax) amax->n = channels * grid.x * 4;
  VASR_LAUNCH(dw_conv_generic_kernel, grid, dim3(256), 0, st, x, ldx, frames_in, w, lens_in, lens_out,
                     channels, kernel, stride, dilation, pad, y, ldy, amax ? amax->p : nullptr, amax ? amax->stride : 0);
  return 0;
}

// max |x| per utterance over columns < lens[b] (or < frames) of x[b][rows][ld]: for tensors whose producer does not
// publish its maxima (port tensors, the fp32 GEMM kernel)
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t ld, int rows, int frames,
                                                   const int32_t* __restrict__ lens, unsigned* __restrict__ amax,
                                                   int amax_stride) {
  const int b = blockIdx.y;
  int n = lens ? lens[b] : frames;
  n = n < frames ? n : frames;
  unsigned m = 0;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float* xr = x + ((int64_t)b * rows + r) * ld;
    for (int t = threadIdx.x; t < n; t += blockDim.x) m = max(m, abs_bits(xr[t]));
  }
  amax_publish(amax, amax_stride, b, blockIdx.x * 4 + (threadIdx.x >> 6), m, threadIdx.x & 63);
}

void launch_amax(const float* x, int64_t ld, int rows, int frames, const int32_t* lens, int batch, AmaxTab* amax,
                 hipStream_t st) {
  const int gx = rows < 64 ? rows : 64;
  amax->n = gx * 4;
  hipLaunchKernelGGL(amax_kernel, dim3(gx, batch), dim3(256), 0, st, x, ld, rows, frames, lens, amax->p, amax->stride);
}

void launch_len_chain(const int64_t* seq, int batch, const LenStep* d_steps, int n_steps, int32_t* lens_tab,
                      float* enc_len, hipStream_t st, const int64_t* wav_len, int hop, int frames_cap) {
  hipLaunchKernelGGL(len_chain_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, seq, batch, d_steps, n_steps,
                     lens_tab, enc_len, wav_len, hop, frames_cap);
}

void launch_repad(const float* src, int64_t src_ld, int rows, int frames, float* dst, int64_t dst_ld,
                  hipStream_t st) {
  dim3 grid((unsigned)((dst_ld + 255) / 256), rows);
  hipLaunchKernelGGL(repad_kernel, grid, dim3(256), 0, st, src, src_ld, frames, dst, dst_ld);
}

}  // namespace vasr
