// DEVTOOLS BUILD ONLY -- a measured prototype, not on the product path (it is bit-identical to the
// shipped GEMM and 1.4 us = 2.8 % faster per 512 -> 512 layer, which does not pay for the conversion it moves into the
// depthwise kernel; tests/test_gpu_round3.py keeps it honest).
//
// Pointwise (1x1) convolution GEMM on PRE-SPLIT activations ("P4" tensors), fp16-split arithmetic (kF16x2 of
// encoder_pw_split.hip: same products, same accumulation order, same epilogue; reference
// nemo/collections/asr/parts/jasper.py:113-132, :374-392).
//
// In encoder_pw_split.hip every workgroup converts its activation tile to fp16 hi / lo planes while it stages it:
// ~50 VALU instructions, 8 LDS writes and 16 row loads per thread and 64-deep K chunk, in the instruction streams of the
// wavefronts that issue the MFMAs.  Here the PRODUCER of the tensor (it would be the Toeplitz depthwise kernel, whose
// epilogue is HBM-bound) has already scaled and split it, and this kernel's staging is an LDS-DMA: no registers, no VALU,
// no LDS write instructions.
//
// P4 layout (same pitch and footprint as the fp32 tensor it replaces): row (b, c) is ld / 4 groups of 16 bytes, group g =
// frames 4 g .. 4 g + 3 as [4 x fp16 hi | 4 x fp16 lo] of s_b * x, s_b the utterance's power-of-two scale (its inverse
// comes in a [B] float table); channels with (c & 2) store the halves SWAPPED ([lo | hi]) -- see the bank argument below.
//
// LDS image of a chunk: 64 rows (channels) x 512 bytes (128 frames), row-major like the tensor, filled by
// global_load_lds_dwordx4 (destination = wave-uniform base + 16 * lane: two rows per instruction; the SOURCE address is
// per lane, which is where the swizzle goes: odd rows are stored with their 16-byte groups XOR 8, i.e. the two 128-byte
// halves of each 256-byte half row exchanged).  B fragments come out of that row-major image through the transposing
// read ds_read_b64_tr_b16 (tools/probes/tr_probe.hip: in every 16-lane group lane i SUPPLIES the address of 8 bytes --
// row i / 4, column quad i % 4 of a 4 x 16 block -- and RECEIVES column i, one element from each of the 4 rows): two
// reads give a lane the 8 consecutive channels of its column, the B operand of v_mfma_f32_32x32x16_f16.  Banks: the 32
// lanes of a half wave read 4 rows x 8 groups x 8 bytes; the groups of a row lie 16 bytes apart (the other half of each
// group is the other plane), so rows r and r + 2 interleave through the swapped halves and rows r and r + 1 through
// the XOR 8: 64 distinct banks.
//
//   tile 512 x 128 on 8 wavefronts (2 x 4 MFMA tiles each) like the shipped kernel; weight fragments three k-steps
//   ahead, B fragments two n-tiles ahead, one LDS-only barrier per chunk with a counted vmcnt for the DMA.
#include <cstdlib>
#include <type_traits>

#include "vasr_internal.h"
#include "vasr_device.h"

namespace vasr {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using v4f = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
#define VASR_LDS __attribute__((address_space(3)))
#define VASR_GLB __attribute__((address_space(1)))

constexpr int BKC = 64, STEPS = BKC / 16, TM = 2, TN = 4, NW = 8, BM = 32 * TM * NW, BN = 32 * TN, NT = 64 * NW;
constexpr int kRowBytes = BN * 4;            // one channel's 128 frames: 32 groups of [hi4 | lo4]
constexpr int kBufBytes = BKC * kRowBytes;   // 32 KB per chunk
constexpr int kLds = 2 * kBufBytes;
#ifndef VASR_P4_ABLATE
#define VASR_P4_ABLATE 0   // dev-only timing ablations (results wrong): 1 no DMA in the main loop, 2 plain ds_read_b128 fragments
#endif

__device__ __forceinline__ f32x16 mma(uint4 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), b, c, 0, 0, 0);
}
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(NT, 2) void pw_gemm_p4_kernel(PwP4Args a, int blocks_m, int tiles_t, int n_blocks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [2][64][512]
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

  const int ksteps = a.K / 16;
  const uint4* __restrict__ ap = a.wt + ((int64_t)((m0 + wm) / 32) * ksteps) * 2 * 64 + lane;
  const int64_t a_tile = (int64_t)ksteps * 2 * 64;
  const int64_t ld4 = a.ldx / 4;                                        // 16-byte groups per row
  const uint4* __restrict__ xb = a.x + (int64_t)b * a.K * ld4 + t0 / 4;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- weights: fragment sets in rotation, three k-steps ahead ----
  constexpr int WSETS = 4;
  uint4 aw[WSETS][TM][2];
  auto aload = [&](int s, uint4 (&dst)[TM][2]) {
    const int sc = s < ksteps ? s : ksteps - 1;   // harmless re-reads past the end
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p) dst[i][p] = ap[i * a_tile + ((int64_t)sc * 2 + p) * 64];
  };

  // ---- activations: LDS-DMA, two rows (1 KB) per instruction, four instructions per wavefront and chunk ----
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds);
  const int dma_row = lane >> 5;                                   // row parity inside the instruction's pair = c & 1
  const int dma_grp = (lane & 31) ^ (dma_row << 3);                // source group that lands in slot lane & 31
  auto dma = [&](int c, int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 2 * (wave * 4 + q);
      const uint4* src = xb + (int64_t)(c * BKC + row + dma_row) * ld4 + dma_grp;
      // Inline asm, not __builtin_amdgcn_global_load_lds: with the builtin in flight hipcc (ROCm 7.2) waits vmcnt(0) at the
      // next use of ANY ordinary load -- here the weight fragments of the very next MFMA -- i.e. it drains the DMA right
      // after issuing it.  An asm load is absent from the compiler's count; its waits for the weight sets then cover
      // four operations more than they name, which completes a DMA at the latest three k-steps after its issue
      // (in-order counter) -- and the chunk-end wait below names the count exactly.  M0 = LDS byte address of lane 0.
      const unsigned dst = lds_base + buf * kBufBytes + row * kRowBytes;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
  };

  // ---- B fragments: transposing reads.  Supply side of lane L: group g = L >> 4 (k-half g >> 1, column half g & 1),
  //      i = L & 15: row r = i >> 2, column quad q = i & 3 ----
  const int tg = lane >> 4, ti = lane & 15, tr = ti >> 2, tq = ti & 3;
  const int t_base = (8 * (tg >> 1) + tr) * kRowBytes + 64 * (tg & 1) + 16 * tq;
  // plane p of row r sits in half p ^ (r >> 1 & 1) of its groups; odd rows have n-tiles 0 <-> 1, 2 <-> 3 exchanged
  VASR_LDS unsigned char* tb[2][2];   // [n-tile parity][plane]
#pragma unroll
  for (int jp = 0; jp < 2; ++jp)
#pragma unroll
    for (int p = 0; p < 2; ++p)
      tb[jp][p] = (VASR_LDS unsigned char*)lds + t_base + ((tr & 1) ? (jp ? -128 : 128) : 0) + 8 * (p ^ ((tr >> 1) & 1));
  auto bread = [&](int buf, int s, int j, int p) -> f16x8 {
    VASR_LDS unsigned char* q = tb[j & 1][p] + buf * kBufBytes + s * (16 * kRowBytes) + 128 * j;
    if (VASR_P4_ABLATE & 2) return __builtin_bit_cast(f16x8, *(VASR_LDS uint4*)((VASR_LDS unsigned char*)lds + buf * kBufBytes + ((s * 2 + p) * 128 + j * 32 + (lane & 31) + 32 * (lane >> 5)) * 16));
    const v4s lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VASR_LDS v4s*)q);
    const v4s hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VASR_LDS v4s*)(q + 4 * kRowBytes));
    return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  const int zf = a.zero_from ? a.zero_from[b] : 0x7fffffff;
  const int nchunks = t0 >= zf ? 0 : a.K / BKC;   // (even: launch_pointwise_p4 checks)
  float out_scale = 1.f;
  if (nchunks) {
    dma(0, 0);
    if (VASR_P4_ABLATE & 1) dma(1, 1);   // (timing ablation: both buffers hold real data, nothing is restaged)
    aload(0, aw[0]);
    aload(1, aw[1]);
    aload(2, aw[2]);
    out_scale = a.x_inv_scale[b] * a.w_inv_scale;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  auto run_chunk = [&](const int c, auto buf_tag, auto last_tag) {
    constexpr int buf = decltype(buf_tag)::value;
    constexpr bool LAST = decltype(last_tag)::value;
    f16x8 bf[3][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) bf[0][p] = bread(buf, 0, 0, p);
#pragma unroll
    for (int p = 0; p < 2; ++p) bf[1][p] = bread(buf, 0, 1, p);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      aload(c * STEPS + s + 3, aw[(s + 3) % WSETS]);
      // the next chunk's rows: behind this k-step's weight request, so that the three weight sets already in flight stay
      // older than the DMA (s_waitcnt counts in order)
      if (!LAST && s == 0 && !(VASR_P4_ABLATE & 1)) dma(c + 1, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      uint4 (&cw)[TM][2] = aw[s % WSETS];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int g = s * TN + j;
        if (g + 2 < STEPS * TN) {
#pragma unroll
          for (int p = 0; p < 2; ++p) bf[(g + 2) % 3][p] = bread(buf, (g + 2) / TN, (g + 2) % TN, p);
        }
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 bh = bf[g % 3][0], bl = bf[g % 3][1];
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mma(cw[i][1], bh, acc[i][j]);   // lo * hi
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mma(cw[i][0], bl, acc[i][j]);   // hi * lo
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mma(cw[i][0], bh, acc[i][j]);   // hi * hi
      }
    }
    // the DMA of the next chunk was followed by the weight requests of k-steps 1 .. 3 (3 x 4 loads): vmcnt(12) = landed
    asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  {
    int c = 0;
    for (; c + 2 < nchunks; c += 2) {
      run_chunk(c, B0{}, std::false_type{});
      run_chunk(c + 1, B1{}, std::false_type{});
    }
    if (nchunks) {
      run_chunk(c, B0{}, std::false_type{});
      run_chunk(c + 1, B1{}, std::true_type{});
    }
  }

  // ---- epilogue (as encoder_pw_split.hip's interior-tile path): un-scale, BN affine, ReLU, rows through LDS ----
  if (a.relu & 2) return;   // debug: skip the epilogue
  const int ylen = a.amax_y.p ? (a.lens_y ? a.lens_y[b] : a.frames) : 0;
  unsigned ymax = 0;
  const float relu_floor = (a.relu & 1) ? 0.f : -__builtin_inff();
  float* stage = reinterpret_cast<float*>(lds) + wave * (2 * 8 * BN);
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
      wave_fence();
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int j = 0; j < TN; ++j) buf[(4 * kh + rr) * BN + 32 * j + l31] = fmaf(acc[i][j][4 * q + rr] * out_scale, sc[rr], sh[rr]);
      wave_fence();
      v4f pv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int f = lane + 64 * k, row = f / (BN / 4), c4 = f % (BN / 4);
        pv[k] = *reinterpret_cast<const v4f*>(buf + row * BN + 4 * c4);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int f = lane + 64 * k, row = f / (BN / 4), c4 = f % (BN / 4);
        const int m = mq + row, t = t0 + 4 * c4;
        const v4f v = __builtin_elementwise_max(pv[k], v4f{relu_floor, relu_floor, relu_floor, relu_floor});
        *reinterpret_cast<v4f*>(a.y + ((int64_t)b * a.M + m) * a.ldy + t) = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned u = abs_bits(v[e]);
          ymax = (t + e < ylen && u > ymax) ? u : ymax;
        }
      }
    }
  }
  if (a.amax_y.p) amax_publish(a.amax_y.p, a.amax_y.stride, b, (mb * tiles_t + nt % tiles_t) * NW + wave, ymax, lane);
}

}  // namespace

bool pointwise_p4_supported(int M, int K, int64_t ldx, int64_t ldy) {
  return M % BM == 0 && K % (2 * BKC) == 0 && ldx % BN == 0 && ldy % 4 == 0 && ldy >= ldx;
}

// Returns 0, a hipError_t, or -1 when the shape is not covered (the caller then runs the fp32-input kernel).
int launch_pointwise_p4(const PwP4Args& a, hipStream_t st, int* amax_n) {
  if (!pointwise_p4_supported(a.M, a.K, a.ldx, a.ldy)) return -1;
  if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.y)) & 15) return -1;
  const int blocks_m = a.M / BM;
  const int tiles_t = (int)(a.ldx / BN);
  const int n_blocks = blocks_m * tiles_t * a.batch;
  if (a.amax_y.p) {
    const int n = blocks_m * tiles_t * NW;
    if (n > a.amax_y.stride) return (int)hipErrorInvalidValue;
    if (amax_n) *amax_n = n;
  }
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(pw_gemm_p4_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  if (attr != hipSuccess) return (int)attr;
  VASR_LAUNCH(pw_gemm_p4_kernel, dim3(n_blocks), dim3(NT), kLds, st, a, blocks_m, tiles_t, n_blocks);
  return 0;
}

// Host restatement of the layout for tests and tools: x [rows][ld] fp32 (one utterance), scale a power of two ->
// out [rows][ld / 4][8] fp16 bit patterns.
void pack_p4_reference(const float* x, int rows, int64_t ld, float scale, unsigned short* out) {
  for (int c = 0; c < rows; ++c)
    for (int64_t g = 0; g < ld / 4; ++g) {
      unsigned short h[4], l[4];
      for (int e = 0; e < 4; ++e) {
        const float v = x[(int64_t)c * ld + 4 * g + e] * scale;
        const _Float16 hh = (_Float16)v;
        const _Float16 ll = (_Float16)(v - (float)hh);
        h[e] = __builtin_bit_cast(unsigned short, hh);
        l[e] = __builtin_bit_cast(unsigned short, ll);
      }
      unsigned short* o = out + ((int64_t)c * (ld / 4) + g) * 8;
      const bool swap = (c & 2) != 0;
      for (int e = 0; e < 4; ++e) { o[(swap ? 4 : 0) + e] = h[e]; o[(swap ? 0 : 4) + e] = l[e]; }
    }
}

}  // namespace vasr
