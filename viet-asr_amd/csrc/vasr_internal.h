// Internal launch interface between vasr_api.cpp and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>

namespace vasr {

// Kernel-SELECTION switches: every product path that the default rules do not reach at a given shape (the packed-FMA
// depthwise, the two-kernel form of a fused sub-block, a pinned GEMM tile, the one-wavefront beam search ...) can be forced,
// so that the tests can hold it to the same goldens.  They exist in the DEVTOOLS build only (libvasr_hip_dev.so), are read
// from the environment ONCE per process (dev_switches(), vasr_api.cpp) and copied into every handle at vasr_create(); in the
// product library the struct is a compile-time constant of defaults and nothing looks at the environment.  (Rounds 1-5 also
// kept timing ablations and one-off experiments behind ~40 more switches and macros inside the kernels; their results are in
// DESIGN_HISTORY.md, their code is in the history -- `git show a54b0b9:viet-asr_amd/csrc`.)  VASR_GEMM and VASR_SLICES --
// documented modes with API equivalents -- are plain getenv at vasr_create().
struct DevSwitches {
  int pw3_tile = 0;             // VASR_PW3_TILE=1..5: pins the split GEMM's tile (512x128, 256x128, 128x64, 64x32, 256x64)
  int pw_lat = 1;               // VASR_PW_LAT=0: small batches stay on the chunked GEMM instead of the latency kernel
  bool dw_pair = true;          // VASR_DW_PAIR=0: the generic depthwise kernel instead of the pair kernel
  bool dw_mfma = true;          // VASR_DW_MFMA=0: packed-FMA depthwise kernels instead of the Toeplitz form
  int dw_upw = 0;               // VASR_DW_UPW=1..8: utterances one wavefront of the Toeplitz kernel walks
  bool fused = true;            // VASR_FUSED=0: depthwise and GEMM of a 256-channel sub-block as two kernels
  int fused_min_tiles = 0;      // VASR_FUSED_MIN_TILES=n: fuse from n tiles on, whatever the round-filling rule says
  int fused_tile = 0;           // VASR_FUSED_TILE=64|128: pins the fused kernel's tile width
  bool fused_residual = true;   // VASR_NO_FUSED_RESIDUAL=1: a block's residual branch as its own GEMM, not as a second K range
  int beam_group = -1;          // VASR_BEAM_GROUP=0|1: always one wavefront per utterance, 4: always four
};
#ifdef VASR_DEVTOOLS
const DevSwitches& dev_switches();
#else
inline const DevSwitches& dev_switches() { static constexpr DevSwitches k{}; return k; }
#endif

// Kernel-duration probe for vasr_profile_begin/end.  When armed, the next VASR_LAUNCH goes out through
// hipExtLaunchKernelGGL with (start, stop) events that take the dispatch packet's own begin / end timestamps, i.e. the
// kernel's execution time as rocprofv3 --kernel-trace reports it.  (Events recorded around a launch span "previous
// kernel done -> this kernel done" and so include the 2-3 us dispatch gap: 13 % on a 20 us depthwise layer.)
struct LaunchProbe { hipEvent_t start = nullptr, stop = nullptr; };
extern thread_local LaunchProbe g_probe;
#define VASR_LAUNCH(kern, grid, block, lds, st, ...)                                                       \
  do {                                                                                                     \
    if (::vasr::g_probe.start) {                                                                           \
      hipExtLaunchKernelGGL(kern, grid, block, lds, st, ::vasr::g_probe.start, ::vasr::g_probe.stop, 0,   \
                            __VA_ARGS__);                                                                  \
      ::vasr::g_probe.start = nullptr;                                                                     \
    } else {                                                                                               \
      hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                         \
    }                                                                                                      \
  } while (0)

// Opt-in of a kernel for more than 64 KB of dynamic LDS, once PER DEVICE (a function-local `static const` ran it
// once per process, for whichever device was current at the first launch).  `done` is one bit per device ordinal, owned by
// the call site (one per kernel instantiation): static std::atomic<uint64_t> done{0};
inline hipError_t dyn_lds_opt_in(const void* kernel, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_relaxed) & bit) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_relaxed);
  return e;
}

// One logical layer issued as several launches (full tiles + tail): the first launch carries the probe's start event, the
// last its stop event, so the bracket covers all of them (and the dispatch gaps between them).
#define VASR_LAUNCH_PART(first, last, kern, grid, block, lds, st, ...)                                     \
  do {                                                                                                     \
    if (::vasr::g_probe.start) {                                                                           \
      hipExtLaunchKernelGGL(kern, grid, block, lds, st, (first) ? ::vasr::g_probe.start : nullptr,         \
                            (last) ? ::vasr::g_probe.stop : nullptr, 0, __VA_ARGS__);                      \
      if (last) ::vasr::g_probe.start = nullptr;                                                           \
    } else {                                                                                               \
      hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                         \
    }                                                                                                      \
  } while (0)

// Activations live in HBM as [B][C][ld] fp32 with the time axis contiguous and
// ld = pad_frames(T): every row starts 512-byte aligned and every (32..128)-frame GEMM tile stays
// inside one utterance.  Columns t >= T are padding (never consumed unmasked).
// (Round 1 padded to 256 frames: T' = 516 -- a 10.3 s clip -- then cost 768 columns, +50 % depthwise tiles and GEMM
// epilogues over T' = 512; at 128 it costs 640.)
constexpr int kTimeTile = 128;
static inline int64_t pad_frames(int64_t t) { return (t + kTimeTile - 1) / kTimeTile * kTimeTile; }


// Per-utterance maxima for the fp16-split arithmetic (encoder_pw_split.hip kF16x2, encoder_dw_mfma.hip): every kernel
// that produces a tensor such a kernel will read publishes max |y| over each utterance's valid frames, one plain store
// per wavefront into its own slot: p[b * stride + slot], slot < n (n = what the producing launch used, set by its
// launcher; stride = capacity per utterance).  The consumer's wavefronts reduce the n slots themselves (vasr_device.h).
// No atomics, no memset: every slot below n is written by exactly one wavefront of the producing launch.
struct AmaxTab {
  unsigned int* p = nullptr;
  int stride = 0;
  int n = 0;
};

// ---- front end (frontend.hip) ----
struct FrontendTables {
  const float* window;   // [512] window already centred/zero padded to n_fft
  const float* tw256;    // [256][2] e^{-2 pi i m/256}
  const float* tw512;    // [257][2] e^{-2 pi i k/512}
  const float* mel_w;    // [64][kMelTaps] packed non-zero filter weights
  const int32_t* mel_lo; // [64] first fft bin of each filter
  int32_t guard_clamp;   // log(max(x, guard)) instead of log(x + guard)  (features.py:269-274)
};
constexpr int kMelTaps = 32;  // >= max non-zeros per Slaney filter at 64 mels / 512 fft (23)

// row_len: nullptr = the reference's batched semantics (every row is `samples` long, reflect padding at the padded
// end, quirk Q5); else each row ends at row_len[b] like an unbatched call
// wav: [batch][samples] float, or (pcm16) int16 PCM scaled by 2^-15 as it is read
void launch_stft_logmel(const FrontendTables& tb, const void* wav, bool pcm16, int batch, int64_t samples,
                        const int64_t* row_len, int hop, float preemph, float log_guard, float* mel, int64_t mel_ld, int frames,
                        hipStream_t st);
void launch_seq_len(const int64_t* len, int batch, int hop, int64_t* seq, hipStream_t st);
// normalize: 1 = per (utterance, mel bin) row, 0 = mask only; "all_features" = mask only, then launch_normalize_all
void launch_normalize_all(float* mel, int64_t mel_ld, const int64_t* seq, int batch, int n_mels, int frames, hipStream_t st);
void launch_normalize(float* mel, int64_t mel_ld, const int64_t* seq, int batch, int n_mels, int frames,
                      int normalize, hipStream_t st);

// ---- encoder (encoder_dw.hip, encoder_pw.hip) ----
struct LenStep { int32_t kernel, stride, dilation, pad; };
// lens_tab[s][b] = mask length seen by the s-th MaskedConv1d of the main chain; row n_steps = after
// the last one; enc_len[b] = the reference's float length (quirk Q3).  wav_len (optional): row n_steps + 1 = the
// output frames an unbatched call on the row would produce, capped at frames_cap (row-independent mode).
// fused path: per-feature normalisation of the log-mel rows + (as extra workgroups of the same launch) seq = ceil(len / hop)
// and the length chain
void launch_normalize_chain(float* mel, int64_t mel_ld, const int64_t* len, int hop, int batch, int n_mels, int frames,
                            int normalize, int64_t* seq, const LenStep* d_steps, int n_steps, int32_t* lens_tab,
                            float* enc_len, const int64_t* wav_len, int frames_cap, hipStream_t st);
void launch_len_chain(const int64_t* seq, int batch, const LenStep* d_steps, int n_steps, int32_t* lens_tab,
                      float* enc_len, hipStream_t st, const int64_t* wav_len = nullptr, int hop = 1, int frames_cap = 0);

void launch_repad(const float* src, int64_t src_ld, int rows, int frames, float* dst, int64_t dst_ld,
                  hipStream_t st);

// depthwise masked conv: y[b][c][t] = sum_k w[c][k] * xm[b][c][t*stride + k*dil - pad],
// xm = x where t < lens_in[b] else 0; y forced to 0 for t >= lens_out[b]; all columns < ldy written.
// amax_y: maxima table of the output (nullptr = not wanted); the call sets amax_y->n (and fails if that exceeds the stride)
int launch_depthwise(const float* x, int64_t ldx, int frames_in, const float* w, const int32_t* lens_in,
                     const int32_t* lens_out, int batch, int channels, int kernel, int stride, int dilation,
                     int pad, float* y, int64_t ldy, hipStream_t st, AmaxTab* amax_y = nullptr);
// slots launch_depthwise may use per utterance for that shape, whatever kernel it picks
int depthwise_amax_slots(int channels, int64_t ldy);

// Toeplitz / MFMA form (encoder_dw_mfma.hip): fp16-split arithmetic, needs the input's maxima table and the per-channel
// tap tables packed by pack_depthwise_taps_f16x2.  stride 1 only.  Returns 0, a hipError_t, or -1 when the shape has
// no instantiation (the caller then uses launch_depthwise).
int depthwise_mfma_table_size(int kernel, int dilation);   // dwords per channel; 0 = shape not covered
float pack_depthwise_taps_f16x2(const float* w, int kernel, int dilation, int tsz, unsigned int* table);   // returns 1 / scale
int launch_depthwise_mfma(const float* x, int64_t ldx, const unsigned int* taps, const float* tap_inv,
                          const int32_t* lens_in, const int32_t* lens_out, AmaxTab amax_x, int batch,
                          int channels, int kernel, int dilation, float* y, int64_t ldy, AmaxTab* amax_y,
                          hipStream_t st);

struct PwArgs {
  const float* wt;        // weights in MFMA fragment order (pack_pointwise_weights), M % 128 == 0, K % 32 == 0
  const float* x;         // [B][K][ldx]  (dual source: [B][K1][ldx])
  const int32_t* lens;    // [B] input mask (nullptr = unmasked)
  const float* x2;        // dual source: [B][K-K1][ldx2], always masked with lens2 (nullptr = single source)
  const int32_t* lens2;
  int32_t K1;
  int64_t ldx2;
  const float* scale;     // [M]
  const float* shift;     // [M]
  const float* res;       // [B][M][ldr] added before the ReLU (nullptr = none)
  float* y;               // [B][M][ldy]
  int32_t M, K, batch;
  int64_t ldx, ldy, ldr;
  int32_t frames;         // valid columns
  int32_t store_cols;     // columns < store_cols are stored: ldy for padded internal buffers, frames for ports
  int32_t m_store;        // rows < m_store are stored (decoder: V+1 of 128)
  int32_t relu;
  // [B] (or nullptr): every input source of utterance b is zero at columns >= zero_from[b] (the depthwise kernel and
  // the masks guarantee it), so a time tile that starts there skips its K loop: its outputs are relu(shift (+ res)).
  // Ragged batches only -- full-length clips never hit it.  Honoured by the split-bf16 kernel.
  const int32_t* zero_from;
  int32_t busy_cus;       // compute units held by a concurrent kernel of the caller's (vasr_set_busy_cus); tile choice only
  // kF16x2 only: maxima tables of x / x2 (inputs) and 1 / (weight scale) of the fp16 pack
  AmaxTab amax_x;
  AmaxTab amax_x2;
  float w_inv_scale;
  // any split arithmetic: publish max |y| over columns < lens_y[b] (nullptr: < frames) into amax_y (p == nullptr: off);
  // launch_pointwise_split reports the slots it used through its amax_n argument
  AmaxTab amax_y;
  const int32_t* lens_y;
};
void launch_pointwise(const PwArgs& a, hipStream_t st);
// host: [cout][cin] row-major -> fragment order [m_pad/32][cin/8][64][4] (zero rows past cout)
void pack_pointwise_weights(const float* w, int cout, int cin, int m_pad, float* out);

// split-operand variants (encoder_pw_split.hip): same PwArgs, a.wt points at the 16-bit fragment pack of the arithmetic
// arith: 0 = 3 x bf16 (six MFMA products per multiply), 1 = 2 x bf16 (three, reduced precision), 2 = 2 x fp16 scaled
// (three; needs amax_x / w_inv_scale and the fp16 pack).  Returns 0 or a hipError_t.
bool pointwise_split_supported(int M, int K, int K1);
double launch_mfma_sustained(int gemm_mode, int n_cu, int steps, float* sink, hipStream_t st);
int launch_pointwise_split(const PwArgs& a, int arith, hipStream_t st, int* amax_n = nullptr);
// the 2 x fp16 arithmetic on the small-batch latency kernel (encoder_pw_lat.hip), bit-identical results; -1: shape not covered
bool pointwise_latency_supported(int M, int K, int K1);
int launch_pointwise_latency(const PwArgs& a, hipStream_t st, int* amax_n);
int pointwise_amax_slots(int M, int64_t ld);   // slots per utterance the split kernel may use for that shape
void pack_pointwise_weights_bf16x3(const float* w, int cout, int cin, int m_pad, unsigned short* out);
float pack_pointwise_weights_f16x2(const float* w, int cout, int cin, int m_pad, unsigned short* out);   // returns 1 / scale
// maxima of a contiguous-per-utterance tensor x[b][rows][ld] over columns < lens[b] (or < frames); sets amax->n (<= 256)
void launch_amax(const float* x, int64_t ld, int rows, int frames, const int32_t* lens, int batch, AmaxTab* amax,
                 hipStream_t st);

// ---- fused depthwise + pointwise sub-block, 256 channels (encoder_fused.hip) ----
struct FusedLaunch {
  const float* x; int64_t ldx;                 // [B][256][ldx] depthwise input
  const int32_t* lens_in; const int32_t* lens_out;
  const float* taps; float dw_l1;              // pack_fused_taps
  AmaxTab amax_x;
  const void* wt; float w_inv_scale;           // f16x2 pack of the 1x1 conv (K = 256, or 512 with the residual folded in)
  const float* scale; const float* shift;
  float* y; int64_t ldy; int32_t frames, relu;
  AmaxTab amax_y; const int32_t* lens_y;
  const float* x2; int64_t ldx2; const int32_t* lens2; AmaxTab amax_x2;   // residual source (nullptr = none)
  int32_t batch, kernel;
  int32_t tile_cols;                           // frames per workgroup: 128 (0 = default) or 64
};
bool fused_dwpw_supported(int channels, int cout, int kernel, int stride, int dilation);
// Which form a 256-channel sub-block takes for `tiles128` tiles of 128 frames (batch x padded frames / 128) on `cus` free
// compute units: 128 or 64 = the fused kernel on tiles of that many frames, 0 = depthwise and GEMM as two kernels.
// One workgroup per CU and tile, so a launch is whole rounds of lock-stepped workgroups (measurements: vasr_api.cpp
// run_encoder): 128-frame tiles from 3/4 of a round up when the last round is >= 80 % full; otherwise
// 64-frame tiles while THOSE fit one round and occupy >= 3/8 of the chip; otherwise two kernels.
static inline int fused_tile_choice(int64_t tiles128, int cus) {
  if (tiles128 <= 0 || cus <= 0) return 0;
  const int64_t rounds = (tiles128 + cus - 1) / cus;
  if (tiles128 >= 3 * (int64_t)cus / 4 && (double)tiles128 >= 0.8 * (double)(rounds * cus)) return 128;
  if (2 * tiles128 >= 3 * (int64_t)cus / 8 && 2 * tiles128 <= cus) return 64;
  return 0;
}
int fused_dwpw_taps_per_pair(int kernel);
// host: [C][K] -> [C / 2][taps_per_pair][2]; returns the bound max_c sum_k |w[c][k]| (rounded up)
float pack_fused_taps(const float* w, int channels, int kernel, float* out);
int launch_fused_dwpw(const FusedLaunch& f, hipStream_t st, int* amax_n);   // 0, a hipError_t, or -1 (shape not covered)

// ---- CTC head / decode (decode.hip) ----
// logits [B][ldm rows][ld] (row v, column t) -> logp [B][T][V] (optional), pred [B][T] (optional)
void launch_logsoftmax_argmax(const float* logits, int64_t row_ld, int64_t batch_stride, int batch, int frames,
                              int num_classes, float* logp, int64_t* pred, hipStream_t st);
void launch_argmax(const float* logp, int batch, int64_t frames, int num_classes, int64_t* pred, hipStream_t st);
// wav_len != nullptr (row-independent mode): row b is collapsed over the frames an unbatched call on wav_len[b]
// samples would have produced -- 1 + wav_len / hop mel frames taken through the conv chain `steps` -- not over `frames`
void launch_ctc_collapse(const int64_t* pred, int batch, int64_t frames, int blank, int32_t* ids,
                         int32_t* id_len, hipStream_t st, const int64_t* wav_len = nullptr, int hop = 0,
                         const LenStep* steps = nullptr, int n_steps = 0);

// ---- audio ingest (audio.hip) ----
void launch_pcm16_to_f32(const short* in, int64_t n, float* out, hipStream_t st);
void launch_resample(const float* x, int64_t ld_in, const int64_t* len_in, int batch, const float* table, int nwin,
                     int num_table, double ratio, float* y, int64_t ld_out, int64_t* len_out, hipStream_t st);

// ---- beam search (beam_wave.hip, beam_group.hip) ----
struct BeamLm {  // device-resident hashed back-off n-gram model
  const void* vocab; int vcap;    // [vcap] 16-byte entries {u64 word hash | 1, i32 word id, u32 flags}, vcap a power of two
  const void* ngram; int ncap;    // [ncap] 16-byte entries {u64 n-gram key | 1, f32 log10 p, f32 log10 back-off}
  const void* trie; int tbuckets; // [tbuckets] 16-byte buckets of two u64 word-prefix keys; nullptr: no unigram list
  int order, bos, eos, unk;
  float alpha, beta, unk_offset;
};
constexpr int kBeamMax = 128;  // beams kept per utterance at most
// beam_wave.hip: one wavefront per utterance, `beam_wave_utts_per_workgroup` utterances per workgroup (= compute unit);
// returns 0 or a hipError_t; row_frames: [batch] frames searched per row, nullptr = all
int launch_beam_search_wave(const float* logp, int batch, int frames, int V1, int space_id, int beam_width,
                            float token_min_logp, float beam_prune_logp, const BeamLm* lm, unsigned int* bp,
                            int32_t* out_ids, int32_t* out_len, float* out_score, hipStream_t st,
                            const int32_t* row_frames = nullptr);
int beam_wave_utts_per_workgroup(int batch);
// beam_group.hip: the latency form -- an utterance on `beam_group_width(batch)` wavefronts of one compute unit (4 below 16
// utterances, else 1 = beam_wave.hip); same results bit for bit.  The entry point of vasr_beam_search_*.
int launch_beam_search_group(const float* logp, int batch, int frames, int V1, int space_id, int beam_width,
                             float token_min_logp, float beam_prune_logp, const BeamLm* lm, unsigned int* bp,
                             int32_t* out_ids, int32_t* out_len, float* out_score, hipStream_t st,
                             const int32_t* row_frames = nullptr);
int beam_group_width(int batch);
size_t beam_wave_lds_bytes();
unsigned long long beam_hash_step(unsigned long long h, unsigned long long v);
unsigned long long beam_hash_init();

}  // namespace vasr
