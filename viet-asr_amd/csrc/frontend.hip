// Mel front end for gfx950: pre-emphasis -> reflect-padded STFT (n_fft 512, hop 160,
// hann 320 centred) -> power -> Slaney mel (sparse) -> log(x + guard) -> per-feature CMVN + mask.
//
// Replaces FilterbankFeatures.forward (reference nemo/collections/asr/parts/features.py:245-301).
//
// One workgroup = 32 consecutive frames of one utterance (8 for small batches, see the kernel).  The 5472-sample
// window those frames cover is staged ONCE into LDS with coalesced loads (62.5 % frame overlap is served
// from LDS, not HBM); each of the 8 wavefronts then runs 4 (1) real 512-point FFTs as a
// 256-point complex radix-4 Stockham FFT (4 LDS-exchanged stages, 4 points per lane) and the
// even/odd split.  The 64 mel filters map one-per-lane; the [64 mel][32 frame] tile goes
// back to HBM as 128-byte row segments.
#include <atomic>

#include "vasr_internal.h"
#include "len_chain.h"

namespace vasr {

namespace {

constexpr int kWaves = 8;                       // 4 wavefronts x 8 frames left the LDS round trips of a frame exposed
constexpr int kThreads = 64 * kWaves;
constexpr int kNfft = 512;

struct cf { float re, im; };
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cf cmul_negi(cf a) { return {a.im, -a.re}; }

// The FFT scratch, power spectrum and tile columns of a frame belong to ONE wavefront, and the LDS pipeline executes a
// wavefront's accesses in issue order: a compiler-level fence is all the frame loop needs (it used six workgroup
// barriers per frame, i.e. the wavefronts of a workgroup kept waiting for each other 48 times per block).
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// grid (ceil(T / kFramesPerBlock), B), block kThreads.  kFramesPerBlock = 32 (four frames per wavefront, one after the other:
// the throughput form, 62.5 % of the staged samples shared) or 8 (one frame per wavefront: a small batch's few workgroups
// -- 21 for a 6.6 s clip -- are a chain of four dependent FFTs each, 16.5 us at batch 1; four times as many workgroups
// of a quarter of the length fill more of the idle chip).  A frame's arithmetic does not depend on the block it is in.
// S = float (the reference's float32 signal) or short: int16 PCM straight from the file, scaled by 2^-15 on the way into LDS
// (AudioSegment._convert_samples_to_float32, parts/segment.py:61-74: an exact conversion and an exact power-of-two product,
// i.e. the bits of converting first -- SURVEY section 8 f1: "int16 -> fp32 fused into the pre-emphasis load").
__device__ __forceinline__ float sample_f32(const float* x, int64_t n) { return x[n]; }
__device__ __forceinline__ float sample_f32(const short* x, int64_t n) { return (float)x[n] * 0x1p-15f; }

template <int kFramesPerBlock, typename S>
__global__ __launch_bounds__(kThreads, 4) void stft_logmel_kernel(FrontendTables tb, const S* __restrict__ wav,
                                                          int64_t samples, const int64_t* __restrict__ row_len,
                                                          int hop, float preemph,
                                                          float log_guard, float* __restrict__ mel,
                                                          int64_t mel_ld, int frames) {
  constexpr int kFramesPerWave = kFramesPerBlock / kWaves;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int seg_len = (kFramesPerBlock - 1) * hop + kNfft;
  float* seg = smem;                                   // [seg_len]
  cf* fftbuf = reinterpret_cast<cf*>(seg + ((seg_len + 3) & ~3));     // [kWaves][256], the stages run in place
  float* pbuf = reinterpret_cast<float*>(fftbuf + kWaves * 256);      // [kWaves][257 + kMelTaps] power spectrum
  float* tile = pbuf + kWaves * (260 + kMelTaps);      // [64][33]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * kFramesPerBlock;
  const S* x = wav + (int64_t)b * samples;
  // row-independent mode (vasr_set_row_independent): the row ends at its own length, as if it were alone in the batch
  const int64_t ns = row_len ? min(row_len[b], samples) : samples;

  // ---- window and twiddles: a lane needs the same few entries for every frame, so they live in registers ----
  const float2* win2 = reinterpret_cast<const float2*>(tb.window);
  const float2* t256 = reinterpret_cast<const float2*>(tb.tw256);
  const float2* t512 = reinterpret_cast<const float2*>(tb.tw512);
  float2 wn[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) wn[r] = win2[lane + 64 * r];
  cf tws[3][3];     // stage Ns = 4, 16, 64: w^step, w^2step, w^3step
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const int Ns = 4 << (2 * g), step = (lane & (Ns - 1)) * (64 / Ns);
#pragma unroll
    for (int m = 0; m < 3; ++m) { const float2 w = t256[(m + 1) * step]; tws[g][m] = {w.x, w.y}; }
  }
  cf twk[5];        // real-FFT split: e^{-2 pi i k/512} for k = lane + 64 i
#pragma unroll
  for (int i = 0; i < 5; ++i) { const float2 w = t512[min(lane + 64 * i, 256)]; twk[i] = {w.x, w.y}; }
  // ---- stage the pre-emphasised, reflect-padded segment ----
  const int64_t p0 = (int64_t)f0 * hop - kNfft / 2;  // sample index of seg[0] before reflection
  for (int i = tid; i < seg_len; i += kThreads) {
    int64_t n = p0 + i;
    if (n < 0) n = -n;                       // torch.stft(center=True, pad_mode="reflect"), features.py:181-188
    if (n >= ns) n = 2 * (ns - 1) - n;
    float v = 0.f;
    if (n >= 0 && n < ns) {
      v = sample_f32(x, n);
      // features.py:254-255  x[:,1:] - preemph * x[:,:-1]  (two roundings, no fma contraction)
      if (preemph >= 0.f && n > 0) v = __fsub_rn(v, __fmul_rn(preemph, sample_f32(x, n - 1)));
    }
    seg[i] = v;
  }
  // per-lane mel filter (lane = filter index)
  float mw[kMelTaps];
#pragma unroll
  for (int i = 0; i < kMelTaps; ++i) mw[i] = tb.mel_w[lane * kMelTaps + i];
  const int mlo = tb.mel_lo[lane];
  float* P = pbuf + wave * (260 + kMelTaps);
  for (int i = lane; i < 260 + kMelTaps; i += 64) P[i] = 0.f;
  __syncthreads();

  // One buffer per wavefront: a stage reads its four points per lane into registers before any lane writes (the
  // wavefront executes the reads as one instruction stream ahead of the writes), so the exchange can be in place.
  cf* buf = fftbuf + wave * 256;

  for (int jf = 0; jf < kFramesPerWave; ++jf) {
    const int fl = wave * kFramesPerWave + jf;  // frame within block
    const float* s = seg + fl * hop;
    // ---- stage Ns = 1 straight from the windowed samples: z[q] = s[2q] + i s[2q+1] ----
    {
      cf v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = lane + 64 * r;
        v[r] = {s[2 * q] * wn[r].x, s[2 * q + 1] * wn[r].y};
      }
      cf t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = cmul_negi(csub(v[1], v[3]));
      buf[4 * lane + 0] = cadd(t0, t2);
      buf[4 * lane + 1] = cadd(t1, t3);
      buf[4 * lane + 2] = csub(t0, t2);
      buf[4 * lane + 3] = csub(t1, t3);
    }
    wave_fence();
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int Ns = 4 << (2 * g);
      const int k = lane & (Ns - 1);
      cf v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = buf[lane + 64 * r];
      wave_fence();
      v[1] = cmul(v[1], tws[g][0]);
      v[2] = cmul(v[2], tws[g][1]);
      v[3] = cmul(v[3], tws[g][2]);
      cf t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = cmul_negi(csub(v[1], v[3]));
      const int j0 = (lane / Ns) * Ns * 4 + k;
      buf[j0] = cadd(t0, t2);
      buf[j0 + Ns] = cadd(t1, t3);
      buf[j0 + 2 * Ns] = csub(t0, t2);
      buf[j0 + 3 * Ns] = csub(t1, t3);
      wave_fence();
    }
    // ---- real-FFT split + power spectrum (features.py:260-263 pow(2).sum(-1)) ----
    const cf* Z = buf;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int k = lane + 64 * i;
      if (k > 256) break;
      cf zk = Z[k & 255];
      cf zc = Z[(256 - k) & 255];
      zc.im = -zc.im;
      cf e = {0.5f * (zk.re + zc.re), 0.5f * (zk.im + zc.im)};
      cf d = {zk.re - zc.re, zk.im - zc.im};
      cf o = {0.5f * d.im, -0.5f * d.re};  // -i/2 * d
      cf xk = cadd(e, cmul(twk[i], o));
      P[k] = xk.re * xk.re + xk.im * xk.im;
    }
    wave_fence();
    // ---- mel projection (features.py:266) over the filter's non-zero bins, log guard "add" (:269-271) ----
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < kMelTaps; ++i) acc = fmaf(mw[i], P[mlo + i], acc);
    // (uniform) log_zero_guard_type: "add" log(x + guard), "clamp" log(max(x, guard))
    tile[lane * (kFramesPerBlock + 1) + fl] = tb.guard_clamp ? logf(fmaxf(acc, log_guard)) : logf(acc + log_guard);
    wave_fence();
  }
  __syncthreads();
  // ---- [64][32] tile -> HBM rows ----
  for (int idx = tid; idx < 64 * kFramesPerBlock; idx += kThreads) {
    const int f = idx / kFramesPerBlock, j = idx % kFramesPerBlock;
    const int t = f0 + j;
    if (t < frames) mel[((int64_t)b * 64 + f) * mel_ld + t] = tile[f * (kFramesPerBlock + 1) + j];
  }
}

__global__ void seq_len_kernel(const int64_t* __restrict__ len, int batch, int hop, int64_t* __restrict__ seq) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  // features.py:238-239  ceil(len.float() / hop).long()
  if (b < batch) seq[b] = (int64_t)ceilf((float)len[b] / (float)hop);   // (seq_of below)
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// normalize_batch("per_feature") + length mask (features.py:17-30, 287-291).  One 256-thread workgroup per (utterance,
// mel bin) row; statistics in double like ATen's CPU Welford accumulator, the four wavefronts' partial sums combined in a
// fixed order.  (Rounds 1-3 gave a row to ONE wavefront: 16 serial trips over a 10 s row, twice -- 8-11 us at batch 1.)
// seq: frames per utterance, or nullptr: computed here from the sample counts `len` (features.py:238-239).
// Workgroups >= rows (normalize_chain_kernel launches ceil(batch / 256) of them) run the encoder's length chain instead
// (len_chain.h) and publish seq: the chain has no business on the critical path of a batch-1 call, where it was a launch
// of its own (5-10 us of dependent scalar arithmetic) in front of the encoder.
__device__ __forceinline__ int64_t seq_of(int64_t len, int hop) { return (int64_t)ceilf((float)len / (float)hop); }

// (kRowsPerWg rows per workgroup, four wavefronts each: the workgroup is alone on its compute unit -- launch_normalize below --, so
// it brings its own occupancy.  `red`: the row group's eight doubles.)
constexpr int kRowsPerWg = 4;
__device__ __forceinline__ void normalize_row(float* __restrict__ x, int n, int frames, int normalize, double* red) {
  const int tid = threadIdx.x & 255, lane = tid & 63, wv = tid >> 6;
  float mean = 0.f, stdv = 1.f;
  if (normalize) {
    double s = 0.0;
    for (int t = tid; t < n; t += 256) s += (double)x[t];
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    const double mu = (((red[0] + red[1]) + red[2]) + red[3]) / (double)n;  // n == 0 -> NaN, like torch .mean() of an empty slice
    double q = 0.0;
    for (int t = tid; t < n; t += 256) { const double d = (double)x[t] - mu; q += d * d; }
    q = wave_sum(q);
    if (lane == 0) red[4 + wv] = q;
    __syncthreads();
    mean = (float)mu;
    stdv = (float)sqrt((((red[4] + red[5]) + red[6]) + red[7]) / (double)(n - 1));  // unbiased; n == 1 -> NaN like torch .std()
    stdv = stdv + 1e-5f;                        // features.py:24-25 CONSTANT
  }
  for (int t = tid; t < frames; t += 256) {
    float v = 0.f;  // pad_value
    if (t < n) v = normalize ? (x[t] - mean) / stdv : x[t];
    x[t] = v;
  }
}

__global__ __launch_bounds__(256 * kRowsPerWg) void normalize_kernel(float* __restrict__ mel, int64_t ld,
                                                                     const int64_t* __restrict__ seq, int rows, int n_mels,
                                                                     int frames, int normalize) {
  __shared__ double red[kRowsPerWg][8];
  const int grp = threadIdx.x >> 8;
  const int row = min((int)blockIdx.x * kRowsPerWg + grp, rows - 1);   // (rows = batch x 64: a multiple of four; clamped, not skipped: barriers)
  const int64_t n64 = seq[row / n_mels];
  normalize_row(mel + (int64_t)row * ld, (int)(n64 < 0 ? 0 : (n64 > frames ? frames : n64)), frames, normalize, red[grp]);
}

// normalize_batch("all_features") (features.py:31-39): ONE mean and one unbiased std per utterance over every mel bin and
// every valid frame, x[b, :, :seq[b]].mean() / .std() + 1e-5.  One workgroup per utterance, after the row kernel has masked the
// padded frames (launched with normalize = 0); statistics in double as above.  No shipped configuration uses it.
__global__ __launch_bounds__(256) void normalize_all_kernel(float* __restrict__ mel, int64_t ld, const int64_t* __restrict__ seq,
                                                            int n_mels, int frames) {
  __shared__ double red[8];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* x = mel + (int64_t)blockIdx.x * n_mels * ld;
  const int64_t n64 = seq[blockIdx.x];
  const int n = (int)(n64 < 0 ? 0 : (n64 > frames ? frames : n64));
  const double cnt = (double)n * (double)n_mels;
  double s = 0.0;
  for (int f = 0; f < n_mels; ++f)
    for (int t = tid; t < n; t += 256) s += (double)x[(int64_t)f * ld + t];
  s = wave_sum(s);
  if (lane == 0) red[wv] = s;
  __syncthreads();
  const double mu = (((red[0] + red[1]) + red[2]) + red[3]) / cnt;       // no valid frame -> NaN, like torch .mean() of an empty slice
  double q = 0.0;
  for (int f = 0; f < n_mels; ++f)
    for (int t = tid; t < n; t += 256) { const double d = (double)x[(int64_t)f * ld + t] - mu; q += d * d; }
  q = wave_sum(q);
  if (lane == 0) red[4 + wv] = q;
  __syncthreads();
  const float mean = (float)mu;
  const float stdv = (float)sqrt((((red[4] + red[5]) + red[6]) + red[7]) / (cnt - 1.0)) + 1e-5f;   // features.py:37-38 CONSTANT
  for (int f = 0; f < n_mels; ++f)
    for (int t = tid; t < n; t += 256) x[(int64_t)f * ld + t] = (x[(int64_t)f * ld + t] - mean) / stdv;
}

__global__ __launch_bounds__(256 * kRowsPerWg) void normalize_chain_kernel(float* __restrict__ mel, int64_t ld,
                                                              const int64_t* __restrict__ len, int hop, int batch, int n_mels,
                                                              int frames, int normalize, int64_t* __restrict__ seq,
                                                              const LenStep* __restrict__ steps, int n_steps,
                                                              int32_t* __restrict__ lens_tab, float* __restrict__ enc_len,
                                                              const int64_t* __restrict__ wav_len, int frames_cap) {
  __shared__ double red[kRowsPerWg][8];
  const int rows = batch * n_mels, row_wgs = (rows + kRowsPerWg - 1) / kRowsPerWg;
  if ((int)blockIdx.x < row_wgs) {
    const int grp = threadIdx.x >> 8;
    const int row = min((int)blockIdx.x * kRowsPerWg + grp, rows - 1);
    const int64_t n64 = seq_of(len[row / n_mels], hop);
    normalize_row(mel + (int64_t)row * ld, (int)(n64 < 0 ? 0 : (n64 > frames ? frames : n64)), frames, normalize, red[grp]);
    return;
  }
  const int b = ((int)blockIdx.x - row_wgs) * 256 * kRowsPerWg + (int)threadIdx.x;
  const int64_t l0 = b < batch ? seq_of(len[b], hop) : 0;
  if (b < batch) seq[b] = l0;
  len_chain_body(b, l0, batch, steps, n_steps, lens_tab, enc_len, wav_len, hop, frames_cap);
}

}  // namespace

void launch_stft_logmel(const FrontendTables& tb, const void* wav, bool pcm16, int batch, int64_t samples,
                        const int64_t* row_len, int hop, float preemph, float log_guard, float* mel, int64_t mel_ld, int frames,
                        hipStream_t st) {
  auto go = [&](auto kern, auto* wav, int fpb) {
    const int seg_len = (fpb - 1) * hop + kNfft;
    const size_t lds = (size_t)((seg_len + 3) & ~3) * 4 + kWaves * 256 * 8 + kWaves * (260 + kMelTaps) * 4 + 64 * (fpb + 1) * 4;
    dim3 grid((frames + fpb - 1) / fpb, batch);
    hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, st, tb, wav, samples, row_len, hop, preemph, log_guard, mel, mel_ld,
                       frames);
  };
  // fewer 32-frame workgroups than half the chip's compute units: the one-frame-per-wavefront form
  const bool small = (int64_t)((frames + 31) / 32) * batch < 128;
  if (pcm16) {
    if (small) go(stft_logmel_kernel<8, short>, static_cast<const short*>(wav), 8);
    else go(stft_logmel_kernel<32, short>, static_cast<const short*>(wav), 32);
  } else {
    if (small) go(stft_logmel_kernel<8, float>, static_cast<const float*>(wav), 8);
    else go(stft_logmel_kernel<32, float>, static_cast<const float*>(wav), 32);
  }
}

void launch_seq_len(const int64_t* len, int batch, int hop, int64_t* seq, hipStream_t st) {
  hipLaunchKernelGGL(seq_len_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, len, batch, hop, seq);
}

// The normalisation kernels keep their statistics in double (ATen's accumulation type: the reference's bits), and on MI355X a
// wavefront's FP64 arithmetic returned WRONG VALUES while a wavefront of a 16-bit MFMA kernel from ANOTHER stream shared its compute
// unit (round 6, tests/devtools/stress_attack.py: the per-row mean / std came out wrong in 1.5-8 % of the calls next to torch's own
// fp16 bmm; profiles/r06_concurrency.txt).  These kernels are small (four wavefronts, 64 bytes of LDS), i.e. exactly what the
// dispatcher slots in beside somebody else's workgroups.  They therefore ask for (almost) a whole compute unit's LDS, which keeps
// every workgroup that needs LDS of its own -- a matrix kernel stages its operands there -- off the unit while one of them runs:
// four rows per 1 024-thread workgroup keep the unit as full as four 256-thread workgroups did.
constexpr int kAloneLds = 152 * 1024;   // + <= 4 160 bytes static; the unit has 160 KB

void launch_normalize_all(float* mel, int64_t mel_ld, const int64_t* seq, int batch, int n_mels, int frames, hipStream_t st) {
  static std::atomic<uint64_t> opted{0};
  (void)dyn_lds_opt_in(reinterpret_cast<const void*>(normalize_all_kernel), kAloneLds, opted);
  hipLaunchKernelGGL(normalize_all_kernel, dim3(batch), dim3(256), kAloneLds, st, mel, mel_ld, seq, n_mels, frames);
}

void launch_normalize(float* mel, int64_t mel_ld, const int64_t* seq, int batch, int n_mels, int frames,
                      int normalize, hipStream_t st) {
  const int rows = batch * n_mels;
  static std::atomic<uint64_t> opted{0};
  (void)dyn_lds_opt_in(reinterpret_cast<const void*>(normalize_kernel), kAloneLds, opted);
  hipLaunchKernelGGL(normalize_kernel, dim3((rows + kRowsPerWg - 1) / kRowsPerWg), dim3(256 * kRowsPerWg), normalize ? kAloneLds : 0, st, mel, mel_ld,
                     seq, rows, n_mels, frames, normalize);
}

void launch_normalize_chain(float* mel, int64_t mel_ld, const int64_t* len, int hop, int batch, int n_mels, int frames,
                            int normalize, int64_t* seq, const LenStep* d_steps, int n_steps, int32_t* lens_tab,
                            float* enc_len, const int64_t* wav_len, int frames_cap, hipStream_t st) {
  const int rows = batch * n_mels;
  static std::atomic<uint64_t> opted{0};
  (void)dyn_lds_opt_in(reinterpret_cast<const void*>(normalize_chain_kernel), kAloneLds, opted);
  const int wg = 256 * kRowsPerWg;
  hipLaunchKernelGGL(normalize_chain_kernel, dim3((rows + kRowsPerWg - 1) / kRowsPerWg + (batch + wg - 1) / wg), dim3(wg), normalize ? kAloneLds : 0, st, mel, mel_ld, len, hop, batch,
                     n_mels, frames, normalize, seq, d_steps, n_steps, lens_tab, enc_len, wav_len, frames_cap);
}

}  // namespace vasr
