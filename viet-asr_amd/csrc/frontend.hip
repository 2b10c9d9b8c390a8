// Mel front end for gfx950: pre-emphasis -> reflect-padded STFT (n_fft 512, hop 160,
// hann 320 centred) -> power -> Slaney mel (sparse) -> log(x + guard) -> per-feature CMVN + mask.
//
// Replaces FilterbankFeatures.forward (reference nemo/collections/asr/parts/features.py:245-301).
//
// One workgroup = 32 consecutive frames of one utterance.  The 5472-sample window those
// frames cover is staged ONCE into LDS with coalesced loads (62.5 % frame overlap is served
// from LDS, not HBM); each of the 4 wavefronts then runs 8 real 512-point FFTs as a
// 256-point complex radix-4 Stockham FFT (4 LDS-exchanged stages, 4 points per lane) and the
// even/odd split.  The 64 mel filters map one-per-lane; the [64 mel][32 frame] tile goes
// back to HBM as 128-byte row segments.
#include "vasr_internal.h"

namespace vasr {

namespace {

constexpr int kFramesPerBlock = 32;
constexpr int kFramesPerWave = 8;
constexpr int kNfft = 512;

struct cf { float re, im; };
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cf cmul_negi(cf a) { return {a.im, -a.re}; }

// The FFT scratch, power spectrum and tile columns of a frame belong to ONE wavefront, and the LDS pipeline executes a
// wavefront's accesses in issue order: a compiler-level fence is all the frame loop needs (it used six workgroup
// barriers per frame, i.e. the four wavefronts kept waiting for each other 48 times per block).
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// grid (ceil(T/32), B), block 256
__global__ __launch_bounds__(256) void stft_logmel_kernel(FrontendTables tb, const float* __restrict__ wav,
                                                          int64_t samples, int hop, float preemph,
                                                          float log_guard, float* __restrict__ mel,
                                                          int64_t mel_ld, int frames) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int seg_len = (kFramesPerBlock - 1) * hop + kNfft;
  float* seg = smem;                                   // [seg_len]
  float* win = seg + ((seg_len + 3) & ~3);             // [512]
  cf* tw256 = reinterpret_cast<cf*>(win + kNfft);      // [256]
  cf* tw512 = tw256 + 256;                             // [257] (+1 pad)
  cf* fftbuf = tw512 + 258;                            // [4 waves][2][256]
  float* pbuf = reinterpret_cast<float*>(fftbuf + 4 * 2 * 256);  // [4 waves][257 + kMelTaps] power spectrum
  float* tile = pbuf + 4 * (260 + kMelTaps);           // [64][33]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * kFramesPerBlock;
  const float* x = wav + (int64_t)b * samples;

  // ---- stage tables + the pre-emphasised, reflect-padded segment ----
  for (int i = tid; i < kNfft; i += 256) win[i] = tb.window[i];
  for (int i = tid; i < 256; i += 256) tw256[i] = {tb.tw256[2 * i], tb.tw256[2 * i + 1]};
  for (int i = tid; i < 257; i += 256) tw512[i] = {tb.tw512[2 * i], tb.tw512[2 * i + 1]};
  const int64_t p0 = (int64_t)f0 * hop - kNfft / 2;  // sample index of seg[0] before reflection
  for (int i = tid; i < seg_len; i += 256) {
    int64_t n = p0 + i;
    if (n < 0) n = -n;                       // torch.stft(center=True, pad_mode="reflect"), features.py:181-188
    if (n >= samples) n = 2 * (samples - 1) - n;
    float v = 0.f;
    if (n >= 0 && n < samples) {
      v = x[n];
      // features.py:254-255  x[:,1:] - preemph * x[:,:-1]  (two roundings, no fma contraction)
      if (preemph >= 0.f && n > 0) v = __fsub_rn(v, __fmul_rn(preemph, x[n - 1]));
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

  cf* bufA = fftbuf + wave * 512;
  cf* bufB = bufA + 256;

  for (int jf = 0; jf < kFramesPerWave; ++jf) {
    const int fl = wave * kFramesPerWave + jf;  // frame within block
    const float* s = seg + fl * hop;
    // ---- stage Ns = 1 straight from the windowed samples: z[q] = s[2q] + i s[2q+1] ----
    {
      cf v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = lane + 64 * r;
        v[r] = {s[2 * q] * win[2 * q], s[2 * q + 1] * win[2 * q + 1]};
      }
      cf t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = cmul_negi(csub(v[1], v[3]));
      bufA[4 * lane + 0] = cadd(t0, t2);
      bufA[4 * lane + 1] = cadd(t1, t3);
      bufA[4 * lane + 2] = csub(t0, t2);
      bufA[4 * lane + 3] = csub(t1, t3);
    }
    wave_fence();
    cf* in = bufA;
    cf* out = bufB;
#pragma unroll
    for (int Ns = 4; Ns < 256; Ns *= 4) {
      const int k = lane & (Ns - 1);
      cf v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = in[lane + 64 * r];
      const int step = k * (64 / Ns);
      v[1] = cmul(v[1], tw256[step]);
      v[2] = cmul(v[2], tw256[2 * step]);
      v[3] = cmul(v[3], tw256[3 * step]);
      cf t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = cmul_negi(csub(v[1], v[3]));
      const int j0 = (lane / Ns) * Ns * 4 + k;
      out[j0] = cadd(t0, t2);
      out[j0 + Ns] = cadd(t1, t3);
      out[j0 + 2 * Ns] = csub(t0, t2);
      out[j0 + 3 * Ns] = csub(t1, t3);
      wave_fence();
      cf* t = in; in = out; out = t;
    }
    // ---- real-FFT split + power spectrum (features.py:260-263 pow(2).sum(-1)) ----
    const cf* Z = in;
    for (int k = lane; k <= 256; k += 64) {
      cf zk = Z[k & 255];
      cf zc = Z[(256 - k) & 255];
      zc.im = -zc.im;
      cf e = {0.5f * (zk.re + zc.re), 0.5f * (zk.im + zc.im)};
      cf d = {zk.re - zc.re, zk.im - zc.im};
      cf o = {0.5f * d.im, -0.5f * d.re};  // -i/2 * d
      cf xk = cadd(e, cmul(tw512[k], o));
      P[k] = xk.re * xk.re + xk.im * xk.im;
    }
    wave_fence();
    // ---- mel projection (features.py:266) over the filter's non-zero bins, log guard "add" (:269-271) ----
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < kMelTaps; ++i) acc = fmaf(mw[i], P[mlo + i], acc);
    tile[lane * (kFramesPerBlock + 1) + fl] = logf(acc + log_guard);
    wave_fence();
  }
  __syncthreads();
  // ---- [64][32] tile -> HBM rows ----
  for (int idx = tid; idx < 64 * kFramesPerBlock; idx += 256) {
    const int f = idx / kFramesPerBlock, j = idx % kFramesPerBlock;
    const int t = f0 + j;
    if (t < frames) mel[((int64_t)b * 64 + f) * mel_ld + t] = tile[f * (kFramesPerBlock + 1) + j];
  }
}

__global__ void seq_len_kernel(const int64_t* __restrict__ len, int batch, int hop, int64_t* __restrict__ seq) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  // features.py:238-239  ceil(len.float() / hop).long()
  if (b < batch) seq[b] = (int64_t)ceilf((float)len[b] / (float)hop);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// normalize_batch("per_feature") + length mask (features.py:17-30, 287-291).  One wavefront per
// (utterance, mel bin) row; statistics in double like ATen's CPU Welford accumulator.
__global__ __launch_bounds__(256) void normalize_kernel(float* __restrict__ mel, int64_t ld,
                                                        const int64_t* __restrict__ seq, int rows, int n_mels,
                                                        int frames, int normalize) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = row / n_mels;
  float* x = mel + (int64_t)row * ld;
  int64_t n64 = seq[b];
  const int n = (int)(n64 < 0 ? 0 : (n64 > frames ? frames : n64));
  float mean = 0.f, stdv = 1.f;
  if (normalize) {
    double s = 0.0;
    for (int t = lane; t < n; t += 64) s += (double)x[t];
    s = wave_sum(s);
    const double mu = s / (double)n;  // n == 0 -> NaN, like torch .mean() of an empty slice
    double q = 0.0;
    for (int t = lane; t < n; t += 64) { const double d = (double)x[t] - mu; q += d * d; }
    q = wave_sum(q);
    mean = (float)mu;
    stdv = (float)sqrt(q / (double)(n - 1));  // unbiased; n == 1 -> NaN like torch .std()
    stdv = stdv + 1e-5f;                        // features.py:24-25 CONSTANT
  }
  for (int t = lane; t < frames; t += 64) {
    float v = 0.f;  // pad_value
    if (t < n) v = normalize ? (x[t] - mean) / stdv : x[t];
    x[t] = v;
  }
}

}  // namespace

void launch_stft_logmel(const FrontendTables& tb, const float* wav, int batch, int64_t samples, int hop,
                        float preemph, float log_guard, float* mel, int64_t mel_ld, int frames,
                        hipStream_t st) {
  const int seg_len = (kFramesPerBlock - 1) * hop + kNfft;
  size_t lds = (size_t)((seg_len + 3) & ~3) * 4 + kNfft * 4 + 256 * 8 + 258 * 8 + 4 * 2 * 256 * 8 +
               4 * (260 + kMelTaps) * 4 + 64 * (kFramesPerBlock + 1) * 4;
  dim3 grid((frames + kFramesPerBlock - 1) / kFramesPerBlock, batch);
  hipLaunchKernelGGL(stft_logmel_kernel, grid, dim3(256), lds, st, tb, wav, samples, hop, preemph, log_guard,
                     mel, mel_ld, frames);
}

void launch_seq_len(const int64_t* len, int batch, int hop, int64_t* seq, hipStream_t st) {
  hipLaunchKernelGGL(seq_len_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, len, batch, hop, seq);
}

void launch_normalize(float* mel, int64_t mel_ld, const int64_t* seq, int batch, int n_mels, int frames,
                      int normalize, hipStream_t st) {
  const int rows = batch * n_mels;
  hipLaunchKernelGGL(normalize_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, mel, mel_ld, seq, rows, n_mels,
                     frames, normalize);
}

}  // namespace vasr
