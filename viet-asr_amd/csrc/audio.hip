// Audio ingest on the device: 16-bit PCM -> float32, and band-limited sample-rate conversion.
//
// Callers of the path (reference infer.py:200 / app.py:66,82 `librosa.load(f, sr=16000)`; AudioSegment
// nemo/collections/asr/parts/segment.py:19-32, 61-74) decode on the host, scale integers by 2^-(bits-1) and
// resample 8 kHz call-centre audio to 16 kHz with librosa's default `kaiser_best` (resampy; third-party, absent:
// parity unpinned).  Doing both on the device halves the host->device bytes (int16 instead of fp32) and keeps
// the 8 -> 16 kHz conversion off the host cores.
//
// The resampler is the interpolated windowed-sinc scheme resampy publishes: a one-sided Kaiser-windowed sinc
// table with 2^9 phases per zero crossing, linear interpolation between table entries, left and right wings
// accumulated per output sample.
//   * resample_phase_kernel: ratios p/q with few distinct output phases (8 -> 16 kHz: p = 2; 48 -> 16 kHz: p = 1).
//     The interpolated filter depends only on t mod p, so a workgroup builds the p wing pairs once in LDS, stages the
//     input span of its 1024 outputs in LDS (zero outside [0, len): that replaces the per-sample bound checks) and
//     every output is two LDS dot products.  Same arithmetic as the generic kernel (fp32 weight, fp32 product, fp64
//     accumulation), 8x faster: the generic form chases two dependent global loads per tap.
//   * resample_kernel: any ratio, one thread per output sample, table gathers from L2.
#include <cstdlib>

#include "vasr_internal.h"

namespace vasr {

namespace {

// out[b][i] = in[b][i] / 32768   (segment.py:61-74: int -> float32 * 1/2^(bits-1))
__global__ __launch_bounds__(256) void pcm16_to_f32_kernel(const short* __restrict__ in, int64_t n,
                                                           float* __restrict__ out) {
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    const short4 v = *reinterpret_cast<const short4*>(in + i4);
    float4 o = make_float4(v.x * (1.0f / 32768.f), v.y * (1.0f / 32768.f), v.z * (1.0f / 32768.f), v.w * (1.0f / 32768.f));
    *reinterpret_cast<float4*>(out + i4) = o;
  } else {
    for (int64_t i = i4; i < n; ++i) out[i] = in[i] * (1.0f / 32768.f);
  }
}

// grid (ceil(n_out/256), B)
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, int64_t ld_in,
                                                       const int64_t* __restrict__ len_in,
                                                       const float2* __restrict__ table, int nwin, int num_table,
                                                       double ratio, float* __restrict__ y, int64_t ld_out,
                                                       int64_t* __restrict__ len_out) {
  const int b = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_orig = len_in[b];
  const int64_t n_out = (int64_t)((double)n_orig * ratio);   // resampy: int(n_orig * ratio) samples are COMPUTED ...
  if (t == 0) len_out[b] = (int64_t)ceil((double)n_orig * ratio);   // ... librosa.resample(fix=True) pads them to ceil(n_orig * ratio)
  if (t >= ld_out) return;
  float* yo = y + (int64_t)b * ld_out;
  if (t >= n_out) { yo[t] = 0.f; return; }
  const float* xi = x + (int64_t)b * ld_in;
  const double scale = ratio < 1.0 ? ratio : 1.0;
  const int index_step = (int)(scale * num_table);
  const double time_register = (double)t / ratio;
  const int64_t n = (int64_t)time_register;
  double acc = 0.0;
  {  // left wing
    const double frac = scale * (time_register - (double)n);
    const double index_frac = frac * num_table;
    const int offset = (int)index_frac;
    const float eta = (float)(index_frac - offset);
    int64_t i_max = (nwin - offset) / index_step;
    if (i_max > n + 1) i_max = n + 1;
    for (int64_t i = 0; i < i_max; ++i) {
      const float2 w = table[offset + i * index_step];
      acc += (double)((w.x + eta * w.y) * xi[n - i]);
    }
  }
  {  // right wing
    const double frac = scale - scale * (time_register - (double)n);
    const double index_frac = frac * num_table;
    const int offset = (int)index_frac;
    const float eta = (float)(index_frac - offset);
    int64_t k_max = (nwin - offset) / index_step;
    if (k_max > n_orig - n - 1) k_max = n_orig - n - 1;
    for (int64_t k = 0; k < k_max; ++k) {
      const float2 w = table[offset + k * index_step];
      acc += (double)((w.x + eta * w.y) * xi[n + k + 1]);
    }
  }
  yo[t] = (float)acc;
}

constexpr int kPhaseTile = 1024;   // outputs per workgroup

// grid (ceil(ld_out/1024), B), block 256.  ratio = p/q in lowest terms; wmax >= taps of either wing.
// Dynamic LDS: float wl[p][wmax], wr[p][wmax], xs[span] (+ int il[p], ir[p]).
__global__ __launch_bounds__(256) void resample_phase_kernel(const float* __restrict__ x, int64_t ld_in,
                                                             const int64_t* __restrict__ len_in,
                                                             const float2* __restrict__ table, int nwin, int num_table,
                                                             double ratio, int p, int wmax, int span,
                                                             float* __restrict__ y, int64_t ld_out,
                                                             int64_t* __restrict__ len_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wl = lds;                       // [p][wmax] left-wing weights (x[n - i])
  float* wr = wl + p * wmax;             // [p][wmax] right-wing weights (x[n + 1 + k])
  float* xs = wr + p * wmax;             // [span]
  int* cnt = reinterpret_cast<int*>(xs + span);   // [2 p] taps per wing and phase

  const int b = blockIdx.y, tid = threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.x * kPhaseTile;
  const int64_t n_orig = len_in[b];
  const int64_t n_out = (int64_t)((double)n_orig * ratio);   // resampy: int(n_orig * ratio) samples are COMPUTED ...
  if (blockIdx.x == 0 && tid == 0) len_out[b] = (int64_t)ceil((double)n_orig * ratio);   // ... librosa pads to ceil(n_orig * ratio)
  const float* xi = x + (int64_t)b * ld_in;
  const double scale = ratio < 1.0 ? ratio : 1.0;
  const int index_step = (int)(scale * num_table);

  // ---- the p interpolated filters: phase of output t is t mod p (same expressions as the generic kernel) ----
  for (int ph = 0; ph < p; ++ph) {
    const int64_t t = t0 + ((ph - t0 % p) + p) % p;          // first output of this tile with that phase
    const double time_register = (double)t / ratio;
    const int64_t n = (int64_t)time_register;
    const double fl = scale * (time_register - (double)n), fr = scale - fl;
    const double ifl = fl * num_table, ifr = fr * num_table;
    const int ol = (int)ifl, orr = (int)ifr;
    const float el = (float)(ifl - ol), er = (float)(ifr - orr);
    const int il = (nwin - ol) / index_step, ir = (nwin - orr) / index_step;
    if (tid == 0) { cnt[2 * ph] = il; cnt[2 * ph + 1] = ir; }
    for (int i = tid; i < wmax; i += 256) {
      float a = 0.f, c = 0.f;
      if (i < il) { const float2 w = table[ol + i * index_step]; a = w.x + el * w.y; }
      if (i < ir) { const float2 w = table[orr + i * index_step]; c = w.x + er * w.y; }
      wl[ph * wmax + i] = a;
      wr[ph * wmax + i] = c;
    }
  }
  // ---- input span of the tile; samples outside [0, n_orig) are zero, which is what the wing bounds amount to ----
  const int64_t n_lo = (int64_t)((double)t0 / ratio) - wmax;
  for (int j = tid; j < span; j += 256) {
    const int64_t g = n_lo + j;
    xs[j] = (g >= 0 && g < n_orig) ? xi[g] : 0.f;
  }
  __syncthreads();

  float* yo = y + (int64_t)b * ld_out;
  for (int r = 0; r < kPhaseTile / 256; ++r) {
    const int64_t t = t0 + tid + 256 * r;
    if (t >= ld_out) break;
    float out = 0.f;
    if (t < n_out) {
      const int ph = (int)(t % p);
      const int64_t n = (int64_t)((double)t / ratio);
      const int base = (int)(n - n_lo);
      const float* a = wl + ph * wmax;
      const float* c = wr + ph * wmax;
      const int il = cnt[2 * ph], ir = cnt[2 * ph + 1];
      double acc = 0.0;
      for (int i = 0; i < il; ++i) acc += (double)(a[i] * xs[base - i]);
      for (int k = 0; k < ir; ++k) acc += (double)(c[k] * xs[base + 1 + k]);
      out = (float)acc;
    }
    yo[t] = out;
  }
}

// Integer up-sampling (ratio = p / 1: 8 -> 16 kHz is p = 2), the BASELINE configs[4] case.  The phase kernel above spends
// ~6 instructions per tap and output (two LDS reads, fp32 multiply, conversion, fp64 add, loop) on 128 taps: 4.4 ms for
// 512 x 30 s -- 0.05 of its HBM roofline.  Here both wings of a phase are ONE filter h[ph][m] over x[n + m]
// (m = -(il - 1) .. ir), stored in LDS as fp64, and a lane produces R = 8 consecutive same-phase outputs, whose inputs
// are consecutive samples (q = 1): a sliding window of R + 7 fp64 samples in registers, per block of 8 taps 8 new
// samples (one LDS read + one conversion each), 8 weights (LDS broadcast reads) and 64 v_fma_f64 -- 1.4 instructions per
// tap and output.  Arithmetic: the same fp32 interpolated weights, fp64 accumulation of the EXACT products (the phase
// kernel rounds every product to fp32 first; the difference is 2^-24 of a term, the oracle comparison is at 2e-6).
constexpr int kUpR = 8, kUpTB = 8;
// grid (ceil(ld_out / tile), B), block 256; tile = 8 p floor(256 / p) outputs.  Dynamic LDS: double h[p][mpad], float xs[span]
__global__ __launch_bounds__(256) void resample_up_kernel(const float* __restrict__ x, int64_t ld_in,
                                                          const int64_t* __restrict__ len_in,
                                                          const float2* __restrict__ table, int nwin, int num_table, int p,
                                                          int wmax, int mpad, int span, int tile,
                                                          float* __restrict__ y, int64_t ld_out,
                                                          int64_t* __restrict__ len_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double* h = reinterpret_cast<double*>(lds_raw);                 // [p][mpad]: taps m = u - (wmax - 1), u = 0 .. mpad - 1
  float* xs = reinterpret_cast<float*>(h + (size_t)p * mpad);     // [span]
  const int b = blockIdx.y, tid = threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.x * tile;                  // a multiple of p: output t0 + ph has phase ph
  const int64_t n_orig = len_in[b];
  const int64_t n_out = n_orig * p;                               // int(n_orig * ratio), ratio = p exactly
  if (blockIdx.x == 0 && tid == 0) len_out[b] = n_out;
  const float* xi = x + (int64_t)b * ld_in;
  const int index_step = num_table;                               // scale = 1 for up-sampling
  // ---- the p filters (same expressions as the generic kernel at t = ph: time_register = ph / p, n = 0) ----
  for (int i = tid; i < p * mpad; i += 256) {
    const int ph = i / mpad, u = i - ph * mpad, m = u - (wmax - 1);
    const double fl = (double)ph / (double)p, fr = 1.0 - fl;
    const double ifl = fl * num_table, ifr = fr * num_table;
    const int ol = (int)ifl, orr = (int)ifr;
    const float el = (float)(ifl - ol), er = (float)(ifr - orr);
    const int il = (nwin - ol) / index_step, ir = (nwin - orr) / index_step;
    float w = 0.f;
    if (m <= 0 && -m < il) { const float2 tw = table[ol + (-m) * index_step]; w = tw.x + el * tw.y; }
    else if (m >= 1 && m - 1 < ir) { const float2 tw = table[orr + (m - 1) * index_step]; w = tw.x + er * tw.y; }
    h[i] = (double)w;
  }
  // ---- input span of the tile: x[n_lo + j], zero outside [0, n_orig) (what the wings' bounds amount to) ----
  const int64_t n_lo = t0 / p - (wmax - 1);
  for (int j = tid; j < span; j += 256) {
    const int64_t g = n_lo + j;
    xs[j] = (g >= 0 && g < n_orig) ? xi[g] : 0.f;
  }
  __syncthreads();
  const int units = (tile / p) / kUpR * p;
  if (tid >= units) return;
  const int ph = tid % p, k = tid / p;
  // outputs t_j = t0 + ph + p (R k + j), inputs n_j = t0 / p + R k + j; tap u reads xs[R k + j + u]
  const double* hp = h + (size_t)ph * mpad;
  const float* xp = xs + kUpR * k;
  double acc[kUpR], xr[kUpR + kUpTB - 1];
#pragma unroll
  for (int j = 0; j < kUpR; ++j) acc[j] = 0.0;
#pragma unroll
  for (int j = 0; j < kUpR - 1; ++j) xr[j] = (double)xp[j];
  for (int u0 = 0; u0 < mpad; u0 += kUpTB) {
#pragma unroll
    for (int e = 0; e < kUpTB; ++e) xr[kUpR - 1 + e] = (double)xp[u0 + kUpR - 1 + e];
#pragma unroll
    for (int e = 0; e < kUpTB; ++e) {
      const double w = hp[u0 + e];
#pragma unroll
      for (int j = 0; j < kUpR; ++j) acc[j] = __builtin_fma(w, xr[e + j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < kUpR - 1; ++j) xr[j] = xr[j + kUpTB];
  }
  float* yo = y + (int64_t)b * ld_out;
#pragma unroll
  for (int j = 0; j < kUpR; ++j) {
    const int64_t t = t0 + ph + (int64_t)p * (kUpR * k + j);
    if (t < ld_out) yo[t] = t < n_out ? (float)acc[j] : 0.f;
  }
}

int64_t gcd64(int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; }

}  // namespace

void launch_pcm16_to_f32(const short* in, int64_t n, float* out, hipStream_t st) {
  const int64_t quads = (n + 3) / 4;
  hipLaunchKernelGGL(pcm16_to_f32_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, in, n, out);
}

void launch_resample(const float* x, int64_t ld_in, const int64_t* len_in, int batch, const float* table, int nwin,
                     int num_table, double ratio, float* y, int64_t ld_out, int64_t* len_out, hipStream_t st) {
  // ratio as p/q: sample rates are integers, so ratio * 48000 * 44100 / ... is overkill -- try denominators up to 1000
  int p = 0, q = 0;   // the reduced ratio p / q (p = the phase count); 0 = no small rational found
  for (int d = 1; d <= 1000 && !p; ++d) {
    const double num = ratio * d;
    const double rn = (double)(int64_t)(num + 0.5);
    if (rn >= 1.0 && rn < 1e6 && (num > rn ? num - rn : rn - num) <= 1e-12 * rn) {
      const int64_t g = gcd64((int64_t)rn, d);
      p = (int)((int64_t)rn / g);
      q = (int)(d / g);
    }
  }
  if (p >= 2 && p <= 16 && q == 1) {   // integer up-sampling: 8 -> 16 kHz
    const int wmax = nwin / num_table + 2;                              // >= taps of either wing
    const int mpad = (2 * wmax + kUpTB - 1) / kUpTB * kUpTB;
    const int tile = kUpR * p * (256 / p);
    const int span = tile / p + mpad + kUpR + kUpTB;
    const size_t lds = sizeof(double) * (size_t)p * mpad + sizeof(float) * span;
    dim3 grid((unsigned)((ld_out + tile - 1) / tile), batch);
    hipLaunchKernelGGL(resample_up_kernel, grid, dim3(256), lds, st, x, ld_in, len_in, reinterpret_cast<const float2*>(table),
                       nwin, num_table, p, wmax, mpad, span, tile, y, ld_out, len_out);
    return;
  }
  if (p > 0) {
    const double scale = ratio < 1.0 ? ratio : 1.0;
    const int index_step = (int)(scale * num_table);
    const int wmax = nwin / index_step + 2;
    const int span = (int)((double)kPhaseTile / ratio) + 2 * wmax + 8;
    const size_t lds = sizeof(float) * ((size_t)2 * p * wmax + span) + sizeof(int) * 2 * p;
    if (lds <= 60 * 1024) {
      dim3 grid((unsigned)((ld_out + kPhaseTile - 1) / kPhaseTile), batch);
      hipLaunchKernelGGL(resample_phase_kernel, grid, dim3(256), lds, st, x, ld_in, len_in,
                         reinterpret_cast<const float2*>(table), nwin, num_table, ratio, p, wmax, span, y, ld_out, len_out);
      return;
    }
  }
  dim3 grid((unsigned)((ld_out + 255) / 256), batch);
  hipLaunchKernelGGL(resample_kernel, grid, dim3(256), 0, st, x, ld_in, len_in, reinterpret_cast<const float2*>(table),
                     nwin, num_table, ratio, y, ld_out, len_out);
}

}  // namespace vasr
