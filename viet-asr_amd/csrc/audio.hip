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
// accumulated per output sample.  One thread per output sample; the table (<= 260 KB fp32 pairs) lives in L2.
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
  const int64_t n_out = (int64_t)((double)n_orig * ratio);   // resampy: int(n_orig * ratio)
  if (t == 0) len_out[b] = n_out;
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

}  // namespace

void launch_pcm16_to_f32(const short* in, int64_t n, float* out, hipStream_t st) {
  const int64_t quads = (n + 3) / 4;
  hipLaunchKernelGGL(pcm16_to_f32_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, in, n, out);
}

void launch_resample(const float* x, int64_t ld_in, const int64_t* len_in, int batch, const float* table, int nwin,
                     int num_table, double ratio, float* y, int64_t ld_out, int64_t* len_out, hipStream_t st) {
  dim3 grid((unsigned)((ld_out + 255) / 256), batch);
  hipLaunchKernelGGL(resample_kernel, grid, dim3(256), 0, st, x, ld_in, len_in, reinterpret_cast<const float2*>(table),
                     nwin, num_table, ratio, y, ld_out, len_out);
}

}  // namespace vasr
