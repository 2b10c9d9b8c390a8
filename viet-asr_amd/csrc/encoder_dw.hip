// Depthwise masked 1-D convolution for gfx950 + the length chain + re-padding helper.
//
// Replaces the groups == channels MaskedConv1d of every separable JasperBlock
// (reference nemo/collections/asr/parts/jasper.py:113-132 mask + conv, :360-373 layer order).
//
// HBM-bound by design (2*K flops per 8 bytes): one wavefront owns one (utterance, channel) row
// segment of 512 outputs.  The masked input window is staged once into LDS with 16-byte
// coalesced loads; every lane then pulls a register window of K+3 samples with ds_read_b128 and
// produces two groups of 4 consecutive outputs, so that each input sample is read from HBM
// exactly once and both the LDS reads and the 16-byte output stores are conflict free / fully
// coalesced.  The K taps of the row's channel are wave-uniform and travel through SGPRs.
#include "vasr_internal.h"

namespace vasr {

namespace {

constexpr int kTile = 512;  // outputs per wavefront (2 groups x 64 lanes x 4)
using v4f = __attribute__((ext_vector_type(4))) float;  // native vector: one ds_read_b128 / global dwordx4

template <int K>
struct DwGeom {
  static constexpr int PAD = K / 2;
  static constexpr int PADL = (PAD + 3) & ~3;
  static constexpr int OFF = PADL - PAD;
  static constexpr int NQ = (OFF + K + 3 + 3) / 4;       // float4 reads per lane per group
  static constexpr int WIN = 256 + 252 + 4 * NQ;         // floats of LDS per wavefront
};

// grid (C/4, B, ceil(ldy/512)), block 256 = 4 wavefronts = 4 channels
template <int K>
__global__ __launch_bounds__(256) void dw_conv_kernel(const float* __restrict__ x, int64_t ldx,
                                                      const float* __restrict__ w,
                                                      const int32_t* __restrict__ lens_in,
                                                      const int32_t* __restrict__ lens_out, int channels,
                                                      float* __restrict__ y, int64_t ldy) {
  using G = DwGeom<K>;
  __shared__ v4f lds4[4 * G::WIN / 4];  // vector-typed so every window access is a provable ds_*_b128
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = blockIdx.x * 4 + wave;
  const int b = blockIdx.y;
  const int t_start = blockIdx.z * kTile;
  const int len_in = lens_in[b];
  const int len_out = lens_out[b];
  const int64_t row = (int64_t)b * channels + c;
  const float* xr = x + row * ldx;
  v4f* win4 = lds4 + wave * (G::WIN / 4);

  // ---- stage masked window: LDS index i <-> frame t_start - PADL + i ----
  for (int i4 = lane; i4 < G::WIN / 4; i4 += 64) {
    const int t = t_start - G::PADL + 4 * i4;
    v4f v = {0.f, 0.f, 0.f, 0.f};
    if (t >= 0 && t < len_in && t + 3 < ldx) {
      v = *reinterpret_cast<const v4f*>(xr + t);
      if (t + 1 >= len_in) v.y = 0.f;     // MaskedConv1d: x.masked_fill(t >= lens, 0)  (jasper.py:113-118)
      if (t + 2 >= len_in) v.z = 0.f;
      if (t + 3 >= len_in) v.w = 0.f;
    }
    win4[i4] = v;
  }
  __syncthreads();

  const float* wc = w + (int64_t)c * K;  // wave-uniform -> scalar loads
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int base = g * 256 + lane * 4;
    float xw[4 * G::NQ];
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
      const v4f v = win4[g * 64 + lane + q];
      xw[4 * q + 0] = v.x; xw[4 * q + 1] = v.y; xw[4 * q + 2] = v.z; xw[4 * q + 3] = v.w;
    }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float wk = wc[k];
      a0 = fmaf(wk, xw[G::OFF + k + 0], a0);
      a1 = fmaf(wk, xw[G::OFF + k + 1], a1);
      a2 = fmaf(wk, xw[G::OFF + k + 2], a2);
      a3 = fmaf(wk, xw[G::OFF + k + 3], a3);
    }
    const int t = t_start + base;
    if (t < ldy) {
      // the following 1x1 MaskedConv1d masks with lens_out: zero here so the GEMM needs no predicate
      v4f o;
      o.x = (t + 0 < len_out) ? a0 : 0.f;
      o.y = (t + 1 < len_out) ? a1 : 0.f;
      o.z = (t + 2 < len_out) ? a2 : 0.f;
      o.w = (t + 3 < len_out) ? a3 : 0.f;
      *reinterpret_cast<v4f*>(y + row * ldy + t) = o;
    }
  }
}

// Any kernel / stride / dilation / row pitch: one thread per output, taps straight from L1/L2.
// Used for the stride-2 prologue block (64 channels) and the dilated K=87 block.
__global__ __launch_bounds__(256) void dw_conv_generic_kernel(const float* __restrict__ x, int64_t ldx,
                                                              int frames_in, const float* __restrict__ w,
                                                              const int32_t* __restrict__ lens_in,
                                                              const int32_t* __restrict__ lens_out,
                                                              int channels, int K, int stride, int dil, int pad,
                                                              float* __restrict__ y, int64_t ldy) {
  const int c = blockIdx.y, b = blockIdx.z;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ldy) return;
  int len_in = lens_in[b];
  if (len_in > frames_in) len_in = frames_in;
  const int64_t row = (int64_t)b * channels + c;
  const float* xr = x + row * ldx;
  const float* wc = w + (int64_t)c * K;
  float acc = 0.f;
  if (t < lens_out[b]) {
    const int s0 = t * stride - pad;
    for (int k = 0; k < K; ++k) {
      const int s = s0 + k * dil;
      if (s >= 0 && s < len_in) acc = fmaf(wc[k], xr[s], acc);
    }
  }
  y[row * ldy + t] = acc;
}

// MaskedConv1d.get_seq_len chain (jasper.py:108-111): lens.to(long) for the mask, then
// (lens + 2p - d(K-1) - 1) / stride + 1 as a FLOAT tensor (quirk Q3).
__global__ void len_chain_kernel(const int64_t* __restrict__ seq, int batch, const LenStep* __restrict__ steps,
                                 int n_steps, int32_t* __restrict__ lens_tab, float* __restrict__ enc_len) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  float lf = (float)seq[b];
  int64_t li = seq[b];
  for (int s = 0; s < n_steps; ++s) {
    if (s > 0) li = (int64_t)lf;  // .to(dtype=torch.long): truncation
    lens_tab[(int64_t)s * batch + b] = (int32_t)li;
    const LenStep st = steps[s];
    lf = (float)(li + 2 * st.pad - st.dilation * (st.kernel - 1) - 1) / (float)st.stride + 1.0f;
  }
  lens_tab[(int64_t)n_steps * batch + b] = (int32_t)(int64_t)lf;
  if (enc_len) enc_len[b] = lf;
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
                 int channels, float* y, int64_t ldy, hipStream_t st) {
  dim3 grid(channels / 4, batch, (unsigned)((ldy + kTile - 1) / kTile));
  hipLaunchKernelGGL(dw_conv_kernel<K>, grid, dim3(256), 0, st, x, ldx, w, li, lo, channels, y, ldy);
}

}  // namespace

void launch_depthwise(const float* x, int64_t ldx, int frames_in, const float* w, const int32_t* lens_in,
                      const int32_t* lens_out, int batch, int channels, int kernel, int stride, int dilation,
                      int pad, float* y, int64_t ldy, hipStream_t st) {
  const bool fast = stride == 1 && dilation == 1 && pad == kernel / 2 && channels % 4 == 0 && ldx % 4 == 0 &&
                    ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(y) & 15) == 0;
  if (fast) {
    switch (kernel) {
      case 33: return launch_dw_t<33>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st);
      case 39: return launch_dw_t<39>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st);
      case 51: return launch_dw_t<51>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st);
      case 63: return launch_dw_t<63>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st);
      case 75: return launch_dw_t<75>(x, ldx, w, lens_in, lens_out, batch, channels, y, ldy, st);
      default: break;
    }
  }
  dim3 grid((unsigned)((ldy + 255) / 256), channels, batch);
  hipLaunchKernelGGL(dw_conv_generic_kernel, grid, dim3(256), 0, st, x, ldx, frames_in, w, lens_in, lens_out,
                     channels, kernel, stride, dilation, pad, y, ldy);
}

void launch_len_chain(const int64_t* seq, int batch, const LenStep* d_steps, int n_steps, int32_t* lens_tab,
                      float* enc_len, hipStream_t st) {
  hipLaunchKernelGGL(len_chain_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, seq, batch, d_steps, n_steps,
                     lens_tab, enc_len);
}

void launch_repad(const float* src, int64_t src_ld, int rows, int frames, float* dst, int64_t dst_ld,
                  hipStream_t st) {
  dim3 grid((unsigned)((dst_ld + 255) / 256), rows);
  hipLaunchKernelGGL(repad_kernel, grid, dim3(256), 0, st, src, src_ld, frames, dst, dst_ld);
}

}  // namespace vasr
