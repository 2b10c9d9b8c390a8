// CTC head epilogue and greedy decode for gfx950.
//
//   logsoftmax_argmax : F.log_softmax(conv_out.transpose(1, 2), dim=-1)   (reference jasper.py:254)
//                       + GreedyCTCDecoder argmax(-1)                      (greedy_ctc_decoder.py:33-36)
//   ctc_collapse      : inner loop of __ctc_decoder_predictions_tensor     (helpers.py:20-31)
//
// The GEMM leaves logits class-major [B][V+1][ld] (time contiguous).  One workgroup takes 64
// consecutive frames, four threads per frame (a quarter of the classes each, coalesced across the
// frames), builds max / sum-exp / arg-max, then the [64][V+1] tile is transposed through
// LDS so that the frame-major log-prob tensor is written as contiguous rows.
#include "vasr_internal.h"

namespace vasr {

namespace {


// grid (ceil(T/64), B), block 256 = 64 frames x 4 class quarters (a wavefront per quarter).  VMAX = classes rounded up to 32,
// VQ = VMAX / 4 per thread: every logit is requested ONCE, all requests in flight together, and max / exp / log-probs /
// arg-max run out of registers.  (Rounds 1-3 walked the classes three times with dependent loads: 14 us per batch-1 call
// at 29 classes, 41 us at the Vietnamese head's 91; round 4's first form -- one lane per frame, all classes in its
// registers -- took 16 us there: 91 exponentials and a 91-iteration transposing store loop with an integer division per
// element, executed by ONE wavefront per 64 frames.)  The arithmetic and its ORDER are those of the one-lane form, so the
// bits are too: the maximum is order-free, the exponentials go through LDS and are summed by one lane per frame in
// ascending class order, the arg-max keeps the first maximum (strict > within a quarter, quarters combined in order).
template <int VMAX>
__global__ __launch_bounds__(256) void logsoftmax_argmax_kernel(const float* __restrict__ logits, int64_t row_ld,
                                                                int64_t batch_stride, int frames, int V,
                                                                float* __restrict__ logp,
                                                                int64_t* __restrict__ pred) {
  constexpr int VQ = VMAX / 4;
  extern __shared__ float tile[];  // [64][VP] | red[4][64] | bestq[4][64] | argq[4][64]
  const int VP = (V + 1) | 1;      // odd pitch: a lane walking its frame's classes meets every bank
  float* red = tile + 64 * VP;
  float* bestq = red + 256;
  int* argq = reinterpret_cast<int*>(bestq + 256);
  const int tid = threadIdx.x, f = tid & 63, b = blockIdx.y;
  const int cq = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t0 = blockIdx.x * 64, t = t0 + f;
  const float* x = logits + (int64_t)b * batch_stride + min(t, frames - 1);
  const int n_t = min(64, frames - t0);
  const int v0 = cq * VQ;
  float xv[VQ];
#pragma unroll
  for (int j = 0; j < VQ; ++j) xv[j] = x[(int64_t)min(v0 + j, V - 1) * row_ld];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < VQ; ++j) if (v0 + j < V) mx = fmaxf(mx, xv[j]);
  red[cq * 64 + f] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[f], red[64 + f]), fmaxf(red[128 + f], red[192 + f]));
#pragma unroll
  for (int j = 0; j < VQ; ++j) if (v0 + j < V) tile[f * VP + v0 + j] = expf(xv[j] - mx);
  __syncthreads();
  if (cq == 0) {
    float s = 0.f;
    for (int v = 0; v < V; ++v) s += tile[f * VP + v];   // ascending, one accumulator: the order of the sum is the result's bits
    red[f] = logf(s);
  }
  __syncthreads();
  const float ls = red[f];
  float best = -INFINITY;
  int arg = 0;
#pragma unroll
  for (int j = 0; j < VQ; ++j) {
    if (v0 + j < V) {
      const float lp = (xv[j] - mx) - ls;
      if (lp > best) { best = lp; arg = v0 + j; }  // strict >: first maximum wins (quirk Q6)
      if (logp) tile[f * VP + v0 + j] = lp;
    }
  }
  bestq[cq * 64 + f] = best;
  argq[cq * 64 + f] = arg;
  __syncthreads();
  if (cq == 0 && pred && t < frames) {
#pragma unroll
    for (int q = 1; q < 4; ++q)
      if (bestq[q * 64 + f] > best) { best = bestq[q * 64 + f]; arg = argq[q * 64 + f]; }
    pred[(int64_t)b * frames + t] = arg;
  }
  if (logp) {
    float* out = logp + ((int64_t)b * frames + t0) * V;
    for (int r = cq; r < n_t; r += 4)
      for (int c = f; c < V; c += 64) out[(int64_t)r * V + c] = tile[r * VP + c];
  }
}

// standalone GreedyCTCDecoder on a frame-major [B*T][V] tensor: 4 lanes... one lane per frame row
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logp, int64_t rows, int V,
                                                     int64_t* __restrict__ pred) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* x = logp + r * V;
  float best = x[0];
  int arg = 0;
  bool nan_hit = best != best;
  for (int v = 1; v < V; ++v) {
    const float f = x[v];
    if (!nan_hit && (f > best || f != f)) { best = f; arg = v; nan_hit = f != f; }  // torch: NaN is a maximum
  }
  pred[r] = arg;
}

// One wavefront per utterance: ballot + popcount stream compaction over all frames (quirk Q4).
__global__ __launch_bounds__(64) void ctc_collapse_kernel(const int64_t* __restrict__ pred, int64_t frames,
                                                          int blank, int32_t* __restrict__ ids,
                                                          int32_t* __restrict__ id_len,
                                                          const int64_t* __restrict__ wav_len, int hop,
                                                          const LenStep* __restrict__ steps, int n_steps) {
  const int lane = threadIdx.x, b = blockIdx.x;
  const int64_t* p = pred + (int64_t)b * frames;
  int32_t* out = ids + (int64_t)b * frames;
  int64_t own = frames;
  if (wav_len) {
    // frames of an unbatched call on this row: torch.stft(center=True) gives 1 + L // hop, every conv
    // floor((t + 2 p - d (K - 1) - 1) / stride) + 1
    int64_t t = 1 + wav_len[b] / hop;
    for (int s = 0; s < n_steps; ++s) {
      const LenStep st = steps[s];
      t = (t + 2 * st.pad - st.dilation * (st.kernel - 1) - 1) / st.stride + 1;
    }
    own = t < frames ? (t < 0 ? 0 : t) : frames;
  }
  int count = 0;
  for (int64_t t0 = 0; t0 < own; t0 += 64) {
    const int64_t t = t0 + lane;
    bool keep = false;
    int cur = blank;
    if (t < own) {
      cur = (int)p[t];
      const int prev = t > 0 ? (int)p[t - 1] : blank;  // previous = blank before the first frame
      keep = (cur != prev || prev == blank) && cur != blank;
    }
    const unsigned long long m = __ballot(keep);
    if (keep) out[count + __popcll(m & ((1ull << lane) - 1ull))] = cur;
    count += __popcll(m);
  }
  if (lane == 0) id_len[b] = count;
}

}  // namespace

void launch_logsoftmax_argmax(const float* logits, int64_t row_ld, int64_t batch_stride, int batch, int frames,
                              int num_classes, float* logp, int64_t* pred, hipStream_t st) {
  dim3 grid((frames + 63) / 64, batch);
  const size_t lds = ((size_t)64 * ((num_classes + 1) | 1) + 3 * 256) * sizeof(float);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, logits, row_ld, batch_stride, frames, num_classes, logp, pred);
  };
  if (num_classes <= 32) go(logsoftmax_argmax_kernel<32>);
  else if (num_classes <= 64) go(logsoftmax_argmax_kernel<64>);
  else if (num_classes <= 96) go(logsoftmax_argmax_kernel<96>);
  else go(logsoftmax_argmax_kernel<128>);
}

void launch_argmax(const float* logp, int batch, int64_t frames, int num_classes, int64_t* pred, hipStream_t st) {
  const int64_t rows = (int64_t)batch * frames;
  hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, logp, rows,
                     num_classes, pred);
}

void launch_ctc_collapse(const int64_t* pred, int batch, int64_t frames, int blank, int32_t* ids, int32_t* id_len,
                         hipStream_t st, const int64_t* wav_len, int hop, const LenStep* steps, int n_steps) {
  hipLaunchKernelGGL(ctc_collapse_kernel, dim3(batch), dim3(64), 0, st, pred, frames, blank, ids, id_len, wav_len,
                     hop, steps, n_steps);
}

}  // namespace vasr
