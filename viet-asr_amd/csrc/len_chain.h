// MaskedConv1d.get_seq_len chain of the encoder, shared by len_chain_kernel (encoder_dw.hip: the per-module entry points)
// and normalize_chain_kernel (frontend.hip: the fused path runs the chain as extra workgroups of the normalisation launch).
#pragma once
#include <hip/hip_runtime.h>

#include "vasr_internal.h"

namespace vasr {

// MaskedConv1d.get_seq_len chain (jasper.py:108-111): lens.to(long) for the mask, then
// (lens + 2p - d(K-1) - 1) / stride + 1 as a FLOAT tensor (quirk Q3).  Utterance b = first_b + threadIdx.x of the calling
// workgroup (every thread of the workgroup must call: the step table goes through LDS behind a barrier).
// seq_out (optional): the chain's input length is written there (the fused path computes it from the sample count here).
__device__ __forceinline__ void len_chain_body(int b, int64_t l0, int batch, const LenStep* __restrict__ steps, int n_steps,
                                               int32_t* __restrict__ lens_tab, float* __restrict__ enc_len,
                                               const int64_t* __restrict__ wav_len, int hop, int frames_cap) {
  // The chain is serial per utterance and one wavefront runs it alone, so its cost is the length of the dependent
  // instruction sequence of one iteration: the step table goes through LDS once (no dependent global load, and no
  // vector-memory wait inside the loop), the arithmetic is 32-bit (frame counts are far below 2^24, where the
  // int64 <-> float conversions -- software sequences on this ISA -- and the int32 ones give the same values), and
  // x / 1.0f is skipped for the stride-1 steps.
  __shared__ LenStep sh_steps[256];
  for (int s = threadIdx.x; s < n_steps && s < 256; s += blockDim.x) sh_steps[s] = steps[s];
  __syncthreads();
  if (b >= batch) return;
  int32_t li = (int32_t)(l0 > 0x7fffff00 ? 0x7fffff00 : l0);
  float lf = (float)li;
  auto advance = [&](int s, const LenStep st) {
    if (s > 0) li = (int32_t)lf;  // .to(dtype=torch.long): truncation
    lens_tab[(int64_t)s * batch + b] = li;
    lf = (float)(li + 2 * st.pad - st.dilation * (st.kernel - 1) - 1);
    if (st.stride != 1) lf = lf / (float)st.stride;
    lf += 1.0f;
  };
  // two loops, not one with `s < 256 ? sh_steps[s] : steps[s]`: that select becomes a flat load, whose wait
  // (vmcnt(0)) also waits for the previous iteration's store to complete -- 660 cycles per step instead of ~60.
  // The LDS loop takes its steps EIGHT at a time: the eight table reads are in flight together, then eight links of the
  // chain run back to back (one LDS round trip per link was ~2/3 of the kernel's 15 us: 86 links for QuartzNet15x5).
  const int n_lds = n_steps < 256 ? n_steps : 256;
  int s = 0;
  for (; s + 8 <= n_lds; s += 8) {
    LenStep blk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) blk[k] = sh_steps[s + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) advance(s + k, blk[k]);
  }
  for (; s < n_lds; ++s) advance(s, sh_steps[s]);
  for (s = n_lds; s < n_steps; ++s) advance(s, steps[s]);
  lens_tab[(int64_t)n_steps * batch + b] = (int32_t)(int64_t)lf;
  if (enc_len) enc_len[b] = lf;
  if (wav_len) {
    // row n_steps + 1: the output frames an UNBATCHED call on this row would produce (row-independent mode) --
    // torch.stft(center=True) gives 1 + L // hop frames, every conv floor((t + 2 p - d (K - 1) - 1) / stride) + 1;
    // the same count ctc_collapse_kernel stops at
    int64_t t = 1 + wav_len[b] / hop;
    for (int s = 0; s < n_steps; ++s) {   // (integer arithmetic, no conversions: its loads pipeline on their own)
      const LenStep st = s < n_lds ? sh_steps[s] : steps[s];
      t = (t + 2 * st.pad - st.dilation * (st.kernel - 1) - 1) / st.stride + 1;
    }
    lens_tab[(int64_t)(n_steps + 1) * batch + b] = (int32_t)(t < 0 ? 0 : (t < frames_cap ? t : frames_cap));
  }
}

}  // namespace vasr
