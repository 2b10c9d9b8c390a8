// Device-side helpers shared by the gfx950 kernels (included from .hip files only).
#pragma once
#include <hip/hip_runtime.h>

namespace vasr {

// Wave-wide unsigned maximum, uniform result: butterfly inside each 16-lane row on DPP (VALU only -- a ds_bpermute
// chain is six dependent LDS round trips at the tail of every wavefront), then the four row maxima through SGPRs.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define VASR_DPP(x, ctrl) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), 0xF, 0xF, false))
  v = max(v, VASR_DPP(v, 0xB1));    // quad_perm [1,0,3,2]
  v = max(v, VASR_DPP(v, 0x4E));    // quad_perm [2,3,0,1]
  v = max(v, VASR_DPP(v, 0x141));   // row_half_mirror
  v = max(v, VASR_DPP(v, 0x140));   // row_mirror: every lane of a row now holds the row's maximum
#undef VASR_DPP
  const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}

__device__ __forceinline__ unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }

// ---- per-utterance maxima (AmaxTab, vasr_internal.h) ----
// Producer side: every wavefront of the producing launch owns ONE slot per utterance it touches and stores its maximum
// there with a plain store (slot < n, n set by the launcher).  Round 2 first used one atomicMax per wavefront into 8
// words per utterance: 32 768 device-scope atomics per depthwise launch land in a handful of cache lines of one memory
// channel and retire at ~1 per ns -- +30 us on a 25 us kernel.  Plain stores to distinct words cost nothing measurable.
__device__ __forceinline__ void amax_publish(unsigned* tab, int stride, int b, int slot, unsigned lane_max, int lane) {
  const unsigned m = wave_max_u32(lane_max);
  if (lane == 0) tab[(int64_t)b * stride + slot] = m;
}
// Consumer side: a wavefront reduces the n slots of utterance b itself (n * 4 bytes from L2, coalesced; no barrier).
__device__ __forceinline__ unsigned amax_read(const unsigned* __restrict__ tab, int stride, int n, int b, int lane) {
  const unsigned* p = tab + (int64_t)b * stride;
  unsigned m = 0;
  for (int i = lane; i < n; i += 64) m = max(m, p[i]);
  return wave_max_u32(m);
}

// The same reduction in two halves, eight slots per lane in flight: amax_request() issues the loads of the first 512
// slots (one per channel of a depthwise layer) and returns; the caller requests whatever else it needs next (a GEMM
// workgroup: its first rows and weight fragments) and calls amax_collect() when it needs the value -- s_waitcnt counts in
// order, so waiting for these OLDER loads leaves the younger ones in flight.  The serial form was four to eight dependent
// round trips to L2 (2-3 us) at the head of every GEMM workgroup, before its first row was even requested.  (Clamped, not
// predicated: a slot read twice does not change a maximum, and the loads stay unconditional.)
__device__ __forceinline__ void amax_request(const unsigned* __restrict__ tab, int stride, int n, int b, int lane,
                                             unsigned (&v)[8]) {
  const unsigned* p = tab + (int64_t)b * stride;
  const int last = max(n, 1) - 1;   // (a table always has a slot 0; n <= 0 does not occur for a published table)
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = p[min(64 * u + lane, last)];
}
__device__ __forceinline__ unsigned amax_collect(const unsigned* __restrict__ tab, int stride, int n, int b, int lane,
                                                 const unsigned (&v)[8]) {
  unsigned m = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) m = max(m, v[u]);
  if (n > 512) {   // several time tiles per channel (long recordings): the rest in further trips
    const unsigned* p = tab + (int64_t)b * stride;
    for (int i0 = 512; i0 < n; i0 += 512) {
      unsigned w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = p[min(i0 + 64 * u + lane, n - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) m = max(m, w[u]);
    }
  }
  return wave_max_u32(m);
}

// power-of-two scale (and its inverse) that puts a maximum of magnitude `amax_bits` (fp32 bit pattern of |x|) into
// [2^14, 2^15): the fp16 planes then have 18 octaves of full 22-bit precision below the maximum
__device__ __forceinline__ void f16_scale(unsigned amax_bits, float* scale, float* inv) {
  int e = (int)(amax_bits >> 23);
  e = e < 16 ? 16 : (e > 254 ? 254 : e);
  *scale = __uint_as_float((unsigned)(268 - e) << 23);   // 2^(141 - e)
  *inv = __uint_as_float((unsigned)(e - 14) << 23);      // 2^(e - 141)
}

}  // namespace vasr
