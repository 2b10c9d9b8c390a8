/* vasr_devtools.h -- measurement and development entry points of libvasr_hip_dev.so (built with -DVASR_DEVTOOLS).
 *
 * NOT part of the drop-in boundary (include/vasr.h): isolated-layer launches for roofline measurements, the host-side
 * weight packers the tests use to feed them, and the bracket-overhead probe.  The same build is the only one that reads
 * the VASR_* kernel-selection switches from the environment (VASR_DW_*, VASR_PW3_TILE, VASR_FUSED*, VASR_NO_*, ...: A/B
 * switches for tests/test_gpu_parity.py::test_alternate_kernel_paths_match_goldens and tools/); the product library
 * libvasr_hip.so exports none of these symbols and ignores those variables. */
#ifndef VASR_DEVTOOLS_H_
#define VASR_DEVTOOLS_H_
#include "vasr.h"
#ifdef __cplusplus
extern "C" {
#endif

/* What one event bracket adds to a bracketed launch: median elapsed time (microseconds) of n brackets around an empty
 * kernel, recorded back to back on `stream` exactly like the profiled launches.  A bracket measures "previous kernel
 * done -> this kernel done", i.e. the kernel plus its dispatch gap; bench.py reports class times both raw and with
 * launches x this value removed (the latter is what rocprofv3 --kernel-trace reports as kernel duration). */
VASR_API int vasr_profile_bracket_overhead(vasr_stream stream, int n, double* out_us);
/* The form a 256-channel separable sub-block takes (csrc/vasr_internal.h fused_tile_choice; pure host arithmetic): 128 / 64 =
 * the fused depthwise + pointwise kernel on tiles of that many frames, 0 = two kernels.  tiles128 = batch x padded frames / 128. */
VASR_API int vasr_fused_tile_choice(int64_t tiles128, int compute_units);
/* Run ONE encoder layer kind in isolation for benchmarking/roofline measurement:
 * kind 0 = depthwise (K, stride 1, dilation 1), 1 = pointwise GEMM (+BN+ReLU epilogue).
 * Buffers are caller provided [B][C][Tp] with Tp = vasr_padded_frames(T). */
VASR_API int vasr_bench_depthwise(const float* d_x, const float* d_w, const int32_t* d_lens, int batch, int channels,
                         int64_t frames, int kernel, float* d_y, vasr_stream stream);
/* Depthwise convolution on the matrix pipe (Toeplitz form, fp16-split arithmetic; stride 1, the (kernel, dilation) pairs
 * of the shipped models): vasr_depthwise_mfma_table_size = dwords per channel of the tap table (0 = shape not covered),
 * vasr_pack_depthwise_taps fills [channels][size] tables and [channels] inverse scales on the host; the bench call runs
 * one layer on [B][C][vasr_padded_frames(T)] buffers (dilation > 1: "same" padding as jasper.py:60-65).  d_amax as in
 * vasr_bench_pointwise_f16x2, amax_stride >= 256 and >= channels * ceil(padded frames / 256) * 4. */
VASR_API int vasr_depthwise_mfma_table_size(int kernel, int dilation);
VASR_API int vasr_pack_depthwise_taps(const float* h_w, int channels, int kernel, int dilation, uint32_t* h_table, float* h_inv);
VASR_API int vasr_bench_depthwise_mfma(const float* d_x, const uint32_t* d_taps, const float* d_tap_inv, const int32_t* d_lens,
                              int batch, int channels, int64_t frames, int kernel, int dilation, float* d_y,
                              uint32_t* d_amax, int amax_stride, vasr_stream stream);
/* Host helper: [cout][cin] row-major weights -> the MFMA fragment order the pointwise kernel streams
 * ([m_pad/32][cin/8][64 lanes][4], rows past cout zero); h_out holds m_pad*cin floats. */
VASR_API int vasr_pack_pointwise(const float* h_w, int cout, int cin, int m_pad, float* h_out);
VASR_API int vasr_bench_pointwise(const float* d_x, const float* d_wt, const float* d_scale, const float* d_shift,
                         int batch, int cin, int cout, int64_t frames, float* d_y, vasr_stream stream);

/* Reference point for the GEMM roofline, like for like: launches `workgroups` x 8 wavefronts that issue nothing but the
 * MFMA stream of the pointwise kernel in arithmetic `gemm_mode` (vasr_set_gemm_mode numbering: 3 = f16x2 -- three
 * v_mfma_f32_32x32x16_f16 per tile pair and k-step on fp16 hi / lo planes, 24 per k-step; 1 = bf16x3 -- six
 * v_mfma_f32_32x32x16_bf16 on three planes, 48 per k-step), same 2 x 4 tile order, operands in registers with the bit
 * statistics of scaled weights and rectified activations; `steps` k-steps per wavefront; *flops = 16-bit flops issued.
 * Timed by the caller; what it sustains is the rate the chip holds under its power limit (box.measured_mfma_tflops). */
VASR_API int vasr_bench_mfma_sustained(int gemm_mode, int workgroups, int steps, float* d_sink, double* flops, vasr_stream stream);

/* 2 x fp16 scaled split variants (fragments: [m_pad/32][cin/16][2][64 lanes][8] fp16 bits; *inv_scale = 1 / the power
 * of two the weights were scaled by).  d_amax: [2][batch][amax_stride] u32 scratch (amax_stride >= 256 and >= cout *
 * frames / 1024) -- table 0 receives per-wavefront maxima of |x| per utterance (the call computes them), table 1 those of
 * |y|; unused slots are zeroed, so max over the last axis is the utterance's maximum. */
VASR_API int vasr_pack_pointwise_f16x2(const float* h_w, int cout, int cin, int m_pad, uint16_t* h_out, float* inv_scale);
VASR_API int vasr_bench_pointwise_f16x2(const float* d_x, const uint16_t* d_w16, float w_inv_scale, const float* d_scale,
                               const float* d_shift, int batch, int cin, int cout, int64_t frames, float* d_y,
                               uint32_t* d_amax, int amax_stride, vasr_stream stream);

/* 3 x bf16 split variants of the two helpers above (fragments: [m_pad/32][cin/16][3][64 lanes][8] bf16 bits). */
VASR_API int vasr_pack_pointwise_bf16x3(const float* h_w, int cout, int cin, int m_pad, uint16_t* h_out);
VASR_API int vasr_bench_pointwise_bf16x3(const float* d_x, const uint16_t* d_w3, const float* d_scale, const float* d_shift,
                                int batch, int cin, int cout, int64_t frames, float* d_y, vasr_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* VASR_DEVTOOLS_H_ */
