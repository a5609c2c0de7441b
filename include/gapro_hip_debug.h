/*
 * gapro_hip_debug.h -- measurement and self-test entry points of libgapro_hip_debug.so.
 *
 * Not part of the product: libgapro_hip.so (include/gapro_hip.h) neither exports nor needs any of these, and nothing
 * under gapro_amd/ loads this library except on request (gapro_amd._lib.load_debug(), used by tests/test_fit_gpu.py's
 * lane-map check, tools/mfma_peak.py, tools/wgloop_peak.py, tools/product_bench.py, tools/pmc_calib.py and bench.py's
 * `peak_measured` key).  Every function takes a gapro_ctx created by libgapro_hip.so's gapro_ctx_create.
 */
#ifndef GAPRO_HIP_DEBUG_H
#define GAPRO_HIP_DEBUG_H

#include "gapro_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* C[16x16] = P^T Q for row-major P, Q with 16 columns and K rows (K % 4 == 0): checks the
 * v_mfma_f64_16x16x4_f64 lane maps the fit kernel relies on. */
int gapro_debug_mfma_tn(gapro_ctx* ctx, void* stream, const double* d_P, const double* d_Q, double* d_C,
                        int32_t K);
/* The fit kernels' special functions (csrc/fit_math.h), elementwise over n doubles: which = 0 erfcx_tab(x) (x >= 0),
 * 1 exp_neg(x) (x <= 0), 2 ndtr_ratio(x) = phi(x) / Phi(x), 3 / 4 log Phi(x) and the ratio of log_ndtr_ratio.
 * (tests/test_fit_gpu.py holds them against SciPy.) */
int gapro_debug_fit_math(gapro_ctx* ctx, void* stream, int64_t n, const double* d_x, double* d_out, int32_t which);
/* Streaming kernels with a known byte count (one double per lane, grid-stride: the access width of the
 * fit kernel), to calibrate the FETCH_SIZE / WRITE_SIZE counters.  mode 0: read n doubles, one atomic
 * partial sum per wave into d_dst[0..4095]; mode 1: copy n doubles. */
int gapro_debug_stream(gapro_ctx* ctx, void* stream, int64_t n, const double* d_src, double* d_dst,
                       int32_t mode);
/* Matrix-core peak of this device, measured: every CU runs waves_per_simd waves per SIMD, each a loop of
 * `iters` x 8 independent 16x16x4 MFMA chains with no memory traffic (kind 0: v_mfma_f64_16x16x4_f64,
 * kind 1: v_mfma_f32_16x16x4_f32, kind 2: 16 chains of v_mfma_f64_4x4x4_4b_f64, kind 3: that instruction with a
 * different A / B register pair per instruction, 8 x 4 accumulators; d_sink then >= 32 device doubles, read); timed
 * with HIP events on `stream`, blocking.  d_sink: one device double.
 * The roofline peak the fit kernels are priced against (the local guide has no FP64 row). */
int gapro_debug_mfma_peak(gapro_ctx* ctx, void* stream, int32_t kind, int32_t iters, int32_t waves_per_simd,
                          double* d_sink, double* out_tflops);
/* The same FP64 loop on n_blocks workgroups (<= 0: every CU x waves_per_simd) with the shader clock it ran at:
 * shader cycles (s_memtime) of one wave over the 100 MHz constant clock.  d_sink: three device doubles.  Shows what
 * the matrix cores sustain when only a part of the chip is busy (tools/mfma_peak.py --clock). */
int gapro_debug_mfma_clock(gapro_ctx* ctx, void* stream, int32_t iters, int32_t waves_per_simd, int32_t n_blocks,
                           double* d_sink, double* out_tflops, double* out_shader_mhz);

/* The chunk loop of the staged kernel's workgroup-tiled products piece by piece (round 3): `blocks` workgroups of 512
 * threads, every wave re-reads the MFMA fragments of a 32 x 64 piece from LDS for two k-steps per iteration and issues
 * 16 v_mfma_f64_16x16x4_f64.  mode bits: 1 = + a workgroup barrier per iteration, 2 = + two 16-byte LDS stores per
 * thread, 4 = + two 16-byte global loads per thread (prefetch distance two iterations), 8 = the 16x16x4 form in the
 * plain loop (0 = four v_mfma_f64_4x4x4_4b_f64 per step, A rotated by DPP), 16 = the 16x16x4 form software-pipelined
 * across the barrier as gemm_wg does it, 32 / 64 = variants of where the loads and stores sit.  d_src: >= (blocks + 1)
 * * 65536 doubles, d_sink: one double.  What the FP64 matrix cores sustain in a loop shaped like the products:
 * ~73 TFLOP/s, against ~48 for chains that feed every instruction the same registers (gapro_debug_mfma_peak). */
int gapro_debug_wgloop(gapro_ctx* ctx, void* stream, int32_t iters, int32_t mode, int32_t blocks, const double* d_src,
                       double* d_sink, double* out_tflops);

/* The staged fit kernel's product engines side by side (round 3): n_wg workgroups, each C = P^T Q on its own three
 * mp x mp matrices of d_slab (n_wg * 3 * mp * mp doubles, filled by the caller), `reps` times.  engine 0: one 32 x 32
 * tile per wave from global memory, 1: 64 x 64 tiles, 2: workgroup-tiled through an LDS ring (gemm_wg) with per-wave
 * strips at the matrix edge.  shape 0: full range, 1: triangular P (range [0, i0 + tile)), 2: lower-triangular output,
 * 3: [max(i0, j0), mp), 4: [j0, mp), 5: lower output with [i0, mp) (the caller zeroes the matching triangles of P / Q).
 * mp a multiple of 32, >= 128.  *out_ms: the launch, HIP events on `stream`, blocking. */
int gapro_debug_product_bench(gapro_ctx* ctx, void* stream, int32_t engine, int32_t shape, int32_t mp, int32_t reps,
                              int32_t n_wg, double* d_slab, float* out_ms);

#ifdef __cplusplus
}
#endif
#endif /* GAPRO_HIP_DEBUG_H */
