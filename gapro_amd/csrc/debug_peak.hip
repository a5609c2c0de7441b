// Matrix-core peak micro-benchmark (debug entry, tools/mfma_peak.py and bench.py's "peak_measured").
//
// The local micro-architecture guide has no FP64 MFMA row; the 78.6 TFLOP/s used as the roofline peak of the
// fit kernels is the public datasheet figure ("verify on the box").  This measures it: every CU
// runs `waves_per_simd` waves per SIMD, each wave a loop of dependent-free v_mfma chains (8 independent
// accumulators, no memory traffic inside the loop), timed with HIP events on the stream they run on.
#include "common.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kChains = 8;

__global__ __launch_bounds__(256) void k_peak_f64(int iters, double* __restrict__ sink) {
  const long long c0 = clock64(), w0 = wall_clock64();  // shader cycles (s_memtime) / 100 MHz constant clock
  d4 acc[kChains];
#pragma unroll
  for (int c = 0; c < kChains; ++c) acc[c] = (d4){0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < kChains; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < kChains; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 12345.678) sink[0] = s;  // keeps the chains alive without a store in the common case
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // sink[1], sink[2]: this wave's shader cycles and 100 MHz ticks
    sink[1] = (double)(clock64() - c0);
    sink[2] = (double)(wall_clock64() - w0);
  }
}

__global__ __launch_bounds__(256) void k_peak_f32(int iters, double* __restrict__ sink) {
  f4 acc[kChains];
#pragma unroll
  for (int c = 0; c < kChains; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + 1e-6f * threadIdx.x, b = 1.0f - 1e-6f * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < kChains; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < kChains; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 12345.678f) sink[0] = s;
}

}  // namespace

static int mfma_peak(gapro_ctx* ctx, void* stream_, int32_t kind, int32_t iters, int32_t waves_per_simd,
                     int32_t n_blocks, double* d_sink, double* out_tflops);

extern "C" int gapro_debug_mfma_peak(gapro_ctx* ctx, void* stream_, int32_t kind, int32_t iters,
                                     int32_t waves_per_simd, double* d_sink, double* out_tflops) {
  return mfma_peak(ctx, stream_, kind, iters, waves_per_simd, 0, d_sink, out_tflops);
}

extern "C" int gapro_debug_mfma_clock(gapro_ctx* ctx, void* stream_, int32_t iters, int32_t waves_per_simd,
                                      int32_t n_blocks, double* d_sink, double* out_tflops, double* out_shader_mhz) {
  if (!out_shader_mhz) return GAPRO_ERR_BAD_ARG;
  const int rc = mfma_peak(ctx, stream_, 0, iters, waves_per_simd, n_blocks, d_sink, out_tflops);
  if (rc != GAPRO_OK) return rc;
  double h[3] = {0.0, 0.0, 0.0};
  GAPRO_HIP_CHECK(ctx, hipMemcpy(h, d_sink, sizeof(h), hipMemcpyDeviceToHost));
  *out_shader_mhz = h[2] > 0.0 ? h[1] / h[2] * 100.0 : 0.0;
  return GAPRO_OK;
}

static int mfma_peak(gapro_ctx* ctx, void* stream_, int32_t kind, int32_t iters, int32_t waves_per_simd,
                     int32_t n_blocks, double* d_sink, double* out_tflops) {
  if (!ctx || !out_tflops || !d_sink || iters <= 0 || waves_per_simd <= 0 || waves_per_simd > 8 || kind < 0 || kind > 1)
    return GAPRO_ERR_BAD_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  hipEvent_t e0, e1;
  GAPRO_HIP_CHECK(ctx, hipEventCreate(&e0));
  GAPRO_HIP_CHECK(ctx, hipEventCreate(&e1));
  const int blocks = n_blocks > 0 ? n_blocks : ctx->n_cu * waves_per_simd;  // 256 threads = one wave per SIMD of a CU
  auto launch = [&](int n) {
    if (kind == 0) hipLaunchKernelGGL(k_peak_f64, dim3(blocks), dim3(256), 0, stream, n, d_sink);
    else hipLaunchKernelGGL(k_peak_f32, dim3(blocks), dim3(256), 0, stream, n, d_sink);
  };
  launch(iters / 8 + 1);  // warm-up (code object load, clocks)
  GAPRO_HIP_CHECK(ctx, hipEventRecord(e0, stream));
  launch(iters);
  GAPRO_HIP_CHECK(ctx, hipEventRecord(e1, stream));
  GAPRO_HIP_CHECK(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  GAPRO_HIP_CHECK(ctx, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  GAPRO_LAUNCH_CHECK(ctx);
  // one 16x16x4 MFMA = 2 * 16 * 16 * 4 FLOP per wave
  const double flop = 2048.0 * kChains * (double)iters * 4.0 * (double)blocks;
  *out_tflops = ms > 0.f ? flop / (ms * 1e-3) / 1e12 : 0.0;
  return GAPRO_OK;
}
