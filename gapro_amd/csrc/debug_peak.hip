// Matrix-core peak micro-benchmark (debug entry, tools/mfma_peak.py and bench.py's "peak_measured").
//
// The local micro-architecture guide has no FP64 MFMA row; the 78.6 TFLOP/s used as the roofline peak of the
// fit kernels is the public datasheet figure ("verify on the box").  This measures it: every CU
// runs `waves_per_simd` waves per SIMD, each wave a loop of dependent-free v_mfma chains (8 independent
// accumulators, no memory traffic inside the loop), timed with HIP events on the stream they run on.
#include "common.h"
#include "../../include/gapro_hip_debug.h"
#include "mfma64.h"

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

// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 blocks per instruction (512 FLOP), 16 accumulator chains
__global__ __launch_bounds__(256) void k_peak_f64_4x4(int iters, double* __restrict__ sink) {
  double acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.0;
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < 16; ++c) s += acc[c];
  if (s == 12345.678) sink[0] = s;
}

// the same instruction as the products use it: 8 x 4 accumulators, a different A / B register pair per instruction,
// the four instructions of one 16 x 16 x 4 step back to back (same B, A rotated)
__global__ __launch_bounds__(256) void k_peak_f64_4x4_tile(int iters, double* __restrict__ sink) {
  double acc[8][4];
  double a[8], b[4];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    a[u] = sink[8 + u] + 1e-9 * threadIdx.x;
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) b[v] = sink[16 + v] - 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u][v] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u], b[v], acc[u][v], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int u = 0; u < 8; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) s += acc[u][v];
  if (s == 12345.678) sink[0] = s;
}

// The chunk loop of gemm_wg (svgp_fit.hip) piece by piece, 512 threads = one workgroup per CU: per iteration every wave
// reads the fragments of a 32 x 64 piece for two k-steps from LDS (6 reads each), rotates the A fragments and issues
// 64 block instructions.  mode bit 0: + a workgroup barrier per iteration; bit 1: + two 16-byte LDS stores; bit 2: + two
// 16-byte global loads per thread and iteration (prefetch distance two iterations); bit 3: the 16x16x4 form instead.
template <bool M16>
__global__ __launch_bounds__(512, 2) void k_peak_wgloop(int iters, int mode, const double* __restrict__ src,
                                                        double* __restrict__ sink) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  __shared__ double lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;  // mode bit 7: full-entropy mantissas in [-1, 1)
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned h2 = h * 3266489917u + 374761393u;
    const double r = ((double)h + (double)h2 * 2.3283064365386963e-10) * 4.656612873077393e-10 - 1.0;
    lds[i] = (mode & 128) ? r : 1.0 + 1e-9 * i;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4;
  const int wi = wave < 4 ? wave : 7 - wave, wj = wave >> 2, sw = (lq & 1) << 4;
  int fa[2], fb[4];
  for (int u = 0; u < 2; ++u) fa[u] = lq * 128 + ((32 * wi + 16 * u + lr) ^ sw);
  for (int v = 0; v < 4; ++v) fb[v] = 1024 + lq * 128 + ((64 * wj + 16 * v + lr) ^ sw);
  const int lds_wr = wave * 128 + ((2 * lane) ^ ((wave & 1) << 4));
  gapro_mfma::d4 acc[2][4];
  for (int u = 0; u < 2; ++u)
    for (int v = 0; v < 4; ++v) acc[u][v] = (gapro_mfma::d4){0.0, 0.0, 0.0, 0.0};
  const double* g = src + (size_t)blockIdx.x * 65536 + threadIdx.x * 2;
  d2 r0 = *(const d2*)g, r1 = *(const d2*)(g + 1024), q0 = *(const d2*)(g + 2048), q1 = *(const d2*)(g + 3072);
  for (int it = 0; it < iters; ++it) {
    const double* st = lds + (it & 1) * 2048;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      double b[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) b[v] = st[fb[v] + 4 * s * 128];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const double a = st[fa[u] + 4 * s * 128];
        if (M16) {
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[v], acc[u][v], 0, 0, 0);
        } else {
          const gapro_mfma::AFrag af = gapro_mfma::make_afrag(a);
#pragma unroll
          for (int v = 0; v < 4; ++v) gapro_mfma::mma16(af, b[v], acc[u][v]);
        }
      }
    }
    if (mode & 2) {
      double* w = lds + ((it + 1) & 1) * 2048 + lds_wr;
      *(d2*)w = (it & 1) ? r1 : r0;
      *(d2*)(w + 1024) = (it & 1) ? q1 : q0;
    }
    if (mode & 4) {
      const double* gp = g + (size_t)((it + 2) & 7) * 8192;
      if (it & 1) { r1 = *(const d2*)gp; q1 = *(const d2*)(gp + 4096); }
      else { r0 = *(const d2*)gp; q0 = *(const d2*)(gp + 4096); }
    }
    if (mode & 1) __syncthreads();
  }
  double s = r0[0] + r1[0] + q0[1] + q1[1];
  for (int u = 0; u < 2; ++u)
    for (int v = 0; v < 4; ++v) s += acc[u][v][0] + acc[u][v][1] + acc[u][v][2] + acc[u][v][3];
  if (s == 12345.678) sink[0] = s;
}

// The same loop software-pipelined across the barrier (mode bit 4): the second k-step's fragments are requested
// before the first k-step's MFMAs, the next chunk's first k-step right behind the barrier and covered by the second
// k-step's MFMAs, so that the matrix pipe never waits for an LDS round trip.
__global__ __launch_bounds__(512, 2) void k_peak_wgloop_sp(int iters, int mode, const double* __restrict__ src,
                                                           double* __restrict__ sink) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  __shared__ double lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;  // mode bit 7: full-entropy mantissas in [-1, 1)
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned h2 = h * 3266489917u + 374761393u;
    const double r = ((double)h + (double)h2 * 2.3283064365386963e-10) * 4.656612873077393e-10 - 1.0;
    lds[i] = (mode & 128) ? r : 1.0 + 1e-9 * i;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4;
  const int wi = wave < 4 ? wave : 7 - wave, wj = wave >> 2, sw = (lq & 1) << 4;
  int fa[2], fb[4];
  for (int u = 0; u < 2; ++u) fa[u] = lq * 128 + ((32 * wi + 16 * u + lr) ^ sw);
  for (int v = 0; v < 4; ++v) fb[v] = 1024 + lq * 128 + ((64 * wj + 16 * v + lr) ^ sw);
  const int lds_wr = wave * 128 + ((2 * lane) ^ ((wave & 1) << 4));
  gapro_mfma::d4 acc[2][4];
  for (int u = 0; u < 2; ++u)
    for (int v = 0; v < 4; ++v) acc[u][v] = (gapro_mfma::d4){0.0, 0.0, 0.0, 0.0};
  const double* g = src + (size_t)blockIdx.x * 65536 + threadIdx.x * 2;
  d2 r0 = *(const d2*)g, r1 = *(const d2*)(g + 1024), q0 = *(const d2*)(g + 2048), q1 = *(const d2*)(g + 3072);
  double a0[2], b0[4], a1[2], b1[4];
  auto rd = [&](const double* st, int s, double (&a)[2], double (&b)[4]) {
#pragma unroll
    for (int v = 0; v < 4; ++v) b[v] = st[fb[v] + 4 * s * 128];
#pragma unroll
    for (int u = 0; u < 2; ++u) a[u] = st[fa[u] + 4 * s * 128];
  };
  auto mm = [&](const double (&a)[2], const double (&b)[4]) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[v], acc[u][v], 0, 0, 0);
  };
  rd(lds, 0, a0, b0);
  // one chunk; (r, q) = the register stage stored this iteration and refilled.  mode bit 5: the stores do not depend on
  // the loads; bit 6: the loads are issued at the end of the iteration instead of at its start
  auto chunk = [&](int it, d2& r, d2& q) {
    const double* st = lds + (it & 1) * 2048;
    const double* sn = lds + ((it + 1) & 1) * 2048;
    const double* gp = g + (size_t)((it + 2) & 7) * 8192;
    if (mode & 2) {
      double* w = lds + ((it + 1) & 1) * 2048 + lds_wr;
      if (mode & 32) {
        *(d2*)w = (d2){1.0, 2.0};
        *(d2*)(w + 1024) = (d2){3.0, 4.0};
      } else {
        *(d2*)w = r;
        *(d2*)(w + 1024) = q;
      }
    }
    if ((mode & 4) && !(mode & 64)) {
      r = *(const d2*)gp;
      q = *(const d2*)(gp + 4096);
    }
    rd(st, 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mm(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    if (mode & 1) __syncthreads();
    rd(sn, 0, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mm(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    if ((mode & 4) && (mode & 64)) {
      r = *(const d2*)gp;
      q = *(const d2*)(gp + 4096);
    }
  };
  for (int it = 0; it + 1 < iters; it += 2) {
    chunk(it, r1, q1);
    chunk(it + 1, r0, q0);
  }
  double s = r0[0] + r1[0] + q0[1] + q1[1] + a0[0] + b0[0];
  for (int u = 0; u < 2; ++u)
    for (int v = 0; v < 4; ++v) s += acc[u][v][0] + acc[u][v][1] + acc[u][v][2] + acc[u][v][3];
  if (s == 12345.678) sink[0] = s;
}

}  // namespace

extern "C" int gapro_debug_wgloop(gapro_ctx* ctx, void* stream_, int32_t iters, int32_t mode, int32_t blocks,
                                  const double* d_src, double* d_sink, double* out_tflops) {
  if (!ctx || !d_src || !d_sink || !out_tflops || iters <= 0 || blocks <= 0) return GAPRO_ERR_BAD_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  hipEvent_t e0, e1;
  GAPRO_HIP_CHECK(ctx, hipEventCreate(&e0));
  GAPRO_HIP_CHECK(ctx, hipEventCreate(&e1));
  auto launch = [&](int n) {
    if (mode & 16) hipLaunchKernelGGL(k_peak_wgloop_sp, dim3(blocks), dim3(512), 0, stream, n, mode, d_src, d_sink);
    else if (mode & 8) hipLaunchKernelGGL(k_peak_wgloop<true>, dim3(blocks), dim3(512), 0, stream, n, mode, d_src, d_sink);
    else hipLaunchKernelGGL(k_peak_wgloop<false>, dim3(blocks), dim3(512), 0, stream, n, mode, d_src, d_sink);
  };
  launch(iters / 8 + 1);
  GAPRO_HIP_CHECK(ctx, hipEventRecord(e0, stream));
  launch(iters);
  GAPRO_HIP_CHECK(ctx, hipEventRecord(e1, stream));
  GAPRO_HIP_CHECK(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  GAPRO_HIP_CHECK(ctx, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  GAPRO_LAUNCH_CHECK(ctx);
  *out_tflops = ms > 0.f ? 2048.0 * 16.0 * 8.0 * (double)iters * (double)blocks / (ms * 1e-3) / 1e12 : 0.0;
  return GAPRO_OK;
}

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
  if (!ctx || !out_tflops || !d_sink || iters <= 0 || waves_per_simd <= 0 || waves_per_simd > 8 || kind < 0 || kind > 3)
    return GAPRO_ERR_BAD_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  hipEvent_t e0, e1;
  GAPRO_HIP_CHECK(ctx, hipEventCreate(&e0));
  GAPRO_HIP_CHECK(ctx, hipEventCreate(&e1));
  const int blocks = n_blocks > 0 ? n_blocks : ctx->n_cu * waves_per_simd;  // 256 threads = one wave per SIMD of a CU
  auto launch = [&](int n) {
    if (kind == 3) hipLaunchKernelGGL(k_peak_f64_4x4_tile, dim3(blocks), dim3(256), 0, stream, n / 2 + 1, d_sink);
    else if (kind == 2) hipLaunchKernelGGL(k_peak_f64_4x4, dim3(blocks), dim3(256), 0, stream, n, d_sink);
    else if (kind == 0) hipLaunchKernelGGL(k_peak_f64, dim3(blocks), dim3(256), 0, stream, n, d_sink);
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
  const double flop = (kind == 3 ? 512.0 * 32 * (double)(iters / 2 + 1) : (kind == 2 ? 512.0 * 16 : 2048.0 * kChains) * (double)iters) *
                      4.0 * (double)blocks;
  *out_tflops = ms > 0.f ? flop / (ms * 1e-3) / 1e12 : 0.0;
  return GAPRO_OK;
}
