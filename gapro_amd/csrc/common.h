// Shared internals of libgapro_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/gapro_hip.h"

constexpr unsigned kTaskRing = 1024;

namespace gapro_mfma {
typedef double d4 __attribute__((ext_vector_type(4)));  // one 16 x 16 MFMA accumulator block (C layout), per lane
}

struct gapro_ctx {
  int device = 0;
  int n_cu = 0;
  std::string last_error;
  gapro_scene_header* h_header_pinned = nullptr;  // pinned staging for the blocking prepare call
  // The fit kernels run on eight library-owned streams ([0] staged / generic kernel, [1] strip kernel, [2] the
  // small-fit strip kernel, [3] the cluster kernel: large fits spread over several CUs, [4] the staged fits whose LDS
  // fits a CU twice, when the launch also has larger ones, [5..7] the wave-per-fit kernels for M_p = 48, 32, 16), so
  // that they share the GPU.
  static constexpr int kFitStreams = 8;
  hipStream_t fit_stream[kFitStreams] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_join[kFitStreams] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_gate = nullptr;  // end of the cluster kernel: the two-per-CU staged launch starts behind it
  // end of the last cluster kernel that used each half of the block-table / barrier-counter staging: the next launch that
  // reuses a half orders its upload and its counter reset behind it, whatever streams the callers use
  hipEvent_t ev_clus_half[2] = {nullptr, nullptr};
  bool clus_half_used[2] = {false, false};
  // cluster kernel: block table staging (pinned host + device) and the clusters' barrier counters, grown on demand
  void* h_cl_stage = nullptr;
  void* d_cl_stage = nullptr;
  unsigned* d_cl_ctl = nullptr;
  size_t cl_stage_bytes = 0;
  size_t cl_ctl_fits = 0;
  unsigned cl_parity = 0;
  // fit-ticket counters (gapro_svgp_fit_batch: the workgroups of a fit kernel take their fits in the order in which they
  // start): kTicketSets sets of kTicketsPerSet counters, one set per launch in turn, zeroed on the launch's stream
  static constexpr unsigned kTicketSets = 64, kTicketsPerSet = 16;
  unsigned* d_tickets = nullptr;
  unsigned ticket_seq = 0;
  // single-scene partition calls stage their one-task batch through this ring (pinned host + device mirror);
  // a slot is reused after kTaskRing further calls, long after the stream has consumed it
  gapro_scene_task* h_task_ring = nullptr;
  gapro_scene_task* d_task_ring = nullptr;
  unsigned task_pos = 0;
  struct gapro_fit_timing* armed_timing = nullptr;  // consumed by the next gapro_svgp_fit_batch
  void* arena = nullptr;  // device-memory arena (devmem.hip), created on first use
};
void gapro_arena_destroy(gapro_ctx* ctx);  // devmem.hip

// HIP events around the kernels of one fit launch, recorded on the streams the kernels run on.
struct gapro_fit_timing {
  static constexpr int kKernels = 5;  // staged, strip, small, cluster, wave-per-fit
  hipEvent_t ev[2 * kKernels] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                 nullptr};  // start / end of each
  bool used[kKernels] = {false, false, false, false, false};
  unsigned* tickets = nullptr;  // the launch's counter set (gapro_ctx::d_tickets): [5..7] = cluster kernel diagnostics
};

// svgp_fit_cluster.hip
int gapro_cluster_size(int Mp, bool all);
size_t gapro_cluster_stage_bytes(int n_fits);
// two steps: the block table is built and uploaded (and the barrier counters zeroed) on `prep_stream`, which the caller
// synchronises before it launches the kernel on `stream` -- so that the kernel is the first thing its stream has to do
// (*out_blocks: grid size; *out_members: workgroups that run a fit, the others are padding and exit at once);
int gapro_prepare_fit_cluster(hipStream_t prep_stream, int n, const int* fit_index, const int* fit_mp, const int* fit_g,
                              void* h_stage, void* d_stage, unsigned* d_ctl, int* out_blocks, int* out_members);
// d_info (may be null): [0] member workgroups started, [1] clusters whose members do NOT share an XCD, [2] clusters
int gapro_launch_fit_cluster(hipStream_t stream, int n_blocks, int feat_dim, void* d_stage, unsigned* d_ctl,
                             unsigned* d_info, const float* d_feats_spp,
                             const int* d_idx, const gapro_fit_desc* d_descs, const double* d_init_mean,
                             const gapro_fit_options& opt, double* d_workspace, float* d_probs, float* d_probs_new,
                             unsigned char* d_labels, float* d_mu, float* d_var, int* d_fit_status, double* d_fit_loss);

// svgp_fit_wave.hip: one wave per fit (M_p <= 48 at feat_dim 6)
int gapro_fit_wave_max_mp(int feat_dim);            // largest padded size it takes at this feature width (0: none)
size_t gapro_fit_wave_lds_bytes(int nb, int feat_dim);
int gapro_fit_wave_per_cu(int nb, int feat_dim);    // fits (waves) per CU of the instantiation for M_p = 16 nb
int gapro_launch_fit_wave(hipStream_t stream, int nb, int n_fits, int n_wg, unsigned* d_ticket, int feat_dim,
                          const float* d_feats_spp, const int* d_idx, const gapro_fit_desc* d_descs,
                          const double* d_init_mean, const gapro_fit_options& opt, double* d_workspace, float* d_probs,
                          float* d_probs_new, unsigned char* d_labels, float* d_mu, float* d_var, int* d_fit_status,
                          double* d_fit_loss);

// Padded size of a fit's M x M matrices: MFMA tiles are 16 wide, so M is rounded up to a multiple of 16 (everything
// scales with M_p^2 M: rounding M = 80 to 96 instead of 80 costs 1.4x the work).  The staged kernel's 32 x 32 wave tiles
// take a last row / column of 16 x 16 tiles where M_p is an odd multiple of 16 (round 3; rounds 1-2 rounded to 32
// beyond 176: 8 % more work on average at M = 200).  From 352 on its 64 x 64 tiles need multiples of 32, and so does
// the cluster kernel (M_p >= 512).
// (M_p = 240 is skipped: 256 runs the workgroup-tiled products, and M = 230 is 5 % faster padded to 256 than to 240)
// Round 4: the padded size is a function of (M, D).  An odd multiple of 16 beyond the strip kernels'
// range is only handed out where the LDS-staged kernel takes the fit (its half-tile edges are the reason the step of 16
// exists); a fit whose points do not fit beside the Cholesky panel (deep features: D = 32 beyond M_p = 208) runs on the
// cluster kernel, which needs multiples of 32, and keeps round 2's padding.  Before, M = 257 .. 336 at D = 32 padded
// to 272 / 304 / 336 fell through to the generic kernel.
constexpr int kPad16MaxMp = 336;
constexpr int kPad16AlwaysMp = 128;           // up to here every multiple of 16 runs on a strip / staged / generic route
constexpr long long kStagedMaxDynLds = 150 * 1024;  // dynamic LDS budget of the staged kernel (svgp_fit.hip kMaxDynLds)
// dynamic LDS bytes of the LDS-staged kernel (512 threads) for a PADDED size beyond kFuseMaxMp = 128:
// Zt | Xt | max(reduction slots, Cholesky block column, operand ring); svgp_fit.hip checks this against its own
// staged_lds_bytes
inline __host__ __device__ long long gapro_staged_lds_bytes_mp(int Mp, int d) {
  const long long red = 8 * 512, panel = (long long)Mp * 17 + 64 * 17, ring = 2 * 2 * 8 * (128 + 16);
  long long s = red > panel ? red : panel;
  s = s > ring ? s : ring;
  return 8LL * (2LL * d * Mp + s);
}
inline __host__ __device__ int gapro_pad_m(int m, int d) {
  if (m < 1) m = 1;
  const int p16 = (m + 15) / 16 * 16, p32 = (m + 31) / 32 * 32;
  if (p16 == p32 || p16 <= kPad16AlwaysMp) return p16;
  if (p16 > kPad16MaxMp || p16 == 240) return p32;
  return (d <= 32 && gapro_staged_lds_bytes_mp(p16, d) <= kStagedMaxDynLds) ? p16 : p32;
}

inline int gapro_fail(gapro_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->last_error = buf;
  return code;
}

#define GAPRO_HIP_CHECK(ctx, expr)                                                              \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return gapro_fail((ctx), GAPRO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                                    \
  } while (0)

#define GAPRO_LAUNCH_CHECK(ctx) GAPRO_HIP_CHECK(ctx, hipGetLastError())
