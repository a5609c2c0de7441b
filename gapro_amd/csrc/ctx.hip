// Context management of libgapro_hip.so.
#include <cstdlib>
#include <cstdio>
#include "common.h"


extern "C" {

int gapro_version(void) { return GAPRO_VERSION; }

// stream priorities of the fit kernels; GAPRO_FIT_PRIO="a,b,c,d,e[,f,g,h]" (0 = greatest, 1 = middle, 2 = least) overrides
static int fit_prio(int k, int least, int greatest) {
  // (rounds 2-3: {0, 0, 2, 0, 0}, the small fits last.  With the fits taken by ticket -- claim_fit, svgp_fit.hip -- one
  // level for all is +2.4 %: small fits beside the cluster kernel load the memory system less than staged ones)
  static int lvl[gapro_ctx::kFitStreams] = {0, 0, 0, 0, 0, 0, 0, 0};
  static bool init = false;
  if (!init) {
    init = true;
    if (const char* e = getenv("GAPRO_FIT_PRIO")) {
      int v[gapro_ctx::kFitStreams] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (sscanf(e, "%d,%d,%d,%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7]) >= 5)
        for (int i = 0; i < gapro_ctx::kFitStreams; ++i) lvl[i] = v[i];
    }
  }
  const int mid = (least + greatest) / 2;
  return lvl[k] <= 0 ? greatest : lvl[k] == 1 ? mid : least;
}

int gapro_ctx_create(int device, gapro_ctx** out) {
  if (!out) return GAPRO_ERR_BAD_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return GAPRO_ERR_HIP;
  gapro_ctx* ctx = new (std::nothrow) gapro_ctx();
  if (!ctx) return GAPRO_ERR_OOM;
  ctx->device = device;
  hipDeviceProp_t prop;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
    delete ctx;
    return GAPRO_ERR_HIP;
  }
  ctx->n_cu = prop.multiProcessorCount;
  if (hipHostMalloc((void**)&ctx->h_header_pinned, sizeof(gapro_scene_header), hipHostMallocDefault) != hipSuccess) {
    delete ctx;
    return GAPRO_ERR_OOM;
  }
  if (hipHostMalloc((void**)&ctx->h_task_ring, kTaskRing * sizeof(gapro_scene_task), hipHostMallocDefault) != hipSuccess ||
      hipMalloc((void**)&ctx->d_task_ring, kTaskRing * sizeof(gapro_scene_task)) != hipSuccess) {
    gapro_ctx_destroy(ctx);
    return GAPRO_ERR_OOM;
  }
  if (hipMalloc((void**)&ctx->d_tickets, gapro_ctx::kTicketSets * gapro_ctx::kTicketsPerSet * sizeof(unsigned)) != hipSuccess) {
    gapro_ctx_destroy(ctx);
    return GAPRO_ERR_OOM;
  }
  // one priority level for all since round 4 (fit_prio)
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  for (int k = 0; k < gapro_ctx::kFitStreams; ++k)
    if (hipStreamCreateWithPriority(&ctx->fit_stream[k], hipStreamNonBlocking, fit_prio(k, prio_least, prio_greatest)) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join[k], hipEventDisableTiming) != hipSuccess) {
      gapro_ctx_destroy(ctx);
      return GAPRO_ERR_HIP;
    }
  if (hipEventCreateWithFlags(&ctx->ev_clus_half[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_clus_half[1], hipEventDisableTiming) != hipSuccess) {
    gapro_ctx_destroy(ctx);
    return GAPRO_ERR_HIP;
  }
  if (hipEventCreateWithFlags(&ctx->ev_gate, hipEventDisableTiming) != hipSuccess) {
    gapro_ctx_destroy(ctx);
    return GAPRO_ERR_HIP;
  }
  *out = ctx;
  return GAPRO_OK;
}

void gapro_ctx_destroy(gapro_ctx* ctx) {
  if (!ctx) return;
  if (ctx->h_header_pinned) (void)hipHostFree(ctx->h_header_pinned);
  if (ctx->h_task_ring) (void)hipHostFree(ctx->h_task_ring);
  if (ctx->d_task_ring) (void)hipFree(ctx->d_task_ring);
  for (int k = 0; k < gapro_ctx::kFitStreams; ++k) {
    if (ctx->ev_join[k]) (void)hipEventDestroy(ctx->ev_join[k]);
    if (ctx->fit_stream[k]) (void)hipStreamDestroy(ctx->fit_stream[k]);
  }
  if (ctx->ev_gate) (void)hipEventDestroy(ctx->ev_gate);
  for (int k = 0; k < 2; ++k)
    if (ctx->ev_clus_half[k]) (void)hipEventDestroy(ctx->ev_clus_half[k]);
  if (ctx->h_cl_stage) (void)hipHostFree(ctx->h_cl_stage);
  if (ctx->d_cl_stage) (void)hipFree(ctx->d_cl_stage);
  if (ctx->d_cl_ctl) (void)hipFree(ctx->d_cl_ctl);
  if (ctx->d_tickets) (void)hipFree(ctx->d_tickets);
  gapro_arena_destroy(ctx);
  delete ctx;
}

const char* gapro_last_error(const gapro_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

void gapro_fit_options_default(gapro_fit_options* opt) {
  if (!opt) return;
  opt->training_iter = 50;
  opt->lr = 0.1;
  opt->jitter = 1e-4;
  opt->min_variance = 1e-6;
  opt->eval_stale_chol = 0;
  opt->reserved = 0;
  opt->psd_retries = 3;
  opt->precision = GAPRO_PRECISION_F64;
  opt->psd_jitter = 1e-8;
}

}  // extern "C"
