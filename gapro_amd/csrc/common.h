// Shared internals of libgapro_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/gapro_hip.h"

struct gapro_ctx {
  int device = 0;
  int n_cu = 0;
  std::string last_error;
  gapro_scene_header* h_header_pinned = nullptr;  // pinned staging for the blocking prepare call
  hipStream_t side_stream = nullptr;              // fit launches fork their second kernel onto it
  hipEvent_t ev_join = nullptr;
};

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
