// Device memory, streams and events owned by the library (round 6): what SURVEY.md 8b describes as the context's
// "workspace arena" -- so that a gen_ps worker needs no torch to run the path.  Through round 5 torch was the plumbing of
// the host side (torch.empty x 16, current_stream x 18, four events in pipeline.py): correct, but `import torch` is 0.75 s
// of every worker's start, and the north-star job at eight GPUs is 0.64 s of work per GPU.  The
// Python face is gapro_amd/devmem.py; the torch-tensor API shims keep using torch's allocator.
//
// Arena: a caching allocator with torch's ordering rule.  A block belongs to the stream it was allocated for; freed, it
// waits in that stream's free list and is handed out again only for that stream -- work on one stream is ordered, so the
// previous owner's kernels have run before the next owner's start, and gapro_dev_free never has to wait for the device.
// (A caller that uses a block on a second stream orders that use with events before it frees, as with torch.)  Sizes are
// rounded up (512 B up to 1 MiB, 2 MiB up to 1 GiB, 64 MiB beyond) so that the per-batch buffers of a worker, whose sizes
// vary by a few per cent from batch to batch, find their blocks again; a free block at most twice the need is taken
// before anything new is allocated.
#include <cstdio>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace {

struct ArenaBlock {
  void* p;
  size_t bytes;
  hipStream_t stream;
};

struct Arena {
  std::mutex mu;
  std::unordered_map<void*, ArenaBlock> live;                    // handed out
  std::map<hipStream_t, std::multimap<size_t, void*>> free_by;  // per stream: size -> block
  size_t reserved = 0, in_use = 0;
};

size_t round_size(size_t n) {
  if (n == 0) n = 1;
  const size_t g = n <= (1u << 20) ? 512 : n <= (1ull << 30) ? (2u << 20) : (64ull << 20);
  return (n + g - 1) / g * g;
}

// Every entry point runs on the context's device whatever the calling thread's current device is: the workers' helper
// threads (workspace pre-allocation, host pools) start on device 0, and a worker of `--devices 0,..,7` owns another one.
#define GAPRO_ON_DEVICE(ctx) GAPRO_HIP_CHECK(ctx, hipSetDevice((ctx)->device))

Arena* arena_of(gapro_ctx* ctx) {
  static std::mutex mk;
  std::lock_guard<std::mutex> lk(mk);
  if (!ctx->arena) ctx->arena = new Arena();
  return (Arena*)ctx->arena;
}

}  // namespace

void gapro_arena_destroy(gapro_ctx* ctx) {
  Arena* a = (Arena*)ctx->arena;
  if (!a) return;
  for (auto& kv : a->live) (void)hipFree(kv.second.p);
  for (auto& s : a->free_by)
    for (auto& kv : s.second) (void)hipFree(kv.second);
  delete a;
  ctx->arena = nullptr;
}

extern "C" {

int gapro_dev_alloc(gapro_ctx* ctx, size_t bytes, void* stream, void** out) {
  if (!ctx || !out) return GAPRO_ERR_BAD_ARG;
  *out = nullptr;
  Arena* a = arena_of(ctx);
  const size_t need = round_size(bytes);
  {
    std::lock_guard<std::mutex> lk(a->mu);
    auto& fl = a->free_by[(hipStream_t)stream];
    auto it = fl.lower_bound(need);
    if (it != fl.end() && it->first <= 2 * need) {
      void* p = it->second;
      const size_t sz = it->first;
      fl.erase(it);
      a->live[p] = ArenaBlock{p, sz, (hipStream_t)stream};
      a->in_use += sz;
      *out = p;
      return GAPRO_OK;
    }
  }
  GAPRO_ON_DEVICE(ctx);
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, need);
  if (e != hipSuccess) {  // out of memory: give the cached blocks back and try once more
    (void)hipGetLastError();
    gapro_dev_trim(ctx);
    e = hipMalloc(&p, need);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return gapro_fail(ctx, GAPRO_ERR_OOM, "gapro_dev_alloc: hipMalloc of %zu bytes failed (%s)", need, hipGetErrorString(e));
  }
  std::lock_guard<std::mutex> lk(a->mu);
  a->live[p] = ArenaBlock{p, need, (hipStream_t)stream};
  a->reserved += need;
  a->in_use += need;
  *out = p;
  return GAPRO_OK;
}

int gapro_dev_free(gapro_ctx* ctx, void* p) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  if (!p) return GAPRO_OK;
  Arena* a = arena_of(ctx);
  std::lock_guard<std::mutex> lk(a->mu);
  auto it = a->live.find(p);
  if (it == a->live.end()) return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_dev_free: %p is not a block of this arena", p);
  const ArenaBlock b = it->second;
  a->live.erase(it);
  a->in_use -= b.bytes;
  a->free_by[b.stream].emplace(b.bytes, b.p);
  return GAPRO_OK;
}

int gapro_dev_trim(gapro_ctx* ctx) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  Arena* a = arena_of(ctx);
  std::vector<std::pair<size_t, void*>> drop;
  {
    std::lock_guard<std::mutex> lk(a->mu);
    for (auto& s : a->free_by) {
      for (auto& kv : s.second) drop.emplace_back(kv.first, kv.second);
      s.second.clear();
    }
    for (auto& d : drop) a->reserved -= d.first;
  }
  if (!drop.empty()) {
    GAPRO_ON_DEVICE(ctx);
    (void)hipDeviceSynchronize();
  }  // a cached block may still be in use by its stream's queued work
  for (auto& d : drop) (void)hipFree(d.second);
  return GAPRO_OK;
}

int gapro_dev_stats(gapro_ctx* ctx, int64_t* reserved_bytes, int64_t* in_use_bytes, int64_t* device_free_bytes,
                    int64_t* device_total_bytes) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  Arena* a = arena_of(ctx);
  {
    std::lock_guard<std::mutex> lk(a->mu);
    if (reserved_bytes) *reserved_bytes = (int64_t)a->reserved;
    if (in_use_bytes) *in_use_bytes = (int64_t)a->in_use;
  }
  if (device_free_bytes || device_total_bytes) {
    size_t fr = 0, tot = 0;
    GAPRO_ON_DEVICE(ctx);
    GAPRO_HIP_CHECK(ctx, hipMemGetInfo(&fr, &tot));
    if (device_free_bytes) *device_free_bytes = (int64_t)fr;
    if (device_total_bytes) *device_total_bytes = (int64_t)tot;
  }
  return GAPRO_OK;
}

int gapro_host_alloc(gapro_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out) return GAPRO_ERR_BAD_ARG;
  *out = nullptr;
  GAPRO_ON_DEVICE(ctx);
  GAPRO_HIP_CHECK(ctx, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return GAPRO_OK;
}

int gapro_host_free(gapro_ctx* ctx, void* p) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  if (p) GAPRO_HIP_CHECK(ctx, hipHostFree(p));
  return GAPRO_OK;
}

int gapro_stream_create(gapro_ctx* ctx, void** out) {
  if (!ctx || !out) return GAPRO_ERR_BAD_ARG;
  hipStream_t s = nullptr;
  GAPRO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GAPRO_HIP_CHECK(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *out = (void*)s;
  return GAPRO_OK;
}

int gapro_stream_destroy(gapro_ctx* ctx, void* stream) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  if (stream) GAPRO_HIP_CHECK(ctx, hipStreamDestroy((hipStream_t)stream));
  return GAPRO_OK;
}

int gapro_stream_sync(gapro_ctx* ctx, void* stream) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  GAPRO_ON_DEVICE(ctx);
  GAPRO_HIP_CHECK(ctx, hipStreamSynchronize((hipStream_t)stream));
  return GAPRO_OK;
}

int gapro_device_sync(gapro_ctx* ctx) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  GAPRO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GAPRO_HIP_CHECK(ctx, hipDeviceSynchronize());
  return GAPRO_OK;
}

int gapro_event_create(gapro_ctx* ctx, int32_t timing, void** out) {
  if (!ctx || !out) return GAPRO_ERR_BAD_ARG;
  hipEvent_t e = nullptr;
  GAPRO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GAPRO_HIP_CHECK(ctx, hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming));
  *out = (void*)e;
  return GAPRO_OK;
}

int gapro_event_destroy(gapro_ctx* ctx, void* ev) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  if (ev) GAPRO_HIP_CHECK(ctx, hipEventDestroy((hipEvent_t)ev));
  return GAPRO_OK;
}

int gapro_event_record(gapro_ctx* ctx, void* ev, void* stream) {
  if (!ctx || !ev) return GAPRO_ERR_BAD_ARG;
  GAPRO_ON_DEVICE(ctx);
  GAPRO_HIP_CHECK(ctx, hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return GAPRO_OK;
}

int gapro_stream_wait_event(gapro_ctx* ctx, void* stream, void* ev) {
  if (!ctx || !ev) return GAPRO_ERR_BAD_ARG;
  GAPRO_ON_DEVICE(ctx);
  GAPRO_HIP_CHECK(ctx, hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
  return GAPRO_OK;
}

int gapro_event_sync(gapro_ctx* ctx, void* ev) {
  if (!ctx || !ev) return GAPRO_ERR_BAD_ARG;
  GAPRO_HIP_CHECK(ctx, hipEventSynchronize((hipEvent_t)ev));
  return GAPRO_OK;
}

int gapro_event_query(gapro_ctx* ctx, void* ev) {  // 1 = complete, 0 = not yet, < 0 = error
  if (!ctx || !ev) return GAPRO_ERR_BAD_ARG;
  const hipError_t e = hipEventQuery((hipEvent_t)ev);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();
    return 0;
  }
  return gapro_fail(ctx, GAPRO_ERR_HIP, "gapro_event_query: %s", hipGetErrorString(e));
}

int gapro_event_elapsed_ms(gapro_ctx* ctx, void* ev_start, void* ev_end, float* out_ms) {
  if (!ctx || !ev_start || !ev_end || !out_ms) return GAPRO_ERR_BAD_ARG;
  GAPRO_HIP_CHECK(ctx, hipEventElapsedTime(out_ms, (hipEvent_t)ev_start, (hipEvent_t)ev_end));
  return GAPRO_OK;
}

// kind: 0 host -> device, 1 device -> host, 2 device -> device.  Asynchronous on `stream` when the host side is pinned
// (gapro_host_alloc); from pageable memory the runtime stages the data before it returns.
int gapro_memcpy_async(gapro_ctx* ctx, void* dst, const void* src, size_t bytes, int32_t kind, void* stream) {
  if (!ctx || kind < 0 || kind > 2 || (bytes && (!dst || !src))) return GAPRO_ERR_BAD_ARG;
  if (!bytes) return GAPRO_OK;
  const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  GAPRO_ON_DEVICE(ctx);
  GAPRO_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, k, (hipStream_t)stream));
  return GAPRO_OK;
}

int gapro_memset_async(gapro_ctx* ctx, void* dst, int32_t value, size_t bytes, void* stream) {
  if (!ctx || (bytes && !dst)) return GAPRO_ERR_BAD_ARG;
  if (!bytes) return GAPRO_OK;
  GAPRO_ON_DEVICE(ctx);
  GAPRO_HIP_CHECK(ctx, hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
  return GAPRO_OK;
}

}  // extern "C"
