// Label-side kernels on either end of the pseudo-label path:
//   * getInstanceInfo            reference gapro/gen_ps_utils.py:195-239  (input producer: GT boxes)
//   * get_miou_scene / cal_iou   reference gapro/eval_ps_labels.py:35-42,100-147 (quality evaluator)
//   * get_scene_sem_conf         reference gapro/eval_ps_labels.py:150-172
// All three are single streaming passes over the point arrays with small per-instance tables: the tables are
// privatised in LDS per workgroup and merged with integer atomics (min / max / add commute, so the results
// are bit-reproducible and independent of the launch shape).
#include "common.h"

#include <algorithm>

namespace {

constexpr int kThreads = 256;
constexpr int kLdsIds = 512;            // instance ids whose tallies fit the per-workgroup LDS table
constexpr int kPairLds = 8192;          // (gt, ps) pair-count bins kept in LDS
constexpr unsigned long long kKeyMax = ~0ull;

inline int grid_for(long long n, int cap = 1024) {
  long long g = (n + kThreads - 1) / kThreads;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// order-preserving map double -> uint64 (and back): integer min/max atomics then order doubles exactly
__device__ inline unsigned long long key_of(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double double_of(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

// ---------------------------------------------------------------------------------------------------
// getInstanceInfo
// ---------------------------------------------------------------------------------------------------
struct InstTable {  // per instance id (device workspace, cap entries of each)
  unsigned long long* lo;   // [cap][3] keys of the coordinate minima
  unsigned long long* hi;   // [cap][3] keys of the coordinate maxima
  unsigned long long* first;  // [cap] smallest point index
  int* rank;                // [cap] box index of a non-empty id, -1 otherwise
  int* max_id;              // [1]
  int* status;              // [1]
};

__device__ inline InstTable inst_table(void* ws, int cap) {
  InstTable t;
  unsigned long long* p = (unsigned long long*)ws;
  t.lo = p;
  t.hi = p + 3 * (size_t)cap;
  t.first = p + 6 * (size_t)cap;
  t.rank = (int*)(p + 7 * (size_t)cap);
  t.max_id = t.rank + cap;
  t.status = t.max_id + 1;
  return t;
}

__global__ __launch_bounds__(kThreads) void k_inst_init(void* ws, int cap) {
  InstTable t = inst_table(ws, cap);
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i < 3 * cap) {
    t.lo[i] = kKeyMax;
    t.hi[i] = 0ull;
  }
  if (i < cap) {
    t.first[i] = kKeyMax;
    t.rank[i] = -1;
  }
  if (i == 0) {
    *t.max_id = -1;
    *t.status = GAPRO_OK;
  }
}

// One pass over the points: per id min/max of xyz, first point index, largest id.  Ids below kLdsIds are
// tallied in LDS first (one flush per workgroup), larger ones go straight to the global table.
__global__ __launch_bounds__(kThreads) void k_inst_minmax(long long n, const double* __restrict__ coords,
                                                          const double* __restrict__ inst, void* ws, int cap) {
  InstTable t = inst_table(ws, cap);
  __shared__ unsigned long long s_lo[kLdsIds * 3], s_hi[kLdsIds * 3], s_first[kLdsIds];
  __shared__ int s_max;
  const int nl = cap < kLdsIds ? cap : kLdsIds;
  for (int j = threadIdx.x; j < nl * 3; j += kThreads) {
    s_lo[j] = kKeyMax;
    s_hi[j] = 0ull;
  }
  for (int j = threadIdx.x; j < nl; j += kThreads) s_first[j] = kKeyMax;
  if (threadIdx.x == 0) s_max = -1;
  __syncthreads();
  const long long stride = (long long)gridDim.x * kThreads;
  int my_max = -1;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const double lab = inst[i];
    if (!(lab >= 0.0)) continue;          // negative ids (and NaN) carry no instance
    if (lab >= (double)cap) {              // more ids than the caller provided room for
      atomicExch(t.status, GAPRO_ERR_BAD_ARG);
      continue;
    }
    const int id = (int)lab;               // instance_label == i_ for integer i_: non-integral labels match nothing
    if ((double)id != lab) continue;
    my_max = id > my_max ? id : my_max;
    const unsigned long long kx = key_of(coords[3 * i]), ky = key_of(coords[3 * i + 1]), kz = key_of(coords[3 * i + 2]);
    if (id < nl) {
      atomicMin(&s_lo[3 * id], kx); atomicMin(&s_lo[3 * id + 1], ky); atomicMin(&s_lo[3 * id + 2], kz);
      atomicMax(&s_hi[3 * id], kx); atomicMax(&s_hi[3 * id + 1], ky); atomicMax(&s_hi[3 * id + 2], kz);
      atomicMin(&s_first[id], (unsigned long long)i);
    } else {
      atomicMin(&t.lo[3 * id], kx); atomicMin(&t.lo[3 * id + 1], ky); atomicMin(&t.lo[3 * id + 2], kz);
      atomicMax(&t.hi[3 * id], kx); atomicMax(&t.hi[3 * id + 1], ky); atomicMax(&t.hi[3 * id + 2], kz);
      atomicMin(&t.first[id], (unsigned long long)i);
    }
  }
  atomicMax(&s_max, my_max);
  __syncthreads();
  for (int j = threadIdx.x; j < nl * 3; j += kThreads) {
    if (s_lo[j] != kKeyMax) {
      atomicMin(&t.lo[j], s_lo[j]);
      atomicMax(&t.hi[j], s_hi[j]);
    }
  }
  for (int j = threadIdx.x; j < nl; j += kThreads)
    if (s_first[j] != kKeyMax) atomicMin(&t.first[j], s_first[j]);
  if (threadIdx.x == 0 && s_max >= 0) atomicMax(t.max_id, s_max);
}

// One workgroup: box index = rank among the non-empty ids (gen_ps_utils.py:207-228), class of the first
// point (ScanNet: -2 unless -100, :236-237), volume = prod(clip(max - min, 0)) multiplied in x, y, z order.
__global__ __launch_bounds__(kThreads) void k_inst_finalize(void* ws, int cap, int scannet_shift,
                                                            const double* __restrict__ sem, double* __restrict__ box,
                                                            double* __restrict__ cls, double* __restrict__ vol,
                                                            gapro_instance_header* __restrict__ header) {
  InstTable t = inst_table(ws, cap);
  __shared__ int s_cnt[kThreads];
  const int per = (cap + kThreads - 1) / kThreads;
  const int b0 = threadIdx.x * per, b1 = min(cap, b0 + per);
  int c = 0;
  for (int id = b0; id < b1; ++id) c += t.first[id] != kKeyMax;
  s_cnt[threadIdx.x] = c;
  __syncthreads();
  int base = 0;
  for (int j = 0; j < (int)threadIdx.x; ++j) base += s_cnt[j];
  for (int id = b0; id < b1; ++id) {
    if (t.first[id] == kKeyMax) continue;
    const int r = base++;
    t.rank[id] = r;
    double lo[3], hi[3];
    for (int k = 0; k < 3; ++k) {
      lo[k] = double_of(t.lo[3 * id + k]);
      hi[k] = double_of(t.hi[3 * id + k]);
      box[6 * r + k] = lo[k];
      box[6 * r + 3 + k] = hi[k];
    }
    double s = sem[t.first[id]];
    if (scannet_shift && s != -100.0) s -= 2.0;
    cls[r] = s;
    vol[r] = fmax(hi[0] - lo[0], 0.0) * fmax(hi[1] - lo[1], 0.0) * fmax(hi[2] - lo[2], 0.0);
  }
  __syncthreads();
  if (threadIdx.x == kThreads - 1) {
    header->instance_num = *t.max_id + 1;
    header->n_boxes = base;
    header->status = *t.status;
  }
}

// corners_label[i] = (min - xyz, max - xyz) of the point's instance as float32, -100 elsewhere (:203,219-220)
__global__ __launch_bounds__(kThreads) void k_inst_corners(long long n, const double* __restrict__ coords,
                                                           const double* __restrict__ inst, const void* ws, int cap,
                                                           float* __restrict__ corners) {
  InstTable t = inst_table((void*)ws, cap);
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const double lab = inst[i];
    const int id = (lab >= 0.0 && lab < (double)cap) ? (int)lab : -1;
    const bool on = id >= 0 && (double)id == lab;
    for (int k = 0; k < 3; ++k) {
      const double x = coords[3 * i + k];
      corners[6 * i + k] = on ? (float)(double_of(t.lo[3 * id + k]) - x) : -100.0f;
      corners[6 * i + 3 + k] = on ? (float)(double_of(t.hi[3 * id + k]) - x) : -100.0f;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// get_miou_scene
// ---------------------------------------------------------------------------------------------------
struct EvalTable {
  int* pair;                    // [(cap_gt + 1) * (cap_ps + 1)] point counts of (gt + 1, ps + 1) pairs, 0 = no id
  unsigned long long* first_gt;  // [cap_gt] first point of a gt id
  unsigned long long* first_ps;  // [cap_ps]
  int* max_gt;                   // [1]
  int* max_ps;                   // [1]
  int* status;                   // [1]
};
__device__ inline EvalTable eval_table(void* ws, int cap_gt, int cap_ps) {
  EvalTable t;
  unsigned long long* p = (unsigned long long*)ws;
  t.first_gt = p;
  t.first_ps = p + cap_gt;
  t.pair = (int*)(p + cap_gt + cap_ps);
  t.max_gt = t.pair + (size_t)(cap_gt + 1) * (cap_ps + 1);
  t.max_ps = t.max_gt + 1;
  t.status = t.max_ps + 1;
  return t;
}

__global__ __launch_bounds__(kThreads) void k_eval_init(void* ws, int cap_gt, int cap_ps) {
  EvalTable t = eval_table(ws, cap_gt, cap_ps);
  const long long nbin = (long long)(cap_gt + 1) * (cap_ps + 1);
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < nbin; i += stride) t.pair[i] = 0;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < cap_gt; i += stride) t.first_gt[i] = kKeyMax;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < cap_ps; i += stride) t.first_ps[i] = kKeyMax;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *t.max_gt = -1;
    *t.max_ps = -1;
    *t.status = GAPRO_OK;
  }
}

// the intersection counts of the reference's one-hot matrix product (eval_ps_labels.py:36) are the counts of
// (gt id, pseudo id) pairs over the points: one histogram pass, LDS-privatised when the table is small
__global__ __launch_bounds__(kThreads) void k_eval_pairs(long long n, const long long* __restrict__ inst,
                                                         const long long* __restrict__ ps_inst, void* ws, int cap_gt,
                                                         int cap_ps) {
  EvalTable t = eval_table(ws, cap_gt, cap_ps);
  __shared__ int s_pair[kPairLds];
  __shared__ unsigned long long s_fg[kLdsIds], s_fp[kLdsIds];
  for (int j = threadIdx.x; j < kLdsIds; j += kThreads) {
    s_fg[j] = kKeyMax;
    s_fp[j] = kKeyMax;
  }
  const int W = cap_ps + 1;
  const long long nbin = (long long)(cap_gt + 1) * W;
  const bool lds = nbin <= kPairLds;
  if (lds)
    for (int j = threadIdx.x; j < (int)nbin; j += kThreads) s_pair[j] = 0;
  __syncthreads();
  int mg = -1, mp = -1;
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const long long g = inst[i], p = ps_inst[i];
    if (g >= cap_gt || p >= cap_ps) {
      atomicExch(t.status, GAPRO_ERR_BAD_ARG);
      continue;
    }
    const int gi = g < 0 ? 0 : (int)g + 1, pi = p < 0 ? 0 : (int)p + 1;  // torch.where(label < 0, 0, label + 1)  :118,124
    if (lds) atomicAdd(&s_pair[gi * W + pi], 1);
    else atomicAdd(&t.pair[(long long)gi * W + pi], 1);
    if (g >= 0) {
      if (g < kLdsIds) atomicMin(&s_fg[g], (unsigned long long)i);
      else atomicMin(&t.first_gt[g], (unsigned long long)i);
      mg = (int)g > mg ? (int)g : mg;
    }
    if (p >= 0) {
      if (p < kLdsIds) atomicMin(&s_fp[p], (unsigned long long)i);
      else atomicMin(&t.first_ps[p], (unsigned long long)i);
      mp = (int)p > mp ? (int)p : mp;
    }
  }
  if (mg >= 0) atomicMax(t.max_gt, mg);
  if (mp >= 0) atomicMax(t.max_ps, mp);
  __syncthreads();
  if (lds)
    for (int j = threadIdx.x; j < (int)nbin; j += kThreads)
      if (s_pair[j]) atomicAdd(&t.pair[j], s_pair[j]);
  for (int j = threadIdx.x; j < kLdsIds; j += kThreads) {
    if (j < cap_gt && s_fg[j] != kKeyMax) atomicMin(&t.first_gt[j], s_fg[j]);
    if (j < cap_ps && s_fp[j] != kKeyMax) atomicMin(&t.first_ps[j], s_fp[j]);
  }
}

// per gt id: max over the pseudo instances of the same class of inter / (|gt| + |ps| - inter + 1e-4), float32
// in the reference's operation order (eval_ps_labels.py:36-40,131-136); class = label of the first point or -1
__global__ __launch_bounds__(kThreads) void k_eval_finalize(void* ws, int cap_gt, int cap_ps,
                                                            const long long* __restrict__ sem,
                                                            const long long* __restrict__ ps_sem,
                                                            float* __restrict__ max_iou, float* __restrict__ gt_cls,
                                                            gapro_eval_header* __restrict__ header) {
  EvalTable t = eval_table(ws, cap_gt, cap_ps);
  const int n_gt = *t.max_gt + 1, n_ps = *t.max_ps + 1, W = cap_ps + 1;
  for (int g = blockIdx.x * kThreads + threadIdx.x; g < n_gt; g += gridDim.x * kThreads) {
    const float cg = t.first_gt[g] == kKeyMax ? -1.0f : (float)sem[t.first_gt[g]];
    long long gt_n = 0;
    for (int p = 0; p <= n_ps; ++p) gt_n += t.pair[(long long)(g + 1) * W + p];
    float best = 0.0f;  // no pseudo instance at all: every IoU is 0
    for (int p = 0; p < n_ps; ++p) {
      const float cp = t.first_ps[p] == kKeyMax ? -1.0f : (float)ps_sem[t.first_ps[p]];
      long long ps_n = 0;
      for (int q = 0; q <= n_gt; ++q) ps_n += t.pair[(long long)q * W + p + 1];
      const float inter = (float)t.pair[(long long)(g + 1) * W + p + 1];
      const float iou = inter / ((float)gt_n + (float)ps_n - inter + 1e-4f);
      const float v = iou * (cg == cp ? 1.0f : 0.0f);
      best = (p == 0 || v > best) ? v : best;
    }
    max_iou[g] = best;
    gt_cls[g] = cg;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    header->n_gt = n_gt;
    header->n_ps = n_ps;
    header->status = *t.status;
  }
}

// ---------------------------------------------------------------------------------------------------
// get_scene_sem_conf
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_sem_conf(long long n, const long long* __restrict__ sem,
                                                       const long long* __restrict__ ps_sem, int C,
                                                       long long* __restrict__ conf) {
  extern __shared__ int s_conf[];
  for (int j = threadIdx.x; j < C * C; j += kThreads) s_conf[j] = 0;
  __syncthreads();
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const long long s = sem[i];
    if (s == -100) continue;                                    // pos_inds  :151
    long long p = ps_sem[i];
    if (p == -100) p = s < 18 ? s + 1 : s - 1;                  // unlabeled pseudo points count as wrong  :157-161
    const long long x = p + (long long)C * s;                    // :163
    if (x >= 0 && x < (long long)C * C) atomicAdd(&s_conf[x], 1);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < C * C; j += kThreads)
    if (s_conf[j]) atomicAdd((unsigned long long*)&conf[j], (unsigned long long)s_conf[j]);
}

}  // namespace

extern "C" {

size_t gapro_instance_info_workspace_bytes(int32_t max_instances) {
  if (max_instances < 1) max_instances = 1;
  return align_up((size_t)max_instances * 7 * sizeof(unsigned long long) + ((size_t)max_instances + 2) * sizeof(int), 256);
}

int gapro_instance_info(gapro_ctx* ctx, void* stream_, int64_t n_points, const double* d_coords,
                        const double* d_instance_label, const double* d_semantic_label, int32_t max_instances,
                        int32_t scannet_class_shift, void* d_workspace, size_t workspace_bytes, double* d_box,
                        double* d_cls, double* d_volume, float* d_corners, gapro_instance_header* d_header,
                        gapro_instance_header* h_header_pinned) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  if (n_points <= 0 || !d_coords || !d_instance_label || !d_semantic_label || max_instances < 1 || !d_workspace ||
      !d_box || !d_cls || !d_volume || !d_header || !h_header_pinned)
    return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_instance_info: bad argument");
  if (workspace_bytes < gapro_instance_info_workspace_bytes(max_instances))
    return gapro_fail(ctx, GAPRO_ERR_WORKSPACE, "gapro_instance_info: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  const int cap = max_instances;
  hipLaunchKernelGGL(k_inst_init, dim3((3 * cap + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, d_workspace, cap);
  hipLaunchKernelGGL(k_inst_minmax, dim3(grid_for(n_points, 512)), dim3(kThreads), 0, stream, (long long)n_points,
                     d_coords, d_instance_label, d_workspace, cap);
  hipLaunchKernelGGL(k_inst_finalize, dim3(1), dim3(kThreads), 0, stream, d_workspace, cap, (int)scannet_class_shift,
                     d_semantic_label, d_box, d_cls, d_volume, d_header);
  if (d_corners)
    hipLaunchKernelGGL(k_inst_corners, dim3(grid_for(n_points)), dim3(kThreads), 0, stream, (long long)n_points,
                       d_coords, d_instance_label, (const void*)d_workspace, cap, d_corners);
  GAPRO_LAUNCH_CHECK(ctx);
  GAPRO_HIP_CHECK(ctx, hipMemcpyAsync(h_header_pinned, d_header, sizeof(gapro_instance_header), hipMemcpyDeviceToHost,
                                      stream));
  return GAPRO_OK;
}

size_t gapro_eval_workspace_bytes(int32_t max_gt, int32_t max_ps) {
  if (max_gt < 1) max_gt = 1;
  if (max_ps < 1) max_ps = 1;
  return align_up(((size_t)max_gt + max_ps) * sizeof(unsigned long long) +
                      ((size_t)(max_gt + 1) * (max_ps + 1) + 3) * sizeof(int), 256);
}

int gapro_eval_miou(gapro_ctx* ctx, void* stream_, int64_t n_points, const int64_t* d_semantic_label,
                    const int64_t* d_instance_label, const int64_t* d_ps_semantic_label,
                    const int64_t* d_ps_instance_label, int32_t max_gt, int32_t max_ps, void* d_workspace,
                    size_t workspace_bytes, float* d_max_iou, float* d_gt_cls, gapro_eval_header* d_header,
                    gapro_eval_header* h_header_pinned) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  if (n_points <= 0 || !d_semantic_label || !d_instance_label || !d_ps_semantic_label || !d_ps_instance_label ||
      max_gt < 1 || max_ps < 1 || !d_workspace || !d_max_iou || !d_gt_cls || !d_header || !h_header_pinned)
    return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_eval_miou: bad argument");
  if (workspace_bytes < gapro_eval_workspace_bytes(max_gt, max_ps))
    return gapro_fail(ctx, GAPRO_ERR_WORKSPACE, "gapro_eval_miou: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  const long long nbin = (long long)(max_gt + 1) * (max_ps + 1);
  hipLaunchKernelGGL(k_eval_init, dim3(grid_for(nbin, 256)), dim3(kThreads), 0, stream, d_workspace, (int)max_gt,
                     (int)max_ps);
  hipLaunchKernelGGL(k_eval_pairs, dim3(grid_for(n_points, 512)), dim3(kThreads), 0, stream, (long long)n_points,
                     (const long long*)d_instance_label, (const long long*)d_ps_instance_label, d_workspace, (int)max_gt,
                     (int)max_ps);
  hipLaunchKernelGGL(k_eval_finalize, dim3(grid_for(max_gt, 64)), dim3(kThreads), 0, stream, d_workspace, (int)max_gt,
                     (int)max_ps, (const long long*)d_semantic_label, (const long long*)d_ps_semantic_label, d_max_iou,
                     d_gt_cls, d_header);
  GAPRO_LAUNCH_CHECK(ctx);
  GAPRO_HIP_CHECK(ctx, hipMemcpyAsync(h_header_pinned, d_header, sizeof(gapro_eval_header), hipMemcpyDeviceToHost, stream));
  return GAPRO_OK;
}

int gapro_eval_sem_confusion(gapro_ctx* ctx, void* stream_, int64_t n_points, const int64_t* d_semantic_label,
                             const int64_t* d_ps_semantic_label, int32_t num_classes, int64_t* d_conf) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  if (n_points <= 0 || !d_semantic_label || !d_ps_semantic_label || num_classes < 1 || num_classes > 128 || !d_conf)
    return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_eval_sem_confusion: bad argument");
  hipStream_t stream = (hipStream_t)stream_;
  const size_t bins = (size_t)num_classes * num_classes;
  GAPRO_HIP_CHECK(ctx, hipMemsetAsync(d_conf, 0, bins * sizeof(int64_t), stream));
  hipLaunchKernelGGL(k_sem_conf, dim3(grid_for(n_points, 256)), dim3(kThreads), bins * sizeof(int), stream,
                     (long long)n_points, (const long long*)d_semantic_label, (const long long*)d_ps_semantic_label,
                     (int)num_classes, (long long*)d_conf);
  GAPRO_LAUNCH_CHECK(ctx);
  return GAPRO_OK;
}

}  // extern "C"
