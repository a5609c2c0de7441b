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

// ---------------------------------------------------------------------------------------------------
// Heuristic labelers: gen_pseudo_label (gen_ps_utils.py:485-569) and gen_pseudo_label_box2mask (:242-290)
// ---------------------------------------------------------------------------------------------------
constexpr int kLabMaxBoxes = 256;
constexpr int kLabChunk = 2048;  // points per workgroup in the multi-box rank scan (256 threads x 8)

struct LabWs {
  unsigned long long* bits;  // [N][W] occupancy bits of every point
  int* raw;                  // [N] label before the superpoint alignment: box, -1 (no box), -2 (rule "none")
  unsigned* chunk;           // [ceil(N / kLabChunk) + 1] multi-box points per chunk -> exclusive offsets
  int* label_count;          // [S][B + 1]
  int* occ_count;            // [S][B]
  int* point_count;          // [S]
  int* label_spp;            // [S]
};
__host__ __device__ inline size_t lab_align(size_t x) { return (x + 255) / 256 * 256; }
__host__ __device__ inline LabWs lab_ws(void* ws, long long n, int S, int B) {
  const int W = (B + 63) / 64;
  char* p = (char*)ws;
  LabWs t;
  t.bits = (unsigned long long*)p; p += lab_align((size_t)n * W * 8);
  t.raw = (int*)p; p += lab_align((size_t)n * 4);
  t.chunk = (unsigned*)p; p += lab_align(((size_t)(n + kLabChunk - 1) / kLabChunk + 1) * 4);
  t.label_count = (int*)p; p += lab_align((size_t)S * (B + 1) * 4);
  t.occ_count = (int*)p; p += lab_align((size_t)S * B * 4);
  t.point_count = (int*)p; p += lab_align((size_t)S * 4);
  t.label_spp = (int*)p;
  return t;
}

// box corners with the 0.005 margin applied in float32 to the float32 box (:502-504), compared in float64
__device__ inline void lab_stage_boxes(const float* __restrict__ box, int B, double* sh) {
  for (int j = threadIdx.x; j < B * 6; j += kThreads) {
    const int c = j % 6;
    sh[j] = (double)(c < 3 ? box[j] - 0.005f : box[j] + 0.005f);
  }
  __syncthreads();
}

// K1: occupancy bits, number of boxes, label for every rule except the multi-box case of "dist"
__global__ __launch_bounds__(kThreads) void k_lab_points(long long n, const double* __restrict__ coords, int B,
                                                         const float* __restrict__ box, const float* __restrict__ vol,
                                                         int rule, LabWs t) {
  extern __shared__ double sh_box[];
  lab_stage_boxes(box, B, sh_box);
  const int W = (B + 63) / 64;
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const double x = coords[3 * i], y = coords[3 * i + 1], z = coords[3 * i + 2];
    int nbb = 0, first = -1, best = -1;
    float best_v = 0.f;
    for (int w = 0; w < W; ++w) {
      unsigned long long bits = 0;
      const int b_end = min(B, (w + 1) * 64);
      for (int b = w * 64; b < b_end; ++b) {
        const double* bx = sh_box + 6 * b;
        const bool in = (x >= bx[0]) & (y >= bx[1]) & (z >= bx[2]) & (x <= bx[3]) & (y <= bx[4]) & (z <= bx[5]);
        if (in) {
          bits |= 1ull << (b - w * 64);
          if (nbb == 0) first = b;
          const float v = vol[b];
          if (nbb == 0 || v < best_v) {  // scatter_min: strict '<', the first minimum wins
            best_v = v;
            best = b;
          }
          ++nbb;
        }
      }
      t.bits[i * W + w] = bits;
    }
    int lab;
    if (nbb == 0) lab = -1;
    else if (nbb == 1) lab = first;
    else lab = rule == 0 ? best : (rule == 2 ? -2 : -3);  // -3: "dist", resolved by k_lab_dist
    t.raw[i] = lab;
  }
}

// multi-box points per chunk, then exclusive offsets (one workgroup)
__global__ __launch_bounds__(kThreads) void k_lab_chunk_count(long long n, LabWs t) {
  const long long base = (long long)blockIdx.x * kLabChunk;
  int c = 0;
  for (int j = threadIdx.x; j < kLabChunk; j += kThreads) {
    const long long i = base + j;
    if (i < n && t.raw[i] == -3) ++c;
  }
  __shared__ int sh[kThreads / 64];
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int j = 0; j < kThreads / 64; ++j) s += sh[j];
    t.chunk[blockIdx.x] = (unsigned)s;
  }
}
__global__ __launch_bounds__(kThreads) void k_lab_chunk_scan(int n_chunks, LabWs t) {
  if (threadIdx.x != 0) return;  // a few dozen chunks per scene
  unsigned run = 0;
  for (int c = 0; c < n_chunks; ++c) {
    const unsigned v = t.chunk[c];
    t.chunk[c] = run;
    run += v;
  }
  t.chunk[n_chunks] = run;
}
// "dist" (:523-531), reproducing the reference's indexing: the k-th multi-box point (k = its rank among the
// multi-box points) is measured from the coordinates of SCENE POINT k, not from its own
__global__ __launch_bounds__(kThreads) void k_lab_dist(long long n, const double* __restrict__ coords, int B,
                                                       const float* __restrict__ box, LabWs t) {
  extern __shared__ double sh_center[];  // [B][3], (lo + hi) / 2 in float32 (:497)
  for (int j = threadIdx.x; j < B * 3; j += kThreads) {
    const int b = j / 3, k = j - 3 * b;
    sh_center[j] = (double)((box[6 * b + k] + box[6 * b + 3 + k]) / 2.0f);
  }
  __shared__ int s_rank[kThreads / 64];
  __syncthreads();
  const int W = (B + 63) / 64;
  const long long base = (long long)blockIdx.x * kLabChunk;
  unsigned run = t.chunk[blockIdx.x];
  // eight sub-blocks of 256 consecutive points: rank = chunk offset + multi-box points before this one
  for (int sub = 0; sub < kLabChunk / kThreads; ++sub) {
    const long long i = base + sub * kThreads + threadIdx.x;
    const bool multi = i < n && t.raw[i] == -3;
    const unsigned long long ball = __ballot(multi);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_rank[w] = __popcll(ball);
    __syncthreads();
    unsigned before = run + __popcll(ball & ((1ull << lane) - 1ull));
    unsigned total = 0;
    for (int j = 0; j < kThreads / 64; ++j) {
      if (j < w) before += s_rank[j];
      total += s_rank[j];
    }
    if (multi) {
      const long long k = before;  // coordinates of scene point k
      const double x = coords[3 * k], y = coords[3 * k + 1], z = coords[3 * k + 2];
      int best = -1;
      double best_d = 0.0;
      for (int b = 0; b < B; ++b) {
        if (!((t.bits[i * W + (b >> 6)] >> (b & 63)) & 1ull)) continue;
        const double dx = x - sh_center[3 * b], dy = y - sh_center[3 * b + 1], dz = z - sh_center[3 * b + 2];
        const double d = dx * dx + dy * dy + dz * dz;
        if (best < 0 || d < best_d) {
          best_d = d;
          best = b;
        }
      }
      t.raw[i] = best;
    }
    run += total;
    __syncthreads();
  }
}

// per-superpoint tallies for the alignment (:539-553 / :277-283)
__global__ __launch_bounds__(kThreads) void k_lab_tally(long long n, const int* __restrict__ spp_inv, int B, int masked,
                                                        LabWs t) {
  const int W = (B + 63) / 64;
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const int s = spp_inv[i], lab = t.raw[i];
    atomicAdd(&t.label_count[(long long)s * (B + 1) + (lab >= 0 ? lab + 1 : 0)], 1);
    if (masked) {
      atomicAdd(&t.point_count[s], 1);
      for (int w = 0; w < W; ++w) {
        unsigned long long bits = t.bits[i * W + w];
        while (bits) {
          const int b = __ffsll((long long)bits) - 1;
          bits &= bits - 1;
          atomicAdd(&t.occ_count[(long long)s * B + 64 * w + b], 1);
        }
      }
    }
  }
}
// spp_align_label (:99-123): argmax of the label counts (first maximum), box rows masked by
// bb_occupancy_spp = mean occupancy >= 0.7 (float32, :545) when `masked`
__global__ __launch_bounds__(kThreads) void k_lab_argmax(int S, int B, int masked, LabWs t) {
  const int s = blockIdx.x * kThreads + threadIdx.x;
  if (s >= S) return;
  const float pc = (float)(t.point_count[s] > 0 ? t.point_count[s] : 1);
  int best = 0, best_c = t.label_count[(long long)s * (B + 1)];
  for (int l = 1; l <= B; ++l) {
    int c = t.label_count[(long long)s * (B + 1) + l];
    if (masked && !((float)t.occ_count[(long long)s * B + l - 1] / pc >= 0.7f)) c = 0;
    if (c > best_c) {
      best_c = c;
      best = l;
    }
  }
  t.label_spp[s] = best;
}
__global__ __launch_bounds__(kThreads) void k_lab_final(long long n, const int* __restrict__ spp_inv, int align,
                                                        const long long* __restrict__ cls, int instance_classes,
                                                        LabWs t, int* __restrict__ sem, int* __restrict__ inst) {
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    int lab = t.raw[i];
    if (align) {
      const int v = t.label_spp[spp_inv[i]];
      lab = v > 0 ? v - 1 : -1;  // :551-553
    }
    sem[i] = lab >= 0 ? (int)cls[lab] : (lab == -1 ? instance_classes : -100);
    inst[i] = lab >= 0 ? lab : -100;
  }
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

size_t gapro_label_heuristic_workspace_bytes(int64_t n_points, int32_t n_spps, int32_t n_boxes) {
  if (n_points < 1) n_points = 1;
  if (n_spps < 1) n_spps = 1;
  if (n_boxes < 1) n_boxes = 1;
  LabWs t = lab_ws(nullptr, n_points, n_spps, n_boxes);
  return (size_t)((char*)t.label_spp - (char*)nullptr) + lab_align((size_t)n_spps * 4);
}

int gapro_label_heuristic(gapro_ctx* ctx, void* stream_, int64_t n_points, const double* d_coords,
                          const int32_t* d_spp_inv, int32_t n_spps, int32_t n_boxes, const float* d_box,
                          const float* d_volume, const int64_t* d_cls, int32_t rule, int32_t align,
                          int32_t instance_classes, void* d_workspace, size_t workspace_bytes, int32_t* d_sem,
                          int32_t* d_inst) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  if (n_points <= 0 || !d_coords || n_boxes <= 0 || n_boxes > kLabMaxBoxes || !d_box || !d_volume || !d_cls ||
      rule < 0 || rule > 3 || !d_workspace || !d_sem || !d_inst || (align && (!d_spp_inv || n_spps <= 0)))
    return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_label_heuristic: bad argument");
  if (workspace_bytes < gapro_label_heuristic_workspace_bytes(n_points, n_spps, n_boxes))
    return gapro_fail(ctx, GAPRO_ERR_WORKSPACE, "gapro_label_heuristic: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  const int B = n_boxes, S = align ? n_spps : 1;
  LabWs t = lab_ws(d_workspace, n_points, n_spps, n_boxes);
  const int box2mask = rule == 3;  // volume rule, alignment without the occupancy mask (:277-283)
  hipLaunchKernelGGL(k_lab_points, dim3(grid_for(n_points, 2048)), dim3(kThreads), (size_t)B * 6 * sizeof(double), stream,
                     (long long)n_points, d_coords, B, d_box, d_volume, box2mask ? 0 : (int)rule, t);
  if (rule == 1) {
    const int n_chunks = (int)((n_points + kLabChunk - 1) / kLabChunk);
    hipLaunchKernelGGL(k_lab_chunk_count, dim3(n_chunks), dim3(kThreads), 0, stream, (long long)n_points, t);
    hipLaunchKernelGGL(k_lab_chunk_scan, dim3(1), dim3(kThreads), 0, stream, n_chunks, t);
    hipLaunchKernelGGL(k_lab_dist, dim3(n_chunks), dim3(kThreads), (size_t)B * 3 * sizeof(double), stream,
                       (long long)n_points, d_coords, B, d_box, t);
  }
  if (align) {
    const int masked = box2mask ? 0 : 1;
    GAPRO_HIP_CHECK(ctx, hipMemsetAsync(t.label_count, 0, (size_t)S * (B + 1) * 4, stream));
    GAPRO_HIP_CHECK(ctx, hipMemsetAsync(t.occ_count, 0, (size_t)S * B * 4, stream));
    GAPRO_HIP_CHECK(ctx, hipMemsetAsync(t.point_count, 0, (size_t)S * 4, stream));
    hipLaunchKernelGGL(k_lab_tally, dim3(grid_for(n_points, 2048)), dim3(kThreads), 0, stream, (long long)n_points,
                       d_spp_inv, B, masked, t);
    hipLaunchKernelGGL(k_lab_argmax, dim3((S + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, S, B, masked, t);
  }
  hipLaunchKernelGGL(k_lab_final, dim3(grid_for(n_points, 2048)), dim3(kThreads), 0, stream, (long long)n_points,
                     d_spp_inv, (int)(align != 0), (const long long*)d_cls, (int)instance_classes, t, d_sem, d_inst);
  GAPRO_LAUNCH_CHECK(ctx);
  return GAPRO_OK;
}

}  // extern "C"
