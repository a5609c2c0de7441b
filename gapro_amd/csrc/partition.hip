// Scene partition kernels: scene statistics, dense superpoint ranks, fused point-in-box
// membership + superpoint pooling, label broadcast.
//
// Replaces reference gapro/gen_ps_utils.py:312-326 (torch.unique, coordinate range) and
// :347-363 (is_within_bb_torch + three torch_scatter means + threshold) and :478-480.
// All of it is HBM-bound streaming work: one coalesced pass over the point arrays, box corners in
// LDS, integer atomics for the per-superpoint tallies (bit-reproducible, order-independent).
#include <stdlib.h>

#include "common.h"

#include <algorithm>

namespace {

constexpr int kThreads = 256;
constexpr int kMaxStatBlocks = 1024;
constexpr int kScanChunk = 2048;  // elements per block in the rank scan (256 threads x 8)

struct StatsPartial {
  double mn[3], mx[3];
  long long smin, smax;
  float fabsmax;
  int pad;
};

struct PrepareWorkspace {  // device layout, all offsets 16-byte aligned
  gapro_scene_header header;
  StatsPartial partials[kMaxStatBlocks];
};

static_assert(sizeof(gapro_scene_task) == 176, "gapro_scene_task layout is part of the ABI (ctypes mirror)");
constexpr size_t kFlagsOffset = (sizeof(PrepareWorkspace) + 255) / 256 * 256;

__device__ inline double wave_min(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline long long wave_min_ll(long long v) {
  for (int o = 32; o > 0; o >>= 1) {
    long long t = __shfl_xor(v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}
__device__ inline long long wave_max_ll(long long v) {
  for (int o = 32; o > 0; o >>= 1) {
    long long t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}
__device__ inline float wave_max_f(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- K1: per-block partial statistics ---------------------------------------------------------
// Every kernel below handles scene blockIdx.y of a batch: its pointers come from tasks[blockIdx.y].
__device__ inline PrepareWorkspace* prep_ws(const gapro_scene_task& t) { return (PrepareWorkspace*)t.prepare_ws; }
__device__ inline unsigned* prep_flags(const gapro_scene_task& t) {
  return (unsigned*)((char*)t.prepare_ws + kFlagsOffset);
}
__device__ inline unsigned* prep_bsum(const gapro_scene_task& t) {
  return (unsigned*)((char*)t.prepare_ws + kFlagsOffset + ((size_t)t.spp_range_cap * sizeof(unsigned) + 255) / 256 * 256);
}

__global__ __launch_bounds__(kThreads) void k_stats(const gapro_scene_task* __restrict__ tasks, int d) {
  const gapro_scene_task& t = tasks[blockIdx.y];
  const long long n = t.n_points;
  const double* __restrict__ coords = t.coords;
  const float* __restrict__ feats = t.feats;
  const long long* __restrict__ spp = (const long long*)t.spp;
  StatsPartial* __restrict__ partials = prep_ws(t)->partials;
  const long long tid = (long long)blockIdx.x * kThreads + threadIdx.x;
  const long long stride = (long long)gridDim.x * kThreads;
  double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  long long smin = 0x7fffffffffffffffLL, smax = -0x7fffffffffffffffLL - 1;
  float fa = 0.f;
  for (long long i = tid; i < n; i += stride) {
    const double x = coords[3 * i], y = coords[3 * i + 1], z = coords[3 * i + 2];
    mn[0] = fmin(mn[0], x); mx[0] = fmax(mx[0], x);
    mn[1] = fmin(mn[1], y); mx[1] = fmax(mx[1], y);
    mn[2] = fmin(mn[2], z); mx[2] = fmax(mx[2], z);
    // fmin / fmax drop NaNs: a non-finite coordinate is carried in the feature statistic (+inf = "scene not finite")
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) fa = INFINITY;
    const long long s = spp[i];
    smin = s < smin ? s : smin;
    smax = s > smax ? s : smax;
  }
  const long long nf = n * d;
  for (long long i = tid; i < nf; i += stride) {
    const float v = feats[i];
    fa = fmaxf(fa, isfinite(v) ? fabsf(v) : INFINITY);  // fmaxf would drop a NaN; the reference would propagate it
  }

  __shared__ StatsPartial sh[kThreads / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int k = 0; k < 3; ++k) { mn[k] = wave_min(mn[k]); mx[k] = wave_max(mx[k]); }
  smin = wave_min_ll(smin); smax = wave_max_ll(smax); fa = wave_max_f(fa);
  if (lane == 0) {
    for (int k = 0; k < 3; ++k) { sh[w].mn[k] = mn[k]; sh[w].mx[k] = mx[k]; }
    sh[w].smin = smin; sh[w].smax = smax; sh[w].fabsmax = fa;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    StatsPartial p = sh[0];
    for (int j = 1; j < kThreads / 64; ++j) {
      for (int k = 0; k < 3; ++k) { p.mn[k] = fmin(p.mn[k], sh[j].mn[k]); p.mx[k] = fmax(p.mx[k], sh[j].mx[k]); }
      p.smin = sh[j].smin < p.smin ? sh[j].smin : p.smin;
      p.smax = sh[j].smax > p.smax ? sh[j].smax : p.smax;
      p.fabsmax = fmaxf(p.fabsmax, sh[j].fabsmax);
    }
    partials[blockIdx.x] = p;
  }
}

// ---- K1b: fold partials into the scene header --------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_stats_final(const gapro_scene_task* __restrict__ tasks, int n_partials,
                                                          gapro_scene_header* __restrict__ headers) {
  const gapro_scene_task& t = tasks[blockIdx.y];
  const long long n = t.n_points, range_cap = t.spp_range_cap;
  PrepareWorkspace* ws = prep_ws(t);
  double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  long long smin = 0x7fffffffffffffffLL, smax = -0x7fffffffffffffffLL - 1;
  float fa = 0.f;
  for (int j = threadIdx.x; j < n_partials; j += kThreads) {
    const StatsPartial& q = ws->partials[j];
    for (int k = 0; k < 3; ++k) { mn[k] = fmin(mn[k], q.mn[k]); mx[k] = fmax(mx[k], q.mx[k]); }
    smin = q.smin < smin ? q.smin : smin;
    smax = q.smax > smax ? q.smax : smax;
    fa = fmaxf(fa, q.fabsmax);
  }
  __shared__ StatsPartial sh[kThreads / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int k = 0; k < 3; ++k) { mn[k] = wave_min(mn[k]); mx[k] = wave_max(mx[k]); }
  smin = wave_min_ll(smin); smax = wave_max_ll(smax); fa = wave_max_f(fa);
  if (lane == 0) {
    for (int k = 0; k < 3; ++k) { sh[w].mn[k] = mn[k]; sh[w].mx[k] = mx[k]; }
    sh[w].smin = smin; sh[w].smax = smax; sh[w].fabsmax = fa;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  StatsPartial p = sh[0];
  for (int j = 1; j < kThreads / 64; ++j) {
    for (int k = 0; k < 3; ++k) { p.mn[k] = fmin(p.mn[k], sh[j].mn[k]); p.mx[k] = fmax(p.mx[k], sh[j].mx[k]); }
    p.smin = sh[j].smin < p.smin ? sh[j].smin : p.smin;
    p.smax = sh[j].smax > p.smax ? sh[j].smax : p.smax;
    p.fabsmax = fmaxf(p.fabsmax, sh[j].fabsmax);
  }
  gapro_scene_header h;
  for (int k = 0; k < 3; ++k) { h.coord_min[k] = p.mn[k]; h.coord_max[k] = p.mx[k]; }
  h.spp_min = p.smin; h.spp_max = p.smax; h.feat_absmax = p.fabsmax;
  // fixed-point exponent: |rint(x 2^k)| * n < 2^61  (same rule as oracle/gen_ps_oracle.py:fixed_point_shift)
  int k = 0;
  if (p.fabsmax > 0.f && isfinite(p.fabsmax)) {
    int e; (void)frexpf(p.fabsmax, &e);
    const int lg = n > 1 ? 64 - __clzll((unsigned long long)(n - 1)) : 0;
    k = 61 - e - lg;
    k = k < -1000 ? -1000 : (k > 1000 ? 1000 : k);
  }
  h.fixed_shift = k;
  h.n_spps = 0;
  h.status = GAPRO_OK;
  const unsigned long long range = (unsigned long long)p.smax - (unsigned long long)p.smin;
  if (range >= (unsigned long long)range_cap) h.status = GAPRO_ERR_SPP_RANGE;
  // a NaN / Inf coordinate or feature: the reference would propagate it into every pooled mean and kernel matrix of
  // the scene; here the scene is reported and skipped (k_pool's fixed-point sum would silently turn a NaN into 0)
  if (!isfinite(p.fabsmax)) h.status = GAPRO_ERR_NOT_FINITE;
  ws->header = h;
  headers[blockIdx.y] = h;
}

// ---- K2: presence flags -------------------------------------------------------------------------
// zero the flag words of the id range actually present (after the statistics are known)
__global__ __launch_bounds__(kThreads) void k_clear_flags(const gapro_scene_task* __restrict__ tasks) {
  const gapro_scene_task& t = tasks[blockIdx.y];
  const PrepareWorkspace* ws = prep_ws(t);
  if (ws->header.status != GAPRO_OK) return;
  const long long range = ws->header.spp_max - ws->header.spp_min + 1;
  unsigned* __restrict__ flags = prep_flags(t);
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < range; i += stride) flags[i] = 0u;
}

__global__ __launch_bounds__(kThreads) void k_flags(const gapro_scene_task* __restrict__ tasks) {
  const gapro_scene_task& t = tasks[blockIdx.y];
  const long long n = t.n_points;
  const long long* __restrict__ spp = (const long long*)t.spp;
  const PrepareWorkspace* ws = prep_ws(t);
  unsigned* __restrict__ flags = prep_flags(t);
  if (ws->header.status != GAPRO_OK) return;
  const long long smin = ws->header.spp_min;
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) flags[spp[i] - smin] = 1u;
}

// ---- K3: exclusive scan of the flags -> dense ranks ---------------------------------------------
__device__ inline unsigned block_exclusive_scan(unsigned v, unsigned* total) {
  __shared__ unsigned wsum[kThreads / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned inc = v;
  for (int o = 1; o < 64; o <<= 1) {
    unsigned t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  unsigned base = 0, tot = 0;
  for (int j = 0; j < kThreads / 64; ++j) {
    if (j < w) base += wsum[j];
    tot += wsum[j];
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(kThreads) void k_scan_blocksum(const gapro_scene_task* __restrict__ tasks) {
  const gapro_scene_task& t = tasks[blockIdx.y];
  const PrepareWorkspace* ws = prep_ws(t);
  const unsigned* __restrict__ flags = prep_flags(t);
  unsigned* __restrict__ bsum = prep_bsum(t);
  if (ws->header.status != GAPRO_OK) return;
  const long long range = ws->header.spp_max - ws->header.spp_min + 1;
  const long long base = (long long)blockIdx.x * kScanChunk;
  if (base >= range) return;
  unsigned s = 0;
  for (int j = 0; j < kScanChunk / kThreads; ++j) {
    const long long i = base + (long long)threadIdx.x * (kScanChunk / kThreads) + j;
    if (i < range) s += flags[i];
  }
  unsigned tot;
  (void)block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kThreads) void k_scan_offsets(const gapro_scene_task* __restrict__ tasks,
                                                           gapro_scene_header* __restrict__ headers) {
  const gapro_scene_task& t = tasks[blockIdx.y];
  PrepareWorkspace* ws = prep_ws(t);
  unsigned* __restrict__ bsum = prep_bsum(t);
  if (ws->header.status != GAPRO_OK) return;
  const long long range = ws->header.spp_max - ws->header.spp_min + 1;
  const int nb = (int)((range + kScanChunk - 1) / kScanChunk);
  unsigned carry = 0;
  for (int b0 = 0; b0 < nb; b0 += kThreads) {
    const int i = b0 + threadIdx.x;
    const unsigned v = i < nb ? bsum[i] : 0u;
    unsigned tot;
    const unsigned ex = block_exclusive_scan(v, &tot);
    if (i < nb) bsum[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) {
    ws->header.n_spps = (int)carry;
    headers[blockIdx.y].n_spps = (int)carry;
  }
}

__global__ __launch_bounds__(kThreads) void k_scan_final(const gapro_scene_task* __restrict__ tasks) {
  const gapro_scene_task& t = tasks[blockIdx.y];
  const PrepareWorkspace* ws = prep_ws(t);
  unsigned* __restrict__ flags_to_rank = prep_flags(t);
  const unsigned* __restrict__ bsum = prep_bsum(t);
  if (ws->header.status != GAPRO_OK) return;
  const long long range = ws->header.spp_max - ws->header.spp_min + 1;
  const long long base = (long long)blockIdx.x * kScanChunk;
  if (base >= range) return;
  constexpr int per = kScanChunk / kThreads;
  unsigned v[per];
  unsigned s = 0;
  for (int j = 0; j < per; ++j) {
    const long long i = base + (long long)threadIdx.x * per + j;
    v[j] = i < range ? flags_to_rank[i] : 0u;
    s += v[j];
  }
  unsigned tot;
  unsigned ex = block_exclusive_scan(s, &tot) + bsum[blockIdx.x];
  for (int j = 0; j < per; ++j) {
    const long long i = base + (long long)threadIdx.x * per + j;
    if (i < range) flags_to_rank[i] = ex;
    ex += v[j];
  }
}

__global__ __launch_bounds__(kThreads) void k_rank_lookup(const gapro_scene_task* __restrict__ tasks) {
  const gapro_scene_task& t = tasks[blockIdx.y];
  const long long n = t.n_points;
  const long long* __restrict__ spp = (const long long*)t.spp;
  const PrepareWorkspace* ws = prep_ws(t);
  const unsigned* __restrict__ rank = prep_flags(t);
  int* __restrict__ spp_inv = t.spp_inv;
  if (ws->header.status != GAPRO_OK) return;
  const long long smin = ws->header.spp_min;
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride)
    spp_inv[i] = (int)rank[spp[i] - smin];
}

// ---- K4: fused membership + pooling ------------------------------------------------------------
// Eight lanes per point.  Lane k of a point adds features k, k+8, ... (so one atomic wave-instruction
// covers 8 points x one contiguous 8*D-byte run of the per-superpoint row instead of 64 scattered
// words: scattered atomics are ~17x slower on gfx950, MI355X_MICROARCH.md "Global float atomics"), lane
// D%8 adds the point count, and lane k tests boxes k, k+8, ...  Box corners (with the +-0.005 margin
// applied in float64, as gen_ps_utils.py:350 does) sit in LDS.
constexpr int kLanesPerPoint = 8;
// zero the integer tallies of every scene (feat_sum, occ_count, point_count)
__global__ __launch_bounds__(kThreads) void k_pool_clear(const gapro_scene_task* __restrict__ tasks, int d) {
  const gapro_scene_task& t = tasks[blockIdx.y];
  const long long S = t.n_spps, nf = S * d, no = S * t.n_boxes;
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < nf + no + S; i += stride) {
    if (i < nf) t.feat_sum[i] = 0;
    else if (i < nf + no) t.occ_count[i - nf] = 0;
    else t.point_count[i - nf - no] = 0;
  }
}

__global__ __launch_bounds__(kThreads) void k_pool(const gapro_scene_task* __restrict__ tasks, int d) {
  const gapro_scene_task& t = tasks[blockIdx.y];
  const long long n = t.n_points;
  const int nb = t.n_boxes, shift = t.fixed_shift;
  const double* __restrict__ coords = t.coords;
  const float* __restrict__ feats = t.feats;
  const int* __restrict__ spp_inv = t.spp_inv;
  const double* __restrict__ boxes = t.boxes;
  unsigned long long* __restrict__ feat_sum = (unsigned long long*)t.feat_sum;
  int* __restrict__ occ_count = t.occ_count;
  int* __restrict__ point_count = t.point_count;
  extern __shared__ double sh_box[];  // [nb][6]: lo xyz, hi xyz
  for (int j = threadIdx.x; j < nb * 6; j += kThreads) {
    const int c = j % 6;
    sh_box[j] = c < 3 ? boxes[j] - 0.005 : boxes[j] + 0.005;
  }
  __syncthreads();
  const int k = threadIdx.x & (kLanesPerPoint - 1);
  const long long ppb = kThreads / kLanesPerPoint;  // points per block per iteration
  const long long stride = (long long)gridDim.x * ppb;
  for (long long i = (long long)blockIdx.x * ppb + (threadIdx.x / kLanesPerPoint); i < n; i += stride) {
    const double x = coords[3 * i], y = coords[3 * i + 1], z = coords[3 * i + 2];
    const int r = spp_inv[i];
    if (k == (d & (kLanesPerPoint - 1))) atomicAdd(&point_count[r], 1);
    const float* f = feats + i * d;
    for (int c = k; c < d; c += kLanesPerPoint) {
      const long long q = __double2ll_rn(ldexp((double)f[c], shift));
      atomicAdd(&feat_sum[(long long)r * d + c], (unsigned long long)q);
    }
    int* occ_row = occ_count + (long long)r * nb;
    for (int b = k; b < nb; b += kLanesPerPoint) {
      const double* bx = sh_box + 6 * b;
      const bool in = (x >= bx[0]) & (y >= bx[1]) & (z >= bx[2]) & (x <= bx[3]) & (y <= bx[4]) & (z <= bx[5]);
      if (in) atomicAdd(&occ_row[b], 1);
    }
  }
}

// ---- K5: per-superpoint finalisation -----------------------------------------------------------
// ---- K4 (round 2): the same pass with the tallies privatised in LDS ------------------------------------------
// k_pool issues ~9 global atomics per point (count, D feature sums, one per containing box) and runs at the L2's
// atomic rate (48 G/s: 8.4 ms for 256 scenes, 3.5 % of the HBM roof; the counters see 2.8x the algorithmic bytes).
// Vertex order is spatially coherent, so a run of a few thousand consecutive points touches only a few dozen
// superpoints: each workgroup takes one such run (kPoolRun points), keeps an open-addressing table
// superpoint rank -> {count, D 64-bit feature sums, n_boxes occupancy counts} in LDS, adds into it with LDS atomics,
// and flushes its non-zero entries with one global atomic each at the end (~30x fewer global atomics).  A point whose
// superpoint finds no slot within kProbe probes falls back to the global atomics.  Integer sums: the result is
// bit-identical to k_pool's whatever the order.
constexpr int kPoolRun = 1024;   // points per workgroup
constexpr int kProbe = 8;

__global__ __launch_bounds__(kThreads) void k_pool_lds(const gapro_scene_task* __restrict__ tasks, int d, int log2_slots,
                                                       int nb_cap) {
  const gapro_scene_task& t = tasks[blockIdx.y];
  const long long n = t.n_points;
  const long long p0 = (long long)blockIdx.x * kPoolRun;
  if (p0 >= n) return;
  const long long p1 = p0 + kPoolRun < n ? p0 + kPoolRun : n;
  const int nb = t.n_boxes, shift = t.fixed_shift;
  const double* __restrict__ coords = t.coords;
  const float* __restrict__ feats = t.feats;
  const int* __restrict__ spp_inv = t.spp_inv;
  const double* __restrict__ boxes = t.boxes;
  unsigned long long* __restrict__ feat_sum = (unsigned long long*)t.feat_sum;
  int* __restrict__ occ_count = t.occ_count;
  int* __restrict__ point_count = t.point_count;
  extern __shared__ double sh_raw[];
  // layout: box corners [nb_cap][6] f64 | fsum [T][d] u64 | keys [T] i32 | cnt [T] u32 | occ [T][nb] u32
  const int T = 1 << log2_slots;
  double* sh_box = sh_raw;
  unsigned long long* fsum = (unsigned long long*)(sh_box + 6 * nb_cap);
  int* keys = (int*)(fsum + (size_t)T * d);
  unsigned* cnt = (unsigned*)(keys + T);
  unsigned* occ = cnt + T;
  for (int j = threadIdx.x; j < nb * 6; j += kThreads) {
    const int c = j % 6;
    sh_box[j] = c < 3 ? boxes[j] - 0.005 : boxes[j] + 0.005;
  }
  for (int j = threadIdx.x; j < T * d; j += kThreads) fsum[j] = 0ull;
  for (int j = threadIdx.x; j < T; j += kThreads) {
    keys[j] = -1;
    cnt[j] = 0u;
  }
  for (int j = threadIdx.x; j < T * nb; j += kThreads) occ[j] = 0u;
  __syncthreads();
  const int k = threadIdx.x & (kLanesPerPoint - 1);
  const int ppb = kThreads / kLanesPerPoint;
  for (long long i = p0 + (threadIdx.x / kLanesPerPoint); i < p1; i += ppb) {
    const double x = coords[3 * i], y = coords[3 * i + 1], z = coords[3 * i + 2];
    const int r = spp_inv[i];
    // slot of superpoint r: the eight lanes of the point probe together (the compare-and-swap is idempotent)
    int slot = -1;
    unsigned h = ((unsigned)r * 2654435761u) >> (32 - log2_slots);
    for (int pr = 0; pr < kProbe; ++pr) {
      const int cur = keys[h];
      if (cur == r) { slot = (int)h; break; }
      if (cur == -1) {
        const int old = atomicCAS(&keys[h], -1, r);
        if (old == -1 || old == r) { slot = (int)h; break; }
      }
      h = (h + 1) & (unsigned)(T - 1);
    }
    const float* f = feats + i * d;
    // the wave's eight points usually belong to ONE superpoint (runs of 50 .. 400 points): then their contributions
    // are summed across the eight point groups by shuffles and lanes 0 .. 7 issue one LDS atomic per feature / box
    // instead of eight to the same address
    const long long i_first = __shfl(i, 0, 64);
    const bool full_wave = i_first + 64 / kLanesPerPoint - 1 < p1;
    const bool one_spp = full_wave && __all(r == __shfl(r, 0, 64) && slot >= 0);
    if (one_spp) {
      const bool head = (threadIdx.x & 63) < kLanesPerPoint;
      if (head && k == (d & (kLanesPerPoint - 1))) atomicAdd(&cnt[slot], (unsigned)(64 / kLanesPerPoint));
      for (int c0 = 0; c0 < d; c0 += kLanesPerPoint) {
        const int c = c0 + k;
        long long q = c < d ? __double2ll_rn(ldexp((double)f[c], shift)) : 0ll;
        q += __shfl_down(q, 8, 64);
        q += __shfl_down(q, 16, 64);
        q += __shfl_down(q, 32, 64);
        if (head && c < d) atomicAdd(&fsum[(size_t)slot * d + c], (unsigned long long)q);
      }
      unsigned* occ_row = occ + (size_t)slot * nb;
      for (int b0 = 0; b0 < nb; b0 += kLanesPerPoint) {
        const int b = b0 + k;
        int in = 0;
        if (b < nb) {
          const double* bx = sh_box + 6 * b;
          in = (x >= bx[0]) & (y >= bx[1]) & (z >= bx[2]) & (x <= bx[3]) & (y <= bx[4]) & (z <= bx[5]);
        }
        in += __shfl_down(in, 8, 64);
        in += __shfl_down(in, 16, 64);
        in += __shfl_down(in, 32, 64);
        if (head && in) atomicAdd(&occ_row[b], (unsigned)in);
      }
    } else if (slot >= 0) {
      if (k == (d & (kLanesPerPoint - 1))) atomicAdd(&cnt[slot], 1u);
      for (int c = k; c < d; c += kLanesPerPoint) {
        const long long q = __double2ll_rn(ldexp((double)f[c], shift));
        atomicAdd(&fsum[(size_t)slot * d + c], (unsigned long long)q);
      }
      unsigned* occ_row = occ + (size_t)slot * nb;
      for (int b = k; b < nb; b += kLanesPerPoint) {
        const double* bx = sh_box + 6 * b;
        const bool in = (x >= bx[0]) & (y >= bx[1]) & (z >= bx[2]) & (x <= bx[3]) & (y <= bx[4]) & (z <= bx[5]);
        if (in) atomicAdd(&occ_row[b], 1u);
      }
    } else {  // table full around this hash: straight to global memory, as k_pool does
      if (k == (d & (kLanesPerPoint - 1))) atomicAdd(&point_count[r], 1);
      for (int c = k; c < d; c += kLanesPerPoint) {
        const long long q = __double2ll_rn(ldexp((double)f[c], shift));
        atomicAdd(&feat_sum[(long long)r * d + c], (unsigned long long)q);
      }
      int* occ_row = occ_count + (long long)r * nb;
      for (int b = k; b < nb; b += kLanesPerPoint) {
        const double* bx = sh_box + 6 * b;
        const bool in = (x >= bx[0]) & (y >= bx[1]) & (z >= bx[2]) & (x <= bx[3]) & (y <= bx[4]) & (z <= bx[5]);
        if (in) atomicAdd(&occ_row[b], 1);
      }
    }
  }
  __syncthreads();
  // flush the non-zero entries
  for (int j = threadIdx.x; j < T; j += kThreads) {
    const int r = keys[j];
    if (r >= 0 && cnt[j]) atomicAdd(&point_count[r], (int)cnt[j]);
  }
  for (int j = threadIdx.x; j < T * d; j += kThreads) {
    const int sl = j / d, c = j - sl * d;
    const int r = keys[sl];
    const unsigned long long v = fsum[j];
    if (r >= 0 && v) atomicAdd(&feat_sum[(long long)r * d + c], v);
  }
  for (int j = threadIdx.x; j < T * nb; j += kThreads) {
    const int sl = j / nb, b = j - sl * nb;
    const int r = keys[sl];
    const unsigned v = occ[j];
    if (r >= 0 && v) atomicAdd(&occ_count[(long long)r * nb + b], (int)v);
  }
}

__global__ __launch_bounds__(kThreads) void k_pool_finalize(const gapro_scene_task* __restrict__ tasks, int d) {
  const gapro_scene_task& t = tasks[blockIdx.y];
  const int n_spps = t.n_spps, nb = t.n_boxes, shift = t.fixed_shift;
  const float thresh = t.thresh_spp_occu;
  const long long* __restrict__ feat_sum = (const long long*)t.feat_sum;
  const int* __restrict__ occ_count = t.occ_count;
  const int* __restrict__ point_count = t.point_count;
  float* __restrict__ feats_spp = t.feats_spp;
  unsigned long long* __restrict__ occ_bits = (unsigned long long*)t.occ_bits;
  int* __restrict__ n_bbs = t.n_bbs;
  const int s = blockIdx.x * kThreads + threadIdx.x;
  if (s >= n_spps) return;
  const int pc = point_count[s] > 0 ? point_count[s] : 1;  // torch_scatter clamps the count at 1
  const float pcf = (float)pc;
  const int words = (nb + 63) / 64;
  int nbb = 0;
  for (int w = 0; w < words; ++w) {
    unsigned long long bits = 0;
    const int b_end = min(nb, (w + 1) * 64);
    for (int b = w * 64; b < b_end; ++b) {
      // scatter-mean of occupancy.float(): one IEEE float32 division of two integers (gen_ps_utils.py:359-362)
      const float mean = (float)occ_count[(long long)s * nb + b] / pcf;
      if (mean >= thresh) bits |= 1ull << (b - w * 64);
    }
    occ_bits[(long long)s * words + w] = bits;
    nbb += __popcll(bits);
  }
  n_bbs[s] = nbb;
  const double pcd = (double)pc;
  for (int k = 0; k < d; ++k)
    feats_spp[(long long)s * d + k] = (float)(ldexp((double)feat_sum[(long long)s * d + k], -shift) / pcd);
}

// ---- K6: superpoint -> point broadcast ----------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_broadcast(const gapro_scene_task* __restrict__ tasks) {
  const gapro_scene_task& t = tasks[blockIdx.y];
  const long long n = t.n_points;
  const int* __restrict__ spp_inv = t.spp_inv;
  const int* __restrict__ sem_spp = t.sem_spp;
  const int* __restrict__ inst_spp = t.inst_spp;
  const float* __restrict__ prob_spp = t.prob_spp;
  int* __restrict__ sem = t.sem;
  int* __restrict__ inst = t.inst;
  float* __restrict__ prob = t.prob;
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const int r = spp_inv[i];
    sem[i] = sem_spp[r];
    inst[i] = inst_spp[r];
    prob[i] = prob_spp[r];
  }
}

inline int grid_for(long long n, int cap = 2048) {
  long long g = (n + kThreads - 1) / kThreads;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

extern "C" {

size_t gapro_partition_prepare_workspace_bytes(int64_t n_points, int64_t spp_range_cap) {
  (void)n_points;
  if (spp_range_cap < 1) spp_range_cap = 1;
  size_t bytes = align_up(sizeof(PrepareWorkspace), 256);
  bytes += align_up((size_t)spp_range_cap * sizeof(unsigned), 256);
  bytes += align_up(((size_t)spp_range_cap / kScanChunk + 2) * sizeof(unsigned), 256);
  return bytes;
}

// ---- batched launches --------------------------------------------------------------------------
static int check_tasks(gapro_ctx* ctx, const char* who, int32_t n_scenes, const gapro_scene_task* h_tasks,
                       gapro_scene_task* d_tasks) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  if (n_scenes <= 0 || n_scenes > 65535 || !h_tasks || !d_tasks)
    return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "%s: bad argument", who);
  return GAPRO_OK;
}

int gapro_partition_prepare_batch(gapro_ctx* ctx, void* stream_, int32_t n_scenes, int32_t feat_dim,
                                  const gapro_scene_task* h_tasks, gapro_scene_task* d_tasks,
                                  gapro_scene_header* d_headers, gapro_scene_header* h_headers_pinned) {
  int rc = check_tasks(ctx, "gapro_partition_prepare_batch", n_scenes, h_tasks, d_tasks);
  if (rc != GAPRO_OK) return rc;
  if (feat_dim <= 0 || !d_headers || !h_headers_pinned)
    return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_partition_prepare_batch: bad argument");
  long long n_max = 0, cap_max = 0;
  for (int i = 0; i < n_scenes; ++i) {
    const gapro_scene_task& t = h_tasks[i];
    if (t.n_points <= 0 || !t.coords || !t.feats || !t.spp || !t.spp_inv || !t.prepare_ws || t.spp_range_cap < 1)
      return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_partition_prepare_batch: scene %d: bad argument", i);
    n_max = std::max<long long>(n_max, t.n_points);
    cap_max = std::max<long long>(cap_max, t.spp_range_cap);
  }
  hipStream_t stream = (hipStream_t)stream_;
  GAPRO_HIP_CHECK(ctx, hipMemcpyAsync(d_tasks, h_tasks, (size_t)n_scenes * sizeof(gapro_scene_task),
                                      hipMemcpyHostToDevice, stream));
  const unsigned ny = (unsigned)n_scenes;
  const int g_stats = grid_for(n_max, kMaxStatBlocks);
  const int g_pts = grid_for(n_max);
  const int g_scan = (int)((cap_max + kScanChunk - 1) / kScanChunk);
  hipLaunchKernelGGL(k_stats, dim3(g_stats, ny), dim3(kThreads), 0, stream, d_tasks, (int)feat_dim);
  hipLaunchKernelGGL(k_stats_final, dim3(1, ny), dim3(kThreads), 0, stream, d_tasks, g_stats, d_headers);
  hipLaunchKernelGGL(k_clear_flags, dim3(grid_for(cap_max, 512), ny), dim3(kThreads), 0, stream, d_tasks);
  hipLaunchKernelGGL(k_flags, dim3(g_pts, ny), dim3(kThreads), 0, stream, d_tasks);
  hipLaunchKernelGGL(k_scan_blocksum, dim3(g_scan, ny), dim3(kThreads), 0, stream, d_tasks);
  hipLaunchKernelGGL(k_scan_offsets, dim3(1, ny), dim3(kThreads), 0, stream, d_tasks, d_headers);
  hipLaunchKernelGGL(k_scan_final, dim3(g_scan, ny), dim3(kThreads), 0, stream, d_tasks);
  hipLaunchKernelGGL(k_rank_lookup, dim3(g_pts, ny), dim3(kThreads), 0, stream, d_tasks);
  GAPRO_LAUNCH_CHECK(ctx);
  GAPRO_HIP_CHECK(ctx, hipMemcpyAsync(h_headers_pinned, d_headers, (size_t)n_scenes * sizeof(gapro_scene_header),
                                      hipMemcpyDeviceToHost, stream));
  return GAPRO_OK;
}

int gapro_partition_pool_batch(gapro_ctx* ctx, void* stream_, int32_t n_scenes, int32_t feat_dim,
                               const gapro_scene_task* h_tasks, gapro_scene_task* d_tasks) {
  int rc = check_tasks(ctx, "gapro_partition_pool_batch", n_scenes, h_tasks, d_tasks);
  if (rc != GAPRO_OK) return rc;
  long long n_max = 0, clear_max = 0;
  int nb_max = 0, s_max = 0;
  for (int i = 0; i < n_scenes; ++i) {
    const gapro_scene_task& t = h_tasks[i];
    if (t.n_points <= 0 || feat_dim <= 0 || t.n_boxes <= 0 || t.n_spps <= 0 || !t.coords || !t.feats || !t.spp_inv ||
        !t.boxes || !t.feat_sum || !t.occ_count || !t.point_count || !t.feats_spp || !t.occ_bits || !t.n_bbs)
      return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_partition_pool_batch: scene %d: bad argument", i);
    n_max = std::max<long long>(n_max, t.n_points);
    nb_max = std::max(nb_max, (int)t.n_boxes);
    s_max = std::max(s_max, (int)t.n_spps);
    clear_max = std::max<long long>(clear_max, (long long)t.n_spps * (feat_dim + t.n_boxes + 1));
  }
  const size_t lds = (size_t)nb_max * 6 * sizeof(double);
  if (lds > 64 * 1024) return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_partition_pool: too many boxes (%d)", nb_max);
  hipStream_t stream = (hipStream_t)stream_;
  GAPRO_HIP_CHECK(ctx, hipMemcpyAsync(d_tasks, h_tasks, (size_t)n_scenes * sizeof(gapro_scene_task),
                                      hipMemcpyHostToDevice, stream));
  const unsigned ny = (unsigned)n_scenes;
  hipLaunchKernelGGL(k_pool_clear, dim3(grid_for(clear_max, 256), ny), dim3(kThreads), 0, stream, d_tasks, (int)feat_dim);
  // LDS-privatised tallies (k_pool_lds): a table of 16 .. 64 slots beside the box corners; GAPRO_POOL_GLOBAL_ATOMICS=1 keeps round 1's kernel (A/B runs, tests)
  static const bool global_only = getenv("GAPRO_POOL_GLOBAL_ATOMICS") != nullptr;
  const int slot_bytes = 8 + 4 * nb_max + 8 * (int)feat_dim;
  // 64 slots hold the ~10 .. 40 superpoints of a run several times over; a small table keeps 5+ workgroups per CU
  int log2_slots = 6;
  while (log2_slots > 4 && lds + ((size_t)slot_bytes << log2_slots) > 60 * 1024) --log2_slots;
  const size_t lds2 = lds + ((size_t)slot_bytes << log2_slots);
  if (!global_only && lds2 <= 60 * 1024) {
    if (lds2 > 48 * 1024)
      GAPRO_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_pool_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds2));
    const unsigned gx = (unsigned)((n_max + kPoolRun - 1) / kPoolRun);
    hipLaunchKernelGGL(k_pool_lds, dim3(gx, ny), dim3(kThreads), lds2, stream, d_tasks, (int)feat_dim, log2_slots,
                       nb_max);
  } else {
    // the whole batch shares the GPU: cap the per-scene grid so that the batch is a few waves of workgroups
    const int cap = n_scenes >= 8 ? 512 : 4096;
    hipLaunchKernelGGL(k_pool, dim3(grid_for(n_max * kLanesPerPoint, cap), ny), dim3(kThreads), lds, stream, d_tasks,
                       (int)feat_dim);
  }
  hipLaunchKernelGGL(k_pool_finalize, dim3((s_max + kThreads - 1) / kThreads, ny), dim3(kThreads), 0, stream, d_tasks,
                     (int)feat_dim);
  GAPRO_LAUNCH_CHECK(ctx);
  return GAPRO_OK;
}

int gapro_broadcast_labels_batch(gapro_ctx* ctx, void* stream_, int32_t n_scenes, const gapro_scene_task* h_tasks,
                                 gapro_scene_task* d_tasks) {
  int rc = check_tasks(ctx, "gapro_broadcast_labels_batch", n_scenes, h_tasks, d_tasks);
  if (rc != GAPRO_OK) return rc;
  long long n_max = 0;
  for (int i = 0; i < n_scenes; ++i) {
    const gapro_scene_task& t = h_tasks[i];
    if (t.n_points <= 0 || !t.spp_inv || !t.sem_spp || !t.inst_spp || !t.prob_spp || !t.sem || !t.inst || !t.prob)
      return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_broadcast_labels_batch: scene %d: bad argument", i);
    n_max = std::max<long long>(n_max, t.n_points);
  }
  hipStream_t stream = (hipStream_t)stream_;
  GAPRO_HIP_CHECK(ctx, hipMemcpyAsync(d_tasks, h_tasks, (size_t)n_scenes * sizeof(gapro_scene_task),
                                      hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(k_broadcast, dim3(grid_for(n_max, n_scenes >= 8 ? 256 : 2048), (unsigned)n_scenes), dim3(kThreads),
                     0, stream, d_tasks);
  GAPRO_LAUNCH_CHECK(ctx);
  return GAPRO_OK;
}

// ---- single-scene forms: the same kernels with a one-task batch staged through the context's ring --
static gapro_scene_task* ring_slot(gapro_ctx* ctx, gapro_scene_task** h_slot) {
  const unsigned i = ctx->task_pos++ % kTaskRing;
  *h_slot = ctx->h_task_ring + i;
  return ctx->d_task_ring + i;
}

int gapro_partition_prepare_async(gapro_ctx* ctx, void* stream_, int64_t n_points, int32_t feat_dim,
                                  const double* d_coords, const float* d_feats, const int64_t* d_spp,
                                  int64_t spp_range_cap, void* d_workspace, size_t workspace_bytes,
                                  int32_t* d_spp_inv, gapro_scene_header* h_header_pinned) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  if (n_points <= 0 || feat_dim <= 0 || !d_coords || !d_feats || !d_spp || !d_workspace || !d_spp_inv ||
      !h_header_pinned || spp_range_cap < 1)
    return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_partition_prepare: bad argument");
  if (workspace_bytes < gapro_partition_prepare_workspace_bytes(n_points, spp_range_cap))
    return gapro_fail(ctx, GAPRO_ERR_WORKSPACE, "gapro_partition_prepare: workspace too small");
  gapro_scene_task* h;
  gapro_scene_task* d = ring_slot(ctx, &h);
  *h = gapro_scene_task{};
  h->n_points = n_points; h->coords = d_coords; h->feats = d_feats; h->spp = d_spp; h->spp_inv = d_spp_inv;
  h->prepare_ws = d_workspace; h->spp_range_cap = spp_range_cap;
  // the header travels through the workspace's own header slot
  return gapro_partition_prepare_batch(ctx, stream_, 1, feat_dim, h, d, &((PrepareWorkspace*)d_workspace)->header,
                                       h_header_pinned);
}

int gapro_partition_prepare(gapro_ctx* ctx, void* stream_, int64_t n_points, int32_t feat_dim,
                            const double* d_coords, const float* d_feats, const int64_t* d_spp,
                            int64_t spp_range_cap, void* d_workspace, size_t workspace_bytes,
                            int32_t* d_spp_inv, gapro_scene_header* h_header) {
  if (!ctx || !h_header) return GAPRO_ERR_BAD_ARG;
  const int rc = gapro_partition_prepare_async(ctx, stream_, n_points, feat_dim, d_coords, d_feats, d_spp,
                                               spp_range_cap, d_workspace, workspace_bytes, d_spp_inv,
                                               ctx->h_header_pinned);
  if (rc != GAPRO_OK) return rc;
  GAPRO_HIP_CHECK(ctx, hipStreamSynchronize((hipStream_t)stream_));
  *h_header = *ctx->h_header_pinned;
  if (h_header->status != GAPRO_OK)
    return gapro_fail(ctx, h_header->status,
                      "gapro_partition_prepare: superpoint id range [%lld, %lld] exceeds capacity %lld",
                      (long long)h_header->spp_min, (long long)h_header->spp_max, (long long)spp_range_cap);
  return GAPRO_OK;
}

int gapro_partition_pool(gapro_ctx* ctx, void* stream_, int64_t n_points, int32_t feat_dim, int32_t n_boxes,
                         int32_t n_spps, int32_t fixed_shift, float thresh_spp_occu, const double* d_coords,
                         const float* d_feats, const int32_t* d_spp_inv, const double* d_boxes,
                         int64_t* d_feat_sum, int32_t* d_occ_count, int32_t* d_point_count, float* d_feats_spp,
                         uint64_t* d_occ_bits, int32_t* d_n_bbs) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  gapro_scene_task* h;
  gapro_scene_task* d = ring_slot(ctx, &h);
  *h = gapro_scene_task{};
  h->n_points = n_points; h->coords = d_coords; h->feats = d_feats; h->spp_inv = (int32_t*)d_spp_inv;
  h->boxes = d_boxes; h->n_boxes = n_boxes; h->n_spps = n_spps; h->fixed_shift = fixed_shift;
  h->thresh_spp_occu = thresh_spp_occu; h->feat_sum = d_feat_sum; h->occ_count = d_occ_count;
  h->point_count = d_point_count; h->feats_spp = d_feats_spp; h->occ_bits = d_occ_bits; h->n_bbs = d_n_bbs;
  return gapro_partition_pool_batch(ctx, stream_, 1, feat_dim, h, d);
}

int gapro_broadcast_labels(gapro_ctx* ctx, void* stream_, int64_t n_points, const int32_t* d_spp_inv,
                           const int32_t* d_sem_spp, const int32_t* d_inst_spp, const float* d_prob_spp,
                           int32_t* d_sem, int32_t* d_inst, float* d_prob) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  gapro_scene_task* h;
  gapro_scene_task* d = ring_slot(ctx, &h);
  *h = gapro_scene_task{};
  h->n_points = n_points; h->spp_inv = (int32_t*)d_spp_inv; h->sem_spp = d_sem_spp; h->inst_spp = d_inst_spp;
  h->prob_spp = d_prob_spp; h->sem = d_sem; h->inst = d_inst; h->prob = d_prob;
  return gapro_broadcast_labels_batch(ctx, stream_, 1, h, d);
}

}  // extern "C"
