// Cluster variant of the batched SVGP fit: ONE GP fit spread over G workgroups (G CUs), for the large overlap
// regions of BASELINE configs[2-4] (M in the hundreds to thousands), where one fit on one CU takes 0.6 s (M = 512)
// to tens of seconds (M = 2048) and the launch waits for its few largest fits while most of the chip idles.
//
// Same algorithm and arithmetic (float64) as svgp_fit_large.hip -- reference gapro/gaussian_process_utils.py:382-445
// and the gpytorch objects it builds (:11-25); hand-derived backward of SURVEY.md Appendix B.5 -- with every matrix
// in the fit's global-memory workspace.  What changes is who does the work and how the phases are separated:
//
//   * a phase's output tiles / elements are dealt to ALL waves / threads of the cluster (cw = g * 8 + wave of G * 8);
//   * phases are separated by a CLUSTER barrier (one monotonic counter per cluster; every workgroup: each wave's
//     vmcnt(0), workgroup barrier, lane 0: agent-scope release, arrive, relaxed sc1 poll, agent-scope acquire,
//     vmcnt(0), workgroup barrier -- the form of /opt/skills/guides/cdna_hip_programming.md Guideline 16), because
//     the per-CU L1s are never refreshed by other CUs' stores and the per-XCD L2s are not coherent with each other;
//   * sums over the data index are two-stage and ordered (partials in the workspace's cluster scratch, summed in a
//     fixed order after a barrier), so a fit is bitwise reproducible for a given G, and G is a function of M only;
//   * Cholesky is right-looking with 64-wide panels: the leader workgroup factors the 64 x 64 diagonal block in LDS
//     (16 x 16 blocks one row per lane in registers, v_readlane pivots) and inverts it, the panel below is one MFMA
//     product with that inverse and the trailing update is a cluster-wide MFMA product: 3 barriers per panel;
//   * L^-1 is built from the panels' 64 x 64 inverses by pairwise merging ([A 0; B C]^-1 = [A^-1 0; -C^-1 B A^-1 C^-1]):
//     log2(M / 64) levels of two cluster-wide MFMA products instead of one serial chain per block column.
//
// Workgroups of one cluster are placed on ids b, b + 8, b + 16, ... (blocks b and b + 8 share an XCD under the
// observed round-robin dispatch, so a cluster's traffic stays in one L2); this is a speed choice only, every
// hand-off is agent-scope.  Co-residency: the grid is at most a few hundred workgroups of one per CU and the
// members of a cluster are adjacent in dispatch order, so a cluster's members become resident together as soon
// as CUs are free; the other fit kernels' workgroups never wait on anything.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "common.h"
#include "fit_layout.h"
#include "epilogue.h"

namespace {
using namespace gapro_fit;

constexpr int NT = kClThreads;  // threads per workgroup
constexpr int NW = NT / 64;     // waves per workgroup
constexpr int NB = 64;          // Cholesky panel width
constexpr int NGH = 20;

typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) double gd;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(3))) double ldsd;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) float gf;  // float32 matrices of the mixed-precision mode

__constant__ double c_gh_t[10] = {0.24534070830090124, 0.7374737285453944, 1.234076215395323,  1.7385377121165861,
                                  2.2549740020892757,  2.7888060584281305, 3.3478545673832163, 3.944764040115625,
                                  4.603682449550744,   5.387480890011233};
__constant__ double c_gh_w[10] = {0.4622436696006101,     0.28667550536283415,    0.1090172060200233,
                                  0.024810520887463643,   0.0032437733422378567,  0.00022833863601635365,
                                  7.80255647853206e-06,   1.0860693707692782e-07, 4.3993409922731747e-10,
                                  2.2293936455341447e-13};

struct Fit {
  int M, T, D, Mp;
  gd* mat[B_COUNT];
  gd* vec[V_COUNT];
  gd *X, *Z, *mZ, *vZ, *gZ, *Xt, *dinv, *dinvT, *scal, *part, *red;
  gd *Zt, *XtT;  // transposed copies [D][Mp] of Z and X
};

constexpr int kMaxPairs = 40;
static_assert((kClusterMaxMp - 64 + 127) / 128 <= kMaxPairs, "pair table of the merge inverse");
struct Shared {
  Fit f;
  int G, g;
  int wave_off;
  int same_xcd;  // every member of the cluster reported the same XCC id: barriers skip the L2 write-back
  unsigned epoch, red_par;
  gu32* count;
  double redw[NW];
  double blk[NB * (NB + 1)];  // Cholesky diagonal block (row stride 65)
  double dblk[16 * 17];
  double dinv[16 * 17];
  double tile[NW * 16 * 17];  // per-wave transpose tiles
  double c, rho_s, rho_l, s, ell, inv_l2;
  double mc, vc, mrs, vrs, mrl, vrl;  // Adam moments of the scalars: every workgroup applies the same updates
  int status, chol_bad;
  int dead;                        // a cluster barrier timed out (here or on another member): barriers fall through
  unsigned long long bar_ticks;    // longest wait at one cluster barrier, 100 MHz ticks (0 = unbounded)
  int pr_lo[kMaxPairs], pr_mid[kMaxPairs], pr_hi[kMaxPairs], pr_t0[kMaxPairs + 1];
#ifdef GAPRO_PROFILE
  unsigned long long prof[28];
  unsigned long long t_last;
  unsigned long long t_start;
#endif
};
__shared__ Shared g_sh;

// diagnostic build only: wall-clock (100 MHz ticks) per phase on the leader, barrier waits included; placed right
// behind cluster barriers, adds no synchronisation of its own
#ifdef GAPRO_PROFILE
__device__ inline void stamp(int id) {
  if (g_sh.g == 0 && threadIdx.x == 0) {
    const unsigned long long t = wall_clock64();
    g_sh.prof[id] += t - g_sh.t_last;
    g_sh.t_last = t;
  }
}
#else
__device__ inline void stamp(int) {}
#endif

// ---- small helpers ---------------------------------------------------------------------------------
__device__ inline double softplus(double x) { return log1p(exp(-fabs(x))) + fmax(x, 0.0); }
__device__ inline double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }
__device__ inline void log_ndtr_ratio(double z, double* lp, double* r) {
  const double rs2 = 0.70710678118654752440;
  if (z < 0.0) {
    const double ex = erfcx(-z * rs2);
    *lp = log(0.5 * ex) - 0.5 * z * z;
    *r = 0.79788456080286535588 / ex;
  } else {
    const double tail = 0.5 * erfc(z * rs2);
    *lp = log1p(-tail);
    *r = exp(-0.5 * z * z) * 0.39894228040143267794 / (1.0 - tail);
  }
}
__device__ inline double lane_bcast(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ inline T* uni_ptr(T* p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
__device__ inline double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline double block_sum(double v) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) g_sh.redw[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < NW; ++w) t += g_sh.redw[w];
  return t;
}

// cluster-wide thread / wave coordinates
__device__ inline int cl_tid() { return g_sh.g * NT + (int)threadIdx.x; }
__device__ inline int cl_threads() { return g_sh.G * NT; }
// wave_off: waves (of the leader) left out of the current product (look-ahead Cholesky); 0 otherwise
__device__ inline int cl_wave() { return uni(g_sh.g * NW + (int)(threadIdx.x >> 6) - g_sh.wave_off); }
__device__ inline int cl_waves() { return uni(g_sh.G * NW - g_sh.wave_off); }

// ---- cluster barrier (Guideline 16: agent-scope release / acquire around one monotonic counter) ------
// The release (buffer_wbl2 sc1: this XCD's dirty L2 lines to memory) is what makes stores visible to a CU behind
// ANOTHER XCD's L2.  When every member of the cluster has reported the same HW_REG_XCC_ID (checked once per launch
// behind a full barrier, never assumed from the block ids), the members share one L2: a store is visible to them once
// it has been acknowledged (vmcnt(0)), and the write-back -- the expensive half of the barrier, it grows with the dirty
// bytes of the whole XCD -- is skipped.  The acquire (buffer_inv sc1: drop this CU's stale L1 lines) stays.
// The wait is BOUNDED: the launch is a plain one (a cooperative launch of thousands of workgroups of four kernels is not
// an option), so that all G members of a cluster are resident together rests on in-order dispatch -- a cluster's
// members are adjacent in block order and the kernel's occupancy is one workgroup per CU -- and on nothing starving a
// member of its CU (CU masks, several processes on one GPU).  If that ever fails, a member that has polled for bar_ticks
// (default 5 s, GAPRO_CLUSTER_BARRIER_TIMEOUT_MS) raises the cluster's give-up word (counter line, third word); every
// member sees it at its next poll, marks itself dead and falls through this and all later barriers; the step loop
// ends, and the fit reports GAPRO_ERR_TIMEOUT like a failed factorisation reports GAPRO_ERR_CHOLESKY: the host maps
// it to the scene, the other fits of the launch are unaffected.
__device__ __noinline__ void cbar() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
  __syncthreads();
  if (g_sh.G > 1 && !g_sh.dead) {
    if (threadIdx.x == 0) {
      if (!g_sh.same_xcd) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the arrive must not overtake the write-back
      }
      const unsigned target = (unsigned)g_sh.G * (++g_sh.epoch);
      __hip_atomic_fetch_add(g_sh.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned polls = 0;
      unsigned long long t0 = 0;
      while (__hip_atomic_load(g_sh.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++polls & 255u) == 0) {  // every ~20 us of waiting: has somebody given up, or is it our turn to?
          const unsigned long long now = wall_clock64();
          if (t0 == 0) t0 = now;
          const bool gave_up = __hip_atomic_load(g_sh.count + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
          if (gave_up || (g_sh.bar_ticks != 0 && now - t0 > g_sh.bar_ticks)) {
            __hip_atomic_store(g_sh.count + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g_sh.dead = 1;
            if (g_sh.status == GAPRO_OK) g_sh.status = GAPRO_ERR_TIMEOUT;
            break;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv sc1: drop this CU's stale L1 lines
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // holds the barrier below until the invalidate is done
    }
    __syncthreads();
  }
}

// Once per launch, behind the first (full) barrier: do all members of the cluster sit on one XCD?
__device__ inline void detect_same_xcd() {
  if (g_sh.G == 1) return;
  unsigned xcc = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) __hip_atomic_fetch_or(g_sh.count + 1, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  cbar();  // full form: same_xcd is still 0
  if (threadIdx.x == 0) {
    const unsigned mask = __hip_atomic_load(g_sh.count + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    g_sh.same_xcd = (mask != 0 && (mask & (mask - 1)) == 0) ? 1 : 0;
  }
  __syncthreads();
}

// Ordered cluster sum of K per-thread partials; the result is the same on every thread of every workgroup.
template <int K>
__device__ __noinline__ void cl_reduce(double (&v)[K]) {
  const int G = g_sh.G;
  // the slot set is read by every thread BEFORE the workgroup barriers of block_sum; thread 0 flips it behind them
  gd* red = g_sh.f.red + (size_t)(g_sh.red_par & 1) * 16 * kClMaxG;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = block_sum(v[k]);
  if (G == 1) return;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[k * kClMaxG + g_sh.g] = v[k];
    g_sh.red_par++;  // two slot sets alternate: a workgroup can be at most one barrier ahead of the slowest reader
  }
  cbar();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double t = 0.0;
    for (int q = 0; q < G; ++q) t += red[k * kClMaxG + q];
    v[k] = t;
  }
}

// ---- TN-form MFMA product, tiles dealt to every wave of the cluster ----------------------------------
//   C[i][j] = sum_{k in [klo,khi)} P[k][i] * Q[k][j] (* qscale[k] if SCALE);  P, Q row-major in k with leading
//   dimension ld; each wave owns (16 TU) x (16 TU) output tiles; `lower_only` enumerates tiles ti >= tj.
//   kr(i0, j0, &klo, &khi): contraction range (multiples of 8); epi(i, j, tile): one 16 x 16 result in C layout.
//   Two register blocks of KS k-steps alternate with no guard in the steady-state body (see svgp_fit.hip).
//   EP / EQ: element types of P and Q in memory (gd = float64, gf = float32: the mixed-precision mode's float32
//   matrices feeding a float64 product are converted when a fragment is consumed, not when it is loaded).
// ORD: tile enumeration, heaviest contraction range first for the product's kr; the cluster's waves take the tiles of
// that order in serpentine rounds (0 .. CW-1, CW-1 .. 0, ...).  dealt round-robin in row-major order a wave whose
// tiles all lie in one tile column (row length a multiple of the wave count) carries up to 1.9x the mean work when the
// range depends on the column.  Which wave computes a tile does not change the tile.
enum { ORD_ROWMAJOR = 0, ORD_ROWS_DESC = 1, ORD_COLMAJOR = 2, ORD_SHELLS = 3 };
template <int ORD>
__device__ inline bool tile_of(int q, int cw, int CW, int ntiles, int mo_tiles, int no_tiles, bool lower_only, int* ti_,
                               int* tj_) {
  const int t = q * CW + ((q & 1) ? CW - 1 - cw : cw);
  if (t >= ntiles || cw < 0) return false;
  int ti, tj;
  if (lower_only) {
    ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    tj = t - ti * (ti + 1) / 2;
  } else if (ORD == ORD_ROWS_DESC) {
    ti = t / no_tiles;
    tj = t - ti * no_tiles;
    ti = mo_tiles - 1 - ti;
  } else if (ORD == ORD_COLMAJOR) {
    tj = t / mo_tiles;
    ti = t - tj * mo_tiles;
  } else if (ORD == ORD_SHELLS) {  // square grids: shell m holds (m, 0 .. m) and (0 .. m-1, m)
    int m = (int)sqrt((double)t);
    while ((m + 1) * (m + 1) <= t) ++m;
    while (m * m > t) --m;
    const int r = t - m * m;
    ti = r <= m ? m : r - m - 1;
    tj = r <= m ? r : m;
  } else {
    ti = t / no_tiles;
    tj = t - ti * no_tiles;
  }
  *ti_ = ti;
  *tj_ = tj;
  return true;
}
// TRIM: the contraction runs over inducing / training points: it stops at M rounded up to the register block (rows
// >= M of both operands are padding; valid outputs keep their bits).
template <int TU, bool SCALE, typename EP = gd, typename EQ = gd, int ORD = ORD_ROWMAJOR, bool TRIM = false,
          typename KRange, typename Epi>
__device__ __noinline__ void gemm_tn(int mo_tiles, int no_tiles, bool lower_only, const EP* __restrict__ P,
                                     const EQ* __restrict__ Q, int ld, const gd* __restrict__ qscale, KRange kr,
                                     Epi epi) {
  mo_tiles = uni(mo_tiles);
  no_tiles = uni(no_tiles);
  lower_only = uni((int)lower_only) != 0;
  ld = uni(ld);
  P = uni_ptr(P);
  Q = uni_ptr(Q);
  qscale = uni_ptr(qscale);
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const int cw = cl_wave(), CW = cl_waves();
  constexpr int TS = 16 * TU;
  constexpr int KS = TU >= 4 ? 1 : 2, KB = 4 * KS;  // 64 x 64 wave tiles: 128 accumulator registers, one k-step per block
  const int ntiles = lower_only ? mo_tiles * (mo_tiles + 1) / 2 : mo_tiles * no_tiles;
  const int rounds = (ntiles + CW - 1) / CW;
#pragma nounroll
  for (int q = 0; q < rounds; ++q) {
    int ti, tj;
    if (!tile_of<ORD>(q, cw, CW, ntiles, mo_tiles, no_tiles, lower_only, &ti, &tj)) continue;
    ti = uni(ti);
    tj = uni(tj);
    const int i0 = ti * TS, j0 = tj * TS;
    int klo, khi;
    kr(i0, j0, &klo, &khi);
    klo = uni(klo);
    khi = uni(khi);
    if (TRIM) {
      const int kmax = uni((g_sh.f.M + KB - 1) / KB * KB);
      khi = khi < kmax ? khi : kmax;
    }
    d4 acc[TU][TU];
#pragma unroll
    for (int u = 0; u < TU; ++u)
#pragma unroll
      for (int v = 0; v < TU; ++v) acc[u][v] = (d4){0.0, 0.0, 0.0, 0.0};
    const EP* pbase = P + (size_t)lq * ld + i0 + lr;
    const EQ* qbase = Q + (size_t)lq * ld + j0 + lr;
    typedef typename std::conditional<std::is_same<EP, gf>::value, float, double>::type VP;
    typedef typename std::conditional<std::is_same<EQ, gf>::value, float, double>::type VQ;
    VP a0[KS][TU], a1[KS][TU];
    VQ b0[KS][TU], b1[KS][TU];
    double s0[KS], s1[KS];
    auto load_block = [&](int k, VP (&a)[KS][TU], VQ (&b)[KS][TU], double (&sc)[KS]) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const EP* pr = pbase + (size_t)(k + 4 * s) * ld;
        const EQ* qr = qbase + (size_t)(k + 4 * s) * ld;
#pragma unroll
        for (int u = 0; u < TU; ++u) a[s][u] = pr[16 * u];
#pragma unroll
        for (int v = 0; v < TU; ++v) b[s][v] = qr[16 * v];
        if (SCALE) sc[s] = qscale[k + 4 * s + lq];
      }
    };
    auto mma_block = [&](VP (&a)[KS][TU], VQ (&b)[KS][TU], double (&sc)[KS]) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        double bs[TU];
#pragma unroll
        for (int v = 0; v < TU; ++v) bs[v] = SCALE ? (double)b[s][v] * sc[s] : (double)b[s][v];
#pragma unroll
        for (int u = 0; u < TU; ++u)
#pragma unroll
          for (int v = 0; v < TU; ++v)
            acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[s][u], bs[v], acc[u][v], 0, 0, 0);
      }
    };
    if (klo < khi) {
      load_block(klo, a0, b0, s0);
      int k = klo;
#pragma nounroll
      for (; k + 2 * KB < khi; k += 2 * KB) {
        load_block(k + KB, a1, b1, s1);
        mma_block(a0, b0, s0);
        load_block(k + 2 * KB, a0, b0, s0);
        mma_block(a1, b1, s1);
      }
      if (k + KB < khi) {
        load_block(k + KB, a1, b1, s1);
        mma_block(a0, b0, s0);
        mma_block(a1, b1, s1);
      } else {
        mma_block(a0, b0, s0);
      }
    }
    run_epilogue<TU * TU>(epi, [&](int b, int* i, int* j) { *i = i0 + 16 * (b / TU); *j = j0 + 16 * (b % TU); },
                          [&](int b) -> const d4& { return acc[b / TU][b % TU]; });
  }
}

// The same product on float32 operands with v_mfma_f32_16x16x4_f32 (the mixed-precision mode's L_S^T A, dA and dL_S
// products): float32 accumulators in the standard C layout, register r -> row 4 (l >> 4) + r, column l & 15.
template <int TU, bool SCALE, int ORD = ORD_ROWMAJOR, bool TRIM = false, typename KRange, typename Epi>
__device__ __noinline__ void gemm_tn_f32(int mo_tiles, int no_tiles, bool lower_only, const gf* __restrict__ P,
                                         const gf* __restrict__ Q, int ld, const gd* __restrict__ qscale, KRange kr,
                                         Epi epi) {
  mo_tiles = uni(mo_tiles);
  no_tiles = uni(no_tiles);
  lower_only = uni((int)lower_only) != 0;
  ld = uni(ld);
  P = uni_ptr(P);
  Q = uni_ptr(Q);
  qscale = uni_ptr(qscale);
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const int cw = cl_wave(), CW = cl_waves();
  constexpr int TS = 16 * TU;
  constexpr int KS = TU >= 4 ? 1 : 2, KB = 4 * KS;  // 64 x 64 wave tiles: 128 accumulator registers, one k-step per block
  const int ntiles = lower_only ? mo_tiles * (mo_tiles + 1) / 2 : mo_tiles * no_tiles;
  const int rounds = (ntiles + CW - 1) / CW;
#pragma nounroll
  for (int q = 0; q < rounds; ++q) {
    int ti, tj;
    if (!tile_of<ORD>(q, cw, CW, ntiles, mo_tiles, no_tiles, lower_only, &ti, &tj)) continue;
    ti = uni(ti);
    tj = uni(tj);
    const int i0 = ti * TS, j0 = tj * TS;
    int klo, khi;
    kr(i0, j0, &klo, &khi);
    klo = uni(klo);
    khi = uni(khi);
    if (TRIM) {
      const int kmax = uni((g_sh.f.M + KB - 1) / KB * KB);
      khi = khi < kmax ? khi : kmax;
    }
    f4 acc[TU][TU];
#pragma unroll
    for (int u = 0; u < TU; ++u)
#pragma unroll
      for (int v = 0; v < TU; ++v) acc[u][v] = (f4){0.f, 0.f, 0.f, 0.f};
    const gf* pbase = P + (size_t)lq * ld + i0 + lr;
    const gf* qbase = Q + (size_t)lq * ld + j0 + lr;
    typedef float VP;
    typedef float VQ;
    VP a0[KS][TU], a1[KS][TU];
    VQ b0[KS][TU], b1[KS][TU];
    float s0[KS], s1[KS];
    auto load_block = [&](int k, VP (&a)[KS][TU], VQ (&b)[KS][TU], float (&sc)[KS]) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const gf* pr = pbase + (size_t)(k + 4 * s) * ld;
        const gf* qr = qbase + (size_t)(k + 4 * s) * ld;
#pragma unroll
        for (int u = 0; u < TU; ++u) a[s][u] = pr[16 * u];
#pragma unroll
        for (int v = 0; v < TU; ++v) b[s][v] = qr[16 * v];
        if (SCALE) sc[s] = (float)qscale[k + 4 * s + lq];
      }
    };
    auto mma_block = [&](VP (&a)[KS][TU], VQ (&b)[KS][TU], float (&sc)[KS]) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        float bs[TU];
#pragma unroll
        for (int v = 0; v < TU; ++v) bs[v] = SCALE ? b[s][v] * sc[s] : b[s][v];
#pragma unroll
        for (int u = 0; u < TU; ++u)
#pragma unroll
          for (int v = 0; v < TU; ++v)
            acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][u], bs[v], acc[u][v], 0, 0, 0);
      }
    };
    if (klo < khi) {
      load_block(klo, a0, b0, s0);
      int k = klo;
#pragma nounroll
      for (; k + 2 * KB < khi; k += 2 * KB) {
        load_block(k + KB, a1, b1, s1);
        mma_block(a0, b0, s0);
        load_block(k + 2 * KB, a0, b0, s0);
        mma_block(a1, b1, s1);
      }
      if (k + KB < khi) {
        load_block(k + KB, a1, b1, s1);
        mma_block(a0, b0, s0);
        mma_block(a1, b1, s1);
      } else {
        mma_block(a0, b0, s0);
      }
    }
#pragma unroll
    for (int u = 0; u < TU; ++u)
#pragma unroll
      for (int v = 0; v < TU; ++v) epi(i0 + 16 * u, j0 + 16 * v, acc[u][v]);
  }
}

// Store a 16x16 accumulator tile (C layout) row-major at Cm[i0.., j0..] and/or transposed at CT[j0.., i0..]; the
// transposed copy goes through the wave's LDS tile so that its global stores are contiguous rows too.  EO: element
// type of the destination (float32 destinations round here).
template <typename EO = gd>
__device__ inline void store_tile(const d4& v, EO* __restrict__ Cm, EO* __restrict__ CT, int ld, int i0, int j0) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  if (Cm) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Cm[(size_t)(i0 + lq + 4 * r) * ld + j0 + lr] = v[r];
  }
  if (CT) {
    ldsd* tile = (ldsd*)g_sh.tile + (threadIdx.x >> 6) * 16 * 17;
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[(lq + 4 * r) * 17 + lr] = v[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) CT[(size_t)(j0 + lq + 4 * r) * ld + i0 + lr] = tile[lr * 17 + lq + 4 * r];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
  }
}
// the same for a float32 MFMA result (register r -> row 4 (l >> 4) + r)
__device__ inline void store_tile_f32(const f4& v, gf* __restrict__ Cm, gf* __restrict__ CT, int ld, int i0, int j0) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  if (Cm) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Cm[(size_t)(i0 + 4 * lq + r) * ld + j0 + lr] = v[r];
  }
  if (CT) {
    ldsd* tile = (ldsd*)g_sh.tile + (threadIdx.x >> 6) * 16 * 17;
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[(4 * lq + r) * 17 + lr] = (double)v[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) CT[(size_t)(j0 + lq + 4 * r) * ld + i0 + lr] = (float)tile[lr * 17 + lq + 4 * r];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
  }
}

__device__ inline double sqdist(const gd* a, const gd* b, int D) {
  double s = 0.0;
  for (int d = 0; d < D; ++d) {
    const double t = a[d] - b[d];
    s += t * t;
  }
  return s;
}

// ---- Cholesky ---------------------------------------------------------------------------------------
// 16 x 16 diagonal block at blk[16 kb.., 16 kb..] (LDS, row stride RS): L_kk in place (upper zeroed) and
// Dinv = L_kk^-1 in g_sh.dinv; one wave, rows in registers, pivots by v_readlane (as svgp_fit.hip).
__device__ __noinline__ void diag16(ldsd* blk, int RS, int kb) {
  const int lane = threadIdx.x & 63, r = lane & 15;
  ldsd* base = blk + (16 * kb) * RS + 16 * kb;
  ldsd* dinv = (ldsd*)g_sh.dinv;
  double rdiag[16];
  double a[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) a[c] = base[r * RS + c];
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    double d = lane_bcast(a[j], j);
    if (!(d > 0.0)) {
      bad = true;
      d = 1e-30;
    }
    const double rs = rsqrt(d);
    rdiag[j] = rs;
    const double lj = (r == j) ? d * rs : a[j] * rs;
    a[j] = lj;
#pragma unroll
    for (int c = j + 1; c < 16; ++c) a[c] -= lj * lane_bcast(lj, c);
  }
  if (bad && lane == 0) g_sh.chol_bad = 1;
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 16; ++c) base[r * RS + c] = (c <= r) ? a[c] : 0.0;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  double x[16], b[16];
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) b[rr] = (rr == r) ? 1.0 : 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    x[q] = (q >= r) ? b[q] * rdiag[q] : 0.0;
#pragma unroll
    for (int rr = q + 1; rr < 16; ++rr) b[rr] = fma(-base[rr * RS + q], x[q], b[rr]);
  }
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 16; ++c) dinv[c * 17 + r] = x[c];  // Dinv[c][r]: lane r holds column r
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
}

// Leader workgroup: factor the w x w (w = 16 nbk <= 64) diagonal block of panel column c0 in LDS, write L_pp, L_pp^T
// and the Dinv blocks, then W = L_pp^-1 (and W^T) into LI / U through the 16-wide block recursion.
__device__ __noinline__ void chol_diag_block(int c0, int w) {
  const Fit& f = g_sh.f;
  const int Mp = f.Mp, nbk = w / 16;
  constexpr int RS = NB + 1;
  ldsd* blk = (ldsd*)g_sh.blk;
  ldsd* dinv = (ldsd*)g_sh.dinv;
  gd* L = f.mat[B_L];
  gd* LT = f.mat[B_LT];
  gd* LI = f.mat[B_LI];
  gd* U = f.mat[B_U];
  const int wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  for (int idx = threadIdx.x; idx < w * w; idx += NT) {
    const int i = idx / w, j = idx - i * w;
    blk[i * RS + j] = L[(size_t)(c0 + i) * Mp + c0 + j];
  }
  __syncthreads();
  for (int kb = 0; kb < nbk; ++kb) {
    if (wave == 0) {
      diag16(blk, RS, kb);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = lane + 64 * e, rr = idx >> 4, cc = idx & 15;
        const int blkid = c0 / 16 + kb;
        f.dinv[(size_t)blkid * 256 + idx] = dinv[rr * 17 + cc];
        f.dinvT[(size_t)blkid * 256 + idx] = dinv[cc * 17 + rr];
      }
    }
    __syncthreads();
    // rows below inside the block: P[i][c] = sum_{q <= c} S[i][16 kb + q] Dinv[c][q]
    const int rows = w - 16 * (kb + 1);
    for (int idx = threadIdx.x; idx < rows * 16; idx += NT) {
      const int i = 16 * (kb + 1) + (idx >> 4), c = idx & 15;
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc += (q <= c) ? blk[i * RS + 16 * kb + q] * dinv[c * 17 + q] : 0.0;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();  // the 16 lanes of a row have read S before any overwrites it
      blk[i * RS + 16 * kb + c] = acc;
    }
    __syncthreads();
    // trailing update inside the block (lower part)
    for (int idx = threadIdx.x; idx < rows * rows; idx += NT) {
      const int i = 16 * (kb + 1) + idx / rows, j = 16 * (kb + 1) + idx % rows;
      if (j <= i) {
        double acc = blk[i * RS + j];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc = fma(-blk[i * RS + 16 * kb + c], blk[j * RS + 16 * kb + c], acc);
        blk[i * RS + j] = acc;
      }
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < w * w; idx += NT) {
    const int i = idx / w, j = idx - i * w;
    L[(size_t)(c0 + i) * Mp + c0 + j] = (j <= i) ? blk[i * RS + j] : 0.0;
    LT[(size_t)(c0 + i) * Mp + c0 + j] = (i <= j) ? blk[j * RS + i] : 0.0;
  }
  __syncthreads();
  // W = L_pp^-1: block column k per wave;  W_kk = Dinv_k,  W_ik = -Dinv_i sum_{j=k}^{i-1} L_ij W_jk
  const int b0 = c0 / 16;
  if (wave < nbk) {
    const int k = wave;
    d4 blkv[NB / 16];
    d4 dk;
#pragma unroll
    for (int r = 0; r < 4; ++r) dk[r] = f.dinv[(size_t)(b0 + k) * 256 + (lq + 4 * r) * 16 + lr];
    store_tile(dk, LI, U, Mp, c0 + 16 * k, c0 + 16 * k);
    blkv[0] = dk;
#pragma unroll
    for (int ii = 1; ii < NB / 16; ++ii) {
      const int i = k + ii;
      if (i < nbk) {
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int jj = 0; jj < ii; ++jj) {
          const ldsd* pa = blk + (16 * i + lr) * RS + 16 * (k + jj) + lq;  // L[16 i + lr][16 (k + jj) + 4 s + lq]
#pragma unroll
          for (int st = 0; st < 4; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * st], blkv[jj][st], acc, 0, 0, 0);
        }
        d4 out = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const double a = -f.dinvT[(size_t)(b0 + i) * 256 + (4 * st + lq) * 16 + lr];  // -Dinv_i[lr][4 st + lq]
          out = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[st], out, 0, 0, 0);
        }
        store_tile(out, LI, U, Mp, c0 + 16 * i, c0 + 16 * k);
        blkv[ii] = out;
      }
    }
  }
  // blocks of W above the diagonal are zero in LI (lower) / below the diagonal in U
  for (int idx = threadIdx.x; idx < w * w; idx += NT) {
    const int i = idx / w, j = idx - i * w;
    if ((j >> 4) > (i >> 4)) {
      LI[(size_t)(c0 + i) * Mp + c0 + j] = 0.0;
      U[(size_t)(c0 + j) * Mp + c0 + i] = 0.0;
    }
  }
}

// squared distance between column i of At[D][Mp] and column j of Bt[D][Mp]; with i wave-uniform the At loads are
// broadcasts and the Bt loads are coalesced over the lanes' consecutive j
__device__ inline double sqdist_t(const gd* At, int i, const gd* Bt, int j, int D, int Mp) {
  double s = 0.0;
  for (int d = 0; d < D; ++d) {
    const double t = At[(size_t)d * Mp + i] - Bt[(size_t)d * Mp + j];
    s += t * t;
  }
  return s;
}

// Kzz + jitter I (lower incl. diagonal; identity on the padded tail; zero above) into B_L.  A wave owns row i (its
// point: broadcast loads) and walks the columns 64 at a time (coalesced loads of the transposed points).
// MX (mixed precision, the reference's split): the kernel matrix is evaluated in float32 arithmetic, as gpytorch
// evaluates it, and handed to the float64 factorisation (K.double()).
// Squared distances between RB consecutive row points i0 .. i0 + RB - 1 of At[D][lda] (i0 wave-uniform: their
// coordinates are the same address in every lane) and column point j of Bt[D][ldb] (coalesced over the lanes).  A
// column coordinate is loaded once for RB pairs and the d-loop is unrolled, so several loads are in flight: with one
// row per pass and a rolled loop the kernel evaluations waited for one load at a time and, at D = 32, re-streamed
// 2 D M_p coordinates per row through a 32 KiB L1.
constexpr int kRowBlock = 4;
// the row block's coordinates, [d][kRowBlock], in the wave's transpose tile (LDS): read back as broadcasts
__device__ inline const ldsd* stage_rows(const gd* __restrict__ At, int lda, int i0, int D) {
  ldsd* zl = (ldsd*)g_sh.tile + (threadIdx.x >> 6) * (16 * 17);
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();  // the previous block's reads are done
  for (int e = lane; e < kRowBlock * D; e += 64) zl[e] = At[(size_t)(e / kRowBlock) * lda + i0 + (e % kRowBlock)];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  return zl;
}
template <typename R>
__device__ inline void sqdist_rows(const ldsd* zl, const gd* __restrict__ Bt, int ldb, int j, int D,
                                   R (&d2)[kRowBlock]) {
#pragma unroll
  for (int r = 0; r < kRowBlock; ++r) d2[r] = (R)0;
#pragma unroll 8
  for (int d = 0; d < D; ++d) {
    const R q = (R)Bt[(size_t)d * ldb + j];
#pragma unroll
    for (int r = 0; r < kRowBlock; ++r) {
      const R t = (R)zl[kRowBlock * d + r] - q;
      d2[r] += t * t;
    }
  }
}
// the same against two column sets at once (kernel gradients: Z_j and X_j)
template <typename R>
__device__ inline void sqdist_rows2(const ldsd* zl, const gd* __restrict__ Bt, const gd* __restrict__ Ct, int ldb, int j,
                                    int D, R (&d2)[kRowBlock], R (&d2x)[kRowBlock]) {
#pragma unroll
  for (int r = 0; r < kRowBlock; ++r) d2[r] = d2x[r] = (R)0;
#pragma unroll 8
  for (int d = 0; d < D; ++d) {
    const R q = (R)Bt[(size_t)d * ldb + j], qx = (R)Ct[(size_t)d * ldb + j];
#pragma unroll
    for (int r = 0; r < kRowBlock; ++r) {
      const R z = (R)zl[kRowBlock * d + r];
      const R t = z - q, tx = z - qx;
      d2[r] += t * t;
      d2x[r] += tx * tx;
    }
  }
}

template <bool MX>
__device__ __noinline__ void build_kzz(double s, double inv_l2, double jitter, double extra) {
  typedef typename std::conditional<MX, float, double>::type R;
  const Fit& f = g_sh.f;
  const int Mp = f.Mp, M = f.M, D = f.D;
  gd* L = f.mat[B_L];
  const gd* Zt = f.Zt;
  const int cw = cl_wave(), CW = cl_waves(), lane = threadIdx.x & 63;
  const float sf = (float)s, hf = -0.5f * (float)inv_l2, jf = (float)jitter;
  for (int i0 = kRowBlock * cw; i0 < Mp; i0 += kRowBlock * CW) {  // M_p is a multiple of 32
    const ldsd* zl = stage_rows(Zt, Mp, i0, D);
    for (int j = lane; j < Mp; j += 64) {
      double v[kRowBlock];
#pragma unroll
      for (int r = 0; r < kRowBlock; ++r) v[r] = (i0 + r >= M && i0 + r == j) ? 1.0 : 0.0;  // identity on the padded tail
      if (i0 < M && j < i0 + kRowBlock && j < M) {  // some row of the block has j <= i < M
        R d2[kRowBlock];
        sqdist_rows<R>(zl, Zt, Mp, j, D, d2);
#pragma unroll
        for (int r = 0; r < kRowBlock; ++r) {
          const int i = i0 + r;
          if (i < M && j <= i) {
            if (MX) {
              float kv = sf * expf(hf * (float)d2[r]);
              if (i == j) kv += jf;  // the variational jitter is added to the float32 matrix ...
              v[r] = (double)kv;
              if (i == j) v[r] += extra;  // ... psd_safe_cholesky's retry jitter to its float64 copy
            } else {
              v[r] = s * exp(-0.5 * inv_l2 * (double)d2[r]);
              if (i == j) v[r] += jitter + extra;
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < kRowBlock; ++r) L[(size_t)(i0 + r) * Mp + j] = v[r];
    }
  }
}

// Right-looking blocked Cholesky of B_L (lower) -> L, L^T, Dinv blocks, and the panels' inverses in LI / U.
__device__ __noinline__ void cholesky_cluster() {
  const Fit& f = g_sh.f;
  const int Mp = f.Mp;
  gd* L = f.mat[B_L];
  gd* LT = f.mat[B_LT];
  const gd* U = f.mat[B_U];
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  // One-panel look-ahead: the leader-only diagonal block of panel p + 1 (~36 us, 7 .. 15 % of a step when everybody
  // waits for it) runs BESIDE the bulk of panel p's trailing update.  Per panel:
  //   (b)  panel solve p                                                 | barrier
  //   (c1) trailing update of the next panel's 64 columns, all members   | barrier
  //   (a') leader: diagonal block p + 1   ||   others: (c2) the rest of the trailing update   | barrier
  // Same three barriers per panel, same MFMAs in the same order per tile (bit-identical factor).
  if (g_sh.g == 0) chol_diag_block(0, Mp < NB ? Mp : NB);
  cbar();
  stamp(1);
  for (int c0 = 0; c0 < Mp; c0 += NB) {
    const int w = (Mp - c0) < NB ? (Mp - c0) : NB, c1 = c0 + w;
    if (c1 >= Mp) break;
    const int w1 = (Mp - c1) < NB ? (Mp - c1) : NB, c2 = c1 + w1;  // the next panel
    // (b) panel below: L[i][c0 + c] = sum_q S[i][c0 + q] W[c][q],  W^T = U (rows q, contiguous in c)
    {
      const int cw = cl_wave(), CW = cl_waves();
      const int row_tiles = (Mp - c1) / 16, wb = w / 16;
      for (int t = cw; t < row_tiles * wb; t += CW) {
        const int rt = t / wb, cb = t - rt * wb;
        const int i0 = c1 + 16 * rt;
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
        // A operand straight from row-major S: A[i = lr][k = 4 st + lq]
        const gd* pa = L + (size_t)(i0 + lr) * Mp + c0 + lq;
        const gd* pb = U + (size_t)(c0 + lq) * Mp + c0 + 16 * cb + lr;
        for (int q = 0; q <= cb; ++q) {  // W[c][q] = 0 for q-block > c-block
#pragma unroll
          for (int st = 0; st < 4; ++st)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[16 * q + 4 * st], pb[(size_t)(16 * q + 4 * st) * Mp], acc, 0, 0, 0);
        }
        // results go to L^T now and to L after every tile of this row block has read S (barrier below)
        store_tile(acc, (gd*)nullptr, LT, Mp, i0, c0 + 16 * cb);
      }
    }
    cbar();
    stamp(2);
    // L rows of the panel from L^T (the in-place overwrite of S must wait for all readers of the row block)
    {
      const long long n = (long long)(Mp - c1) * w;
      for (long long idx = cl_tid(); idx < n; idx += cl_threads()) {
        const int i = c1 + (int)(idx / w), c = c0 + (int)(idx % w);
        L[(size_t)i * Mp + c] = LT[(size_t)c * Mp + i];
      }
    }
    // trailing update S[i][j] -= sum_{q in panel} L[i][q] L[j][q] (reads L^T only), 32 x 32 tiles
    auto upd = [=](int i, int j, const d4& v, gd* S) {
      const int ln = threadIdx.x & 63, c = ln & 15, g4 = ln >> 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gd* p = S + (size_t)(i + g4 + 4 * r) * Mp + j + c;
        *p = *p - v[r];
      }
    };
    // (c1) the next panel's block column: rows >= c1, columns [c1, c2)
    {
      gd* S = L + (size_t)c1 * Mp + c1;
      gemm_tn<2, false>((Mp - c1) / 32, w1 / 32, false, LT + c1, LT + c1, Mp, nullptr,
                        [=](int, int, int* lo, int* hi) { *lo = c0; *hi = c1; },
                        [=](int i, int j, const d4& v) { upd(i, j, v, S); });
    }
    cbar();
    stamp(3);
    // (a') | (c2): the leader factors the next diagonal block while the other members update columns >= c2
    const bool alone = g_sh.G == 1;
    if (!alone) {  // the leader's waves take no tiles of the product below (set and cleared between barriers)
      if (threadIdx.x == 0) g_sh.wave_off = NW;
      __syncthreads();
    }
    if (g_sh.g == 0) chol_diag_block(c1, w1);
    if (c2 < Mp && (alone || g_sh.g != 0)) {
      gd* S = L + (size_t)c2 * Mp + c2;
      const int mt = (Mp - c2) / 32;
      gemm_tn<2, false>(mt, mt, true, LT + c2, LT + c2, Mp, nullptr,
                        [=](int, int, int* lo, int* hi) { *lo = c0; *hi = c1; },
                        [=](int i, int j, const d4& v) { upd(i, j, v, S); });
    }
    cbar();
    if (!alone) {
      if (threadIdx.x == 0) g_sh.wave_off = 0;
      __syncthreads();
    }
    stamp(1);
  }
}

// LI = L^-1 and U = LI^T from the panels' inverses by pairwise merging; KX serves as the temporary.
__device__ __noinline__ void tri_inverse_cluster() {
  const Fit& f = g_sh.f;
  const int Mp = f.Mp;
  const gd* LT = f.mat[B_LT];
  gd* LI = f.mat[B_LI];
  gd* U = f.mat[B_U];
  gd* Tm = f.mat[B_KX];
  Shared& sh = g_sh;
  for (int s = NB; s < Mp; s *= 2) {
    // pairs [lo, mid) + [mid, hi) with lo a multiple of 2 s
    int np = 0;
    for (int lo = 0; lo + s < Mp; lo += 2 * s) ++np;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t0 = 0, k = 0;
      for (int lo = 0; lo + s < Mp; lo += 2 * s, ++k) {
        const int mid = lo + s, hi = (lo + 2 * s) < Mp ? (lo + 2 * s) : Mp;
        sh.pr_lo[k] = lo; sh.pr_mid[k] = mid; sh.pr_hi[k] = hi; sh.pr_t0[k] = t0;
        t0 += ((hi - mid) / 32) * (s / 32);
      }
      sh.pr_t0[k] = t0;
    }
    __syncthreads();
    const int ntot = sh.pr_t0[np];
    const int cw = cl_wave(), CW = cl_waves();
    const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    // two passes over the same tile enumeration: T = B A^-1, then X = -C^-1 T
    for (int pass = 0; pass < 2; ++pass) {
      for (int t = cw; t < ntot; t += CW) {
        int k = 0;
        while (sh.pr_t0[k + 1] <= t) ++k;
        const int lo = uni(sh.pr_lo[k]), mid = uni(sh.pr_mid[k]);
        const int tl = t - uni(sh.pr_t0[k]);
        const int nct = s / 32;
        // pass 0's range shrinks with the tile column and the wave count is usually a multiple of the row length:
        // the columns are rotated by the row, so that a wave does not own one column (1.3 .. 1.9x the mean work)
        const int ti = tl / nct, tj = pass == 0 ? (tl - ti * nct + ti) % nct : tl - ti * nct;
        const int i0 = mid + 32 * ti, j0 = lo + 32 * tj;  // output tile rows in [mid, hi), columns in [lo, mid)
        int klo, khi;
        const gd *Pp, *Qp;
        if (pass == 0) {  // T[i][j] = sum_k B[i][k] Ainv[k][j], k in [lo, mid): P[k][i] = LT[k][i], Q[k][j] = LI[k][j] (k >= j)
          Pp = LT; Qp = LI; klo = j0; khi = mid;
        } else {          // X[i][j] = -sum_k Cinv[i][k] T[k][j], k in [mid, hi): P[k][i] = U[k][i] (k <= i), Q = T
          Pp = U; Qp = Tm; klo = mid; khi = i0 + 32;
        }
        d4 acc[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int v = 0; v < 2; ++v) acc[u][v] = (d4){0.0, 0.0, 0.0, 0.0};
        const gd* pb = Pp + (size_t)lq * Mp + i0 + lr;
        const gd* qb = Qp + (size_t)lq * Mp + j0 + lr;
#pragma unroll 2
        for (int kk = klo; kk < khi; kk += 4) {
          const double a0 = pb[(size_t)kk * Mp], a1 = pb[(size_t)kk * Mp + 16];
          const double b0 = qb[(size_t)kk * Mp], b1 = qb[(size_t)kk * Mp + 16];
          acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            if (pass == 0) {
              store_tile(acc[u][v], Tm, (gd*)nullptr, Mp, i0 + 16 * u, j0 + 16 * v);
            } else {
              d4 neg;
#pragma unroll
              for (int r = 0; r < 4; ++r) neg[r] = -acc[u][v][r];
              store_tile(neg, LI, U, Mp, i0 + 16 * u, j0 + 16 * v);
            }
          }
      }
      cbar();
    }
  }
}

// KX[k][n] = s exp(-|Z_k - P_n|^2 / (2 l^2)) for k < M, n < ncols, zero elsewhere; Pt = the points transposed [D][ldp].
// MX: float32 arithmetic, float32 storage.
template <bool MX>
__device__ __noinline__ void build_kx(const gd* Pt, int ldp, int ncols, double s, double inv_l2) {
  typedef typename std::conditional<MX, float, double>::type R;
  const Fit& f = g_sh.f;
  const int Mp = f.Mp, M = f.M, D = f.D;
  gd* KX = f.mat[B_KX];
  gf* KXf = (gf*)f.mat[B_KX];
  const gd* Zt = f.Zt;
  const int cw = cl_wave(), CW = cl_waves(), lane = threadIdx.x & 63;
  const float sf = (float)s, hf = -0.5f * (float)inv_l2;
  for (int k0 = kRowBlock * cw; k0 < Mp; k0 += kRowBlock * CW) {
    const ldsd* zl = stage_rows(Zt, Mp, k0, D);
    for (int c = lane; c < Mp; c += 64) {
      R d2[kRowBlock];
      const bool in = k0 < M && c < ncols;
      if (in) sqdist_rows<R>(zl, Pt, ldp, c, D, d2);
#pragma unroll
      for (int r = 0; r < kRowBlock; ++r) {
        const bool ok = in && k0 + r < M;
        if (MX)
          KXf[(size_t)(k0 + r) * Mp + c] = ok ? sf * expf(hf * (float)d2[r]) : 0.f;
        else
          KX[(size_t)(k0 + r) * Mp + c] = ok ? s * exp(-0.5 * inv_l2 * (double)d2[r]) : 0.0;
      }
    }
  }
}

// Stage 1 of the ordered column sums: plane p of the scratch gets, for row group rg and column c,
//   sum over rows r = rg, rg + RG, ... of term(r, c).  RG = max(1, CT / Mp); a column's partials are summed in
//   row-group order by col_final.  The scratch plane holds RG * Mp <= max(CT, Mp) doubles.
template <typename Term>
__device__ inline void col_partials(int plane, int Mp, Term term) {
  const int CT = cl_threads(), ct = cl_tid();
  const int RG = CT / Mp > 0 ? CT / Mp : 1;
  gd* out = g_sh.f.part + (size_t)plane * cluster_plane_doubles(Mp);
  if (CT >= Mp) {
    const int rg = ct / Mp, c = ct - rg * Mp;
    if (rg < RG) {
      double acc = 0.0;
      for (int r = rg; r < Mp; r += RG) acc += term(r, c);
      out[(size_t)rg * Mp + c] = acc;
    }
  } else {
    for (int c = ct; c < Mp; c += CT) {
      double acc = 0.0;
      for (int r = 0; r < Mp; ++r) acc += term(r, c);
      out[c] = acc;
    }
  }
}
__device__ inline double col_final(int plane, int Mp, int c) {
  const int CT = cl_threads();
  const int RG = CT / Mp > 0 ? CT / Mp : 1;
  const gd* in = g_sh.f.part + (size_t)plane * cluster_plane_doubles(Mp);
  double acc = 0.0;
  for (int rg = 0; rg < RG; ++rg) acc += in[(size_t)rg * Mp + c];
  return acc;
}

// A = LI KX (+ A^T), B^T = A^T LS (+ B) over the first `ncols` columns, then the column partials of mu and var.
// MX: A = (L^-1 K_ZX in float64) rounded to float32 (gpytorch: interp_term ... .to(float32)); B in float32 on
// v_mfma_f32 from the float32 A and L_S; the column sums read the float32 matrices and accumulate in float64.
template <bool MX, int TU>
__device__ __noinline__ void forward_products(int ncols) {
  const Fit& f = g_sh.f;
  const int Mp = f.Mp;
  constexpr int TS = 16 * TU;
  const int mt = Mp / TS, nt = (ncols + TS - 1) / TS;
  const gd* vm = f.vec[V_M];
  if (MX) {
    gf* A = (gf*)f.mat[B_A];
    gf* AT = (gf*)f.mat[B_AT];
    gf* BM = (gf*)f.mat[B_BM];
    gf* BMT = (gf*)f.mat[B_BMT];
    gemm_tn<TU, false, gd, gf, ORD_ROWS_DESC, true>(mt, nt, false, f.mat[B_U], (const gf*)f.mat[B_KX], Mp, nullptr,
                              [=](int i0, int, int* lo, int* hi) { *lo = 0; *hi = i0 + TS; },
                              [=](int i, int n, const d4& v) { store_tile<gf>(v, A, AT, Mp, i, n); });
    cbar();
    stamp(20);
    gemm_tn_f32<TU, false, ORD_COLMAJOR, true>(nt, mt, false, A, (const gf*)f.mat[B_LS], Mp, nullptr,
                          [=](int, int j0, int* lo, int* hi) { *lo = j0; *hi = Mp; },
                          [=](int n, int j, const f4& v) { store_tile_f32(v, BMT, BM, Mp, n, j); });
    cbar();
    stamp(21);
    col_partials(0, Mp, [=](int r, int c) { return vm[r] * (double)A[(size_t)r * Mp + c]; });
    col_partials(1, Mp, [=](int r, int c) {
      const double a = A[(size_t)r * Mp + c], b = BM[(size_t)r * Mp + c];
      return b * b - a * a;
    });
  } else {
    gd* A = f.mat[B_A];
    gd* AT = f.mat[B_AT];
    gd* BM = f.mat[B_BM];
    gd* BMT = f.mat[B_BMT];
    gemm_tn<TU, false, gd, gd, ORD_ROWS_DESC, true>(mt, nt, false, f.mat[B_U], f.mat[B_KX], Mp, nullptr,
                      [=](int i0, int, int* lo, int* hi) { *lo = 0; *hi = i0 + TS; },
                      [=](int i, int n, const d4& v) { store_tile(v, A, AT, Mp, i, n); });
    cbar();
    stamp(20);
    gemm_tn<TU, false, gd, gd, ORD_COLMAJOR, true>(nt, mt, false, A, f.mat[B_LS], Mp, nullptr,
                      [=](int, int j0, int* lo, int* hi) { *lo = j0; *hi = Mp; },
                      [=](int n, int j, const d4& v) { store_tile(v, BMT, BM, Mp, n, j); });
    cbar();
    stamp(21);
    col_partials(0, Mp, [=](int r, int c) { return vm[r] * A[(size_t)r * Mp + c]; });
    col_partials(1, Mp, [=](int r, int c) {
      const double a = A[(size_t)r * Mp + c], b = BM[(size_t)r * Mp + c];
      return b * b - a * a;
    });
  }
  cbar();
}

// Kernel gradients of one Adam step; see the call site.  A function of its own: its per-row accumulators must not
// live in the registers of the step loop.  MX: float32 arithmetic (the kernel's backward runs in float32 in the
// reference: G_Kzz and G_KX arrive through .float()), KX is a float32 matrix, sums of the float32 terms in float64.
template <int DMAX, bool MX>
__device__ __noinline__ void kernel_grads(const gd* __restrict__ G, const gd* __restrict__ GT,
                                          const gd* __restrict__ GKX, const gd* __restrict__ KX, double s,
                                          double inv_l2, double (&ks)[2]) {
  typedef typename std::conditional<MX, float, double>::type R;
  const Fit& f = g_sh.f;
  const int M = f.M, Mp = f.Mp, D = f.D;
  const int cw = cl_wave(), CW = cl_waves(), lane = threadIdx.x & 63;
  const gd* Zt = f.Zt;
  const gd* Xc = f.XtT;
  const gf* KXf = (const gf*)KX;
  const R sr = (R)s, hr = (R)(-0.5) * (R)inv_l2;
  double k0 = 0.0, k1 = 0.0;
  // kRowBlock rows per pass: a column's points are loaded once for the block, and the block's 4 x kRowBlock matrix
  // elements are requested together (one row per pass waited ~1.7 us per 64 pairs for four dependent-free loads)
  for (int i0 = kRowBlock * cw; i0 < M; i0 += kRowBlock * CW) {
    R acc[kRowBlock][DMAX], wsum[kRowBlock];
#pragma unroll
    for (int r = 0; r < kRowBlock; ++r) {
      wsum[r] = (R)0;
#pragma unroll
      for (int d = 0; d < DMAX; ++d) acc[r][d] = (R)0;
    }
    const ldsd* zl = stage_rows(Zt, Mp, i0, D);
    for (int j = lane; j < M; j += 64) {
      R gsym[kRowBlock], gk[kRowBlock], kx[kRowBlock];
#pragma unroll
      for (int r = 0; r < kRowBlock; ++r) {
        const size_t o = (size_t)(i0 + r < M ? i0 + r : i0) * Mp + j;
        gsym[r] = (R)(0.5 * (G[o] + GT[o]));
        gk[r] = (R)GKX[o];
        kx[r] = MX ? (R)KXf[o] : (R)KX[o];
      }
      R zj[DMAX], xj[DMAX];
#pragma unroll
      for (int d = 0; d < DMAX; ++d) {
        zj[d] = d < D ? (R)Zt[(size_t)d * Mp + j] : (R)0;
        xj[d] = d < D ? (R)Xc[(size_t)d * Mp + j] : (R)0;
      }
#pragma unroll
      for (int r = 0; r < kRowBlock; ++r) {
        if (i0 + r < M) {
          R d2 = (R)0, d2x = (R)0;
#pragma unroll
          for (int d = 0; d < DMAX; ++d) {
            const R zi = d < D ? (R)zl[kRowBlock * d + r] : (R)0;
            const R a = zi - zj[d], bx = zi - xj[d];
            d2 += a * a;
            d2x += bx * bx;
          }
          const R e = MX ? (R)expf((float)(hr * d2)) : (R)exp((double)(hr * d2));
          const R w = gsym[r] * sr * e, wx = gk[r] * kx[r];
          k0 += (double)(gsym[r] * e + gk[r] * kx[r] / sr);
          k1 += (double)(w * d2 + wx * d2x);
          // sum_j w_j (Z_i - P_j) = Z_i sum_j w_j - sum_j w_j P_j : only the weighted point sums are accumulated
          wsum[r] += (R)2 * w + wx;
#pragma unroll
          for (int d = 0; d < DMAX; ++d) acc[r][d] += (R)2 * w * zj[d] + wx * xj[d];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kRowBlock; ++r) {
      const double wsum_d = wave_sum((double)wsum[r]);
#pragma unroll
      for (int d = 0; d < DMAX; ++d) {
        if (d < D && i0 + r < M) {
          const double a = wave_sum((double)acc[r][d]);
          if (lane == 0) {
            const double gz = -inv_l2 * (wsum_d * (double)(R)zl[kRowBlock * d + r] - a);
            f.gZ[(size_t)(i0 + r) * D + d] = MX ? (double)(float)gz : gz;
          }
        }
      }
    }
  }
  ks[0] = k0;
  ks[1] = k1;
}

// The same gradients for deep features (8 < D <= 32), in two phases.  One pass with D = 32 needs the 32 weighted point
// sums AND the row's point in registers and spilled (it took 2/3 of a step at M = 224).  Here the elementwise phase only
// forms the weights, W2 = 2 w in place of G and WX = wx in place of G_KX (row sums of both in V_MU), and the weighted
// point sums  sum_j W2[i][j] Z_j + WX[i][j] X_j  are an M x M x D product on the matrix cores: A fragments straight
// from the row-major weights (lane (i, k) reads W[i][k]: 32-byte segments, L2-resident, the product is tiny), B
// fragments from the row-major points.
template <bool MX>
__device__ __noinline__ void kernel_grads_deep(gd* __restrict__ G, const gd* __restrict__ GT, gd* __restrict__ GKX,
                                               const gd* __restrict__ KX, double s, double inv_l2, double (&ks)[2]) {
  typedef typename std::conditional<MX, float, double>::type R;
  const Fit& f = g_sh.f;
  const int M = f.M, Mp = f.Mp, D = f.D;
  const int cw = cl_wave(), CW = cl_waves(), lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const gd* Zt = f.Zt;
  const gd* Xc = f.XtT;
  const gf* KXf = (const gf*)KX;
  gd* wsum_v = f.vec[V_MU];
  const R sr = (R)s, hr = (R)(-0.5) * (R)inv_l2;
  double k0 = 0.0, k1 = 0.0;
  for (int i0 = kRowBlock * cw; i0 < M; i0 += kRowBlock * CW) {
    R wsum[kRowBlock];
#pragma unroll
    for (int r = 0; r < kRowBlock; ++r) wsum[r] = (R)0;
    const ldsd* zl = stage_rows(Zt, Mp, i0, D);
    for (int j = lane; j < M; j += 64) {
      R d2[kRowBlock], d2x[kRowBlock];
      sqdist_rows2<R>(zl, Zt, Xc, Mp, j, D, d2, d2x);
#pragma unroll
      for (int r = 0; r < kRowBlock; ++r) {
        if (i0 + r < M) {
          const size_t o = (size_t)(i0 + r) * Mp + j;
          const R gsym = (R)(0.5 * (G[o] + GT[o]));
          const R gk = (R)GKX[o];
          const R kx = MX ? (R)KXf[o] : (R)KX[o];
          const R e = MX ? (R)expf((float)(hr * d2[r])) : (R)exp((double)(hr * d2[r]));
          const R w = gsym * sr * e, wx = gk * kx;
          k0 += (double)(gsym * e + gk * kx / sr);
          k1 += (double)(w * d2[r] + wx * d2x[r]);
          wsum[r] += (R)2 * w + wx;
          G[o] = (double)((R)2 * w);
          GKX[o] = (double)wx;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kRowBlock; ++r) {
      const double wsum_d = wave_sum((double)wsum[r]);
      if (lane == 0 && i0 + r < M) wsum_v[i0 + r] = wsum_d;
    }
  }
  ks[0] = k0;
  ks[1] = k1;
  cbar();  // the weights of every row are visible to the whole cluster
  const int mt16 = (M + 15) / 16, dt16 = (D + 15) / 16, k4 = (M + 3) / 4 * 4;
  const gd* Zr = f.Z;
  const gd* Xr = f.X;
  for (int t = cw; t < mt16 * dt16; t += CW) {
    const int ti = t / dt16, td = t - ti * dt16;
    const int i0 = 16 * ti, d0 = 16 * td;
    const int ia = i0 + lr, dn = d0 + lr;
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
    const gd* w2 = G + (size_t)(ia < M ? ia : 0) * Mp;
    const gd* wxr = GKX + (size_t)(ia < M ? ia : 0) * Mp;
#pragma unroll 4
    for (int k = lq; k < k4; k += 4) {
      const bool kin = k < M;
      const double a2 = (kin && ia < M) ? w2[k] : 0.0, ax = (kin && ia < M) ? wxr[k] : 0.0;
      const double bz = (kin && dn < D) ? Zr[(size_t)k * D + dn] : 0.0;
      const double bxv = (kin && dn < D) ? Xr[(size_t)k * D + dn] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, bz, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, bxv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + lq + 4 * r;
      if (i < M && dn < D) {
        const double gz = -inv_l2 * (wsum_v[i] * Zr[(size_t)i * D + dn] - acc[r]);
        f.gZ[(size_t)i * D + dn] = MX ? (double)(float)gz : gz;
      }
    }
  }
}

template <bool MX, int TU>
__device__ __noinline__ void fit_body(const gapro_fit_options& opt, const gapro_fit_desc& desc, float* __restrict__ o_probs,
                         float* __restrict__ o_probs_new, unsigned char* __restrict__ o_labels,
                         float* __restrict__ o_mu, float* __restrict__ o_var, double* loss_out) {
  const Fit& f = g_sh.f;
  Shared& sh = g_sh;
  const int M = f.M, Mp = f.Mp, D = f.D, T = f.T;
  constexpr int TS = 16 * TU;
  const int mt = Mp / TS;
  const double Nd = (double)M;
  const double jitter = opt.jitter;
  gd* LS = f.mat[B_LS];
  gd* LST = f.mat[B_LST];
  gd* MLS = f.mat[B_MLS];
  gd* VLS = f.mat[B_VLS];
  gd* A = f.mat[B_A];
  gd* AT = f.mat[B_AT];
  gd* BM = f.mat[B_BM];
  gd* BMT = f.mat[B_BMT];
  gd* GA = f.mat[B_GA];
  gd* GKX = f.mat[B_GKX];
  gd* GKXT = f.mat[B_GKXT];
  gd* KX = f.mat[B_KX];
  gd* vm = f.vec[V_M];
  gd* gmu = f.vec[V_GMU];
  gd* gv = f.vec[V_GV];
  // mixed precision: the float32 matrices live in the first half of their float64 slot (leading dimension Mp)
  gf* LSf = (gf*)LS;
  gf* LSTf = (gf*)LST;
  gf* MLSf = (gf*)MLS;
  gf* VLSf = (gf*)VLS;
  gf* Af = (gf*)A;
  gf* ATf = (gf*)AT;
  gf* BMf = (gf*)BM;
  gf* BMTf = (gf*)BMT;
  gf* GAf = (gf*)GA;
  gd* GAT = f.mat[B_GLS];  // G_A^T: the G_LS slot is otherwise unused here (G_LS never leaves the registers)
  gf* GATf = (gf*)GAT;
  gd* PmT = BM;            // Phi(L^T G_L)^T: B is dead once G_A is formed
  const int ct = cl_tid(), CT = cl_threads();
  double last_loss = 0.0;

  auto refresh_hypers = [&]() {
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MX) {  // float32 parameters and float32 softplus
        const float rs = (float)sh.rho_s, rl = (float)sh.rho_l;
        const float sv = log1pf(expf(-fabsf(rs))) + fmaxf(rs, 0.f), lv = log1pf(expf(-fabsf(rl))) + fmaxf(rl, 0.f);
        sh.s = (double)sv;
        sh.ell = (double)lv;
        sh.inv_l2 = (double)(1.0f / (lv * lv));
      } else {
        sh.s = softplus(sh.rho_s);
        sh.ell = softplus(sh.rho_l);
        sh.inv_l2 = 1.0 / (sh.ell * sh.ell);
      }
    }
    __syncthreads();
  };
  // gpytorch's psd_safe_cholesky around the factorisation (see svgp_fit.hip: cholesky_psd_safe)
  auto factorize = [&]() {
    double extra = 0.0;
    for (int attempt = 0;; ++attempt) {
      build_kzz<MX>(sh.s, sh.inv_l2, jitter, extra);
      cbar();
      stamp(0);
      cholesky_cluster();
      // the leader's flag -> everybody (one ordered sum)
      double bad[1] = {(threadIdx.x == 0 && sh.g == 0 && sh.chol_bad) ? 1.0 : 0.0};
      cl_reduce(bad);
      if (threadIdx.x == 0) sh.chol_bad = 0;
      __syncthreads();
      if (bad[0] == 0.0) break;
      if (attempt >= opt.psd_retries) {
        if (threadIdx.x == 0) sh.status = GAPRO_ERR_CHOLESKY;
        break;
      }
      extra = opt.psd_jitter * pow(10.0, (double)attempt);
    }
    stamp(8);
    tri_inverse_cluster();
    stamp(4);
  };

  for (int step = 1; step <= opt.training_iter; ++step) {
    refresh_hypers();  // (workgroup barriers inside: sh.dead below is read uniformly)
    if (sh.dead) break;  // a cluster barrier timed out: nothing after it is synchronised, stop computing
    const double s = sh.s, ell = sh.ell, inv_l2 = sh.inv_l2, c = sh.c;
    // ------------------------------- forward -------------------------------
    factorize();
    build_kx<MX>(f.XtT, Mp, M, s, inv_l2);
    cbar();
    stamp(5);
    forward_products<MX, TU>(M);
    stamp(6);
    // quadrature: 16 lanes per training point, lane q < 10 evaluates the symmetric node pair +-t_q of the 20-point
    // Gauss-Hermite rule (BernoulliLikelihood.expected_log_prob), the 16 lanes are summed by xor-shuffles
    double sums[4] = {0.0, 0.0, 0.0, 0.0};  // E, g_c, gv_sum, KL part
    {
      const int q = ct & 15, per_pass = CT >> 4;
      for (int n0 = 0; n0 < Mp; n0 += per_pass) {
        const int n = n0 + (ct >> 4);
        double E = 0.0, dmu = 0.0, dvar = 0.0, y = 0.0, sd = 1.0;
        bool clamped = false;
        const bool on = n < M;
        if (on) {
          const double mu = col_final(0, Mp, n) + c;
          const double vraw = s + jitter + col_final(1, Mp, n);
          clamped = vraw < opt.min_variance;
          const double var = clamped ? opt.min_variance : vraw;
          sd = sqrt(2.0 * var);
          y = f.vec[V_Y][n];
          if (q < NGH / 2) {
            const double t = c_gh_t[q], w = c_gh_w[q];
            double lp, r;
            log_ndtr_ratio(y * (mu - sd * t), &lp, &r);
            E += w * lp; dmu += w * r; dvar -= w * t * r;
            log_ndtr_ratio(y * (mu + sd * t), &lp, &r);
            E += w * lp; dmu += w * r; dvar += w * t * r;
          }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          E += __shfl_xor(E, o, 64);
          dmu += __shfl_xor(dmu, o, 64);
          dvar += __shfl_xor(dvar, o, 64);
        }
        if (q == 0 && n < Mp) {
          double g1 = 0.0, g2 = 0.0;
          if (on) {
            const double ipi = 0.56418958354775628695;
            sums[0] += ipi * E;
            g1 = -(ipi * dmu * y) / Nd;
            g2 = clamped ? 0.0 : -(ipi * dvar * y / sd) / Nd;
          }
          gmu[n] = g1;
          gv[n] = g2;
          sums[1] += g1;
          sums[2] += g2;
        }
      }
    }
    if (step == opt.training_iter) {  // the KL term: only the ELBO VALUE needs it, and only the last one is reported
      const int cw = cl_wave(), CW = cl_waves(), lane = threadIdx.x & 63;
      for (int i = cw; i < M; i += CW)
        for (int j = lane; j <= i; j += 64) {
          const double v = MX ? (double)LSf[(size_t)i * Mp + j] : LS[(size_t)i * Mp + j];
          sums[3] += v * v;
          if (i == j) sums[3] -= log(v * v);
        }
      for (int i = ct; i < M; i += CT) sums[3] += vm[i] * vm[i];
    }
    cl_reduce(sums);  // includes a cluster barrier: gmu / gv are visible to everybody behind it
    const double g_c = sums[1], gv_sum = sums[2];
    last_loss = -(sums[0] / Nd - 0.5 * (sums[3] - Nd) / Nd);
    stamp(9);

    // ------------------------------- backward ------------------------------
    const double b1 = 0.9, b2 = 0.999, aeps = 1e-8;
    const double bc1 = 1.0 - pow(b1, (double)step), bc2s = sqrt(1.0 - pow(b2, (double)step));
    const double step_size = opt.lr / bc1;
    // G_m partials (through A^T) and G_A read L_S; the Adam update of L_S is fused into the G_LS product of the
    // NEXT phase, which runs beside G_KX = LI^T G_A
    if (MX) {
      const float step_f = (float)step_size, bc2s_f = (float)bc2s, Ndf = (float)Nd;
      col_partials(2, Mp, [=](int r, int cc) { return gmu[r] * (double)ATf[(size_t)r * Mp + cc]; });
      // G_A = m g_mu^T + L_S (2 B g_v) - 2 A g_v on v_mfma_f32 (float32 operands, float32 epilogue)
      gemm_tn_f32<TU, false, ORD_ROWS_DESC, true>(mt, mt, false, LSTf, BMf, Mp, nullptr,
                            [=](int i0, int, int* lo, int* hi) { *lo = 0; *hi = i0 + TS; },
                            [=](int i0, int n0, const f4& v) {
                              const int ln = threadIdx.x & 63, lr = ln & 15, lq = ln >> 4;
                              const int n = n0 + lr;
                              const float gvn = (float)gv[n], gmn = (float)gmu[n];
                              f4 ga;
#pragma unroll
                              for (int r = 0; r < 4; ++r) {
                                const int i = i0 + 4 * lq + r;  // float32 C layout
                                const float a = Af[(size_t)i * Mp + n];
                                ga[r] = 2.0f * gvn * v[r] + (float)vm[i] * gmn - 2.0f * a * gvn;
                              }
                              store_tile_f32(ga, GAf, GATf, Mp, i0, n0);
                            });
      cbar();
      stamp(10);
      // G_LS (lower) + KL' with Adam on the float32 L_S in the epilogue
      gemm_tn_f32<TU, true, ORD_ROWMAJOR, true>(mt, mt, true, ATf, BMTf, Mp, gv, [=](int, int, int* lo, int* hi) { *lo = 0; *hi = Mp; },
                           [=](int i0, int j0, const f4& v) {
                             const int ln = threadIdx.x & 63, lr = ln & 15, lq = ln >> 4;
                             const int j = j0 + lr;
                             f4 newv;
#pragma unroll
                             for (int r = 0; r < 4; ++r) {
                               const int i = i0 + 4 * lq + r;
                               const size_t o = (size_t)i * Mp + j;
                               float lnew = 0.f;
                               if (j <= i && i < M) {
                                 const float l = LSf[o];
                                 const float g = 2.0f * v[r] + (l - (i == j ? 1.0f / l : 0.f)) / Ndf;
                                 const float m1 = 0.9f * MLSf[o] + 0.1f * g;
                                 const float m2 = 0.999f * VLSf[o] + 0.001f * g * g;
                                 MLSf[o] = m1;
                                 VLSf[o] = m2;
                                 lnew = l - step_f * m1 / (sqrtf(m2) / bc2s_f + 1e-8f);
                                 LSf[o] = lnew;
                               }
                               newv[r] = lnew;
                             }
                             store_tile_f32(newv, (gf*)nullptr, LSTf, Mp, i0, j0);
                           });
      // G_KX = LI^T G_A in float64 (the float32 G_A enters through .double())
      gemm_tn<TU, false, gd, gf, ORD_ROWMAJOR, true>(mt, mt, false, f.mat[B_LI], GAf, Mp, nullptr,
                                [=](int i0, int, int* lo, int* hi) { *lo = i0; *hi = Mp; },
                                [=](int i, int n, const d4& v) { store_tile(v, GKX, (gd*)nullptr, Mp, i, n); });
      // Pm = Phi(L^T G_L) = Phi(-G_A A^T) in float64 (see the float64 branch), transposed -> the whole B slot (the
      // float32 B in its first half is dead)
      gemm_tn<TU, false, gf, gf, ORD_ROWMAJOR, true>(mt, mt, true, GATf, ATf, Mp, nullptr,
                                [=](int, int, int* lo, int* hi) { *lo = 0; *hi = Mp; },
                                [=](int i0, int j0, const d4& v) {
                                  const int ln = threadIdx.x & 63, lr = ln & 15, lq = ln >> 4;
                                  d4 pv;
#pragma unroll
                                  for (int r = 0; r < 4; ++r) {
                                    const int i = i0 + lq + 4 * r, j = j0 + lr;
                                    pv[r] = (j < i) ? -v[r] : (j == i ? -0.5 * v[r] : 0.0);
                                  }
                                  store_tile(pv, (gd*)nullptr, PmT, Mp, i0, j0);
                                });
      cbar();
      stamp(11);
    } else {
    col_partials(2, Mp, [=](int r, int cc) { return gmu[r] * AT[(size_t)r * Mp + cc]; });
    // (two-phase epilogue, epilogue.h: the loads of a tile's four blocks are issued before any of its stores)
    struct GaPre { double a[4], m[4], gvn, gmn; };
    gemm_tn<TU, false, gd, gd, ORD_ROWS_DESC, true>(mt, mt, false, LST, BM, Mp, nullptr,
                      [=](int i0, int, int* lo, int* hi) { *lo = 0; *hi = i0 + TS; },
                      two_phase_epi<4>(
                      [=](int i0, int n0) {
                        const int ln = threadIdx.x & 63, lr = ln & 15, lq = ln >> 4;
                        const int n = n0 + lr;
                        GaPre p;
                        p.gvn = gv[n];
                        p.gmn = gmu[n];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                          const int i = i0 + lq + 4 * r;
                          p.a[r] = A[(size_t)i * Mp + n];
                          p.m[r] = vm[i];
                        }
                        return p;
                      },
                      [=](int i0, int n0, const d4& v, const GaPre& p) {
                        d4 ga;
#pragma unroll
                        for (int r = 0; r < 4; ++r) ga[r] = 2.0 * p.gvn * v[r] + p.m[r] * p.gmn - 2.0 * p.a[r] * p.gvn;
                        store_tile(ga, GA, GAT, Mp, i0, n0);
                      }));
    cbar();
    stamp(10);
    // G_LS[i][j] = sum_n A[i][n] 2 g_v[n] B[j][n] (lower) + KL', Adam on L_S in the epilogue (L_S^T through the
    // wave's transpose tile: 128-byte rows)
    struct LsPre { double l[4], m1[4], m2[4]; };
    gemm_tn<TU, true, gd, gd, ORD_ROWMAJOR, true>(mt, mt, true, AT, BMT, Mp, gv, [=](int, int, int* lo, int* hi) { *lo = 0; *hi = Mp; },
                     two_phase_epi<4>(
                     [=](int i0, int j0) {
                       const int ln = threadIdx.x & 63, lr = ln & 15, lq = ln >> 4;
                       const int j = j0 + lr;
                       LsPre p;
#pragma unroll
                       for (int r = 0; r < 4; ++r) {  // unconditional: every (i, j) of a tile lies inside the M_p x M_p slots
                         const size_t o = (size_t)(i0 + lq + 4 * r) * Mp + j;
                         p.l[r] = LS[o];
                         p.m1[r] = MLS[o];
                         p.m2[r] = VLS[o];
                       }
                       return p;
                     },
                     [=](int i0, int j0, const d4& v, const LsPre& p) {
                       const int ln = threadIdx.x & 63, lr = ln & 15, lq = ln >> 4;
                       const int j = j0 + lr;
                       d4 newv;
#pragma unroll
                       for (int r = 0; r < 4; ++r) {
                         const int i = i0 + lq + 4 * r;
                         const size_t o = (size_t)i * Mp + j;
                         double lnew = 0.0;
                         if (j <= i && i < M) {
                           const double l = p.l[r];
                           const double g = 2.0 * v[r] + (l - (i == j ? 1.0 / l : 0.0)) / Nd;
                           const double m1 = b1 * p.m1[r] + (1.0 - b1) * g;
                           const double m2 = b2 * p.m2[r] + (1.0 - b2) * g * g;
                           MLS[o] = m1;
                           VLS[o] = m2;
                           lnew = l - step_size * m1 / (sqrt(m2) / bc2s + aeps);
                           LS[o] = lnew;
                         }
                         newv[r] = lnew;
                       }
                       store_tile(newv, (gd*)nullptr, LST, Mp, i0, j0);  // LST[j][i]; zeros above the diagonal
                     }));
    // G_KX = LI^T G_A
    gemm_tn<TU, false, gd, gd, ORD_ROWMAJOR, true>(mt, mt, false, f.mat[B_LI], GA, Mp, nullptr,
                      [=](int i0, int, int* lo, int* hi) { *lo = i0; *hi = Mp; },
                      [=](int i, int n, const d4& v) { store_tile(v, GKX, (gd*)nullptr, Mp, i, n); });
    // The Cholesky backward pass needs Pm = Phi(L^T G_L) with G_L = -tril(L^-T G_A A^T).  Row i of L^T X only reads
    // rows k >= i of X, so the lower triangle of L^T tril(X) is the lower triangle of L^T X = -G_A A^T:
    //   Pm = Phi(-G_A A^T)   -- no G_L, no product with L^T, and two cluster barriers fewer per step (rounds 1-2 formed
    // G_L = -tril(G_KX A^T) and L^T G_L in phases of their own).  Stored transposed (the P operand of W) -> B buffer.
    gemm_tn<TU, false, gd, gd, ORD_ROWMAJOR, true>(mt, mt, true, GAT, AT, Mp, nullptr, [=](int, int, int* lo, int* hi) { *lo = 0; *hi = Mp; },
                      [=](int i0, int j0, const d4& v) {
                        const int ln = threadIdx.x & 63, lr = ln & 15, lq = ln >> 4;
                        d4 pv;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                          const int i = i0 + lq + 4 * r, j = j0 + lr;
                          pv[r] = (j < i) ? -v[r] : (j == i ? -0.5 * v[r] : 0.0);
                        }
                        store_tile(pv, (gd*)nullptr, PmT, Mp, i0, j0);
                      });
    cbar();
    stamp(11);
    }
    // G_Kzz (unsymmetrised) = L^-T Pm L^-1, associated as L^-T (Pm L^-1): W = Pm L^-1 is a product of two
    // lower-triangular matrices (M^3 / 3, lower itself), S = L^-T W costs 2 M^3 / 3 -- 1.0 M^3 where (L^-T Pm) L^-1
    // spends 2/3 + 1 (svgp_fit.hip has the same order).
    // W = Pm L^-1 (lower tiles only; zeros above the diagonal inside them) -> BMT buffer.  S reads W[k][j] for
    // k >= max(i0, j0) only, i.e. lower tiles: what the upper tiles of the buffer hold does not matter.
    // (j0 <= k < i0 + tile: Pm^T[k][i] = 0 for k > i, L^-1[k][j] = 0 for k < j)
    gd* Wm = BMT;
    gemm_tn<TU, false, gd, gd, ORD_ROWMAJOR, true>(mt, mt, true, PmT, f.mat[B_LI], Mp, nullptr,
                      [=](int i0, int j0, int* lo, int* hi) { *lo = j0; *hi = i0 + TS; },
                      [=](int i0, int j0, const d4& v) {
                        const int ln = threadIdx.x & 63, lr = ln & 15, lq = ln >> 4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                          const int i = i0 + lq + 4 * r, j = j0 + lr;
                          Wm[(size_t)i * Mp + j] = (j <= i) ? v[r] : 0.0;
                        }
                      });
    cbar();
    stamp(14);
    // S = L^-T W -> G in the BM buffer, G^T in the GKXT buffer
    gd* G = BM;
    gd* GT = GKXT;
    gemm_tn<TU, false, gd, gd, ORD_SHELLS, true>(mt, mt, false, f.mat[B_LI], Wm, Mp, nullptr,
                      [=](int i0, int j0, int* lo, int* hi) { *lo = i0 > j0 ? i0 : j0; *hi = Mp; },
                      [=](int i, int j, const d4& v) { store_tile(v, G, GT, Mp, i, j); });
    cbar();
    stamp(15);
    // kernel gradients in one pass (kernel_grads): a wave owns row i and walks its columns j 64 at a time (Z_i in
    // registers, Z_j / X_j: coalesced loads from the transposed copies):
    //   w  = sym(G)[i][j] s E_ij  ->  G_s += sym(G) E,  G_l += w d2,   G_Z[i] += 2 w (Z_i - Z_j)
    //   wx = G_KX[i][j] KX_ij     ->  G_s += G_KX KX / s, G_l += wx d2x, G_Z[i] += wx (Z_i - X_j)
    double ks[2] = {0.0, 0.0};
    if (D <= 8) kernel_grads<8, MX>(G, GT, GKX, KX, s, inv_l2, ks);
    else kernel_grads_deep<MX>(G, GT, GKX, KX, s, inv_l2, ks);
    cl_reduce(ks);  // its barrier also publishes G_Z: every entry was computed from the OLD Z
    stamp(16);
    const double g_s = ks[0] + gv_sum;
    const double g_l = ks[1] / (ell * ell * ell);
    stamp(17);

    // ------------------------------- Adam (Z, m, scalars; L_S was updated in the G_LS epilogue) --------------
    // torch.optim.Adam; under MX on float32 parameters in float32 arithmetic (values kept in the float64 slots)
    auto adam = [&](double p, double& m1, double& m2, double g) {
      if (MX) {
        const float gf32 = (float)g;
        const float a1 = 0.9f * (float)m1 + 0.1f * gf32;
        const float a2 = 0.999f * (float)m2 + 0.001f * gf32 * gf32;
        m1 = (double)a1;
        m2 = (double)a2;
        return (double)((float)p - (float)step_size * a1 / (sqrtf(a2) / (float)bc2s + 1e-8f));
      }
      m1 = b1 * m1 + (1.0 - b1) * g;
      m2 = b2 * m2 + (1.0 - b2) * g * g;
      return p - step_size * m1 / (sqrt(m2) / bc2s + aeps);
    };
    for (int idx = ct; idx < M * D; idx += CT) {
      const int i = idx / D, d = idx - i * D;
      double m1 = f.mZ[idx], m2 = f.vZ[idx];
      const double znew = adam(f.Z[idx], m1, m2, f.gZ[idx]);
      f.Z[idx] = znew;
      f.Zt[(size_t)d * Mp + i] = znew;
      f.mZ[idx] = m1;
      f.vZ[idx] = m2;
    }
    for (int i = ct; i < M; i += CT) {
      const double g = col_final(2, Mp, i) + vm[i] / Nd;
      f.vec[V_GM][i] = g;
      double m1 = f.vec[V_MM][i], m2 = f.vec[V_VM][i];
      vm[i] = adam(vm[i], m1, m2, g);
      f.vec[V_MM][i] = m1;
      f.vec[V_VM][i] = m2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // every workgroup keeps the scalars and applies the same update
      sh.c = adam(sh.c, sh.mc, sh.vc, g_c);
      const double sg_s = MX ? (double)(1.0f / (1.0f + expf(-(float)sh.rho_s))) : sigmoid(sh.rho_s);
      const double sg_l = MX ? (double)(1.0f / (1.0f + expf(-(float)sh.rho_l))) : sigmoid(sh.rho_l);
      sh.rho_s = adam(sh.rho_s, sh.mrs, sh.vrs, g_s * sg_s);
      sh.rho_l = adam(sh.rho_l, sh.mrl, sh.vrl, g_l * sg_l);
    }
    cbar();
    stamp(18);
  }

  // ------------------------------- prediction ------------------------------
  refresh_hypers();
  if (!(opt.eval_stale_chol && opt.training_iter > 0)) factorize();
  const double s = sh.s, inv_l2 = sh.inv_l2, c = sh.c;
  for (int t0 = 0; t0 < T; t0 += Mp) {
    const int nc = (T - t0) < Mp ? (T - t0) : Mp;
    build_kx<MX>(f.Xt + t0, round_up(T > 0 ? T : 1, 32), nc, s, inv_l2);
    cbar();
    forward_products<MX, TU>(nc);
    for (int n = ct; n < nc; n += CT) {
      const double mu = col_final(0, Mp, n) + c;
      const double var = fmax(s + jitter + col_final(1, Mp, n), opt.min_variance);
      const double p = 0.5 * erfc(-(mu / sqrt(1.0 + var)) * 0.70710678118654752440);
      const float pf = (float)p;                       // pred_probs            :432
      const bool lab = pf >= 0.5f;                     // pred_labels           :433
      const long long o = desc.out_offset + t0 + n;
      o_probs[o] = pf;
      o_probs_new[o] = lab ? pf : 1.0f - pf;           // pred_probs_new        :438
      o_labels[o] = lab ? 1 : 0;
      o_mu[o] = (float)mu;                             // pred_mu               :435
      o_var[o] = (float)var;                           // pred_variance         :436
      if ((!isfinite(mu) || !isfinite(var)) && sh.status == GAPRO_OK) sh.status = GAPRO_ERR_NOT_FINITE;
    }
    cbar();
  }
  // status of the cluster = the first error any member saw
  double st[1] = {(threadIdx.x == 0 && sh.status != GAPRO_OK) ? (double)(-sh.status) : 0.0};
  {
    // max over members, through an ordered sum of one-hot encodings would be overkill: errors are rare -> reduce sum
    cl_reduce(st);
  }
  stamp(19);
#ifdef GAPRO_PROFILE
  if (sh.g == 0 && threadIdx.x == 0)
  {
    for (int i = 0; i < 28; ++i) f.scal[24 + i] = (double)sh.prof[i];
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    f.scal[24 + 25] = (double)sh.t_start;  // timeline of the launch: tools/fit_timeline.py
    f.scal[24 + 26] = (double)wall_clock64();
    f.scal[24 + 27] = (double)(((xcc & 15u) << 16) | (hwid & 0xFFFFu));
  }
#endif
  if (sh.g == 0 && threadIdx.x == 0) {
    if (sh.status == GAPRO_OK && st[0] != 0.0) sh.status = GAPRO_ERR_NOT_FINITE;
    f.scal[S_C] = sh.c;
    f.scal[S_RS] = sh.rho_s;
    f.scal[S_RL] = sh.rho_l;
    f.scal[S_MC] = sh.mc; f.scal[S_VC] = sh.vc;
    f.scal[S_MRS] = sh.mrs; f.scal[S_VRS] = sh.vrs;
    f.scal[S_MRL] = sh.mrl; f.scal[S_VRL] = sh.vrl;
    f.scal[S_LOSS] = last_loss;
    *loss_out = last_loss;
  }
}

struct ClBlock {
  int fit;  // index into the launch's descriptor array, -1 = padding block
  int g, G;
  int ctl;  // index of the cluster's barrier counter
};

__global__ __launch_bounds__(NT, 2) void k_svgp_fit_cluster(const ClBlock* __restrict__ blocks, int D,
                                                          const float* __restrict__ feats_spp,
                                                          const int* __restrict__ idx,
                                                          const gapro_fit_desc* __restrict__ descs,
                                                          const double* __restrict__ init_mean, gapro_fit_options opt,
                                                          double* __restrict__ ws, unsigned* __restrict__ ctl,
                                                          float* __restrict__ o_probs, float* __restrict__ o_probs_new,
                                                          unsigned char* __restrict__ o_labels, float* __restrict__ o_mu,
                                                          float* __restrict__ o_var, int* __restrict__ o_status,
                                                          double* __restrict__ o_loss, unsigned long long bar_ticks,
                                                          unsigned* info) {
  const ClBlock cb = blocks[blockIdx.x];
  if (cb.fit < 0) return;
  if (info && threadIdx.x == 0) atomicAdd(info, 1u);  // diagnostics (gapro_fit_timing_cluster_info)
  // test bit 15 of gapro_fit_options.reserved: the last member of every cluster never arrives (as if it had not been
  // given a CU) -- the others must time out and report GAPRO_ERR_TIMEOUT instead of hanging
  if ((opt.reserved & 32768) && cb.G > 1 && cb.g == cb.G - 1) return;
  const gapro_fit_desc desc = descs[cb.fit];
  Shared& sh = g_sh;
  Fit& f = sh.f;
  const Layout lay = make_layout(desc.m1 + desc.m2, desc.t, D);
  gd* base = (gd*)(ws + desc.ws_offset);
  if (threadIdx.x == 0) {
    f.M = desc.m1 + desc.m2;
    f.T = desc.t;
    f.D = D;
    f.Mp = lay.Mp;
    for (int b = 0; b < B_COUNT; ++b) f.mat[b] = base + lay.mat + (long long)b * lay.Mp * lay.Mp;
    for (int v = 0; v < V_COUNT; ++v) f.vec[v] = base + lay.vec + (long long)v * lay.Mp;
    f.X = base + lay.xz;
    f.Z = f.X + (long long)lay.Mp * D;
    f.mZ = f.Z + (long long)lay.Mp * D;
    f.vZ = f.mZ + (long long)lay.Mp * D;
    f.gZ = f.vZ + (long long)lay.Mp * D;
    f.Xt = base + lay.xt;
    f.dinv = base + lay.dinv;
    f.dinvT = f.dinv + (long long)lay.Mp * 16;
    f.scal = base + lay.scal;
    f.part = base + lay.cl;
    f.red = f.part + 3LL * cluster_plane_doubles(lay.Mp);
    f.Zt = base + lay.cl + cluster_part_doubles(lay.Mp);
    f.XtT = f.Zt + (long long)lay.Mp * D;
    sh.G = cb.G;
    sh.g = cb.g;
    sh.epoch = 0;
    sh.red_par = 0;
    sh.same_xcd = 0;
    sh.wave_off = 0;
    sh.count = (gu32*)(ctl + (size_t)cb.ctl * 32);
    sh.c = sh.rho_s = sh.rho_l = 0.0;
    sh.mc = sh.vc = sh.mrs = sh.vrs = sh.mrl = sh.vrl = 0.0;
    sh.status = GAPRO_OK;
    sh.chol_bad = 0;
    sh.dead = 0;
    sh.bar_ticks = bar_ticks;
#ifdef GAPRO_PROFILE
    for (int i = 0; i < 28; ++i) sh.prof[i] = 0;
    sh.t_last = wall_clock64();
    sh.t_start = sh.t_last;
#endif
  }
  __syncthreads();
  const int M = f.M, Mp = f.Mp;
  const int ct = cl_tid(), CT = cl_threads();
  // zero everything the kernel reads before writing (parameters / Adam state, padded operand tails)
  for (long long i = ct; i < lay.total; i += CT) base[i] = 0.0;
  cbar();
  if (!(opt.reserved & 256)) detect_same_xcd();  // debug bit 8: always the full barrier
  if (info && threadIdx.x == 0 && cb.g == 0 && cb.G > 1) {  // diagnostics (gapro_fit_timing_cluster_info)
    atomicAdd(info + 2, 1u);
    if (!sh.same_xcd) atomicAdd(info + 1, 1u);
  }
  const int* my_idx = idx + desc.idx_offset;
  for (int e = ct; e < M * D; e += CT) {
    const int i = e / D, d = e - i * D;
    const double v = (double)feats_spp[(size_t)my_idx[i] * D + d];  // train_x = cat(b1_feats, b2_feats)  :395
    f.X[e] = v;
    f.Z[e] = v;  // inducing points initialised to train_x  (:14)
    f.Zt[(size_t)d * Mp + i] = v;
    f.XtT[(size_t)d * Mp + i] = v;
  }
  for (int e = ct; e < f.T * D; e += CT) {
    const int i = e / D, d = e - i * D;
    f.Xt[(size_t)d * lay.Tp + i] = (double)feats_spp[(size_t)my_idx[M + i] * D + d];  // intersect_feats :386, [D][Tp]
  }
  for (int i = ct; i < M; i += CT) {
    f.vec[V_Y][i] = i < desc.m1 ? -1.0 : 1.0;  // train_y  :396-398
    f.vec[V_M][i] = init_mean ? init_mean[desc.idx_offset + i] : 0.0;
    if (opt.precision == GAPRO_PRECISION_MIXED) {  // float32 matrices in the first half of their slots
      ((gf*)f.mat[B_LS])[(size_t)i * Mp + i] = 1.0f;
      ((gf*)f.mat[B_LST])[(size_t)i * Mp + i] = 1.0f;
    } else {
      f.mat[B_LS][(size_t)i * Mp + i] = 1.0;  // chol_variational_covar = I
      f.mat[B_LST][(size_t)i * Mp + i] = 1.0;
    }
  }
  cbar();
  // wave tiles of the M^3 products: 32 x 32, or 64 x 64 where M_p allows it (twice the FLOP per operand byte; the
  // products of many concurrent fits stream their operands from HBM) -- debug bit 5 of gapro_fit_options.reserved
  const bool wide = (opt.reserved & 32) && Mp % 64 == 0;
  if (opt.precision == GAPRO_PRECISION_MIXED) {
    if (wide) fit_body<true, 4>(opt, desc, o_probs, o_probs_new, o_labels, o_mu, o_var, &o_loss[desc.slot]);
    else fit_body<true, 2>(opt, desc, o_probs, o_probs_new, o_labels, o_mu, o_var, &o_loss[desc.slot]);
  } else {
    if (wide) fit_body<false, 4>(opt, desc, o_probs, o_probs_new, o_labels, o_mu, o_var, &o_loss[desc.slot]);
    else fit_body<false, 2>(opt, desc, o_probs, o_probs_new, o_labels, o_mu, o_var, &o_loss[desc.slot]);
  }
  __syncthreads();
  if (sh.g == 0 && threadIdx.x == 0) {
    int st = sh.status;
    if (sh.dead) st = GAPRO_ERR_TIMEOUT;
    if (st == GAPRO_OK && !isfinite(o_loss[desc.slot]) && opt.training_iter > 0) st = GAPRO_ERR_NOT_FINITE;
    o_status[desc.slot] = st;
    f.scal[S_STATUS] = (double)st;
  }
}

}  // namespace

// Smallest padded M routed to this kernel by default (a fit below stays on one workgroup of the LDS-staged / strip
// kernels).  Tunable for A/B runs through GAPRO_CLUSTER_MIN_MP (read once).
int gapro_cluster_min_mp() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("GAPRO_CLUSTER_MIN_MP");
    v = e ? atoi(e) : kClusterDefaultMinMp;
    if (v < kClusterMinMp) v = kClusterMinMp;
  }
  return v;
}

// Workgroups a fit of padded size Mp is spread over by gapro_svgp_fit_batch; 0 = not a cluster fit.  `all`: every fit
// the kernel can take (debug bit 4 of gapro_fit_options.reserved: precision sweeps through one kernel).
int gapro_cluster_size(int Mp, bool all) {
  if (!cluster_capable(Mp) || (!all && Mp < gapro_cluster_min_mp())) return 0;
  // a pure function of Mp (never of the batch: the summation order inside a fit depends on G); the two environment
  // variables are tuning knobs for tools/bench_fit.py
  static double unit = 0.0;
  static int pow2 = 1;
  if (unit == 0.0) {
    const char* e = getenv("GAPRO_CLUSTER_UNIT");
    const char* r = getenv("GAPRO_CLUSTER_ROUND");
    if (r) pow2 = strcmp(r, "ceil") != 0;
    unit = e && atof(e) >= kClusterMinUnit ? atof(e) : 384.0;  // (the scratch planes are sized for kClusterMinUnit)
  }
  return cluster_g(Mp, unit, pow2 != 0);
}

// Internal launcher used by gapro_svgp_fit_batch (svgp_fit.hip).  `fits` = the n cluster fits' indices into the device
// descriptor array d_descs (already uploaded), with their padded sizes; h_stage / d_stage: staging for the block
// table (>= bytes returned by gapro_cluster_stage_bytes), d_ctl: >= 128 bytes per fit, zeroed here.
size_t gapro_cluster_stage_bytes(int n_fits) { return (size_t)n_fits * kClMaxG * sizeof(ClBlock) + 8 * kClMaxG * sizeof(ClBlock); }

int gapro_prepare_fit_cluster(hipStream_t stream, int n, const int* fit_index, const int* fit_mp, const int* fit_g,
                              void* h_stage, void* d_stage, unsigned* d_ctl, int* out_blocks, int* out_members) {
  *out_blocks = *out_members = 0;
  if (n <= 0) return GAPRO_OK;
  // windows of 8 clusters (one per XCD label): block base + 8 j + x is member j of the window's x-th cluster
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    return fit_g[a] != fit_g[b] ? fit_g[a] > fit_g[b] : fit_mp[a] > fit_mp[b];
  });
  ClBlock* hb = (ClBlock*)h_stage;
  int nb = 0;
  // slot x of a window is XCD x: the window's clusters (largest first) go to the slots that carry the least work so
  // far (M_p^3 per cluster), so that no XCD gets the largest cluster of every window
  double load[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int w0 = 0; w0 < n; w0 += 8) {
    const int cnt = std::min(8, n - w0);
    const int Gw = fit_g[order[w0]];
    int slot_fit[8], slot_ctl[8];
    {
      int slots[8] = {0, 1, 2, 3, 4, 5, 6, 7};
      std::stable_sort(slots, slots + 8, [&](int a, int b) { return load[a] < load[b]; });
      for (int x = 0; x < 8; ++x) slot_fit[x] = -1;
      for (int c = 0; c < cnt; ++c) {
        const int fi = order[w0 + c];
        slot_fit[slots[c]] = fi;
        slot_ctl[slots[c]] = w0 + c;  // the cluster's barrier counter line
        load[slots[c]] += (double)fit_mp[fi] * fit_mp[fi] * fit_mp[fi];
      }
    }
    for (int j = 0; j < Gw; ++j)
      for (int x = 0; x < 8; ++x) {
        ClBlock b;
        b.fit = -1; b.g = 0; b.G = 0; b.ctl = 0;
        if (slot_fit[x] >= 0) {
          const int fi = slot_fit[x];
          if (j < fit_g[fi]) {
            b.fit = fit_index[fi];
            b.g = j;
            b.G = fit_g[fi];
            b.ctl = slot_ctl[x];
          }
        }
        if (b.fit >= 0) ++*out_members;
        hb[nb++] = b;
      }
  }
  *out_blocks = nb;
  if (hipMemsetAsync(d_ctl, 0, (size_t)n * 128, stream) != hipSuccess) return GAPRO_ERR_HIP;
  if (hipMemcpyAsync(d_stage, h_stage, (size_t)nb * sizeof(ClBlock), hipMemcpyHostToDevice, stream) != hipSuccess)
    return GAPRO_ERR_HIP;
  return GAPRO_OK;
}

int gapro_launch_fit_cluster(hipStream_t stream, int nb, int feat_dim, void* d_stage, unsigned* d_ctl, unsigned* d_info,
                             const float* d_feats_spp, const int* d_idx, const gapro_fit_desc* d_descs,
                             const double* d_init_mean, const gapro_fit_options& opt, double* d_workspace, float* d_probs,
                             float* d_probs_new, unsigned char* d_labels, float* d_mu, float* d_var, int* d_fit_status,
                             double* d_fit_loss) {
  if (nb <= 0) return GAPRO_OK;
  // longest wait at one cluster barrier before the cluster gives up (cbar): 5 s unless the environment says otherwise
  static long long timeout_ms = -1;
  if (timeout_ms < 0) {
    const char* e = getenv("GAPRO_CLUSTER_BARRIER_TIMEOUT_MS");
    timeout_ms = e && atoll(e) >= 0 ? atoll(e) : 5000;
  }
  const unsigned long long bar_ticks = (unsigned long long)timeout_ms * 100000ull;  // wall_clock64: 100 MHz
  hipLaunchKernelGGL(k_svgp_fit_cluster, dim3(nb), dim3(NT), 0, stream, (const ClBlock*)d_stage, feat_dim, d_feats_spp,
                     d_idx, d_descs, d_init_mean, opt, d_workspace, d_ctl, d_probs, d_probs_new, d_labels, d_mu, d_var,
                     d_fit_status, d_fit_loss, bar_ticks, d_info);
  return hipGetLastError() == hipSuccess ? GAPRO_OK : GAPRO_ERR_HIP;
}
