// Workspace layout of one GP fit (doubles), shared by every fit kernel of libgapro_hip.so.
#pragma once
#include "common.h"

namespace gapro_fit {

enum MatId {
  B_LS = 0, B_LST, B_MLS, B_VLS, B_GLS, B_L, B_LT, B_LI, B_U, B_KX, B_A, B_AT, B_BM, B_BMT, B_GA, B_GKX, B_GKXT,
  B_COUNT
};
enum VecId { V_Y = 0, V_M, V_MM, V_VM, V_GM, V_MU, V_VAR, V_GMU, V_GV, V_COUNT };
constexpr int kScalars = 64;
// scalars kept in the workspace tail (also visible to tests)
enum ScalId { S_C = 0, S_RS, S_RL, S_MC, S_MRS, S_MRL, S_VC, S_VRS, S_VRL, S_LOSS, S_STATUS };

inline __host__ __device__ int round_up(int x, int a) { return (x + a - 1) / a * a; }

// Fits run by the cluster kernel (svgp_fit_cluster.hip: one fit over G workgroups) exchange partial sums through a
// scratch region at the end of their workspace: 3 column-partial planes of max(G * 512, Mp) doubles + 2 x 16 scalar
// slots per member, then the transposed copies Zt[D][Mp], Xt[D][Mp] of the inducing / training points.
constexpr int kClMaxG = 32;                      // largest cluster (workgroups per fit)
constexpr int kClThreads = 512;                  // threads per cluster workgroup
constexpr int kClusterMinMp = 64;                // smallest padded M the cluster kernel takes (64-wide panels)
constexpr int kClusterDefaultMinMp = 512;        // default routing threshold (gapro_cluster_min_mp)
// workgroups a fit of padded size Mp is spread over: the work grows with Mp^3 while a launch's other fits finish in
// a fraction of a second, so the largest fits get the most CUs (powers of two; one CU up to Mp = 384)
inline __host__ __device__ int cluster_g(int Mp, double unit = 384.0, bool pow2 = true) {
  const double work = (double)Mp * Mp * Mp / (unit * unit * unit);
  int g = 1;
  if (pow2) {
    while (g < kClMaxG && (double)g < work) g *= 2;
  } else {
    while (g < kClMaxG && (double)g < work) ++g;
  }
  return g;
}
// largest padded M the cluster kernel takes: its pairwise-merge inverse keeps one record per pair of 64-wide panels in
// LDS (kMaxPairs = 40 in svgp_fit_cluster.hip: ceil((M_p - 64) / 128) pairs at the first level)
constexpr int kClusterMaxMp = 5120;
inline __host__ __device__ bool cluster_capable(int Mp) {
  return Mp >= kClusterMinMp && Mp % 32 == 0 && Mp <= kClusterMaxMp;
}
// a plane of the ordered two-stage column sums: sized for the largest cluster ANY policy can give a fit of this size
// (gapro_cluster_size: the work unit is tunable down to kClusterMinUnit), so that the layout does not depend on the
// policy -- and not for 32 workgroups whatever the size: M_p = 64 .. 128 fits, thousands per batch and never on the
// cluster kernel by default, carried 400 KB of scratch each
constexpr double kClusterMinUnit = 64.0;
inline __host__ __device__ long long cluster_plane_doubles(int Mp) {
  const long long t = (long long)cluster_g(Mp, kClusterMinUnit, true) * kClThreads;
  return Mp > t ? Mp : t;
}
inline __host__ __device__ long long cluster_part_doubles(int Mp) {
  return 3 * cluster_plane_doubles(Mp) + 2 * 16 * kClMaxG;
}
inline __host__ __device__ long long cluster_scratch_doubles(int Mp, int d) {
  if (!cluster_capable(Mp)) return 0;
  return cluster_part_doubles(Mp) + 2LL * Mp * d;
}

struct Layout {
  int Mp, Tp, D;
  long long mat, vec, xz, xt, dinv, scal, cl, total;
};
inline __host__ __device__ Layout make_layout(int m, int t, int d) {
  Layout L;
  L.Mp = gapro_pad_m(m, d);
  L.Tp = round_up(t > 0 ? t : 1, 32);
  L.D = d;
  L.mat = 0;
  L.vec = L.mat + (long long)B_COUNT * L.Mp * L.Mp;
  L.xz = L.vec + (long long)V_COUNT * L.Mp;
  L.xt = L.xz + 5LL * L.Mp * d;  // X, Z, mZ, vZ, gZ
  L.dinv = L.xt + (long long)L.Tp * d;
  L.scal = L.dinv + 2LL * L.Mp * 16;  // Dinv and Dinv^T blocks
  L.cl = L.scal + kScalars;
  L.total = L.cl + cluster_scratch_doubles(L.Mp, d);
  L.total = (L.total + 1) / 2 * 2;
  return L;
}

}  // namespace gapro_fit
