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

// Fits spread over several workgroups (svgp_fit_cluster.hip) exchange partial sums through a scratch region at the
// end of their workspace: 3 column-partial planes of max(kClMaxThreads, Mp) doubles + 2 x 16 scalar slots per member,
// then the transposed copies Zt[D][Mp], Xt[D][Mp] of the inducing / training points (coalesced kernel evaluation).
constexpr int kClMaxG = 32;                      // largest cluster (workgroups per fit)
constexpr int kClThreads = 512;                  // threads per cluster workgroup
constexpr int kClMaxThreads = kClMaxG * kClThreads;
constexpr int kClusterMinMp = 192;               // smallest padded M that may be routed to the cluster kernel (its
                                                 // scratch exists from here on; tiles are 32 wide: M_p % 32 == 0)
constexpr int kClusterDefaultMinMp = 416;        // default routing threshold (gapro_cluster_min_mp)
inline __host__ __device__ long long cluster_part_doubles(int Mp) {
  const long long w = Mp > kClMaxThreads ? Mp : kClMaxThreads;
  return 3 * w + 2 * 16 * kClMaxG;
}
inline __host__ __device__ long long cluster_scratch_doubles(int Mp, int d) {
  if (Mp < kClusterMinMp) return 0;
  return cluster_part_doubles(Mp) + 2LL * Mp * d;
}

struct Layout {
  int Mp, Tp, D;
  long long mat, vec, xz, xt, dinv, scal, cl, total;
};
inline __host__ __device__ Layout make_layout(int m, int t, int d) {
  Layout L;
  L.Mp = gapro_pad_m(m);
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
