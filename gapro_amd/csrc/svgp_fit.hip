// Batched whitened-SVGP classifier fit for gfx950 (MI355X): one workgroup = one GP fit, the whole Adam loop inside
// one launch.  This file holds the single-workgroup MFMA kernels -- k_svgp_fit<WPS, KMIN> (LDS-staged, 128 < M_p <= 512),
// k_svgp_fit_strip<D> (strip-streaming, M_p <= 128; built a second time with 256 threads per fit as svgp_fit_small.hip
// for M_p <= 64) -- and the host entry gapro_svgp_fit_batch that routes every fit of a launch (strip / staged / cluster
// / generic kernel).  svgp_fit_cluster.hip spreads one large fit over several workgroups; svgp_fit_large.hip is the
// generic fallback (D > 32) with the same arithmetic.
//
// Replaces reference gapro/gaussian_process_utils.py:382-445 (fit_gp_spp) and the gpytorch objects it builds
// (:11-25): CholeskyVariationalDistribution + whitened VariationalStrategy with learned inducing locations,
// ConstantMean, ScaleKernel(RBFKernel), BernoulliLikelihood (20-point Gauss-Hermite), VariationalELBO,
// Adam(lr=0.1) x training_iter (:416-423), then prediction (:426-438).  There is no autograd on the device: the
// backward pass is the hand-derived one of SURVEY.md Appendix B.5, restated and checked against torch autograd in
// oracle/svgp_oracle.py.
//
// Arithmetic: float64 throughout (DESIGN.md "Precision").  Every M x M x M contraction is an MFMA product on
// v_mfma_f64_16x16x4_f64.  The default form is TN, C[i][j] = sum_k P[k][i] Q[k][j], both operands row-major in k so
// that fragment loads are 128-byte row segments; up to M_p = 256 (KMIN) the products that contract over the COLUMNS of
// A, B, G_A, Pm read those matrices as they are (two consecutive k per 16-byte load), so that no transposed copy is
// ever written.
//
// Per Adam step (round 3's step: 7.67 M^3 executed; G_L is never formed):
//   forward : K_ZZ tiles are evaluated on the fly inside the blocked left-looking Cholesky (16-wide block columns,
//             MFMA updates reading L^T, block column in LDS, 16 x 16 diagonal block factored AND inverted in registers by
//             one wave) -> L^T (the row-major factor L is not written: nothing reads it);  LI = L^-1 (one 16-wide block
//             column per wave, blocks kept in registers) -> LI, LI^T;  KX = k(Z, X);  A = LI KX;  B = LS^T A;
//             mu = A^T m + c;  var = s + eps + colsum(B^2) - colsum(A^2)  (column sums fused into the product epilogues)
//             g_mu, g_v from the 20-point Gauss-Hermite rule of log Phi(y f)
//   backward: G_m, G_c;  G_A = m g_mu^T + LS (2 B g_v) - 2 A diag(g_v);  G_LS = tril(A (2 B g_v)^T) + KL' with the Adam
//             update of LS fused into the epilogue;  G_KX^T = G_A^T LI (only the transposed G_KX is ever read);
//             Pm = Phi(L^T G_L) = Phi(-G_A A^T)   (tril(L^T tril(X)) = tril(L^T X) and L^T LI^T = I);
//             G_Kzz = LI^T (Pm LI): W = Pm LI is lower (M^3 / 3), S = LI^T W (2 M^3 / 3);  one fused pass turns G_Kzz and
//             G_KX into G_s, G_l, G_Z (kernel values recomputed from the LDS copies of Z and X) and applies Adam to Z
//   Adam    : torch.optim.Adam defaults (beta 0.9 / 0.999, eps 1e-8), lr 0.1 on {Z, m, tril(LS), c, rho_s, rho_l}
// Inducing points Z and training points X live transposed in LDS ([d][i]) so that a thread that owns column j reads
// its own point conflict-free and the row point as an LDS broadcast.
//
// Built a third time with GAPRO_DEBUG_TU (svgp_fit_debug.hip -> libgapro_hip_debug.so): only the debug entry points
// of include/gapro_hip_debug.h (product-engine bench, MFMA lane-map self test, counter-calibration streams).
#include <math.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "common.h"
#include "fit_layout.h"
#include "fit_math.h"
#include "epilogue.h"
#ifdef GAPRO_DEBUG_TU
#include "../../include/gapro_hip_debug.h"
#include "mfma64.h"
#endif

void gapro_launch_fit_large(hipStream_t stream, int n_fits, int feat_dim, const float* d_feats_spp, const int* d_idx,
                            const gapro_fit_desc* d_descs, const double* d_init_mean, const gapro_fit_options& opt,
                            double* d_workspace, float* d_probs, float* d_probs_new, unsigned char* d_labels,
                            float* d_mu, float* d_var, int* d_fit_status, double* d_fit_loss);

namespace {
using namespace gapro_fit;

#ifndef GAPRO_NT
#define GAPRO_NT 512
#endif
constexpr int NT = GAPRO_NT;  // threads per fit (the small-fit translation unit builds this file with 256)
constexpr int NW = NT / 64;   // waves per fit
constexpr int kMaxMpLds = 512;          // largest padded M the LDS-staged kernel takes
constexpr int kMaxDynLds = 150 * 1024;  // dynamic LDS budget (160 KiB per CU minus the static part)
constexpr int kRedSlots = 8;            // values reduced across row groups per pass
#ifndef GAPRO_WAVES_PER_SIMD
#define GAPRO_WAVES_PER_SIMD 4           // 2 workgroups of 8 waves per CU -> 128 VGPRs per lane
#endif
constexpr int kWavesPerSimd = GAPRO_WAVES_PER_SIMD;
#ifndef GAPRO_TU4_MIN
#define GAPRO_TU4_MIN 320
#endif
// 64 x 64 wave tiles from this M_p on (one-per-CU build, D = 6; multiples of 32).  Round 3, 512 fits: M_p = 320 +2.4 %,
// 288 -2.2 % against 32 x 32 tiles; with the copy-free forms, one per CU, M_p = 192 / 224: -15 % (and one per CU with
// 32 x 32 tiles is 7 .. 10 % behind two per CU there).
constexpr int kTu4MinMp = GAPRO_TU4_MIN;
constexpr int kKminMaxMp = 256;         // largest M_p with the copy-free product forms (fit_body's KMIN)
#ifndef GAPRO_GEMM_RING
#define GAPRO_GEMM_RING 2
#endif

using gapro_mfma::d4;
// LDS pointers carry their address space explicitly: ds_read/ds_write instead of flat accesses, and no
// generic->local casts for the optimiser to trip over.
typedef __attribute__((address_space(3))) double ldsd;
// Workspace pointers are typed as global memory: global_load/global_store with counted vmcnt waits (a
// generic pointer compiles to flat_* accesses, whose completion is unordered and forces vmcnt(0)).
typedef __attribute__((address_space(1))) double gd;

// numpy.polynomial.hermite.hermgauss(20): positive nodes (ascending) and their weights; the rule is
// symmetric.  Printed with repr() from NumPy 2.2.
__constant__ double c_gh_t[10] = {0.24534070830090124, 0.7374737285453944, 1.234076215395323,  1.7385377121165861,
                                  2.2549740020892757,  2.7888060584281305, 3.3478545673832163, 3.944764040115625,
                                  4.603682449550744,   5.387480890011233};
__constant__ double c_gh_w[10] = {0.4622436696006101,     0.28667550536283415,    0.1090172060200233,
                                  0.024810520887463643,   0.0032437733422378567,  0.00022833863601635365,
                                  7.80255647853206e-06,   1.0860693707692782e-07, 4.3993409922731747e-10,
                                  2.2293936455341447e-13};


// ---- workspace layout (doubles); identical to svgp_fit_large.hip -----------------------------------
// (enums, Layout and make_layout: fit_layout.h, shared by every fit kernel)

// dynamic LDS of the staged kernel: Zt[D][Mp] | Pt[D][Mp] | scratch
constexpr int kTileDoubles = NW * 16 * 17;  // per-wave transpose tiles at the start of the scratch
constexpr int kFuseMaxMp = 128;            // up to this size column sums are fused into GEMM epilogues
// workgroup-tiled products (gemm_wg, beyond kFuseMaxMp): operand chunks in LDS
constexpr int kWgTile = 128;                  // output tile edge of the workgroup
constexpr int kWgKC = 8;                      // k rows per chunk
constexpr int kWgRow = kWgTile + 16;          // LDS row stride of a chunk (doubles): the two k rows a 32-lane group of a
                                              // fragment read touches start 128 bytes apart modulo the 256-byte bank row
constexpr int kWgStage = 2 * kWgKC * kWgRow;  // doubles per LDS stage: P chunk | Q chunk
constexpr int kWgRingDoubles = 2 * kWgStage;  // two stages
inline __host__ __device__ int part_doubles(int Mp) { return Mp <= kFuseMaxMp ? 3 * (Mp / 16) * Mp : 0; }
inline __host__ __device__ int scratch_doubles(int Mp) {
  const int a = kRedSlots * NT;                   // cross-group reductions / quadrature partials
  const int b = Mp * 17 + 64 * 17;                // Cholesky block column (row stride 17) + slack
  // transpose tiles + per-tile-row column sums (M_p <= kFuseMaxMp), or the products' operand ring (beyond it; the
  // transpose tiles then sit inside the ring's second stage, which is idle while a tile's epilogue runs)
  static_assert(kTileDoubles <= kWgStage, "the transpose tiles alias one stage of the operand ring");
  const int c = Mp <= kFuseMaxMp ? kTileDoubles + part_doubles(Mp) : kWgRingDoubles;
  int m = a > b ? a : b;
  return m > c ? m : c;
}
inline __host__ __device__ long long staged_lds_bytes(int m, int d) {
  const int Mp = gapro_pad_m(m, d);
  return 8LL * (2LL * d * Mp + scratch_doubles(Mp));
}
// (gapro_pad_m decides from gapro_staged_lds_bytes_mp, common.h: the two must agree beyond kFuseMaxMp)
inline __host__ __device__ bool staged_ok(int m, int d) {
  return gapro_pad_m(m, d) <= kMaxMpLds && d <= 32 && staged_lds_bytes(m, d) <= kMaxDynLds;
}


struct Fit {
  int M, T, D, Mp;
  int M1;  // train_y = -1 for the first M1 training points, +1 for the others (gaussian_process_utils.py:396-398)
  gd* mat[B_COUNT];
  gd* vec[V_COUNT];
  gd *X, *Z, *mZ, *vZ, *gZ, *Xt, *dinv, *dinvT, *scal;
};

#ifdef GAPRO_PROFILE
constexpr int kProfSlots = 28;
#endif
struct Shared {
  Fit f;
#ifdef GAPRO_PROFILE
  unsigned long long prof[kProfSlots];
  unsigned long long t_last;
  unsigned long long t_start;
#endif
  double red[NW];
  double dblk[16 * 17];
  double dinv[16 * 17];
  double c, rho_s, rho_l, s, ell, inv_l2;
  int status;
  int chol_bad;  // a pivot of the factorisation in progress was not positive (see factorize: the jitter retries)
};
__shared__ Shared g_sh;  // one fit per workgroup

#ifdef GAPRO_PROFILE
__device__ inline void prof_stamp(int id) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = wall_clock64();
    g_sh.prof[id] += t - g_sh.t_last;
    g_sh.t_last = t;
  }
}
#else
__device__ inline void prof_stamp(int) {}
#endif

// ---- small helpers ---------------------------------------------------------------------------------
__device__ inline double softplus(double x) { return log1p(exp(-fabs(x))) + fmax(x, 0.0); }
__device__ inline double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }

// log Phi(z) and r(z) = phi(z)/Phi(z), both tails stable.  The oracle (oracle/svgp_oracle.py) branches on the sign of z
// -- erfcx for z < 0, erfc for z >= 0 -- and the ten quadrature nodes of a point straddle zero, so a wave ran both
// branches (erfcx, log, erfc, log1p, exp: the likelihood phase was 6 .. 10 % of a step of the strip kernels, 4 .. 7 % of
// the staged kernel's).  Here both signs share t = erfcx(|z| / sqrt 2) and e = exp(-z^2 / 2):
//   z <  0:  Phi = e t / 2          log Phi = log(t / 2) - z^2 / 2     r = sqrt(2 / pi) / t
//   z >= 0:  Phi = 1 - e t / 2      log Phi = log(1 - e t / 2)         r = e / (sqrt(2 pi) Phi)
// -- three library calls and two selects.  erfc(x) = e t to the last bit or two; log(1 - tail) instead of log1p(-tail)
// is absolutely accurate to 1e-16, and log Phi only enters the reported ELBO value.
__device__ inline void log_ndtr_ratio(double z, double* lp, double* r) {
  const double rs2 = 0.70710678118654752440;
  const double t = gapro_fit_math::lik_erfcx(fabs(z) * rs2);
  const double hz2 = 0.5 * z * z;
  const double e = gapro_fit_math::rbf_exp(-hz2);
  const bool neg = z < 0.0;
  const double phi_pos = 1.0 - 0.5 * e * t;  // Phi(z) for z >= 0
  *lp = log(neg ? 0.5 * t : phi_pos) - (neg ? hz2 : 0.0);
  *r = (neg ? 0.79788456080286535588 : e * 0.39894228040143267794) / (neg ? t : phi_pos);  // one division
}

// r(z) alone, the same bits as log_ndtr_ratio's: log Phi only enters the ELBO VALUE, which is reported after the last
// step and read by nobody before it -- 49 of 50 steps need no log (a quarter of the instructions of an evaluation)
__device__ inline double ndtr_ratio(double z) {
  const double t = gapro_fit_math::lik_erfcx(fabs(z) * 0.70710678118654752440);
  const double e = gapro_fit_math::rbf_exp(-0.5 * z * z);
  const bool neg = z < 0.0;
  return (neg ? 0.79788456080286535588 : e * 0.39894228040143267794) / (neg ? t : 1.0 - 0.5 * e * t);
}
// one symmetric pair of Gauss-Hermite nodes (t, weight w) of a point: sums for E (want_e only), dE/dmu, dE/dvar
__device__ inline void gh_pair(double y, double mu, double sd, double t, double w, bool want_e, double* E, double* dmu,
                               double* dvar) {
#ifdef GAPRO_LIK_ALWAYS_LOG  // A/B builds
  want_e = true;
#endif
  if (want_e) {  // workgroup-uniform
    double lp, r;
    log_ndtr_ratio(y * (mu - sd * t), &lp, &r);
    *E += w * lp; *dmu += w * r; *dvar -= w * t * r;
    log_ndtr_ratio(y * (mu + sd * t), &lp, &r);
    *E += w * lp; *dmu += w * r; *dvar += w * t * r;
  } else {
    double r = ndtr_ratio(y * (mu - sd * t));
    *dmu += w * r; *dvar -= w * t * r;
    r = ndtr_ratio(y * (mu + sd * t));
    *dmu += w * r; *dvar += w * t * r;
  }
}

// value of `v` in lane `lane` (wave-uniform, compile-time after unrolling): v_readlane, no LDS crossbar
__device__ inline double lane_bcast(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Function arguments of non-kernel functions arrive in VGPRs; these make wave-uniform values scalar again
// so that loop control and address arithmetic run on the scalar unit.
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ inline T* uni_ptr(T* p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}

// Which fit of its kernel's list this workgroup runs.  Workgroup b of a launch runs on XCD b % 8 whatever that XCD is
// busy with, so with fit = blockIdx.x an XCD's share of a kernel's fits is fixed before the launch starts, and the XCDs
// ended a launch up to 55 ms apart (tools/fit_timeline.py) even with their shares balanced by cost.  With a ticket
// counter (one per kernel of a launch, zeroed by gapro_svgp_fit_batch) a workgroup takes the next fit of the
// longest-first list when it STARTS; the launcher over-subscribes the grid, the workgroups left without a fit exit at
// once, and an XCD that frees up earlier simply starts more of them.  A fit's result does not depend on who runs it.
__device__ inline int claim_fit(unsigned* ticket) {
  if (!ticket) return blockIdx.x;
  __shared__ int s_claim;
  if (threadIdx.x == 0) s_claim = (int)atomicAdd(ticket, 1u);
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(s_claim);
}

__device__ inline double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Deterministic block sum (fixed tree), result broadcast to every thread.
__device__ inline double block_sum(double v) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) g_sh.red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < NW; ++w) t += g_sh.red[w];
  return t;
}

// squared distance between staged points: At[d][i] and Bt[d][j], leading dimension Mp
__device__ inline double sqdist_t(const ldsd* At, int i, const ldsd* Bt, int j, int D, int Mp) {
  double s = 0.0;
  for (int d = 0; d < D; ++d) {
    const double t = At[d * Mp + i] - Bt[d * Mp + j];
    s += t * t;
  }
  return s;
}

// stage n points [n][D] (global, row-major) transposed into LDS dst[D][Mp]; columns >= n are zeroed
__device__ inline void stage_points_t(ldsd* dst, const gd* src, int n, int D, int Mp) {
  for (int e = threadIdx.x; e < D * Mp; e += NT) {
    const int d = e / Mp, i = e - d * Mp;
    dst[e] = i < n ? src[(size_t)i * D + d] : 0.0;
  }
}

// ---- TN-form MFMA product ---------------------------------------------------------------------------
//   C[i][j] = sum_{k in [klo,khi)} P[k][i] * Q[k][j] (* qscale[k] if SCALE),  ld = leading dimension
// Each wave owns (16 TU) x (16 TU) output tiles, round-robin; `lower_only` enumerates tiles ti >= tj.
// kr(i0, j0, &klo, &khi) restricts the contraction (multiples of 16) to where triangular operands are
// non-zero.  epi(i0, j0, tile) consumes one 16x16 result tile in MFMA C layout.  Fragments of the next
// 8-deep block are loaded before the MFMAs of the current one are issued.
// MFMA f64 16x16x4 lane maps (cdna_hip_programming.md section 3): A[i = l & 15][k = l >> 4],
// B[k = l >> 4][j = l & 15], C/D register r -> row (l >> 4) + 4 r, col l & 15.
// ORD: the order in which the tiles are enumerated, heaviest contraction range first for the product's kr (a wave
// takes the tiles of that order in serpentine rounds: 0..7, 7..0, ...; with triangular operands a round-robin deal in
// row-major order leaves the slowest wave up to 1.8x the mean work, e.g. the same tile column for every tile of a wave
// when a row has 8 tiles).  Which wave computes a tile does not change the tile: results are bit-identical.
enum { ORD_ROWMAJOR = 0,   // equal ranges, or ranges shrinking with the tile row
       ORD_ROWS_DESC = 1,  // ranges growing with the tile row: last row first
       ORD_COLMAJOR = 2,   // ranges shrinking with the tile column
       ORD_SHELLS = 3 };   // ranges shrinking with max(row, column) (square tile grids): shells m = 0, 1, ...
// TRIM: the contraction index runs over inducing / training points; rows >= M of both operands are padding whose
// products with every valid output vanish, so the range stops at M rounded up to the register block (M = 230 padded to
// 256: 9 % fewer loads and MFMAs; valid outputs keep their bits, x + 0 * y = x).
// TU = 4 (the staged kernel's full-register build, M_p >= 352): 64 x 64 wave tiles, i.e. 8 instead of 16 operand columns
// fetched per 16 x 16 output block -- no fragment of these products is ever found in L2 (the hit rate does not move
// with the tile order; 256 concurrent fits stream ~100 MB per Adam step each at M = 384, together the ~6.3 TB/s the
// HBM delivers), so the bytes per block are what a product costs.  The extents are then given in 32 x 32 units and an
// odd count leaves a last row / column of 32 x 32 tiles, dealt after the full ones.
// PK / QK (round 3): the operand is stored with the contraction index along its ROWS' contiguous direction -- P[i][k]
// instead of P[k][i], Q[j][k] instead of Q[k][j] -- so that products of the forms X Y^T and X Y run on the matrices as
// they are and nobody has to write (and read back) a transposed copy: A^T, B^T, G_A^T, Pm^T each cost a full matrix
// of HBM writes per Adam step plus a pass through the waves' LDS transpose tiles.  A lane fetches two consecutive k of
// its row with one 16-byte load (64 contiguous bytes per row and instruction); with any operand in this form the k of
// MFMA step e = 0, 1 of a block of 8 is k0 + 2 (lane >> 4) + e for BOTH operands (a k-major operand then reads rows
// k0 + 2 lq and k0 + 2 lq + 1) -- a fixed permutation of the contraction order inside a block, so these products do not
// have the bits of the k-major form, but every product has ONE form in all kernels' variants that must agree.
typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) d2 lds_d2;
typedef __attribute__((address_space(1))) d2 g_d2;

// Round 6 (profiles/r06_spill_traffic.md): every out-of-line product saved and restored the 41 callee-saved VGPRs it
// uses -- ~410 scratch stores and as many loads per wave and Adam step; the stores are 9 % of the bytes the two-per-CU
// staged fits write at M = 160 (the reloads are served by L2), and those fits sit on the memory roof.  Inlining the products into the KERNEL was 3 % slower (the
// register allocator then carries the kernel's long-lived values through every k loop); inlining them into ONE
// out-of-line function per Adam step (step_fn in fit_body, through gemm_tn_in) pays the callee-saved traffic once per
// step: M = 160 -3.3 %, 200 -1.7 %, 256 +0.7 %, 320 / 384 +-0 in time, bit-identical.  The strip kernels' tail products keep
// the out-of-line form (their caller holds 80 accumulator registers across them).  -DGAPRO_NO_STEP_FN restores the
// per-product calls everywhere.
#ifndef GAPRO_NO_STEP_FN
#define GAPRO_STEP_FN
#endif
template <int TU, bool SCALE, int KS = 2, int ORD = ORD_ROWMAJOR, bool TRIM = false, int PK = 0, int QK = 0,
          typename KRange, typename Epi>
__device__ __forceinline__ void gemm_tn_body(int mo_tiles, int no_tiles, bool lower_only, const gd* __restrict__ P,
                                          const gd* __restrict__ Q, int ld, const gd* __restrict__ qscale, KRange kr,
                                          Epi epi) {
  mo_tiles = uni(mo_tiles);
  no_tiles = uni(no_tiles);
  lower_only = uni((int)lower_only) != 0;
  ld = uni(ld);
  P = uni_ptr(P);
  Q = uni_ptr(Q);
  qscale = uni_ptr(qscale);
  const int wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int lr = lane & 15, lq = lane >> 4;
  // One output tile of TV x TV blocks at (i0, j0).  KSV = k-steps (of 4) per register block; two blocks alternate (one
  // in flight): 2 under the 128-VGPR budget of the LDS-staged kernel, 4 in the strip kernels' tail (256 VGPRs: +1.5 %;
  // in the staged kernel -3..-6 %), 1 with the 128 accumulator registers of a 64 x 64 tile.  Everything between the
  // issue of a block's loads and its MFMAs is straight-line code: s_waitcnt counts memory operations in issue order,
  // and behind a join of two paths ("prefetch only if there is a next block") the compiler falls back to waiting for
  // everything, i.e. for the block it has just requested -- no load would ever overlap an MFMA.  So the steady-state
  // loop has no guard in its body (the last one or two blocks are peeled off behind it), and the column scale of the
  // SCALE form is applied when a block is consumed, not when it is loaded (a multiply at load time is a wait at load
  // time).
  auto tile = [&](auto tv_tag, int i0, int j0) {
    constexpr int TV = decltype(tv_tag)::value;
    constexpr bool MINOR = PK || QK;
    constexpr int KSV = MINOR ? 2 : (TV >= 4 ? 1 : KS), KB = 4 * KSV;
    int klo, khi;
    kr(i0, j0, &klo, &khi);
    klo = uni(klo);
    khi = uni(khi);
    if (TRIM) {
      const int kmax = uni((g_sh.f.M + 7) / 8 * 8);  // a multiple of every KB in use but the strip tail's 16 (M_p there)
      khi = khi < kmax ? khi : kmax;
    }
    d4 acc[TV][TV];
#pragma unroll
    for (int u = 0; u < TV; ++u)
#pragma unroll
      for (int v = 0; v < TV; ++v) acc[u][v] = (d4){0.0, 0.0, 0.0, 0.0};
    const gd* pbase = PK ? P + (size_t)(i0 + lr) * ld + 2 * lq : P + (size_t)(MINOR ? 2 * lq : lq) * ld + i0 + lr;
    const gd* qbase = QK ? Q + (size_t)(j0 + lr) * ld + 2 * lq : Q + (size_t)(MINOR ? 2 * lq : lq) * ld + j0 + lr;
    double a0[KSV][TV], b0[KSV][TV], a1[KSV][TV], b1[KSV][TV];
    double s0[KSV], s1[KSV];
    auto load_block = [&](int k, double (&a)[KSV][TV], double (&b)[KSV][TV], double (&sc)[KSV]) {
      if constexpr (MINOR) {
#pragma unroll
        for (int u = 0; u < TV; ++u) {
          if constexpr (PK) {
            const d2 t = *(const g_d2*)(pbase + (size_t)(16 * u) * ld + k);
            a[0][u] = t[0];
            a[1][u] = t[1];
          } else {
            a[0][u] = pbase[(size_t)k * ld + 16 * u];
            a[1][u] = pbase[(size_t)(k + 1) * ld + 16 * u];
          }
        }
#pragma unroll
        for (int v = 0; v < TV; ++v) {
          if constexpr (QK) {
            const d2 t = *(const g_d2*)(qbase + (size_t)(16 * v) * ld + k);
            b[0][v] = t[0];
            b[1][v] = t[1];
          } else {
            b[0][v] = qbase[(size_t)k * ld + 16 * v];
            b[1][v] = qbase[(size_t)(k + 1) * ld + 16 * v];
          }
        }
        if (SCALE) {
          sc[0] = qscale[k + 2 * lq];
          sc[1] = qscale[k + 2 * lq + 1];
        }
      } else {
#pragma unroll
        for (int s = 0; s < KSV; ++s) {
          const gd* pr = pbase + (size_t)(k + 4 * s) * ld;
          const gd* qr = qbase + (size_t)(k + 4 * s) * ld;
#pragma unroll
          for (int u = 0; u < TV; ++u) a[s][u] = pr[16 * u];
#pragma unroll
          for (int v = 0; v < TV; ++v) b[s][v] = qr[16 * v];
          if (SCALE) sc[s] = qscale[k + 4 * s + lq];
        }
      }
    };
    auto mma_block = [&](double (&a)[KSV][TV], double (&b)[KSV][TV], double (&sc)[KSV]) {
#pragma unroll
      for (int s = 0; s < KSV; ++s) {
        double bs[TV];
#pragma unroll
        for (int v = 0; v < TV; ++v) bs[v] = SCALE ? b[s][v] * sc[s] : b[s][v];
#pragma unroll
        for (int u = 0; u < TV; ++u)
#pragma unroll
          for (int v = 0; v < TV; ++v)
            acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][u], bs[v], acc[u][v], 0, 0, 0);
      }
    };
    if (klo < khi) {  // khi - klo is a multiple of KB (tile-aligned ranges, trimmed to a multiple of 8)
      load_block(klo, a0, b0, s0);
      int k = klo;
#pragma nounroll
      for (; k + 2 * KB < khi; k += 2 * KB) {  // steady state: both prefetches are real, no guard in the body
        load_block(k + KB, a1, b1, s1);
        mma_block(a0, b0, s0);
        load_block(k + 2 * KB, a0, b0, s0);
        mma_block(a1, b1, s1);
      }
      if (k + KB < khi) {  // two blocks left
        load_block(k + KB, a1, b1, s1);
        mma_block(a0, b0, s0);
        mma_block(a1, b1, s1);
      } else {  // one block left
        mma_block(a0, b0, s0);
      }
    }
    run_epilogue<TV * TV>(epi, [&](int b, int* i, int* j) { *i = i0 + 16 * (b / TV); *j = j0 + 16 * (b % TV); },
                          [&](int b) -> const d4& { return acc[b / TV][b % TV]; });
  };
  // full tiles: fo x fn of TU x TU blocks.  TU >= 2 takes its extents in HALF tiles (TU = 2: 16 x 16, TU = 4: 32 x 32)
  // and an odd count leaves a last row / column of half-size tiles: M_p need not be a multiple of the tile (round 3:
  // M_p in steps of 16 up to 336, see gapro_pad_m)
  constexpr int TE = TU >= 2 ? TU / 2 : TU;  // blocks per side of an edge tile
  const int fo = TU >= 2 ? mo_tiles / 2 : mo_tiles, fn = TU >= 2 ? no_tiles / 2 : no_tiles;
  const int odd_i = TU >= 2 ? (mo_tiles & 1) : 0, odd_j = (TU >= 2 && !lower_only) ? (no_tiles & 1) : 0;
  const int nfull = lower_only ? fo * (fo + 1) / 2 : fo * fn;
  const int nrow = odd_i ? (lower_only ? mo_tiles : no_tiles) : 0;  // edge row: tiles (mo_tiles - 1, 0 ..) in half tiles
  const int ncol = odd_j ? mo_tiles - odd_i : 0;                    // edge column: tiles (0 .., no_tiles - 1) above it
  const int ntiles = nfull + nrow + ncol;
  const int rounds = (ntiles + NW - 1) / NW;
#pragma nounroll
  for (int q = 0; q < rounds; ++q) {
    const int t = q * NW + ((q & 1) ? NW - 1 - wave : wave);
    if (t >= ntiles) continue;
    if (t >= nfull) {  // half-size edge tiles, a quarter of a full tile's work each
      const int e = t - nfull;
      const int ih = e < nrow ? mo_tiles - 1 : e - nrow, jh = e < nrow ? e : no_tiles - 1;
      tile(std::integral_constant<int, TE>{}, 16 * TE * ih, 16 * TE * jh);
      continue;
    }
    int ti, tj;
    if (lower_only) {
      ti = 0;
      while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
      tj = t - ti * (ti + 1) / 2;
    } else if (ORD == ORD_ROWS_DESC) {
      ti = t / fn;
      tj = t - ti * fn;
      ti = fo - 1 - ti;
    } else if (ORD == ORD_COLMAJOR) {
      tj = t / fo;
      ti = t - tj * fo;
    } else if (ORD == ORD_SHELLS) {
      int m = 0;
      while ((m + 1) * (m + 1) <= t) ++m;
      const int r = t - m * m;
      ti = r <= m ? m : r - m - 1;
      tj = r <= m ? r : m;
    } else {
      ti = t / fn;
      tj = t - ti * fn;
    }
    tile(std::integral_constant<int, TU>{}, 16 * TU * ti, 16 * TU * tj);
  }
}

// the product as a function of its own (the strip kernels' tail products, the debug engines) ...
template <int TU, bool SCALE, int KS = 2, int ORD = ORD_ROWMAJOR, bool TRIM = false, int PK = 0, int QK = 0,
          typename KRange, typename Epi>
__device__ __noinline__ void gemm_tn(int mo_tiles, int no_tiles, bool lower_only, const gd* __restrict__ P,
                                     const gd* __restrict__ Q, int ld, const gd* __restrict__ qscale, KRange kr,
                                     Epi epi) {
  gemm_tn_body<TU, SCALE, KS, ORD, TRIM, PK, QK>(mo_tiles, no_tiles, lower_only, P, Q, ld, qscale, kr, epi);
}
// ... and inlined into its caller (the staged kernel's step function)
template <int TU, bool SCALE, int KS = 2, int ORD = ORD_ROWMAJOR, bool TRIM = false, int PK = 0, int QK = 0,
          typename KRange, typename Epi>
__device__ __forceinline__ void gemm_tn_in(int mo_tiles, int no_tiles, bool lower_only, const gd* __restrict__ P,
                                        const gd* __restrict__ Q, int ld, const gd* __restrict__ qscale, KRange kr,
                                        Epi epi) {
#ifdef GAPRO_STEP_FN
  gemm_tn_body<TU, SCALE, KS, ORD, TRIM, PK, QK>(mo_tiles, no_tiles, lower_only, P, Q, ld, qscale, kr, epi);
#else
  gemm_tn<TU, SCALE, KS, ORD, TRIM, PK, QK>(mo_tiles, no_tiles, lower_only, P, Q, ld, qscale, kr, epi);
#endif
}

// ---- TN-form MFMA product, workgroup-tiled through an LDS ring (round 3) ---------------------------------
// The same contract as gemm_tn at TU = 1 -- extents in 16 x 16 blocks, kr(i0, j0) = the contraction range of the block
// at (i0, j0), epi(i0, j0, block) per block, every block accumulated over ascending k in steps of 4 -- and therefore
// the same bits.  What changes is where the operand fragments come from.  With one tile per wave fed from global
// memory every wave fetches its own fragments: 4 FLOP per byte at 32 x 32, and the 256 .. 512 concurrent fits of a
// launch, whose working sets (17 matrices each) hit neither L2 nor the Infinity Cache, move 6.3 TB/s at M = 256 -- the
// staged kernel sat ON the HBM roof (profiles/r03_probes.md).  Here the eight waves of the workgroup share one
// 128 x 128 output tile: an 8-row chunk of both operands (8 KiB each) is fetched ONCE per workgroup, one 16-byte load
// per thread and operand (wave w fetches row w: 1 KiB contiguous), held in registers for two iterations (the
// prefetch distance), written to one of two LDS stages and read from there as MFMA fragments by every wave: 16 FLOP
// per byte of global traffic.  One barrier per chunk.  Wave w owns the 32 x 64 piece at rows 32 r, columns 64 (w >> 2)
// of the tile, r = w & 3 for the first four waves and 3 - (w & 3) for the others: waves w and w + 4 share a SIMD, so
// with triangular operands every SIMD gets a long and a short contraction range.  A piece skips the chunks outside the
// hull of its blocks' ranges (the extra rows inside the hull multiply structural zeros of a triangular operand: x + 0 y
// = x, the bits stay).
// LDS image of a chunk: row k at k * kWgRow doubles with kWgRow = 128 + 16 (padded rows, no swizzle): the two k rows a
// 32-lane group of a fragment read touches start 128 bytes apart modulo the 256-byte bank row, i.e. they land in
// different halves of it (conflict-free ds_read_b64).
template <bool SCALE, bool TRIM, int PF, typename KRange, typename Epi>
__device__ __noinline__ void gemm_wg(int rows16, int cols16, bool lower_only, const gd* __restrict__ P,
                                     const gd* __restrict__ Q, int ld, const gd* __restrict__ qscale, KRange kr,
                                     Epi epi, ldsd* ring) {
  static_assert(NT == 512 || !sizeof(Epi), "gemm_wg: eight waves, one k row of a chunk per wave");
  static_assert(PF == 2 || PF == 4, "gemm_wg: register stages");
  rows16 = uni(rows16);
  cols16 = uni(cols16);
  lower_only = uni((int)lower_only) != 0;
  ld = uni(ld);
  P = uni_ptr(P);
  Q = uni_ptr(Q);
  qscale = uni_ptr(qscale);
  const int wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const int rows = 16 * rows16, cols = 16 * cols16;
  const int nti = (rows + kWgTile - 1) / kWgTile, ntj = (cols + kWgTile - 1) / kWgTile;
  const int wi = wave < 4 ? wave : 7 - wave, wj = wave >> 2;
  const int kmax = TRIM ? uni((g_sh.f.M + 7) / 8 * 8) : ld;
  // loader: wave w moves k row w of a chunk, lane l the 16 bytes at columns 2 l, 2 l + 1 of the tile
  const int lcol = 2 * lane;
  const int lds_wr = wave * kWgRow + lcol;
  // fragment reads: element (k = 4 s + lq, column c + lr) of a chunk; block, k-step and stage are immediate offsets
  const int fa = lq * kWgRow + 32 * wi + lr;
  const int fb = kWgKC * kWgRow + lq * kWgRow + 64 * wj + lr;
#pragma nounroll
  for (int ti = 0; ti < nti; ++ti) {
#pragma nounroll
    for (int tj = 0; tj < (lower_only ? ti + 1 : ntj); ++tj) {
      const int I0 = kWgTile * ti, J0 = kWgTile * tj;
      // Contraction range of the workgroup tile and of this wave's piece: the hull of their blocks' ranges.  Every kr
      // of the fit is monotone (lo and hi never decrease with the block row or the block column), so a rectangle's
      // hull is [lo of its first block, hi of its last block]: two calls instead of one per block (64 per tile and
      // wave cost ~3 us of scalar work per tile, a quarter of a tile's time at M_p = 256).  For a lower-triangular
      // output the blocks above the diagonal are not part of it; leaving them in the hull only widens it, and what a
      // wider hull adds are products with structural zeros.
      int klo = 1 << 30, khi = 0, plo = 1 << 30, phi = 0;
      unsigned on_mask = 0;  // bit 4 u + v: block (u, v) of this wave's piece is part of the output
      if (rows - I0 >= kWgTile && cols - J0 >= kWgTile) {  // a full tile: every piece, every block inside the matrix
        int lo, hi, d;
        kr(I0, J0, &lo, &d);
        kr(I0 + kWgTile - 16, J0 + kWgTile - 16, &d, &hi);
        hi = hi < kmax ? hi : kmax;
        if (lo < hi) { klo = lo; khi = hi; }
        const int pi0 = I0 + 32 * wi, pj0 = J0 + 64 * wj;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (!(lower_only && pi0 + 16 * u < pj0 + 16 * v)) on_mask |= 1u << (4 * u + v);
        if (on_mask) {
          kr(pi0, pj0, &lo, &d);
          kr(pi0 + 16, pj0 + 48, &d, &hi);
          hi = hi < kmax ? hi : kmax;
          if (lo < hi) { plo = lo; phi = hi; }
        }
      } else {  // a tile at the matrix edge: block by block
#pragma nounroll
        for (int bi = 0; bi < kWgTile / 16; ++bi) {
          const int ib = I0 + 16 * bi;
          if (ib >= rows) break;
#pragma nounroll
          for (int bj = 0; bj < kWgTile / 16; ++bj) {
            const int jb = J0 + 16 * bj;
            if (jb >= cols) break;
            if (lower_only && ib < jb) continue;
            const bool mine = (bi >> 1) == wi && (bj >> 2) == wj;
            if (mine) on_mask |= 1u << (4 * (bi & 1) + (bj & 3));  // also with an empty range: epi sees a zero block
            int lo, hi;
            kr(ib, jb, &lo, &hi);
            hi = hi < kmax ? hi : kmax;
            if (lo >= hi) continue;
            klo = lo < klo ? lo : klo;
            khi = hi > khi ? hi : khi;
            if (mine) {
              plo = lo < plo ? lo : plo;
              phi = hi > phi ? hi : phi;
            }
          }
        }
      }
      klo = uni(klo) & ~(kWgKC - 1);
      khi = uni(khi);
      plo = uni(plo) & ~(kWgKC - 1);
      phi = uni(phi);
      on_mask = (unsigned)uni((int)on_mask);
      const int nch = klo < khi ? (khi - klo + kWgKC - 1) / kWgKC : 0;  // workgroup-uniform
      d4 acc[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = (d4){0.0, 0.0, 0.0, 0.0};
      if (nch > 0) {
        const bool pin = I0 + lcol < ld, qin = J0 + lcol < ld;  // columns beyond the matrix: blocks that are dropped
        const gd* pg = P + (size_t)wave * ld + (pin ? I0 + lcol : 0);
        const gd* qg = Q + (size_t)wave * ld + (qin ? J0 + lcol : 0);
        // Register stages: chunk n of the tile waits in stage n % PF from its request until it goes to LDS, PF - 1 .. PF
        // iterations later (the chunk index is clamped to the last one, so that every load is unconditional and the
        // waits stay counted).  An iteration cannot be shorter than the memory latency / PF: with one workgroup per CU
        // and two stages the loop ran at the HBM latency, not at the matrix rate.  With SCALE the row's scale rides
        // along (one address per wave) and is applied when the row goes to LDS: b * scale, the same product the
        // per-wave form computes on every fragment.
        d2 rp[PF], rq[PF];
        double rs[PF];
        auto gload = [&](int n, d2& p, d2& q, double& sc) {
          n = n < nch ? n : nch - 1;
          const size_t o = (size_t)(klo + kWgKC * n) * ld;
#ifdef GAPRO_WG_NOLOAD  // timing experiment: no operand traffic (results are wrong)
          p = (d2){1.0 + (double)o, 2.0};
          q = (d2){0.5, 0.25};
#else
          p = *(const g_d2*)(pg + o);
          q = *(const g_d2*)(qg + o);
#endif
          if (SCALE) sc = qscale[klo + kWgKC * n + wave];
        };
        auto lstore = [&](int stage, const d2& p, const d2& q, double sc) {
          ldsd* base = ring + stage * kWgStage + lds_wr;
          *(lds_d2*)base = p;
          *(lds_d2*)(base + kWgKC * kWgRow) = SCALE ? (d2){q[0] * sc, q[1] * sc} : q;
        };
        // Fragments of one k-step: 2 of A, 4 of B.  Two sets alternate, software-pipelined across the barrier: the
        // second step's set is requested before the first step's MFMAs, the NEXT chunk's first step right behind the
        // barrier, under the second step's MFMAs -- the matrix pipe never waits for an LDS round trip, and a barrier
        // costs what the waves' skew costs (tools/wgloop_peak.py: 47 -> 64 TFLOP/s for this loop with every CU running
        // it).  Every block of a piece that has work is computed, also the ones whose result is dropped (columns
        // beyond the matrix, blocks above the diagonal of a lower-triangular output): a guard per block costs two
        // taken branches per MFMA, and a block's result depends on its own accumulator only.
        double fa0[2], fb0[4], fa1[2], fb1[4];
        auto rd = [&](int stage, int s, double (&a)[2], double (&b)[4]) {
          const ldsd* st = ring + stage * kWgStage + 4 * s * kWgRow;
#pragma unroll
          for (int v = 0; v < 4; ++v) b[v] = st[fb + 16 * v];
#pragma unroll
          for (int u = 0; u < 2; ++u) a[u] = st[fa + 16 * u];
        };
        auto mm = [&](int k0, const double (&a)[2], const double (&b)[4]) {
          if (k0 < plo || k0 >= phi) return;  // wave-uniform: the piece's own contraction range
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v)
#ifdef GAPRO_WG_NOMFMA  // timing experiment: no matrix work (results are wrong)
              acc[u][v][0] += a[u] * b[v];
#else
              acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[v], acc[u][v], 0, 0, 0);
#endif
        };
        // chunk c: CUR = its LDS stage, SLOT = the register stage holding chunk c + 1, refilled with chunk c + 1 + PF
        auto chunk = [&](auto cur_tag, auto slot_tag, int c) {
          constexpr int CUR = decltype(cur_tag)::value, SLOT = decltype(slot_tag)::value;
          const int k0 = klo + kWgKC * c;
          lstore(CUR ^ 1, rp[SLOT], rq[SLOT], rs[SLOT]);
          gload(c + 1 + PF, rp[SLOT], rq[SLOT], rs[SLOT]);
          rd(CUR, 1, fa1, fb1);
          __builtin_amdgcn_sched_barrier(0);
          mm(k0, fa0, fb0);
          __builtin_amdgcn_sched_barrier(0);
          __syncthreads();
          rd(CUR ^ 1, 0, fa0, fb0);
          __builtin_amdgcn_sched_barrier(0);
          mm(k0, fa1, fb1);
          __builtin_amdgcn_sched_barrier(0);
        };
        using I0t = std::integral_constant<int, 0>;
        using I1t = std::integral_constant<int, 1>;
        using I2t = std::integral_constant<int, 2>;
        using I3t = std::integral_constant<int, 3>;
        // prologue: chunks 0 .. PF - 1 requested, chunk 0 on to LDS, chunk PF requested, first fragments read
#pragma unroll
        for (int n = 0; n < PF; ++n) gload(n, rp[n], rq[n], rs[n]);
        lstore(0, rp[0], rq[0], rs[0]);
        gload(PF, rp[0], rq[0], rs[0]);
        __syncthreads();
        rd(0, 0, fa0, fb0);
        int c = 0;
        if constexpr (PF == 2) {
#pragma nounroll
          for (; c + 1 < nch; c += 2) {  // the LDS stages and the register stages alternate by name
            chunk(I0t{}, I1t{}, c);
            chunk(I1t{}, I0t{}, c + 1);
          }
          if (c < nch) chunk(I0t{}, I1t{}, c);
        } else {
#pragma nounroll
          for (; c + 3 < nch; c += 4) {
            chunk(I0t{}, I1t{}, c);
            chunk(I1t{}, I2t{}, c + 1);
            chunk(I0t{}, I3t{}, c + 2);
            chunk(I1t{}, I0t{}, c + 3);
          }
          if (c < nch) chunk(I0t{}, I1t{}, c);
          if (c + 1 < nch) chunk(I1t{}, I2t{}, c + 1);
          if (c + 2 < nch) chunk(I0t{}, I3t{}, c + 2);
        }
      }
      // (no barrier here: what slower waves may still do with the ring is the read-ahead of a chunk that does not
      // exist, and the next tile's prologue has a barrier between its first LDS store and everything else)
      run_epilogue<8>(epi,
                      [&](int b, int* i, int* j) {
                        const bool on = (on_mask >> b) & 1u;
                        *i = on ? I0 + 32 * wi + 16 * (b >> 2) : -1;
                        *j = J0 + 64 * wj + 16 * (b & 3);
                      },
                      [&](int b) -> const d4& { return acc[b >> 2][b & 3]; });
    }
  }
}

// Store a 16x16 accumulator tile (C layout) row-major at Cm[i0.., j0..] and/or transposed at CT[j0.., i0..].
// The transposed copy goes through a per-wave LDS tile so that its global stores are 128-byte rows too.
__device__ inline void store_tile(const d4& v, gd* __restrict__ Cm, gd* __restrict__ CT, int ld, int i0, int j0,
                                  ldsd* tile /* per-wave 16x17 */) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  if (Cm) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Cm[(size_t)(i0 + lq + 4 * r) * ld + j0 + lr] = v[r];
  }
  if (CT) {
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

// Factor the 16x16 diagonal block S (LDS panel rows 0..15, row stride 17) = L_kk L_kk^T and invert L_kk.
// One wave: lane r (mod 16) holds row r in registers; column pivots and multipliers travel by
// v_readlane broadcasts, so the 16 dependent elimination steps never wait on LDS.  The inverse is a
// forward substitution, one column per lane, reading L_kk as LDS broadcasts.  Results: L_kk and
// Dinv = L_kk^-1 in g_sh.dblk / g_sh.dinv, and in global memory (L^T diagonal block, Dinv, Dinv^T).  (Round 3: L itself
// is no longer written by these kernels -- since Pm = Phi(-G_A A^T) nothing reads it; everything works from L^T.)
__device__ __noinline__ void diag_factor_invert(ldsd* panel, int kb) {
  const Fit& f = g_sh.f;
  const int Mp = f.Mp;
  kb = uni(kb);
  gd* LT = f.mat[B_LT];
  const int lane = threadIdx.x & 63, r = lane & 15;
#ifdef GAPRO_PROFILE
  const unsigned long long tp0 = wall_clock64();
#endif
  double rdiag[16];  // 1 / L[j][j] (wave-uniform)
  {
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = panel[r * 17 + c];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      double d = lane_bcast(a[j], j);
      if (!(d > 0.0)) {  // not positive definite (or NaN): flag it, keep going with a tiny pivot
        bad = true;
        d = 1e-30;
      }
      const double rs = rsqrt(d);
      rdiag[j] = rs;
      const double lj = (r == j) ? d * rs : a[j] * rs;  // column j of L: rows >= j are meaningful
      a[j] = lj;
#pragma unroll
      for (int c = j + 1; c < 16; ++c) a[c] -= lj * lane_bcast(lj, c);  // only rows r >= c are used later
    }
    if (bad && lane == 0) g_sh.chol_bad = 1;
    if (lane < 16) {
#pragma unroll
      for (int c = 0; c < 16; ++c) g_sh.dblk[r * 17 + c] = (c <= r) ? a[c] : 0.0;  // L_kk
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
#ifdef GAPRO_PROFILE
  const unsigned long long tp1 = wall_clock64();
#endif
  {
    // column r of Dinv by forward substitution, right-looking: x[q] = b[q] / L[q][q], then every later row takes
    // its term, b[rr] -= L[rr][q] x[q] (15 - q independent updates, L_kk read as LDS broadcasts).  The phase is
    // bound by VALU issue (every lane of the wave executes every instruction), so the terms are bare FMAs: x[q] = 0
    // for q < r makes the terms of the rows above the column vanish without a select per term (1.83 -> 0.99 us per
    // block).  Feeding these FMAs from the factor loop's own broadcasts (one fused pass, no LDS reads) was slower:
    // 4.05 us per block against 3.4 for the two loops.
    double x[16], b[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) b[rr] = (rr == r) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      x[q] = (q >= r) ? b[q] * rdiag[q] : 0.0;
#pragma unroll
      for (int rr = q + 1; rr < 16; ++rr) b[rr] = fma(-g_sh.dblk[rr * 17 + q], x[q], b[rr]);
    }
    if (lane < 16) {
#pragma unroll
      for (int c = 0; c < 16; ++c) g_sh.dinv[c * 17 + r] = x[c];  // Dinv[c][r]: lane r holds column r
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
#ifdef GAPRO_PROFILE
  const unsigned long long tp2 = wall_clock64();
#endif
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = lane + 64 * e;
    const int rr = idx >> 4, cc = idx & 15;
    LT[(size_t)(16 * kb + rr) * Mp + 16 * kb + cc] = g_sh.dblk[cc * 17 + rr];
    f.dinv[(size_t)kb * 256 + idx] = g_sh.dinv[rr * 17 + cc];
    f.dinvT[(size_t)kb * 256 + idx] = g_sh.dinv[cc * 17 + rr];
  }
#ifdef GAPRO_PROFILE
  if (lane == 0) {  // wave-0-only sub-phases of the diagonal block (they overlap slot 5, not part of the total)
    const unsigned long long tp3 = wall_clock64();
    g_sh.prof[22] += tp1 - tp0;  // factor
    g_sh.prof[23] += tp2 - tp1;  // inverse
    g_sh.prof[24] += tp3 - tp2;  // stores issued
  }
#endif
}

// acc -= sum_{q < Q} L[i][q] L[j][q] for one 16x16 tile of the Cholesky update (operands from L^T, TN form), Q a
// multiple of 16.  Blocks of four k-steps (eight loads) alternate between two register sets with no guard in the
// steady-state body, so that the loads of a block are in flight during the MFMAs of the previous one (see gemm_tn).
__device__ inline void chol_update_tile(d4& acc, const gd* pa, const gd* pb, int Mp, int Q) {
  const size_t st = (size_t)4 * Mp;
  auto ld = [&](int q, double (&a)[4], double (&b)[4]) {
    const size_t o = (size_t)q * Mp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = pa[o + e * st];
      b[e] = pb[o + e * st];
    }
  };
  auto mm = [&](double (&a)[4], double (&b)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[e], b[e], acc, 0, 0, 0);
  };
  if (Q <= 0) return;
  double a0[4], b0[4], a1[4], b1[4];
  ld(0, a0, b0);
  int q = 0;
#pragma nounroll
  for (; q + 32 < Q; q += 32) {
    ld(q + 16, a1, b1);
    mm(a0, b0);
    ld(q + 32, a0, b0);
    mm(a1, b1);
  }
  if (q + 16 < Q) {
    ld(q + 16, a1, b1);
    mm(a0, b0);
    mm(a1, b1);
  } else {
    mm(a0, b0);
  }
}

// ---- Cholesky of Kzz + jitter I, fused with the kernel evaluation -----------------------------------
// Left-looking, 16-wide block columns.  Block column kb:
//   (1) S = Kzz[:, kb] - L[:, <kb] L[kb, <kb]^T : the Kzz tile is evaluated from the staged inducing
//       points straight into the MFMA accumulator, the update reads L^T (TN form); S goes to an LDS panel
//   (2) wave 0 factors the 16x16 diagonal block held one row per lane in registers (cross-lane
//       broadcasts, no LDS round trips) and inverts it (column per lane)
//   (3) panel below = S * Dinv^T, computed in LDS, then written once to L^T (rows)
// The padded tail (index >= M) is an identity block.  Strict upper triangle of L stays zero.
template <int DC>
__device__ __noinline__ void cholesky_fused(const ldsd* Zt, ldsd* panel, double s, double inv_l2, double jitter) {
  const Fit& f = g_sh.f;
  Shared& sh = g_sh;
  const int Mp = f.Mp, M = f.M, D = DC ? DC : f.D, nb = Mp / 16;
  gd* LT = f.mat[B_LT];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int lr = lane & 15, lq = lane >> 4;
  for (int kb = 0; kb < nb; ++kb) {
    // (1)
    for (int ib = kb + wave; ib < nb; ib += NW) {
      d4 acc;
      const int col = 16 * kb + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ib + lq + 4 * r;
        double v = 0.0;
        if (row < M && col < M) {
          v = s * gapro_fit_math::rbf_exp(-0.5 * inv_l2 * sqdist_t(Zt, row, Zt, col, D, Mp));
          if (row == col) v += jitter;
        } else if (row == col) {
          v = 1.0;
        }
        acc[r] = v;
      }
      const gd* pa = LT + (size_t)lq * Mp + 16 * ib + lr;
      const gd* pb = LT + (size_t)lq * Mp + 16 * kb + lr;
      chol_update_tile(acc, pa, pb, Mp, 16 * kb);
      ldsd* dst = panel + (16 * (ib - kb)) * 17;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(lq + 4 * r) * 17 + lr] = acc[r];
    }
    __syncthreads();
    prof_stamp(0);
    // (2)
    if (wave == 0) diag_factor_invert(panel, kb);
    __syncthreads();
    prof_stamp(5);
    // (3) rows below the diagonal block, in LDS: P[i][c] <- sum_{q <= c} S[i][q] Dinv[c][q]
    const int rows_below = Mp - 16 * (kb + 1);
    ldsd* pb = panel + 16 * 17;
    for (int idx = threadIdx.x; idx < rows_below * 16; idx += NT) {
      const int i = idx >> 4, c = idx & 15;
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc += (q <= c) ? pb[i * 17 + q] * sh.dinv[c * 17 + q] : 0.0;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();  // the 16 lanes of a row have all read S before any overwrites it
      pb[i * 17 + c] = acc;
    }
    __syncthreads();
    if (rows_below > 0)
      for (int idx = threadIdx.x; idx < rows_below * 16; idx += NT) {  // L^T rows: contiguous in i
        const int c = idx / rows_below, i = idx - c * rows_below;
        LT[(size_t)(16 * kb + c) * Mp + 16 * (kb + 1) + i] = pb[i * 17 + c];
      }
    __syncthreads();
    prof_stamp(18);
  }
}

// ---- the same factorisation with a one-column look-ahead (strip kernels) ------------------------------
// In cholesky_fused the other waves idle while wave 0 factors the diagonal block (~4.5 us per block column, more
// than the whole MFMA update of a column).  Here two LDS panels alternate: while wave 0 factors the diagonal block
// of column kb, the other waves already build column kb + 1 -- kernel tile minus the contributions of the columns
// < kb, which are final in L^T -- and once column kb's panel has been scaled, its rank-16 contribution is
// subtracted straight from LDS (both operands are rows of the scaled panel) while the panel is written to L / L^T.
// Every accumulator sees the same MFMAs in the same order as in cholesky_fused (the detour of the partial sums
// through LDS is exact), so the factor is bit-identical.
inline __host__ __device__ int chol_panel_doubles(int Mp) { return Mp * 17 + 64 * 17; }
template <int DC>
__device__ __noinline__ void cholesky_fused_lookahead(const ldsd* Zt, ldsd* panels, double s, double inv_l2,
                                                      double jitter) {
  const Fit& f = g_sh.f;
  Shared& sh = g_sh;
  const int Mp = f.Mp, M = f.M, D = DC ? DC : f.D, nb = Mp / 16;
  gd* LT = f.mat[B_LT];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int lr = lane & 15, lq = lane >> 4;
  const int PS = chol_panel_doubles(Mp);
  // block (ib, kb) of Kzz + jitter I minus the contributions of the block columns < qb, into panel dst
  auto build = [&](int ib, int kb, int qb, ldsd* dst_panel) {
    d4 acc;
    const int col = 16 * kb + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * ib + lq + 4 * r;
      double v = 0.0;
      if (row < M && col < M) {
        v = s * gapro_fit_math::rbf_exp(-0.5 * inv_l2 * sqdist_t(Zt, row, Zt, col, D, Mp));
        if (row == col) v += jitter;
      } else if (row == col) {
        v = 1.0;
      }
      acc[r] = v;
    }
    const gd* pa = LT + (size_t)lq * Mp + 16 * ib + lr;
    const gd* pb = LT + (size_t)lq * Mp + 16 * kb + lr;
    chol_update_tile(acc, pa, pb, Mp, 16 * qb);
    ldsd* dst = dst_panel + (16 * (ib - kb)) * 17;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(lq + 4 * r) * 17 + lr] = acc[r];
  };
  for (int ib = wave; ib < nb; ib += NW) build(ib, 0, 0, panels);
  __syncthreads();
  for (int kb = 0; kb < nb; ++kb) {
    ldsd* cur = panels + (kb & 1) * PS;
    ldsd* nxt = panels + ((kb + 1) & 1) * PS;
    // (2) diagonal block on wave 0 | column kb + 1 without the contribution of column kb on the other waves
    if (wave == 0) {
      // the serial chain everybody waits for.  In the 256-thread build a SIMD hosts one wave of each of the CU's two
      // fits: the chain goes ahead of the other fit's wave (+2..3 % at M <= 64; with 512 threads it is -0.6 %)
      if (NW <= 4) __builtin_amdgcn_s_setprio(3);
      diag_factor_invert(cur, kb);
      if (NW <= 4) __builtin_amdgcn_s_setprio(0);
    } else {
      for (int ib = kb + wave; ib < nb; ib += NW - 1) build(ib, kb + 1, kb, nxt);
    }
    __syncthreads();
    prof_stamp(5);
    // (3) rows below the diagonal block, in LDS: P[i][c] <- sum_{q <= c} S[i][q] Dinv[c][q]
    const int rows_below = Mp - 16 * (kb + 1);
    ldsd* pb = cur + 16 * 17;
    for (int idx = threadIdx.x; idx < rows_below * 16; idx += NT) {
      const int i = idx >> 4, c = idx & 15;
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc += (q <= c) ? pb[i * 17 + q] * sh.dinv[c * 17 + q] : 0.0;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();  // the 16 lanes of a row have all read S before any overwrites it
      pb[i * 17 + c] = acc;
    }
    __syncthreads();
    // column kb's contribution to column kb + 1, both operands from the scaled panel (block row kb + 1 is its top)
    for (int ib = kb + 1 + wave; ib < nb; ib += NW) {
      ldsd* dst = nxt + (16 * (ib - kb - 1)) * 17;
      const ldsd* la = pb + (16 * (ib - kb - 1) + lr) * 17 + lq;
      const ldsd* lb = pb + lr * 17 + lq;
      d4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = dst[(lq + 4 * r) * 17 + lr];
#pragma unroll
      for (int st = 0; st < 4; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-la[4 * st], lb[4 * st], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(lq + 4 * r) * 17 + lr] = acc[r];
    }
    if (rows_below > 0)
      for (int idx = threadIdx.x; idx < rows_below * 16; idx += NT) {  // L^T rows: contiguous in i
        const int c = idx / rows_below, i = idx - c * rows_below;
        LT[(size_t)(16 * kb + c) * Mp + 16 * (kb + 1) + i] = pb[i * 17 + c];
      }
    __syncthreads();
    prof_stamp(18);
  }
}

// ---- the look-ahead with the next column's partial sums in REGISTERS (staged kernel, round 4) --------------------
// The LDS look-ahead above needs two panels (87 KB at M_p = 256), which the staged kernel does not have beside Z, X
// and its two-workgroups-per-CU budget of 72 KB.  Here the tiles of column kb + 1 that a wave builds while wave 0
// factors the diagonal block of column kb stay in that wave's registers (at most TMAX accumulators of 8 VGPRs: block
// rows kb + w, kb + w + 7, ... for wave w >= 1), take column kb's rank-16 contribution from the scaled panel, and are
// written to the ONE panel once every reader of column kb is done with it.  Same MFMAs in the same order per
// accumulator: bit-identical to cholesky_fused.  Four barriers per block column, as there.
template <int DC, int TMAX>
__device__ __noinline__ void cholesky_fused_lookahead_reg(const ldsd* Zt, ldsd* panel, double s, double inv_l2,
                                                          double jitter) {
  const Fit& f = g_sh.f;
  Shared& sh = g_sh;
  const int Mp = f.Mp, M = f.M, D = DC ? DC : f.D, nb = Mp / 16;
  gd* LT = f.mat[B_LT];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int lr = lane & 15, lq = lane >> 4;
  // block (ib, kb) of Kzz + jitter I minus the contributions of the block columns < qb
  auto build = [&](int ib, int kb, int qb) -> d4 {
    d4 acc;
    const int col = 16 * kb + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * ib + lq + 4 * r;
      double v = 0.0;
      if (row < M && col < M) {
        v = s * gapro_fit_math::rbf_exp(-0.5 * inv_l2 * sqdist_t(Zt, row, Zt, col, D, Mp));
        if (row == col) v += jitter;
      } else if (row == col) {
        v = 1.0;
      }
      acc[r] = v;
    }
    const gd* pa = LT + (size_t)lq * Mp + 16 * ib + lr;
    const gd* pb = LT + (size_t)lq * Mp + 16 * kb + lr;
    chol_update_tile(acc, pa, pb, Mp, 16 * qb);
    return acc;
  };
  for (int ib = wave; ib < nb; ib += NW) {
    const d4 acc = build(ib, 0, 0);
    ldsd* dst = panel + (16 * ib) * 17;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(lq + 4 * r) * 17 + lr] = acc[r];
  }
  __syncthreads();
  d4 nx[TMAX];
  for (int kb = 0; kb < nb; ++kb) {
    // (2) diagonal block on wave 0 | column kb + 1 without the contribution of column kb on the other waves
    if (wave == 0) {
      diag_factor_invert(panel, kb);
    } else {
#pragma unroll
      for (int t = 0; t < TMAX; ++t) {
        const int ib = kb + wave + t * (NW - 1);
        if (ib < nb) nx[t] = build(ib, kb + 1, kb);
      }
    }
    __syncthreads();
    prof_stamp(5);
    // (3) rows below the diagonal block, in LDS: P[i][c] <- sum_{q <= c} S[i][q] Dinv[c][q]
    const int rows_below = Mp - 16 * (kb + 1);
    ldsd* pb = panel + 16 * 17;
    for (int idx = threadIdx.x; idx < rows_below * 16; idx += NT) {
      const int i = idx >> 4, c = idx & 15;
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc += (q <= c) ? pb[i * 17 + q] * sh.dinv[c * 17 + q] : 0.0;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();  // the 16 lanes of a row have all read S before any overwrites it
      pb[i * 17 + c] = acc;
    }
    __syncthreads();
    // column kb's contribution to this wave's tiles of column kb + 1, both operands from the scaled panel
    if (wave != 0) {
#pragma unroll
      for (int t = 0; t < TMAX; ++t) {
        const int ib = kb + wave + t * (NW - 1);
        if (ib < nb) {
          const ldsd* la = pb + (16 * (ib - kb - 1) + lr) * 17 + lq;
          const ldsd* lb = pb + lr * 17 + lq;
#pragma unroll
          for (int st = 0; st < 4; ++st)
            nx[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-la[4 * st], lb[4 * st], nx[t], 0, 0, 0);
        }
      }
    }
    if (rows_below > 0)
      for (int idx = threadIdx.x; idx < rows_below * 16; idx += NT) {  // L^T rows: contiguous in i
        const int c = idx / rows_below, i = idx - c * rows_below;
        LT[(size_t)(16 * kb + c) * Mp + 16 * (kb + 1) + i] = pb[i * 17 + c];
      }
    __syncthreads();  // every reader of the scaled column kb is done: the panel takes column kb + 1
    if (wave != 0) {
#pragma unroll
      for (int t = 0; t < TMAX; ++t) {
        const int ib = kb + wave + t * (NW - 1);
        if (ib < nb) {
          ldsd* dst = panel + (16 * (ib - kb - 1)) * 17;
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[(lq + 4 * r) * 17 + lr] = nx[t][r];
        }
      }
    }
    __syncthreads();
    prof_stamp(18);
  }
}

// ---- gpytorch's psd_safe_cholesky around either factorisation -----------------------------------------------
// VariationalStrategy._cholesky_factor calls psd_safe_cholesky(K_ZZ.double() + jitter I): when the factorisation meets
// a non-positive pivot it is repeated on K + j I with j = psd_jitter * 10^i (settings.cholesky_jitter: 1e-8 for
// float64), i = 0 .. psd_retries - 1 (settings.cholesky_max_tries = 3), and only then gives up (NotPSDError -> here
// GAPRO_ERR_CHOLESKY for this fit; the other fits of the launch are unaffected).  The extra jitter lives inside the
// factorisation only: it is not added to the k_xx term of the predictive variance, as in gpytorch.  A function of its
// own so that the retry state is not live across the step loop of the callers.
// LOOKAHEAD: 0 = cholesky_fused, 1 = two LDS panels (strip kernels), 2 / 3 = the next column in registers, at most 3 / 5
// tiles per wave (staged kernel: M_p <= 256 / <= 512)
template <int DC, int LOOKAHEAD>
__device__ __noinline__ void cholesky_psd_safe(const ldsd* Zt, ldsd* scratch, double s, double inv_l2, double jitter,
                                               int retries, double psd_jitter) {
  double extra = 0.0;
  for (int attempt = 0;; ++attempt) {
    if (LOOKAHEAD == 1) cholesky_fused_lookahead<DC>(Zt, scratch, s, inv_l2, jitter + extra);
    else if (LOOKAHEAD == 2) cholesky_fused_lookahead_reg<DC, 3>(Zt, scratch, s, inv_l2, jitter + extra);
    else if (LOOKAHEAD == 3) cholesky_fused_lookahead_reg<DC, 5>(Zt, scratch, s, inv_l2, jitter + extra);
    else cholesky_fused<DC>(Zt, scratch, s, inv_l2, jitter + extra);
    const int bad = g_sh.chol_bad;  // both factorisations end with a workgroup barrier
    if (!bad) return;               // the common case: one LDS read, no extra barrier
    __syncthreads();
    if (threadIdx.x == 0) {
      g_sh.chol_bad = 0;
      if (attempt >= retries) g_sh.status = GAPRO_ERR_CHOLESKY;
    }
    __syncthreads();
    if (attempt >= retries) return;
    extra = psd_jitter * pow(10.0, (double)attempt);
  }
}

// ---- the strip kernels' triangular inverse: one block column per wave as a straight-line chain ---------------
// Same products in the same order as tri_inverse<8> (bit-identical), but instantiated per column length so that
// there is no guard inside the chain: the L operands of block row ii + 1 (and its Dinv^T) are requested before the
// MFMAs of row ii and are in flight while row ii is computed and stored.  In the guarded form every block waits
// for its own loads and, s_waitcnt being what it is behind a join, for the stores of the block before.
template <int CNT>
__device__ inline void inv_column(int k, ldsd* tile) {
  const Fit& f = g_sh.f;
  const int Mp = f.Mp;
  const gd* LT = f.mat[B_LT];
  gd* LI = f.mat[B_LI];
  gd* U = f.mat[B_U];
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  constexpr int MAXR = CNT > 1 ? CNT - 1 : 1;
  d4 blk[CNT];
  double A0[MAXR * 4], A1[MAXR * 4], D0[4], D1[4];
  // operands of block row ii: L[16(k+ii)+lr][16(k+jj)+4s+lq] for jj < ii, and -Dinv_{k+ii}^T
  auto ldrow = [&](int ii, double (&A)[MAXR * 4], double (&D)[4]) {
    const int i = k + ii;
#pragma unroll
    for (int jj = 0; jj < MAXR; ++jj)
      if (jj < ii) {
        const gd* pa = LT + (size_t)(16 * (k + jj) + lq) * Mp + 16 * i + lr;
#pragma unroll
        for (int st = 0; st < 4; ++st) A[jj * 4 + st] = pa[(size_t)(4 * st) * Mp];
      }
#pragma unroll
    for (int st = 0; st < 4; ++st) D[st] = f.dinvT[(size_t)i * 256 + (4 * st + lq) * 16 + lr];
  };
  d4 dk;
#pragma unroll
  for (int r = 0; r < 4; ++r) dk[r] = f.dinv[(size_t)k * 256 + (lq + 4 * r) * 16 + lr];
  if (CNT > 1) ldrow(1, A1, D1);
  store_tile(dk, LI, U, Mp, 16 * k, 16 * k, tile);
  blk[0] = dk;
#pragma unroll
  for (int ii = 1; ii < CNT; ++ii) {
    if (ii + 1 < CNT) {
      if ((ii + 1) & 1) ldrow(ii + 1, A1, D1);
      else ldrow(ii + 1, A0, D0);
    }
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int jj = 0; jj < MAXR; ++jj)
      if (jj < ii) {
#pragma unroll
        for (int st = 0; st < 4; ++st)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64((ii & 1) ? A1[jj * 4 + st] : A0[jj * 4 + st], blk[jj][st], acc, 0, 0, 0);
      }
    d4 out = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int st = 0; st < 4; ++st)
      out = __builtin_amdgcn_mfma_f64_16x16x4f64(-((ii & 1) ? D1[st] : D0[st]), acc[st], out, 0, 0, 0);
    store_tile(out, LI, U, Mp, 16 * (k + ii), 16 * k, tile);
    blk[ii] = out;
  }
}
__device__ __noinline__ void tri_inverse_strip(ldsd* tiles) {
  const int nb = g_sh.f.Mp / 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  ldsd* tile = tiles + wave * 16 * 17;
  for (int k = wave; k < nb; k += NW) {
    switch (nb - k) {
      case 1: inv_column<1>(k, tile); break;
      case 2: inv_column<2>(k, tile); break;
      case 3: inv_column<3>(k, tile); break;
      case 4: inv_column<4>(k, tile); break;
#if GAPRO_NT >= 320
      case 5: inv_column<5>(k, tile); break;
      case 6: inv_column<6>(k, tile); break;
      case 7: inv_column<7>(k, tile); break;
      default: inv_column<8>(k, tile); break;
#else
      default: break;  // M_p <= 64 on the small-fit route
#endif
    }
  }
}

// ---- LI = L^-1 (lower) and U = LI^T, one 16-wide block column per wave -----------------------------
//   LI_kk = Dinv_k;   LI_ik = -Dinv_i * sum_{j=k}^{i-1} L_ij LI_jk   (i > k)
// Block columns are independent.  For nb <= NBR the blocks of the column stay in registers (an MFMA
// result in C layout is directly the B operand of the next product: register r holds rows 4r + lane/16);
// longer columns re-read their own blocks from memory after a workgroup-scope fence.
template <int NBR>
__device__ __noinline__ void tri_inverse(ldsd* tiles) {
  const Fit& f = g_sh.f;
  const int Mp = f.Mp, nb = Mp / 16;
  const gd* LT = f.mat[B_LT];
  gd* LI = f.mat[B_LI];
  gd* U = f.mat[B_U];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int lr = lane & 15, lq = lane >> 4;
  ldsd* tile = tiles + wave * 16 * 17;
  // Which wave takes which block column.  NBR > 0 (M_p <= 128): round-robin, at most one column per wave.  Long
  // columns (NBR == 0, round 4): a column of n blocks is a serial chain of n (n - 1) / 2 block products, and dealt
  // round-robin wave 0 got columns 0, 8, 16, ... -- 1.5x the mean work at M_p = 384, with the whole workgroup waiting
  // for it at the barrier behind this phase.  Longest-processing-time-first instead: columns in ascending k (descending
  // cost) each go to the wave with the least work so far (every wave computes the same table; the result of a column
  // does not depend on who computes it: bit-identical).
  unsigned long long mine = 0;  // bit k: this wave computes block column k (nb <= 32)
  if (NBR == 0) {
    int load[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) load[w] = 0;
    for (int k = 0; k < nb; ++k) {
      int best = 0;
#pragma unroll
      for (int w = 1; w < NW; ++w) best = load[w] < load[best] ? w : best;
      const int n = nb - k;
#pragma unroll
      for (int w = 0; w < NW; ++w) load[w] += (w == best) ? n * (n - 1) / 2 + 1 : 0;
      if (best == wave) mine |= 1ull << k;
    }
  }
  for (int k = NBR > 0 ? wave : 0; k < nb; k += NBR > 0 ? NW : 1) {
    if (NBR == 0 && !((mine >> k) & 1ull)) continue;
    d4 blk[NBR > 0 ? NBR : 1];
    d4 dk;
#pragma unroll
    for (int r = 0; r < 4; ++r) dk[r] = f.dinv[(size_t)k * 256 + (lq + 4 * r) * 16 + lr];
    store_tile(dk, LI, U, Mp, 16 * k, 16 * k, tile);
    if (NBR > 0) blk[0] = dk;
    if (NBR > 0) {
      // relative row index ii = i - k is a compile-time constant: every register index is static
#pragma unroll
      for (int ii = 1; ii < NBR; ++ii) {
        const int i = k + ii;
        if (i < nb) {
          d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int jj = 0; jj < ii; ++jj) {
            const gd* pa = LT + (size_t)(16 * (k + jj) + lq) * Mp + 16 * i + lr;  // L[16i+lr][16(k+jj)+4s+lq]
#pragma unroll
            for (int sstep = 0; sstep < 4; ++sstep)
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[(size_t)(4 * sstep) * Mp], blk[jj][sstep], acc, 0, 0, 0);
          }
          d4 out = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int sstep = 0; sstep < 4; ++sstep) {
            const double a = -f.dinvT[(size_t)i * 256 + (4 * sstep + lq) * 16 + lr];  // -Dinv_i[lr][4s + lq]
            out = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[sstep], out, 0, 0, 0);
          }
          store_tile(out, LI, U, Mp, 16 * i, 16 * k, tile);
          blk[ii] = out;
        }
      }
    } else {
      for (int i = k + 1; i < nb; ++i) {
        // nacc = -sum_{j = k}^{i - 1} L_ij LI_jk through the Cholesky update's double-buffered block loop (round 4: the
        // loop here issued the eight loads of a block and waited for them before its four MFMAs, one memory round trip
        // per block of a chain of up to nb (nb - 1) / 2 blocks).  -(a) b accumulated is exactly -(a b accumulated), and
        // (-Dinv) x = Dinv (-x): the same bits as before.
        d4 nacc = (d4){0.0, 0.0, 0.0, 0.0};
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        chol_update_tile(nacc, LT + (size_t)(16 * k + lq) * Mp + 16 * i + lr, LI + (size_t)(16 * k + lq) * Mp + 16 * k + lr,
                         Mp, 16 * (i - k));
        d4 out = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sstep = 0; sstep < 4; ++sstep) {
          const double a = f.dinvT[(size_t)i * 256 + (4 * sstep + lq) * 16 + lr];  // Dinv_i[lr][4s + lq]
          out = __builtin_amdgcn_mfma_f64_16x16x4f64(a, nacc[sstep], out, 0, 0, 0);
        }
        store_tile(out, LI, U, Mp, 16 * i, 16 * k, tile);
      }
    }
  }
}

// ---- elementwise passes, thread-per-column ------------------------------------------------------------
// Threads are laid out as G row groups x Mp columns (Mp <= NT).  A thread owns one column and walks rows
// group, group+G, ...; per-column partial results of the G groups are combined in a fixed order through
// `red` ([kRedSlots][NT] doubles).
struct ColMap {
  int G, col, grp;
  bool active;
};
__device__ inline ColMap col_map(int Mp) {
  ColMap c;
  c.G = NT / Mp;
  c.col = threadIdx.x % Mp;
  c.grp = threadIdx.x / Mp;
  c.active = c.grp < c.G;
  return c;
}

// KX[k][n] = s exp(-|Z_k - P_n|^2 / (2 l^2)) for k < M, n < ncols, zero elsewhere (Pt: staged points)
template <int DC>
__device__ __noinline__ void build_kx(const ldsd* Zt, const ldsd* Pt, int ncols, double s, double inv_l2) {
  const Fit& f = g_sh.f;
  const int Mp = f.Mp, M = f.M, D = DC ? DC : f.D;
  gd* KX = f.mat[B_KX];
  const ColMap cm = col_map(Mp);
  if (!cm.active) return;
  const int n = cm.col;
  for (int k = cm.grp; k < Mp; k += cm.G) {
    double v = 0.0;
    if (k < M && n < ncols) v = s * gapro_fit_math::rbf_exp(-0.5 * inv_l2 * sqdist_t(Zt, k, Pt, n, D, Mp));
    KX[(size_t)k * Mp + n] = v;
  }
}

// mu[n] = sum_i m[i] A[i][n],  var[n] = s + jitter + sum_i (BM[i][n]^2 - A[i][n]^2)
__device__ __noinline__ void mean_var(double s, double jitter, ldsd* red) {
  const Fit& f = g_sh.f;
  const int Mp = f.Mp;
  const gd* A = f.mat[B_A];
  const gd* BM = f.mat[B_BM];
  const gd* m = f.vec[V_M];
  const ColMap cm = col_map(Mp);
  double pm = 0.0, pv = 0.0;
  if (cm.active) {
    for (int i = cm.grp; i < Mp; i += cm.G) {
      const double a = A[(size_t)i * Mp + cm.col], b = BM[(size_t)i * Mp + cm.col];
      pm += m[i] * a;
      pv += b * b - a * a;
    }
  }
  red[threadIdx.x] = pm;
  red[NT + threadIdx.x] = pv;
  __syncthreads();
  if (threadIdx.x < Mp) {
    double sm = 0.0, sv = 0.0;
    for (int g = 0; g < cm.G; ++g) {
      sm += red[g * Mp + threadIdx.x];
      sv += red[NT + g * Mp + threadIdx.x];
    }
    f.vec[V_MU][threadIdx.x] = sm;
    f.vec[V_VAR][threadIdx.x] = s + jitter + sv;
  }
  __syncthreads();
}

// out[c] = sum_r w[r] Mtx[r][c]
__device__ __noinline__ void weighted_colsum(const gd* Mtx, const gd* w, int Mp, gd* out, ldsd* red) {
  const ColMap cm = col_map(Mp);
  double p = 0.0;
  if (cm.active)
    for (int r = cm.grp; r < Mp; r += cm.G) p += w[r] * Mtx[(size_t)r * Mp + cm.col];
  red[threadIdx.x] = p;
  __syncthreads();
  if (threadIdx.x < Mp) {
    double sacc = 0.0;
    for (int g = 0; g < cm.G; ++g) sacc += red[g * Mp + threadIdx.x];
    out[threadIdx.x] = sacc;
  }
  __syncthreads();
}

// Expected log-likelihood terms: 20-point Gauss-Hermite of log Phi(y f), ten threads per point (one per
// symmetric node pair).  Writes g_mu[n] = -dE/dmu / N and g_v[n] = -dE/dvar / N (0 where the variance
// was clamped), returns sum_n E_n when want_e.
__device__ __noinline__ double quadrature(double c, double min_variance, double Nd, bool want_e, ldsd* red,
                                          double* g_c, double* gv_sum) {
  const Fit& f = g_sh.f;
  const int Mp = f.Mp, M = f.M;
  gd* gmu = f.vec[V_GMU];
  gd* gv = f.vec[V_GV];
  double e_tot = 0.0, gc_part = 0.0, gvs_part = 0.0;
  constexpr int PPR = NT / 10;  // points per round
  const int q = threadIdx.x % 10, nl = threadIdx.x / 10;
  for (int n0 = 0; n0 < M; n0 += PPR) {
    const int n = n0 + nl;
    double E = 0.0, dmu = 0.0, dvar = 0.0;
    const bool on = nl < PPR && n < M;
    if (on) {
      const double mu = f.vec[V_MU][n] + c;
      const double vraw = f.vec[V_VAR][n];
      const double var = vraw < min_variance ? min_variance : vraw;
      const double sd = sqrt(2.0 * var);
      const double y = n < f.M1 ? -1.0 : 1.0;  // train_y, not read from memory (global-memory latency in this chain)
      const double t = c_gh_t[q], w = c_gh_w[q];
      gh_pair(y, mu, sd, t, w, want_e, &E, &dmu, &dvar);
    }
    red[threadIdx.x] = E;
    red[NT + threadIdx.x] = dmu;
    red[2 * NT + threadIdx.x] = dvar;
    __syncthreads();
    if (on && q == 0) {
      double se = 0.0, sm = 0.0, sv = 0.0;
      for (int qq = 0; qq < 10; ++qq) {
        se += red[threadIdx.x + qq];
        sm += red[NT + threadIdx.x + qq];
        sv += red[2 * NT + threadIdx.x + qq];
      }
      const double ipi = 0.56418958354775628695;  // 1/sqrt(pi)
      const double vraw = f.vec[V_VAR][n];
      const bool clamped = vraw < min_variance;
      const double var = clamped ? min_variance : vraw;
      const double y = n < f.M1 ? -1.0 : 1.0;  // train_y, not read from memory (global-memory latency in this chain)
      const double g1 = -(ipi * sm * y) / Nd;
      const double g2 = clamped ? 0.0 : -(ipi * sv * y / sqrt(2.0 * var)) / Nd;
      gmu[n] = g1;
      gv[n] = g2;
      e_tot += ipi * se;
      gc_part += g1;
      gvs_part += g2;
    }
    __syncthreads();
  }
  for (int n = M + threadIdx.x; n < Mp; n += NT) {
    gmu[n] = 0.0;
    gv[n] = 0.0;
  }
  *g_c = block_sum(gc_part);
  *gv_sum = block_sum(gvs_part);
  return want_e ? block_sum(e_tot) : 0.0;
}

// Fused kernel-gradient pass + Adam on Z.  The thread owning column j accumulates over rows i:
//   zz: w  = sym(G)[i][j] s E_ij  ->  G_s += sym(G) E,  G_l += w d2,   G_Z[j] += 2 w (Z_j - Z_i)
//   zx: wx = G_KX[j][n=i] KX_jn   ->  G_s += G_KX E,    G_l += wx d2,  G_Z[j] += wx (Z_j - X_i)
// (sym(G) o K is symmetric, so the sum over i of column j equals the row sum of the oracle's formula.)
#ifdef GAPRO_X_NOEXP
#define KG_EXP(x) (1.0 + (x))
#else
#define KG_EXP(x) gapro_fit_math::rbf_exp(x)
#endif
template <int DMAX, bool ZX, int U, int DC>
__device__ __noinline__ void kernel_grads_adam_z(ldsd* Zt, const ldsd* Xt, const gd* Gm, const gd* GTm,
                                                 const gd* GKXT, double s, double inv_l2, double step_size,
                                                 double bc2s, ldsd* red, double* gs_out, double* gl_out,
                                                 const ldsd* zx_rows = nullptr) {
  const Fit& f = g_sh.f;
  const int Mp = f.Mp, M = f.M, D = DC ? DC : f.D;
  const ColMap cm = col_map(Mp);
  const int j = cm.col;
  double acc[DMAX];
#pragma unroll
  for (int d = 0; d < DMAX; ++d) acc[d] = 0.0;
  double gs = 0.0, gl = 0.0;
  if (DMAX > 8) {
    // Wide features (deep features, D = 32): the difference vectors are never held in registers.  With
    //   sum_i w_i (Z_j - P_i) = Z_j sum_i w_i - sum_i w_i P_i
    // only acc[d] = sum_i w_i P_i[d] and the scalar sum of the weights are accumulated; Z_j, Z_i, X_i are
    // re-read from LDS (conflict-free / broadcast), which keeps the pass inside a 128-VGPR budget.
    double wsum = 0.0;
    if (cm.active && j < M) {
      for (int i = cm.grp; i < M; i += cm.G) {
        const size_t o = (size_t)i * Mp + j;
        const double gsym = 0.5 * (Gm[o] + GTm[o]);
        const double g3 = ZX ? GKXT[o] : 0.0;  // G_KX[j][i]
        double d2 = 0.0, d2x = 0.0;
#pragma unroll 8
        for (int d = 0; d < D; ++d) {
          const double zjd = Zt[d * Mp + j];
          const double a = zjd - Zt[d * Mp + i];
          d2 += a * a;
          if (ZX) {
            const double bx = zjd - Xt[d * Mp + i];
            d2x += bx * bx;
          }
        }
        const double e = KG_EXP(-0.5 * inv_l2 * d2);
        const double w = gsym * s * e;
        gs += gsym * e;
        gl += w * d2;
        double wx = 0.0;
        if (ZX) {
          const double ex = KG_EXP(-0.5 * inv_l2 * d2x);
          wx = g3 * s * ex;
          gs += g3 * ex;
          gl += wx * d2x;
        }
        wsum += 2.0 * w + wx;
#pragma unroll
        for (int d = 0; d < DMAX; ++d)
          if (d < D) acc[d] += 2.0 * w * Zt[d * Mp + i] + (ZX ? wx * Xt[d * Mp + i] : 0.0);
      }
    }
#pragma unroll
    for (int d = 0; d < DMAX; ++d) acc[d] = (d < D) ? wsum * Zt[d * Mp + j] - acc[d] : 0.0;
  } else if (cm.active && j < M) {
    double zj[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) zj[d] = (d < D) ? Zt[d * Mp + j] : 0.0;
    // the (up to) 3 U operand loads of U rows are issued before the first exp: one memory round trip per
    // U rows instead of one per row (U is chosen per calling kernel to fit its register budget; a distinct U
    // also keeps the two kernels from sharing one instantiation compiled for the tighter budget)
    for (int i0 = cm.grp; i0 < M; i0 += cm.G * U) {
      double g1[U], g2[U], g3[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * cm.G;
        const bool act = i < M;
        const size_t o = (size_t)(act ? i : 0) * Mp + j;
        g1[u] = act ? Gm[o] : 0.0;
        g2[u] = act ? GTm[o] : 0.0;
        g3[u] = (ZX && act) ? GKXT[o] : 0.0;  // G_KX[j][i]
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * cm.G;
        if (u * cm.G >= M) break;  // wave-uniform: no row of this or any later u exists (small M)
        if (i < M) {
          double t[DMAX];
          double d2 = 0.0;
#pragma unroll
          for (int d = 0; d < DMAX; ++d) {
            t[d] = (d < D) ? zj[d] - Zt[d * Mp + i] : 0.0;
            d2 += t[d] * t[d];
          }
          const double e = KG_EXP(-0.5 * inv_l2 * d2);
          const double gsym = 0.5 * (g1[u] + g2[u]);
          const double w = gsym * s * e;
          gs += gsym * e;
          gl += w * d2;
#pragma unroll
          for (int d = 0; d < DMAX; ++d) acc[d] += 2.0 * w * t[d];
          if (ZX) {
            double d2x = 0.0;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
              t[d] = (d < D) ? zj[d] - Xt[d * Mp + i] : 0.0;
              d2x += t[d] * t[d];
            }
            const double ex = KG_EXP(-0.5 * inv_l2 * d2x);
            const double wx = g3[u] * s * ex;
            gs += g3[u] * ex;
            gl += wx * d2x;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) acc[d] += wx * t[d];
          }
        }
      }
    }
  }
  prof_stamp(20);
  *gs_out = block_sum(gs);
  *gl_out = block_sum(gl);
  prof_stamp(21);
  // combine the row groups, kRedSlots feature dimensions at a time, then Adam on Z (and its LDS copy)
  const double b1 = 0.9, b2 = 0.999, aeps = 1e-8;
#pragma unroll
  for (int dc = 0; dc < DMAX; dc += kRedSlots) {
    if (dc < D) {
#pragma unroll
      for (int e = 0; e < kRedSlots; ++e)
        if (dc + e < DMAX) red[e * NT + threadIdx.x] = acc[dc + e];
      __syncthreads();
      if (threadIdx.x < M) {
#pragma unroll
        for (int e = 0; e < kRedSlots; ++e) {
          const int d = dc + e;
          if (d < D && d < DMAX) {
            double sacc = 0.0;
            for (int g = 0; g < cm.G; ++g) sacc += red[e * NT + g * Mp + threadIdx.x];
            const size_t zi = (size_t)threadIdx.x * D + d;
            if (!ZX) sacc += zx_rows ? zx_rows[threadIdx.x * 8 + d] : f.gZ[zi];  // zx part of the strip loop (raw sums)
            const double grad = -inv_l2 * sacc;
            f.gZ[zi] = grad;
            const double m1 = b1 * f.mZ[zi] + (1.0 - b1) * grad;
            const double m2 = b2 * f.vZ[zi] + (1.0 - b2) * grad * grad;
            f.mZ[zi] = m1;
            f.vZ[zi] = m2;
            const double znew = f.Z[zi] - step_size * m1 / (sqrt(m2) / bc2s + aeps);
            f.Z[zi] = znew;
            Zt[d * Mp + threadIdx.x] = znew;
          }
        }
      }
      __syncthreads();
    }
  }
}

// one product of the staged kernel: workgroup-tiled through LDS (WG; extents and ranges per 16 x 16 block) or one
// tile per wave from global memory (gemm_tn)
// (PK / QK: operands with the contraction index along their rows, see gemm_tn; lower-triangular outputs only -- the
// workgroup-tiled form takes k-major operands)
template <int WG, int TU, bool SCALE, int ORD, int PK = 0, int QK = 0, typename KRange, typename Epi>
__device__ inline void product(int mo, int no, bool lower, const gd* __restrict__ P, const gd* __restrict__ Q, int ld,
                               const gd* __restrict__ qs, KRange kr, Epi epi, ldsd* ring) {
  if constexpr (WG == 0) {
    gemm_tn_in<TU, SCALE, 2, ORD, true, PK, QK>(mo, no, lower, P, Q, ld, qs, kr, epi);
  } else {
    // Workgroup-tiled products take the part of the output that whole 128 x 128 tiles cover; what is left at the
    // matrix edge (an L of up to 96 rows / columns, M_p a multiple of 32) goes to the per-wave products as 32 x 32
    // tiles: a workgroup tile with 32 valid rows costs as much as a full one (one wave strip in four has work), which
    // made M_p = 288 25 % slower than the per-wave form.  Same blocks, same k order: the split does not change a bit.
    // (extents rounded up to the strips' 32: the per-wave 32 x 32 form computes a last half-filled tile in full, M_p is
    // a multiple of 32 here)
    const int rows = (16 * mo + 31) / 32 * 32, cols = (16 * no + 31) / 32 * 32;
    const int R = rows / kWgTile * kWgTile, Cc = cols / kWgTile * kWgTile;
    constexpr bool kMajor = !(PK || QK);  // the workgroup-tiled form takes k-major operands only
    if constexpr (kMajor) {
      if (R > 0 && Cc > 0 && !lower) gemm_wg<SCALE, true, WG>(R / 16, Cc / 16, false, P, Q, ld, qs, kr, epi, ring);
    }
    auto strip_t = [&](auto tu_tag, int r0, int c0, int nr, int nc, bool low) {
      constexpr int TUS = decltype(tu_tag)::value;  // 2: 32 x 32 wave tiles, 4: 64 x 64 (extents in half tiles)
      if (nr <= 0 || nc <= 0) return;
      // (row-major tile order: the shell order of some products enumerates SQUARE tile grids only)
      gemm_tn_in<TUS, SCALE, 2, ORD_ROWMAJOR, true, PK, QK>(
          nr / (8 * TUS), nc / (8 * TUS), low, PK ? P + (size_t)r0 * ld : P + r0, QK ? Q + (size_t)c0 * ld : Q + c0, ld, qs,
          [=](int i0, int j0, int* lo, int* hi) {
            int l0, h0, l1, h1;  // a wave tile's range: the hull of its 16 x 16 blocks' (kr is monotone; gemm_tn trims hi)
            kr(r0 + i0, c0 + j0, &l0, &h0);
            kr(r0 + i0 + 16 * TUS - 16, c0 + j0 + 16 * TUS - 16, &l1, &h1);
            *lo = l0;
            *hi = h1;
          },
          shifted_epi(epi, r0, c0));
    };
    auto strip = [&](int r0, int c0, int nr, int nc, bool low) {  // rows [r0, r0 + nr) x columns [c0, c0 + nc)
      strip_t(std::integral_constant<int, 2>{}, r0, c0, nr, nc, low);
    };
    if (lower) {
      // Lower-triangular outputs (G_LS with its Adam epilogue, G_L, Pm) stay per-wave as a whole: a diagonal workgroup
      // tile computes 64 blocks for the 36 it needs, and these are the products with the heaviest epilogues, which the
      // eight waves of a tile then run in lockstep (fit-level, M = 256 two per CU: G_LS 15.6 -> 22.5, Pm 4.0 -> 5.3 ms
      // per fit with the tiled form; G_KX, G, G_A the other way).  64 x 64 wave tiles where round 2 used them.
      if constexpr (WG == 4) {
        if (rows >= 352) {
          strip_t(std::integral_constant<int, 4>{}, 0, 0, rows, cols, true);
          return;
        }
      }
      strip(0, 0, rows, cols, true);
      return;
    }
    if constexpr (kMajor) {
      strip(R, 0, rows - R, cols, false);   // bottom strip, full width
      strip(0, Cc, R, cols - Cc, false);    // right strip above it
    } else {
      strip(0, 0, rows, cols, false);
    }
  }
}

// WG: 0 = one tile per wave (gemm_tn), 2 / 4 = workgroup-tiled products with that many register stages (gemm_wg)
// KMIN: the products that contract over the columns of A, B, G_A and Pm read those matrices as they are (gemm_tn's PK /
// QK forms) and no transposed copy of them is written -- the build for M_p <= 256, where two workgroups share a CU and a
// launch of such fits sits on the HBM roof: 160: +4 .. 8 %, 256: +5 .. 7 % fits/s.  Those loads touch 16 half cache
// lines per instruction where the k-major form touches 4 whole ones, and the lower-triangular products are 25 .. 30 %
// slower with them; with one workgroup per CU (M_p >= 288) that costs what the copies cost (320: -1 %, 384: -3 %,
// 448: +2 %), so the larger fits keep the copies.
template <int TU, int DMAX, int DC, int WG = 0, bool KMIN = false, int EG = 4>
__device__ void fit_body(const gapro_fit_options& opt, ldsd* Zt, ldsd* Pt, ldsd* scratch, const gapro_fit_desc& desc, float* __restrict__ o_probs, float* __restrict__ o_probs_new,
                         unsigned char* __restrict__ o_labels, float* __restrict__ o_mu, float* __restrict__ o_var,
                         double* loss_out) {
  const Fit& f = g_sh.f;
  Shared& sh = g_sh;
  const int M = f.M, Mp = f.Mp, D = DC ? DC : f.D, T = f.T;
  constexpr int TS = 16 * TU;
  constexpr int TSB = TU >= 2 ? 8 * TU : TS;  // unit of gemm_tn's extents (TU >= 2: half tiles, see there)
  const int mt = Mp / TSB;
  const double Nd = (double)M;  // num_data = train_y.numel() (gaussian_process_utils.py:414)
  const double jitter = opt.jitter;
  gd* LS = f.mat[B_LS];
  gd* LST = f.mat[B_LST];
  gd* MLS = f.mat[B_MLS];
  gd* VLS = f.mat[B_VLS];
  gd* A = f.mat[B_A];
  gd* AT = KMIN ? nullptr : f.mat[B_AT];
  gd* BM = f.mat[B_BM];
  gd* BMT = f.mat[B_BMT];  // !KMIN only
  gd* GA = f.mat[B_GA];
  gd* GKXT = f.mat[B_GKXT];
  gd* GAT = KMIN ? nullptr : GKXT;  // !KMIN: G_A^T lives in the G_KX^T slot until G_KX is formed
  gd* vm = f.vec[V_M];
  gd* gmu = f.vec[V_GMU];
  gd* gv = f.vec[V_GV];
  // per-wave transpose tile; with the workgroup-tiled products the operand ring starts the scratch and the tiles alias
  // its second stage (gemm_wg: no stage is live while epilogues run, and a barrier precedes the next use of that stage)
  ldsd* ring = scratch;
  ldsd* tile = scratch + (WG ? kWgStage : 0) + (threadIdx.x >> 6) * 16 * 17;
  static_assert(WG == 0 || TU == 1, "workgroup-tiled products take their extents and ranges per 16 x 16 block");
  double last_loss = 0.0;
#ifdef GAPRO_PROFILE
  // diagnostic build only: per-phase wall-clock shares (100 MHz ticks), see tools/bench_fit.py --profile
  auto stamp = [&](int id) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long t = wall_clock64();
      sh.prof[id] += t - sh.t_last;
      sh.t_last = t;
    }
  };
#else
  auto stamp = [&](int) {};
#endif
  // between the products of the merged backward phase: nothing in the product build; the diagnostic build either keeps
  // the per-product stamps (and with them the barriers: GAPRO_PROFILE_SPLIT) or charges the whole phase to slot 10
#if defined(GAPRO_PROFILE) && defined(GAPRO_PROFILE_SPLIT)
#define STAMP_MERGED(id) do { __syncthreads(); stamp(id); } while (0)
#elif defined(GAPRO_NO_MERGED_BWD)  // A/B builds: the three products as three barrier-separated phases (round 3)
#define STAMP_MERGED(id) do { __syncthreads(); } while (0)
#else
  // merged only in the KMIN instantiations (M_p <= 256).  Beyond, G_A^T lives in the G_KX^T slot until G_KX^T is formed
  // (GAT below), so Pm -- which reads it -- must be complete before G_KX^T starts; and with one workgroup per CU and
  // 64 x 64 tiles the merge measured +-0 there anyway.
#define STAMP_MERGED(id) do { if constexpr (!KMIN) __syncthreads(); } while (0)
#endif

  auto refresh_hypers = [&]() {
    __syncthreads();
    if (threadIdx.x == 0) {
      sh.s = softplus(sh.rho_s);
      sh.ell = softplus(sh.rho_l);
      sh.inv_l2 = 1.0 / (sh.ell * sh.ell);
    }
    __syncthreads();
  };
  auto factorize = [&]() {
    stamp(19);
#ifdef GAPRO_NO_CHOL_LOOKAHEAD
    cholesky_psd_safe<DC, 0>(Zt, scratch, sh.s, sh.inv_l2, jitter, opt.psd_retries, opt.psd_jitter);
#else
    // The register look-ahead (next column's tiles built while wave 0 factors the diagonal block) beyond M_p = 256:
    // one workgroup per CU there, nobody else fills the CU while seven waves wait for the diagonal block (512 fits:
    // M = 320 -1.9 %, 384 -1.6 %, 448 -1.0 % in time, bit-identical).  Up to 256 two workgroups share a CU and hide
    // each other's serial stretches: 160 +-0, 200 +1.3 %, 256 +1 % -- not used there.
    cholesky_psd_safe<DC, KMIN ? 0 : 3>(Zt, scratch, sh.s, sh.inv_l2, jitter, opt.psd_retries, opt.psd_jitter);
#endif
    stamp(1);
    if (Mp <= 128)
      tri_inverse<8>(scratch);
    else
      tri_inverse<0>(scratch);
    __syncthreads();
    stamp(2);
  };
  // Column sums fused into the GEMM epilogues: every 16x16 result tile leaves the partial sum of its 16 rows per
  // column in part_x[tile_row][column]; summed later in tile order.  The partials live in LDS up to M_p = kFuseMaxMp
  // and beyond it in the G_A slot of the workspace (free until the backward pass; 3/16 of a matrix): a separate pass
  // over A and B (mean_var: two more matrix reads per step, 3 % of a step at M = 257 .. 464) is not needed.
  const bool fuse = Mp <= kFuseMaxMp;
  ldsd* part_m = scratch + kTileDoubles;       // sum_i m[i] A[i][n]   (later reused for G_m partials)
  ldsd* part_a = part_m + (Mp / 16) * Mp;      // sum_i A[i][n]^2
  ldsd* part_b = part_a + (Mp / 16) * Mp;      // sum_j B[j][n]^2
  gd* gpart_m = f.mat[B_GA];
  gd* gpart_a = gpart_m + (size_t)(Mp / 16) * Mp;
  gd* gpart_b = gpart_a + (size_t)(Mp / 16) * Mp;
  // A = LI * KX and B = LS^T A over ncols columns (row-major only: the products that contract over the columns of A
  // and B read them with the contraction index along the rows, gemm_tn's PK / QK forms); then mu (without c) and var
  auto forward_products = [&](int ncols, double s_, double jitter_) {
    const int nt = (ncols + TSB - 1) / TSB;
    // A[i][n] = sum_k U[k][i] KX[k][n],  U[k][i] = LI[i][k] = 0 for k > i
    product<WG, TU, false, ORD_ROWS_DESC>(mt, nt, false, f.mat[B_U], f.mat[B_KX], Mp, nullptr,
                       [=](int i0, int, int* lo, int* hi) { *lo = 0; *hi = i0 + TS; },
                       [=](int i, int n, const d4& v) {
                         if constexpr (KMIN) store_tile(v, A, nullptr, Mp, i, n, tile);
                         else store_tile(v, A, AT, Mp, i, n, tile);
                         {
                           const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
                           double pm = 0.0, pa = 0.0;
#pragma unroll
                           for (int r = 0; r < 4; ++r) {
                             pm += vm[i + lq + 4 * r] * v[r];
                             pa += v[r] * v[r];
                           }
                           pm += __shfl_xor(pm, 16, 64);
                           pa += __shfl_xor(pa, 16, 64);
                           pm += __shfl_xor(pm, 32, 64);
                           pa += __shfl_xor(pa, 32, 64);
                           if (lq == 0) {
                             if (fuse) {
                               part_m[(i >> 4) * Mp + n + lr] = pm;
                               part_a[(i >> 4) * Mp + n + lr] = pa;
                             } else {
                               gpart_m[(size_t)(i >> 4) * Mp + n + lr] = pm;
                               gpart_a[(size_t)(i >> 4) * Mp + n + lr] = pa;
                             }
                           }
                         }
                       }, ring);
    __syncthreads();
    if constexpr (KMIN) {
    // B[j][n] = sum_i LS[i][j] A[i][n],  LS[i][j] = 0 for i < j
    product<WG, TU, false, ORD_ROWMAJOR>(mt, nt, false, LS, A, Mp, nullptr,
                       [=](int j0, int, int* lo, int* hi) { *lo = j0; *hi = Mp; },
                       [=](int j, int n, const d4& v) {
                         store_tile(v, BM, nullptr, Mp, j, n, tile);
                         {
                           const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
                           double pb = 0.0;
#pragma unroll
                           for (int r = 0; r < 4; ++r) pb += v[r] * v[r];
                           pb += __shfl_xor(pb, 16, 64);
                           pb += __shfl_xor(pb, 32, 64);
                           if (lq == 0) {
                             if (fuse) part_b[(j >> 4) * Mp + n + lr] = pb;
                             else gpart_b[(size_t)(j >> 4) * Mp + n + lr] = pb;
                           }
                         }
                       }, ring);
    } else {
    // BMT[n][j] = sum_i A[i][n] LS[i][j],  LS[i][j] = 0 for i < j
    product<WG, TU, false, ORD_COLMAJOR>(nt, mt, false, A, LS, Mp, nullptr,
                       [=](int, int j0, int* lo, int* hi) { *lo = j0; *hi = Mp; },
                       [=](int n, int j, const d4& v) {
                         store_tile(v, BMT, BM, Mp, n, j, tile);
                         {
                           const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
#pragma unroll
                           for (int r = 0; r < 4; ++r) {
                             double pb = v[r] * v[r];
                             pb += __shfl_xor(pb, 1, 64);
                             pb += __shfl_xor(pb, 2, 64);
                             pb += __shfl_xor(pb, 4, 64);
                             pb += __shfl_xor(pb, 8, 64);
                             if (lr == 0) {
                               if (fuse) part_b[(j >> 4) * Mp + n + lq + 4 * r] = pb;
                               else gpart_b[(size_t)(j >> 4) * Mp + n + lq + 4 * r] = pb;
                             }
                           }
                         }
                       }, ring);
    }
    __syncthreads();
    for (int n = threadIdx.x; n < nt * TSB; n += NT) {
      double sm = 0.0, sa = 0.0, sb = 0.0;
      if (fuse) {
        for (int tq = 0; tq < Mp / 16; ++tq) {
          sm += part_m[tq * Mp + n];
          sa += part_a[tq * Mp + n];
          sb += part_b[tq * Mp + n];
        }
      } else {
        for (int tq = 0; tq < Mp / 16; ++tq) {
          sm += gpart_m[(size_t)tq * Mp + n];
          sa += gpart_a[(size_t)tq * Mp + n];
          sb += gpart_b[(size_t)tq * Mp + n];
        }
      }
      f.vec[V_MU][n] = sm;
      f.vec[V_VAR][n] = s_ + jitter_ + (sb - sa);
    }
    __syncthreads();
  };

#ifdef GAPRO_STEP_FN
  // one Adam step as ONE out-of-line function with the products inlined into it: the callee-saved registers are saved
  // once per step instead of once per product call (see gemm_tn_in)
  auto step_fn = [&](int step) __attribute__((noinline)) {
#else
  for (int step = 1; step <= opt.training_iter; ++step) {
#endif
    refresh_hypers();
    const double s = sh.s, ell = sh.ell, inv_l2 = sh.inv_l2, c = sh.c;
    const bool last = step == opt.training_iter;
    // ------------------------------- forward -------------------------------
    factorize();
    build_kx<DC>(Zt, Pt, M, s, inv_l2);
    __syncthreads();
    stamp(3);
    forward_products(M, s, jitter);
    stamp(4);
    double g_c, gv_sum;
    const double e_sum = quadrature(c, opt.min_variance, Nd, last, scratch, &g_c, &gv_sum);
    if (last) {  // the ELBO value is only reported, never used by the optimiser
      double kl_part = 0.0;
      for (int idx = threadIdx.x; idx < M * M; idx += NT) {
        const int i = idx / M, j = idx - i * M;
        if (j <= i) {
          const double v = LS[(size_t)i * Mp + j];
          kl_part += v * v;
          if (i == j) kl_part -= log(v * v);
        }
      }
      for (int i = threadIdx.x; i < M; i += NT) kl_part += vm[i] * vm[i];
      const double kl = 0.5 * (block_sum(kl_part) - Nd);
      last_loss = -(e_sum / Nd - kl / Nd);
    }
    stamp(6);

    // ------------------------------- backward ------------------------------
    const double b1 = 0.9, b2 = 0.999, aeps = 1e-8;
    const double bc1 = 1.0 - pow(b1, (double)step), bc2s = sqrt(1.0 - pow(b2, (double)step));
    const double step_size = opt.lr / bc1;
    // G_m = A g_mu (+ m / N, added with the Adam update below): fused into the G_A epilogue; the partials of the
    // 16-column tiles go to LDS, or beyond M_p = kFuseMaxMp to the G_KX slot (written two phases later)
    gd* gpart_g = f.mat[B_GKX];
    // G_A[i][n] = 2 g_v[n] sum_j LS[i][j] BM[j][n] + m[i] g_mu[n] - 2 A[i][n] g_v[n]
    // (two-phase epilogue: the loads of A, m, g_mu, g_v for a group of blocks are issued together, see two_phase_epi)
    struct GaPre { double a[4], m[4], gvn, gmn; };
    product<WG, TU, false, ORD_ROWS_DESC>(mt, mt, false, LST, BM, Mp, nullptr,
                       [=](int i0, int, int* lo, int* hi) { *lo = 0; *hi = i0 + TS; },
                       two_phase_epi<EG>(
                       [=](int i0, int n0) {
                         const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
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
                         const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
                         const int n = n0 + lr;
                         const double gvn = p.gvn, gmn = p.gmn;
                         d4 ga;
#pragma unroll
                         for (int r = 0; r < 4; ++r) {
                           const int i = i0 + lq + 4 * r;
                           const double a = p.a[r];
                           ga[r] = 2.0 * gvn * v[r] + p.m[r] * gmn - 2.0 * a * gvn;
                           if constexpr (KMIN) GA[(size_t)i * Mp + n] = ga[r];
                           double pg = a * gmn;
                           pg += __shfl_xor(pg, 1, 64);
                           pg += __shfl_xor(pg, 2, 64);
                           pg += __shfl_xor(pg, 4, 64);
                           pg += __shfl_xor(pg, 8, 64);
                           if (lr == 0) {
                             if (fuse) part_m[(n0 >> 4) * Mp + i] = pg;
                             else gpart_g[(size_t)(n0 >> 4) * Mp + i] = pg;
                           }
                         }
                         if constexpr (!KMIN) store_tile(ga, GA, GAT, Mp, i0, n0, tile);
                       }), ring);
    __syncthreads();
    for (int i = threadIdx.x; i < Mp; i += NT) {
      double sg = 0.0;
      if (fuse) {
        for (int tq = 0; tq < Mp / 16; ++tq) sg += part_m[tq * Mp + i];
      } else {
        for (int tq = 0; tq < Mp / 16; ++tq) sg += gpart_g[(size_t)tq * Mp + i];
      }
      f.vec[V_GM][i] = sg;
    }
    __syncthreads();
    stamp(7);
    // G_LS[i][j] = sum_n A[i][n] 2 g_v[n] BM[j][n] (lower) + KL', Adam on LS fused in the epilogue
    struct LsPre { double l[4], m1[4], m2[4]; };
    auto gls_epi = two_phase_epi<EG>(
                      [=](int i0, int j0) {
                        const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
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
                        const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
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
                        store_tile(newv, nullptr, LST, Mp, i0, j0, tile);  // LST[j][i]; zeros above the diagonal
                      });
    auto gls_range = [=](int, int, int* lo, int* hi) { *lo = 0; *hi = Mp; };
    if constexpr (KMIN)  // A and B as they are: both with the contraction index n along their rows
      product<WG, TU, true, ORD_ROWMAJOR, 1, 1>(mt, mt, true, A, BM, Mp, gv, gls_range, gls_epi, ring);
    else
      product<WG, TU, true, ORD_ROWMAJOR>(mt, mt, true, AT, BMT, Mp, gv, gls_range, gls_epi, ring);
    // (round 4) no barrier here, nor after Pm: G_LS, Pm and G_KX^T all read A, B, G_A, LI as the G_A phase left them
    // and write disjoint matrices (Pm now goes to the L slot, which no single-workgroup MFMA kernel writes since
    // round 3, instead of the B buffer G_LS is still reading), so a wave that has finished its G_LS tiles goes straight
    // on to its Pm and G_KX^T tiles.  Three barrier-separated phases whose tiles do not divide evenly among eight waves
    // (36 lower 32 x 32 tiles at M_p = 256: 4.5 rounds, the slowest wave sets the pace of each) become ONE phase of
    // 36 + 36 + 64 tiles, with the heavy Adam epilogue of G_LS under other waves' MFMAs.  Same tiles, same k order:
    // the bits do not change.
    STAMP_MERGED(8);
    // The Cholesky backward pass needs Pm = Phi(L^T G_L) with G_L = -tril(L^-T G_A A^T) = -tril(G_KX A^T).  Row i of
    // L^T X only reads rows k >= i of X, so the lower triangle of L^T tril(X) is the lower triangle of L^T X, and with
    // X = -L^-T G_A A^T:   Pm = Phi(-G_A A^T)   -- no G_L, no product with L^T (rounds 1-2 and the first half of round 3
    // formed G_L and L^T G_L: 1.33 M^3 where this is 1.0 M^3, one phase and one matrix write more).
    // -> B buffer (dead after G_LS).  KMIN: Pm as it is, from G_A and A as they are (contraction index n along their
    // rows); otherwise Pm^T (the k-major P operand of W: Pm^T[k][i] = Pm[i][k], non-zero for k <= i) from G_A^T and A^T
    gd* Pm = f.mat[B_L];
    auto pm_range = [=](int, int, int* lo, int* hi) { *lo = 0; *hi = Mp; };
    if constexpr (KMIN) {
      product<WG, TU, false, ORD_ROWMAJOR, 1, 1>(mt, mt, true, GA, A, Mp, nullptr, pm_range,
                       [=](int i0, int j0, const d4& v) {
                         const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
#pragma unroll
                         for (int r = 0; r < 4; ++r) {
                           const int i = i0 + lq + 4 * r, j = j0 + lr;
                           Pm[(size_t)i * Mp + j] = (j < i) ? -v[r] : (j == i ? -0.5 * v[r] : 0.0);
                         }
                       }, ring);
    } else {
      product<WG, TU, false, ORD_ROWMAJOR>(mt, mt, true, GAT, AT, Mp, nullptr, pm_range,
                       [=](int i0, int j0, const d4& v) {
                         const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
                         d4 pv;
#pragma unroll
                         for (int r = 0; r < 4; ++r) {
                           const int i = i0 + lq + 4 * r, j = j0 + lr;
                           pv[r] = (j < i) ? -v[r] : (j == i ? -0.5 * v[r] : 0.0);
                         }
                         store_tile(pv, nullptr, Pm, Mp, i0, j0, tile);
                       }, ring);
    }
    STAMP_MERGED(9);
    // G_KX^T = G_A^T LI (only the transposed form is used: kernel gradients); formed as the product whose
    // output IS the transposed matrix, so that the epilogue is plain row stores   (Q = LI[k][i], non-zero for k >= i)
    product<WG, TU, false, ORD_COLMAJOR>(mt, mt, false, GA, f.mat[B_LI], Mp, nullptr,
                       [=](int, int i0, int* lo, int* hi) { *lo = i0; *hi = Mp; },
                       [=](int n, int i, const d4& v) { store_tile(v, GKXT, nullptr, Mp, n, i, tile); }, ring);
    __syncthreads();
    stamp(10);
    // G_Kzz (unsymmetrised) = L^-T Pm L^-1, associated as L^-T (Pm L^-1) (round 3): W = Pm L^-1 is a product of two
    // lower-triangular matrices (M^3 / 3, lower-triangular itself), S = L^-T W then costs 2 M^3 / 3 -- 1.0 M^3 where
    // (L^-T Pm) L^-1, rounds 1-2's order, spends 2/3 + 1.  Same value in exact arithmetic; the rounding differs at 1e-16.
    stamp(11);
    // W = Pm L^-1 (lower) -> the G_LS slot, which nothing else writes in this kernel: its upper blocks ARE zero, as the
    // hulls of S's ranges assume   (j0 <= k < i0 + tile: Pm^T[k][i] = 0 for k > i, L^-1[k][j] = 0 for k < j)
    gd* Wm = f.mat[B_GLS];
    auto w_epi = [=](int i0, int j0, const d4& v) {
                         const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
#pragma unroll
                         for (int r = 0; r < 4; ++r) {
                           const int i = i0 + lq + 4 * r, j = j0 + lr;
                           Wm[(size_t)i * Mp + j] = (j <= i) ? v[r] : 0.0;
                         }
                       };
    auto w_range = [=](int i0, int j0, int* lo, int* hi) { *lo = j0; *hi = i0 + TS; };
    if constexpr (KMIN)
      product<WG, TU, false, ORD_ROWMAJOR, 1, 0>(mt, mt, true, Pm, f.mat[B_LI], Mp, nullptr, w_range, w_epi, ring);
    else
      product<WG, TU, false, ORD_ROWMAJOR>(mt, mt, true, Pm, f.mat[B_LI], Mp, nullptr, w_range, w_epi, ring);
    __syncthreads();
    stamp(12);
    // S = L^-T W -> G in the BM buffer, G^T in the A buffer   (k >= max(i0, j0))
    gd* G = BM;
    gd* GT = A;
    product<WG, TU, false, ORD_SHELLS>(mt, mt, false, f.mat[B_LI], Wm, Mp, nullptr,
                       [=](int i0, int j0, int* lo, int* hi) { *lo = i0 > j0 ? i0 : j0; *hi = Mp; },
                       [=](int i, int j, const d4& v) { store_tile(v, G, GT, Mp, i, j, tile); }, ring);
    __syncthreads();
    stamp(13);
    // kernel gradients + Adam on Z
    double g_s, g_l;
    kernel_grads_adam_z<DMAX, true, (DMAX <= 8 ? 2 : 1), DC>(Zt, Pt, G, GT, GKXT, s, inv_l2, step_size, bc2s, scratch, &g_s,
                                                         &g_l);
    g_s += gv_sum;
    g_l /= (ell * ell * ell);
    stamp(14);

    // ------------------------------- Adam (m, scalars) ----------------------
    auto adam_upd = [&](double p, double& m1, double& m2, double g) {
      m1 = b1 * m1 + (1.0 - b1) * g;
      m2 = b2 * m2 + (1.0 - b2) * g * g;
      return p - step_size * m1 / (sqrt(m2) / bc2s + aeps);
    };
    for (int i = threadIdx.x; i < M; i += NT) {
      const double g = f.vec[V_GM][i] + vm[i] / Nd;
      f.vec[V_GM][i] = g;
      double m1 = f.vec[V_MM][i], m2 = f.vec[V_VM][i];
      vm[i] = adam_upd(vm[i], m1, m2, g);
      f.vec[V_MM][i] = m1;
      f.vec[V_VM][i] = m2;
    }
    if (threadIdx.x == 0) {
      double m1, m2;
      m1 = f.scal[S_MC]; m2 = f.scal[S_VC];
      sh.c = adam_upd(sh.c, m1, m2, g_c);
      f.scal[S_MC] = m1; f.scal[S_VC] = m2;
      m1 = f.scal[S_MRS]; m2 = f.scal[S_VRS];
      sh.rho_s = adam_upd(sh.rho_s, m1, m2, g_s * sigmoid(sh.rho_s));
      f.scal[S_MRS] = m1; f.scal[S_VRS] = m2;
      m1 = f.scal[S_MRL]; m2 = f.scal[S_VRL];
      sh.rho_l = adam_upd(sh.rho_l, m1, m2, g_l * sigmoid(sh.rho_l));
      f.scal[S_MRL] = m1; f.scal[S_VRL] = m2;
    }
    __syncthreads();
    stamp(16);
  #ifdef GAPRO_STEP_FN
  };
#pragma nounroll
  for (int step = 1; step <= opt.training_iter; ++step) step_fn(step);
#else
  }
#endif

  // ------------------------------- prediction ------------------------------
  refresh_hypers();
  if (!(opt.eval_stale_chol && opt.training_iter > 0)) factorize();
  const double s = sh.s, inv_l2 = sh.inv_l2, c = sh.c;
  for (int t0 = 0; t0 < T; t0 += Mp) {
    const int nc = (T - t0) < Mp ? (T - t0) : Mp;
    __syncthreads();
    stage_points_t(Pt, f.Xt + (size_t)t0 * D, nc, D, Mp);
    __syncthreads();
    build_kx<DC>(Zt, Pt, nc, s, inv_l2);
    __syncthreads();
    forward_products(nc, s, jitter);
    for (int n = threadIdx.x; n < nc; n += NT) {
      const double mu = f.vec[V_MU][n] + c;
      const double var = fmax(f.vec[V_VAR][n], opt.min_variance);
      const double p = 0.5 * erfc(-(mu / sqrt(1.0 + var)) * 0.70710678118654752440);
      const float pf = (float)p;                       // pred_probs            :432
      const bool lab = pf >= 0.5f;                     // pred_labels           :433
      const long long o = desc.out_offset + t0 + n;
      o_probs[o] = pf;
      o_probs_new[o] = lab ? pf : 1.0f - pf;           // pred_probs_new        :438
      o_labels[o] = lab ? 1 : 0;
      o_mu[o] = (float)mu;                             // pred_mu               :435
      o_var[o] = (float)var;                           // pred_variance         :436
      if ((!isfinite(mu) || !isfinite(var)) && sh.status == GAPRO_OK) sh.status = GAPRO_ERR_NOT_FINITE;  // first error wins
    }
    __syncthreads();
  }
  stamp(17);
#ifdef GAPRO_PROFILE
  if (threadIdx.x == 0)
{
      for (int i = 0; i < kProfSlots; ++i) f.scal[24 + i] = (double)sh.prof[i];
      unsigned xcc, hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      f.scal[24 + 25] = (double)sh.t_start;  // timeline of the launch: tools/fit_timeline.py
      f.scal[24 + 26] = (double)wall_clock64();
      f.scal[24 + 27] = (double)(((xcc & 15u) << 16) | (hwid & 0xFFFFu));
    }
#endif
  if (threadIdx.x == 0) {
    f.scal[S_C] = sh.c;
    f.scal[S_RS] = sh.rho_s;
    f.scal[S_RL] = sh.rho_l;
    f.scal[S_LOSS] = last_loss;
    *loss_out = last_loss;
  }
}

// Common prologue of the fit kernels: workspace pointers, parameter initialisation
// (gaussian_process_utils.py:386-403 and gpytorch's parameter inits), staging of Z and X into LDS.
__device__ inline void fit_setup(const gapro_fit_desc& desc, int D, const float* __restrict__ feats_spp,
                                 const int* __restrict__ idx, const double* __restrict__ init_mean,
                                 double* __restrict__ ws, ldsd* Zt, ldsd* Pt) {
  Shared& sh = g_sh;
  Fit& f = sh.f;
  const Layout lay = make_layout(desc.m1 + desc.m2, desc.t, D);
  gd* base = (gd*)(ws + desc.ws_offset);
  if (threadIdx.x == 0) {
    f.M = desc.m1 + desc.m2;
    f.M1 = desc.m1;
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
    sh.c = 0.0;
    sh.rho_s = 0.0;
    sh.rho_l = 0.0;
    sh.status = GAPRO_OK;
    sh.chol_bad = 0;
#ifdef GAPRO_PROFILE
    for (int i = 0; i < kProfSlots; ++i) sh.prof[i] = 0;
    sh.t_last = wall_clock64();
    sh.t_start = sh.t_last;
#endif
  }
  const int M = desc.m1 + desc.m2, Mp = lay.Mp;
  // zero everything the kernel reads before writing: parameters/Adam state, padded operand tails
  for (long long i = threadIdx.x; i < lay.total; i += NT) base[i] = 0.0;
  __syncthreads();
  const int* my_idx = idx + desc.idx_offset;
  for (int e = threadIdx.x; e < M * D; e += NT) {
    const int i = e / D, d = e - i * D;
    const double v = (double)feats_spp[(size_t)my_idx[i] * D + d];  // train_x = cat(b1_feats, b2_feats)  :395
    f.X[e] = v;
    f.Z[e] = v;  // inducing points initialised to train_x  (:14)
  }
  for (int e = threadIdx.x; e < desc.t * D; e += NT) {
    const int i = e / D, d = e - i * D;
    f.Xt[e] = (double)feats_spp[(size_t)my_idx[M + i] * D + d];  // intersect_feats  :386
  }
  for (int i = threadIdx.x; i < M; i += NT) {
    f.vec[V_Y][i] = i < desc.m1 ? -1.0 : 1.0;  // train_y  :396-398
    f.vec[V_M][i] = init_mean ? init_mean[desc.idx_offset + i] : 0.0;
    f.mat[B_LS][(size_t)i * Mp + i] = 1.0;  // chol_variational_covar = I
    f.mat[B_LST][(size_t)i * Mp + i] = 1.0;
  }
  __syncthreads();
  stage_points_t(Zt, f.Z, M, D, Mp);
  stage_points_t(Pt, f.X, M, D, Mp);
  __syncthreads();
}

__device__ inline void fit_epilogue(const gapro_fit_desc& desc, const gapro_fit_options& opt, int* o_status,
                                    double* o_loss) {
  __syncthreads();
  if (threadIdx.x == 0) {
    int st = g_sh.status;
    if (st == GAPRO_OK && !isfinite(o_loss[desc.slot]) && opt.training_iter > 0) st = GAPRO_ERR_NOT_FINITE;
    o_status[desc.slot] = st;
    g_sh.f.scal[S_STATUS] = (double)st;
  }
}

// WPS = waves per SIMD the register budget is sized for: kWavesPerSimd (two workgroups per CU) for a full
// launch, 2 (one workgroup per CU, 256 VGPRs, no spills in the body) when the launch has fewer fits than CUs
// KMIN: the copy-free product forms (fit_body), for fits up to M_p = kKminMaxMp -- a function of M_p alone, so that a fit
// has the same bits in every build; the launcher (gapro_svgp_fit_batch) sends a fit to the instantiation of its M_p.
// One form per kernel: with both bodies in one kernel either loses ~2 % (registers, code size).
template <int WPS, bool KMIN>
__global__ __launch_bounds__(NT, WPS) void k_svgp_fit(int n_fits, int D, const float* __restrict__ feats_spp,
                                                 const int* __restrict__ idx, const gapro_fit_desc* __restrict__ descs,
                                                 const double* __restrict__ init_mean, gapro_fit_options opt,
                                                 double* __restrict__ ws, float* __restrict__ o_probs,
                                                 float* __restrict__ o_probs_new, unsigned char* __restrict__ o_labels,
                                                 float* __restrict__ o_mu, float* __restrict__ o_var,
                                                 int* __restrict__ o_status, double* __restrict__ o_loss,
                                                 unsigned* ticket) {
  extern __shared__ double dyn_lds[];
  const int fit = claim_fit(ticket);
  if (fit >= n_fits) return;
  const gapro_fit_desc desc = descs[fit];
  const int Mp = gapro_pad_m(desc.m1 + desc.m2, D);
  ldsd* Zt = (ldsd*)dyn_lds;
  ldsd* Pt = Zt + D * Mp;
  ldsd* scratch = Pt + D * Mp;
  fit_setup(desc, D, feats_spp, idx, init_mean, ws, Zt, Pt);
  double* loss_slot = &o_loss[desc.slot];
  // the reference's two feature widths (xyz+rgb = 6, deep features = 32) get a compile-time D: the distance
  // loops unroll and their LDS reads are issued together; any other D <= 32 runs the generic body
#define GAPRO_FIT_KM(TUV, DM, DCV, WGV)                                                                       \
  fit_body<TUV, DM, DCV, WGV, KMIN, (WPS == 2 ? 4 : 1)>(opt, Zt, Pt, scratch, desc, o_probs, o_probs_new, o_labels, o_mu, \
                                                        o_var, loss_slot)
#define GAPRO_FIT_BODY(DM, DCV)                                                                              \
  do {                                                                                                       \
    if (Mp > kFuseMaxMp && Mp % 32 == 0 && !(opt.reserved & 131072) &&                                       \
        ((opt.reserved & 8192) ? !(WPS == 2 && (opt.reserved & 16384)) : Mp % 128 == 0))                      \
      /* workgroup-tiled products (gemm_wg; bit-identical to the per-wave ones) where whole 128 x 128 tiles   \
         cover the matrix: M_p = 256, 384 (+6 % / +4 % fits/s, a quarter less traffic; neutral to -7 % at     \
         the other sizes: DESIGN 6.0).  Bit 13: every M_p > 128 that is a multiple of 32 (bit 14: not in the  \
         one-per-CU build); bit 17: nowhere */                                                                \
      GAPRO_FIT_KM(1, DM, DCV, (WPS == 2 ? 4 : 2));                                                            \
    else if (WPS == 2 && DM == 6 && !KMIN && Mp >= kTu4MinMp && Mp % 32 == 0 && !(opt.reserved & 4096)) {   \
      if constexpr (WPS == 2 && DM == 6 && !KMIN)                                                            \
        fit_body<4, DM, DCV>(opt, Zt, Pt, scratch, desc, o_probs, o_probs_new, o_labels, o_mu, o_var, loss_slot); \
    } else if (Mp >= 128)                                                                                    \
      GAPRO_FIT_KM(2, DM, DCV, 0);                                                                             \
    else                                                                                                     \
      GAPRO_FIT_KM(1, DM, DCV, 0);                                                                             \
  } while (0)
  if (D == 6) GAPRO_FIT_BODY(6, 6);
  else if (D == 32) GAPRO_FIT_BODY(32, 32);
  else GAPRO_FIT_BODY(32, 0);
#undef GAPRO_FIT_BODY
#undef GAPRO_FIT_KM
  fit_epilogue(desc, opt, o_status, o_loss);
}

// =====================================================================================================
// Strip-streaming variant (M_p <= 128): the data-dependent intermediates never touch global memory.
//
// Everything between the triangular factors and the gradients is column-wise in the data index n:
//   KX[:,n] -> A[:,n] = LI KX[:,n] -> B[:,n] = LS^T A[:,n] -> mu_n, var_n -> g_mu_n, g_v_n
//   -> G_A[:,n] = m g_mu_n + LS (2 g_v_n B[:,n]) - 2 g_v_n A[:,n] -> G_KX[:,n] = LI^T G_A[:,n]
// and the only couplings across n are sums:  G_LS += A[:,n] (2 g_v_n B[:,n])^T,  G_L -= G_KX[:,n] A[:,n]^T,
// G_m += A[:,n] g_mu_n, and the kernel-gradient sums.  So the training points are streamed in strips of
// SW = 32 columns held in three LDS buffers (row stride RS = 34: conflict-free for both MFMA operand
// patterns), the two M x M gradient matrices are accumulated as MFMA tiles that stay in registers for
// the whole step (each wave owns up to five 16x16 lower tiles of each), and KX, A, A^T, B, B^T, G_A, G_KX^T
// -- 9 matrix writes and ~13 matrix reads per step in the staged kernel -- disappear from HBM traffic.
// What still streams from L2/HBM per strip are the fixed operands U, LS, LS^T, LI (triangular halves).
// =====================================================================================================
constexpr int SW = 32;  // strip width (data columns)
constexpr int RS = 34;  // LDS row stride of a strip buffer (doubles)
constexpr int kStripMaxMp = 128;
constexpr int kAccTiles = 5;  // lower 16x16 tiles of an 8x8-block matrix: 36 over 8 waves

inline __host__ __device__ int strip_region_doubles(int Mp) {
  const int a = 3 * Mp * RS, b = scratch_doubles(Mp), c = 2 * (Mp * 17 + 64 * 17);  // c: two Cholesky panels
  const int m = a > b ? a : b;
  return m > c ? m : c;
}
inline __host__ __device__ long long strip_lds_bytes(int m, int d) {
  const int Mp = gapro_pad_m(m, d);
  // (+ 8 Mp: the per-row kernel-gradient sums of the fused zx pass, narrow features only)
  return 8LL * (2LL * d * Mp + strip_region_doubles(Mp) + 3 * NT + Mp + 4 * SW + 32 + (d <= 8 ? 8 * Mp : 0));
}
inline __host__ __device__ bool strip_ok(int m, int d) {
  return gapro_pad_m(m, d) <= kStripMaxMp && d <= 32 && strip_lds_bytes(m, d) <= kMaxDynLds;
}

enum { K_LE = 0, K_GE = 1 };
// One strip product: out[16 rb .. ][16 ct ..] = sum_k P[k][16 rb + i] * Sin[k][16 ct + n], k restricted to
// k < 16 (rb+1) (K_LE: P upper-triangular in (k,i)) or k >= 16 rb (K_GE).  Wave rb owns row block rb and both
// 16-column tiles of the strip (at most 8 k-blocks, 16 MFMA groups).  The A operand streams from global memory
// (TN rows); the fixed operand matrices of 256 concurrent fits do not stay in L2, so a fetch costs ~1 us under
// load: ALL of the wave's A fragments (<= 32 loads) are issued up front and the MFMAs consume them in order,
// paying that latency once per product instead of once per k-block.  The B operand comes from the LDS strip buffer.
constexpr int kStripBlocks = kStripMaxMp / 16 + 1;
template <int MODE, typename Epi>
__device__ __noinline__ void strip_gemm(const gd* __restrict__ P, int Mp, const ldsd* Sin, int nbk, Epi epi) {
  // its own function on purpose: the caller keeps 80 accumulator registers alive across it (callee-saved
  // VGPRs), and in here the 36 operand loads must all be in flight at once without a spill between them
  P = uni_ptr(P);
  Mp = uni(Mp);
  nbk = uni(nbk);
  const int wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int lr = lane & 15, lq = lane >> 4;
  // One row block per wave, BOTH column tiles of the strip: every A fragment is fetched once per workgroup (dealing
  // the 2 nbk tiles out one by one balances the MFMAs better, 9 blocks per wave instead of up to 16, but fetches
  // every fragment twice, and the strip products wait for their operands, not for the matrix cores).
  const int rb = wave;
  if (rb >= nbk) return;
  const int kb0 = MODE == K_LE ? 0 : rb, n = MODE == K_LE ? rb + 1 : nbk - rb;  // first k-block, block count
  const size_t sa = (size_t)4 * Mp;
  double a[kStripBlocks][4];
#pragma unroll
  for (int blk = 0; blk < kStripBlocks; ++blk) {
    if (blk < n) {
      const gd* pa = P + (size_t)(16 * (kb0 + blk) + lq) * Mp + 16 * rb + lr;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[blk][ks] = pa[ks * sa];
    }
  }
  d4 acc0 = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int blk = 0; blk < kStripBlocks; ++blk) {
    if (blk < n) {
      const ldsd* pb = Sin + (16 * (kb0 + blk) + lq) * RS + lr;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[blk][ks], pb[4 * ks * RS], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[blk][ks], pb[4 * ks * RS + 16], acc1, 0, 0, 0);
      }
    }
  }
  epi(rb, 0, acc0);
  epi(rb, 1, acc1);
}

__device__ inline void lower_tile(int t, int* ti, int* tj) {
  int i = 0;
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  *ti = i;
  *tj = t - i * (i + 1) / 2;
}

// The register-hungry, MFMA-free parts of a strip are separate functions: values that live across a call
// (the gradient tiles) are kept in callee-saved VGPRs instead of being spilled around inlined libm code.
template <int DC>
__device__ __noinline__ void strip_fill_kx(ldsd* Cs, const ldsd* Zt, const ldsd* Xpts, int n0, int nc, double s,
                                           double inv_l2) {
  const Fit& f = g_sh.f;
  const int Mp = f.Mp, M = f.M, D = DC ? DC : f.D;
  for (int idx = threadIdx.x; idx < Mp * SW; idx += NT) {
    const int k = idx / SW, n = idx - k * SW;
    double v = 0.0;
    if (k < M && n < nc) v = s * gapro_fit_math::rbf_exp(-0.5 * inv_l2 * sqdist_t(Zt, k, Xpts, n0 + n, D, Mp));
    Cs[k * RS + n] = v;
  }
  __syncthreads();
}

// mu_s[n] = sum_i m[i] As[i][n],  var_s[n] = s + jitter + sum_i (Bs[i][n]^2 - As[i][n]^2) for the SW strip columns
__device__ __noinline__ void strip_mean_var(const ldsd* As, const ldsd* Bs, const ldsd* m_s, ldsd* sred, ldsd* mu_s,
                                            ldsd* var_s, double s, double jitter) {
  const int Mp = g_sh.f.Mp;
  const int n = threadIdx.x % SW, p = threadIdx.x / SW;  // NT / SW row groups
  double pm = 0.0, pv = 0.0;
  for (int i = p; i < Mp; i += NT / SW) {
    const double a = As[i * RS + n], b = Bs[i * RS + n];
    pm += m_s[i] * a;
    pv += b * b - a * a;
  }
  sred[threadIdx.x] = pm;
  sred[NT + threadIdx.x] = pv;
  __syncthreads();
  if (threadIdx.x < SW) {
    double sm = 0.0, sv = 0.0;
    for (int g = 0; g < NT / SW; ++g) {
      sm += sred[g * SW + threadIdx.x];
      sv += sred[NT + g * SW + threadIdx.x];
    }
    mu_s[threadIdx.x] = sm;
    var_s[threadIdx.x] = s + jitter + sv;
  }
  __syncthreads();
}

#if GAPRO_NT >= 10 * 32
// Likelihood gradients of the strip columns (ten threads per column, one per symmetric Gauss-Hermite pair):
// gmu_s / gv_s for the strip, and this thread's contributions to sum E, sum g_mu, sum g_v in out3[0..2].
__device__ __noinline__ void strip_likelihood(const ldsd* mu_s, const ldsd* var_s, ldsd* gmu_s, ldsd* gv_s, ldsd* sred,
                                              int n0, int nc, double c, double min_variance, double Nd,
                                              bool want_e, double* out3) {
  const Fit& f = g_sh.f;
  const int q = threadIdx.x % 10, nl = threadIdx.x / 10;
  double E = 0.0, dmu = 0.0, dvar = 0.0;
  const bool on = nl < nc;
  if (on) {
    const double mu = mu_s[nl] + c;
    const double vraw = var_s[nl];
    const double var = vraw < min_variance ? min_variance : vraw;
    const double sd = sqrt(2.0 * var);
    const double y = n0 + nl < f.M1 ? -1.0 : 1.0;  // train_y (see quadrature)
    const double t = c_gh_t[q], w = c_gh_w[q];
    gh_pair(y, mu, sd, t, w, want_e, &E, &dmu, &dvar);
  }
  sred[threadIdx.x] = E;
  sred[NT + threadIdx.x] = dmu;
  sred[2 * NT + threadIdx.x] = dvar;
  __syncthreads();
  double e_add = 0.0, gc_add = 0.0, gv_add = 0.0;
  if (on && q == 0) {
    double se = 0.0, sm = 0.0, sv = 0.0;
    for (int qq = 0; qq < 10; ++qq) {
      se += sred[threadIdx.x + qq];
      sm += sred[NT + threadIdx.x + qq];
      sv += sred[2 * NT + threadIdx.x + qq];
    }
    const double ipi = 0.56418958354775628695;  // 1/sqrt(pi)
    const double vraw = var_s[nl];
    const bool clamped = vraw < min_variance;
    const double var = clamped ? min_variance : vraw;
    const double y = n0 + nl < f.M1 ? -1.0 : 1.0;  // train_y (see quadrature)
    const double g1 = -(ipi * sm * y) / Nd;
    const double g2 = clamped ? 0.0 : -(ipi * sv * y / sqrt(2.0 * var)) / Nd;
    gmu_s[nl] = g1;
    gv_s[nl] = g2;
    e_add = ipi * se;
    gc_add = g1;
    gv_add = g2;
  }
  if ((int)threadIdx.x >= nc && threadIdx.x < SW) {
    gmu_s[threadIdx.x] = 0.0;
    gv_s[threadIdx.x] = 0.0;
  }
  out3[0] = e_add;
  out3[1] = gc_add;
  out3[2] = gv_add;
  __syncthreads();
}
#else
// 256 threads cover 25 columns per pass: a 32-column strip takes two
__device__ __noinline__ void strip_likelihood(const ldsd* mu_s, const ldsd* var_s, ldsd* gmu_s, ldsd* gv_s, ldsd* sred,
                                              int n0, int nc, double c, double min_variance, double Nd,
                                              bool want_e, double* out3) {
  const Fit& f = g_sh.f;
  constexpr int kCols = NT / 10;  // columns per pass: 51 with 512 threads (one pass per strip), 25 with 256
  const int q = threadIdx.x % 10, nl0 = threadIdx.x / 10;
  double e_add = 0.0, gc_add = 0.0, gv_add = 0.0;
  for (int cbase = 0; cbase < nc; cbase += kCols) {
    const int nl = cbase + nl0;
    double E = 0.0, dmu = 0.0, dvar = 0.0;
    const bool on = nl0 < kCols && nl < nc;
    if (on) {
      const double mu = mu_s[nl] + c;
      const double vraw = var_s[nl];
      const double var = vraw < min_variance ? min_variance : vraw;
      const double sd = sqrt(2.0 * var);
      const double y = n0 + nl < f.M1 ? -1.0 : 1.0;  // train_y (see quadrature)
      const double t = c_gh_t[q], w = c_gh_w[q];
      gh_pair(y, mu, sd, t, w, want_e, &E, &dmu, &dvar);
    }
    sred[threadIdx.x] = E;
    sred[NT + threadIdx.x] = dmu;
    sred[2 * NT + threadIdx.x] = dvar;
    __syncthreads();
    if (on && q == 0) {
      double se = 0.0, sm = 0.0, sv = 0.0;
      for (int qq = 0; qq < 10; ++qq) {
        se += sred[threadIdx.x + qq];
        sm += sred[NT + threadIdx.x + qq];
        sv += sred[2 * NT + threadIdx.x + qq];
      }
      const double ipi = 0.56418958354775628695;  // 1/sqrt(pi)
      const double vraw = var_s[nl];
      const bool clamped = vraw < min_variance;
      const double var = clamped ? min_variance : vraw;
      const double y = n0 + nl < f.M1 ? -1.0 : 1.0;  // train_y (see quadrature)
      const double g1 = -(ipi * sm * y) / Nd;
      const double g2 = clamped ? 0.0 : -(ipi * sv * y / sqrt(2.0 * var)) / Nd;
      gmu_s[nl] = g1;
      gv_s[nl] = g2;
      e_add += ipi * se;
      gc_add += g1;
      gv_add += g2;
    }
    if (kCols < SW) __syncthreads();  // another pass may follow and reuses sred
  }
  if ((int)threadIdx.x >= nc && threadIdx.x < SW) {
    gmu_s[threadIdx.x] = 0.0;
    gv_s[threadIdx.x] = 0.0;
  }
  out3[0] = e_add;
  out3[1] = gc_add;
  out3[2] = gv_add;
  __syncthreads();
}
#endif

#if GAPRO_NT >= 320
// Adam on LS for the wave's lower tiles in straight-line code (see the comment at its twin inside the kernel, used by
// the 256-thread build): a function of its own so that its registers (a tile in flight, a tile being updated, the
// division and square-root sequences) are allocated apart from the strip loop, whose accumulator tiles arrive here
// by value.  R = rounds of the workgroup; a slot past the wave's last tile works on the wave's first tile with
// its stores redirected to the B slot of the workspace, which this kernel never uses (its B lives in LDS).
template <int R>
__device__ __noinline__ void adam_ls_tiles(d4 g0, d4 g1, d4 g2, d4 g3, d4 g4, double Nd, double step_size, double bc2s,
                                           ldsd* tile) {
  const Fit& f = g_sh.f;
  const int Mp = f.Mp, M = f.M, nbk = Mp / 16, nt_acc = nbk * (nbk + 1) / 2;
  gd* LS = f.mat[B_LS];
  gd* LST = f.mat[B_LST];
  gd* MLS = f.mat[B_MLS];
  gd* VLS = f.mat[B_VLS];
  gd* Pm = f.mat[B_BM];  // the dead buffer
  const int wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const double b1 = 0.9, b2 = 0.999, aeps = 1e-8;
  const d4 gls[5] = {g0, g1, g2, g3, g4};
  double lsv[4], m1v[4], m2v[4], lsn[4], m1n[4], m2n[4];
  auto slot_tile = [&](int q, int* ti, int* tj) {
    const int t = wave + NW * q;
    const bool valid = t < nt_acc;
    lower_tile(valid ? t : wave, ti, tj);
    return valid;
  };
  auto load_tile = [&](int q, double (&l)[4], double (&a1)[4], double (&a2)[4]) {
    int ti, tj;
    slot_tile(q, &ti, &tj);
    const int j = 16 * tj + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t o = (size_t)(16 * ti + lq + 4 * r) * Mp + j;
      l[r] = LS[o];
      a1[r] = MLS[o];
      a2[r] = VLS[o];
    }
  };
  load_tile(0, lsv, m1v, m2v);
#pragma unroll
  for (int q = 0; q < R; ++q) {
    if (q + 1 < R) load_tile(q + 1, lsn, m1n, m2n);
    int ti, tj;
    const bool valid = slot_tile(q, &ti, &tj);
    gd* wLS = uni_ptr(valid ? LS : Pm);
    gd* wMLS = uni_ptr(valid ? MLS : Pm);
    gd* wVLS = uni_ptr(valid ? VLS : Pm);
    gd* wLST = uni_ptr(valid ? LST : Pm);
    const int j = 16 * tj + lr;
    d4 newv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * ti + lq + 4 * r;
      const size_t o = (size_t)i * Mp + j;
      const bool act = j <= i && i < M;
      const double l = act ? lsv[r] : 1.0;
      const double g = gls[q][r] + (l - (i == j ? 1.0 / l : 0.0)) / Nd;
      const double m1 = b1 * m1v[r] + (1.0 - b1) * g;
      const double m2 = b2 * m2v[r] + (1.0 - b2) * g * g;
      const double lnew = l - step_size * m1 / (sqrt(m2) / bc2s + aeps);
      wMLS[o] = act ? m1 : m1v[r];
      wVLS[o] = act ? m2 : m2v[r];
      wLS[o] = act ? lnew : lsv[r];
      newv[r] = act ? lnew : 0.0;
    }
    store_tile(newv, nullptr, wLST, Mp, 16 * ti, 16 * tj, tile);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lsv[r] = lsn[r];
      m1v[r] = m1n[r];
      m2v[r] = m2n[r];
    }
  }
}
#endif

// Kernel gradients through KX, fused into the strip loop (round 5; narrow features): while G_KX[:, n0 .. n0 + 32) is in
// LDS, W_zx = G_KX o KX is formed element by element (kernel value recomputed from the staged points, as the gradient
// pass after the loop did) and summed into per-row accumulators zacc[k][0..5] = sum_n W[k][n] (Z_k - X_n),
// [6] = sum_n G_KX E, [7] = sum_n W d2.  Four adjacent lanes share a row (eight columns each) and are combined by two
// DPP hops.  Before: the strip was stored as G_KX^T rows to the workspace (7 % of a step) and read back by the
// gradient pass, whose zx half was another ~4 % of memory round trips.
template <int DC>
__device__ __noinline__ void strip_kgrad_zx(const ldsd* Gs, const ldsd* Zt, const ldsd* Xpts, ldsd* zacc, int n0, int nc,
                                            double s, double inv_l2) {
  const Fit& f = g_sh.f;
  const int Mp = f.Mp, M = f.M;
  constexpr int D = DC;
  const int row = threadIdx.x >> 2, cg = threadIdx.x & 3;
  double acc[D];
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = 0.0;
  double gs = 0.0, gl = 0.0;
  if (row < M) {
    double zk[D];
#pragma unroll
    for (int d = 0; d < D; ++d) zk[d] = Zt[d * Mp + row];
#pragma unroll 2
    for (int j = 0; j < SW / 4; ++j) {
      const int n = cg * (SW / 4) + j;
      if (n < nc) {
        double t[D];
        double d2 = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          t[d] = zk[d] - Xpts[d * Mp + n0 + n];
          d2 += t[d] * t[d];
        }
        const double e = KG_EXP(-0.5 * inv_l2 * d2);
        const double g = Gs[row * RS + n];
        const double wx = g * s * e;
        gs += g * e;
        gl += wx * d2;
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] += wx * t[d];
      }
    }
  }
  using gapro_fit_math::dpp_mov;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    acc[d] += dpp_mov<0xB1>(acc[d]);  // quad_perm(1, 0, 3, 2)
    acc[d] += dpp_mov<0x4E>(acc[d]);  // quad_perm(2, 3, 0, 1)
  }
  gs += dpp_mov<0xB1>(gs);
  gs += dpp_mov<0x4E>(gs);
  gl += dpp_mov<0xB1>(gl);
  gl += dpp_mov<0x4E>(gl);
  if (cg == 0 && row < M) {
    ldsd* z = zacc + row * 8;
#pragma unroll
    for (int d = 0; d < D; ++d) z[d] += acc[d];
    z[6] += gs;
    z[7] += gl;
  }
}

template <int DMAX, int DC>
__device__ void fit_body_strip(const gapro_fit_options& opt, ldsd* Zt, ldsd* Pt, ldsd* region,
                               const gapro_fit_desc& desc, float* __restrict__ o_probs,
                               float* __restrict__ o_probs_new, unsigned char* __restrict__ o_labels,
                               float* __restrict__ o_mu, float* __restrict__ o_var, double* loss_out) {
  const Fit& f = g_sh.f;
  Shared& sh = g_sh;
  const int M = f.M, Mp = f.Mp, D = DC ? DC : f.D, T = f.T;
  const int nbk = Mp / 16, nt_acc = nbk * (nbk + 1) / 2;
  const double Nd = (double)M;
  const double jitter = opt.jitter;
  ldsd* As = region;
  ldsd* Bs = As + Mp * RS;
  ldsd* Cs = Bs + Mp * RS;
  ldsd* scratch = region;  // outside the strip loop the same memory serves Cholesky, tiles and reductions
  ldsd* sred = region + strip_region_doubles(Mp);  // 3 * NT
  ldsd* m_s = sred + 3 * NT;                       // Mp
  ldsd* mu_s = m_s + Mp;                           // SW each
  ldsd* var_s = mu_s + SW;
  ldsd* gmu_s = var_s + SW;
  ldsd* gv_s = gmu_s + SW;
  // narrow features (the reference's xyz + rgb): the zx kernel gradients are summed inside the strip loop
  constexpr bool kFuseKg = DC > 0 && DC <= 6 && !(DMAX > 8);
  ldsd* zacc = gv_s + SW + 32;  // [Mp][8], behind the alignment slack of the vectors
  gd* LS = f.mat[B_LS];
  gd* LST = f.mat[B_LST];
#if GAPRO_NT < 320
  gd* MLS = f.mat[B_MLS];
  gd* VLS = f.mat[B_VLS];
#endif
#if GAPRO_NT < 320
  gd* dead = f.mat[B_BM];   // never read: target of the redirected stores of the straight-line Adam code
#endif
  gd* Pm = f.mat[B_GA];     // Pm^T for the tail products
  gd* T1T = f.mat[B_BMT];
  gd* Gb = f.mat[B_A];
  gd* GTb = f.mat[B_AT];
  gd* GKXT = f.mat[B_GKXT];
  gd* vm = f.vec[V_M];
  const int wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int lr = lane & 15, lq = lane >> 4;
  ldsd* tile = scratch + wave * 16 * 17;
  double last_loss = 0.0;
  // LDS offsets of this lane's operand rows for the accumulator tiles the wave owns
  int offA[kAccTiles], offB[kAccTiles];
#pragma unroll
  for (int q = 0; q < kAccTiles; ++q) {
    int ti = 0, tj = 0;
    if (wave + NW * q < nt_acc) lower_tile(wave + NW * q, &ti, &tj);
    offA[q] = (16 * ti + lr) * RS + lq;
    offB[q] = (16 * tj + lr) * RS + lq;
  }
#ifdef GAPRO_PROFILE
  auto stamp = [&](int id) { prof_stamp(id); };
#else
  auto stamp = [&](int) {};
#endif
  auto refresh_hypers = [&]() {
    __syncthreads();
    if (threadIdx.x == 0) {
      sh.s = softplus(sh.rho_s);
      sh.ell = softplus(sh.rho_l);
      sh.inv_l2 = 1.0 / (sh.ell * sh.ell);
    }
    __syncthreads();
  };
  auto factorize = [&]() {
    stamp(19);
    cholesky_psd_safe<DC, 1>(Zt, scratch, sh.s, sh.inv_l2, jitter, opt.psd_retries, opt.psd_jitter);
    stamp(1);
    tri_inverse_strip(scratch);
    __syncthreads();
    stamp(2);
  };
  // forward part of one strip: Cs = KX(:, n0..), As = LI Cs, Bs = LS^T As, mu_s / var_s for the strip columns
  auto strip_forward = [&](const ldsd* Xpts, int n0, int nc, double s, double inv_l2) {
    strip_fill_kx<DC>(Cs, Zt, Xpts, n0, nc, s, inv_l2);
    stamp(3);
    strip_gemm<K_LE>(f.mat[B_U], Mp, Cs, nbk, [=](int rb, int ct, const d4& v) {
      const int ln = threadIdx.x & 63, c = ln & 15, g4 = ln >> 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) As[(16 * rb + g4 + 4 * r) * RS + 16 * ct + c] = v[r];
    });
    __syncthreads();
    stamp(4);
    strip_gemm<K_GE>(LS, Mp, As, nbk, [=](int rb, int ct, const d4& v) {
      const int ln = threadIdx.x & 63, c = ln & 15, g4 = ln >> 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) Bs[(16 * rb + g4 + 4 * r) * RS + 16 * ct + c] = v[r];
    });
    __syncthreads();
    stamp(7);
    strip_mean_var(As, Bs, m_s, sred, mu_s, var_s, s, jitter);
    stamp(9);
  };

  for (int step = 1; step <= opt.training_iter; ++step) {
    refresh_hypers();
    const double s = sh.s, ell = sh.ell, inv_l2 = sh.inv_l2, c = sh.c;
    const bool last = step == opt.training_iter;
    factorize();
    for (int i = threadIdx.x; i < Mp; i += NT) m_s[i] = vm[i];
    if (kFuseKg)
      for (int i = threadIdx.x; i < 8 * Mp; i += NT) zacc[i] = 0.0;
    d4 gls[kAccTiles], gl[kAccTiles];
#pragma unroll
    for (int q = 0; q < kAccTiles; ++q) {
      gls[q] = (d4){0.0, 0.0, 0.0, 0.0};
      gl[q] = (d4){0.0, 0.0, 0.0, 0.0};
    }
    double gm_acc = 0.0, e_tot = 0.0, gc_part = 0.0, gvs_part = 0.0;
    __syncthreads();
    stamp(3);

    for (int n0 = 0; n0 < M; n0 += SW) {
      const int nc = (M - n0) < SW ? (M - n0) : SW;
      strip_forward(Pt, n0, nc, s, inv_l2);
      // ---- likelihood gradients of the strip columns
      {
        double part[3];
        strip_likelihood(mu_s, var_s, gmu_s, gv_s, sred, n0, nc, c, opt.min_variance, Nd, last, part);
        e_tot += part[0];
        gc_part += part[1];
        gvs_part += part[2];
      }
      stamp(10);
      // ---- G_LS += A GB^T (register tiles), G_m += A g_mu
      double sc2[SW / 4];
#pragma unroll
      for (int ks = 0; ks < SW / 4; ++ks) sc2[ks] = 2.0 * gv_s[4 * ks + lq];
#pragma unroll
      for (int q = 0; q < kAccTiles; ++q) {
        const int t = wave + NW * q;
        if (t < nt_acc) {
          const ldsd* pa = As + offA[q];
          const ldsd* pb = Bs + offB[q];
          // GB = 2 B diag(g_v) is never materialised: the factor rides on the A operand (k index = n)
#pragma unroll
          for (int ks = 0; ks < SW / 4; ++ks)
            gls[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * ks] * sc2[ks], pb[4 * ks], gls[q], 0, 0, 0);
        }
      }
      if (threadIdx.x < Mp) {
        const ldsd* pa = As + threadIdx.x * RS;
        for (int n = 0; n < SW; ++n) gm_acc += pa[n] * gmu_s[n];
      }
      stamp(11);
      // ---- G_A strip -> Cs:  m g_mu^T + LS GB - 2 A diag(g_v)     (LS[i][j] = LST[j][i], j <= i)
      strip_gemm<K_LE>(LST, Mp, Bs, nbk, [=](int rb, int ct, const d4& v) {
        const int ln = threadIdx.x & 63, c = ln & 15, g4 = ln >> 4;
        const int n = 16 * ct + c;
        const double gvn = gv_s[n], gmn = gmu_s[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * rb + g4 + 4 * r;
          Cs[i * RS + n] = 2.0 * gvn * v[r] + m_s[i] * gmn - 2.0 * As[i * RS + n] * gvn;
        }
      });
      __syncthreads();
      stamp(12);
      // ---- G_KX strip -> Bs:  LI^T G_A   (P = LI[k][i], k >= i)
      strip_gemm<K_GE>(f.mat[B_LI], Mp, Cs, nbk, [=](int rb, int ct, const d4& v) {
        const int ln = threadIdx.x & 63, c = ln & 15, g4 = ln >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) Bs[(16 * rb + g4 + 4 * r) * RS + 16 * ct + c] = v[r];
      });
      // ---- L^T G_L -= G_A A^T (register tiles).  The Cholesky backward pass needs Phi(L^T G_L) with
      // G_L = -tril(L^-T G_A A^T); row i of L^T X reads rows k >= i of X only, so the lower triangle of L^T tril(X) is
      // the lower triangle of L^T X = -G_A A^T: no G_L, no product with L^T (the staged kernel has the same identity).
      // Reads the G_A and A strips, not G_KX: no barrier between the G_KX product and this.
#pragma unroll
      for (int q = 0; q < kAccTiles; ++q) {
        const int t = wave + NW * q;
        if (t < nt_acc) {
          const ldsd* pa = Cs + offA[q];
          const ldsd* pb = As + offB[q];
#pragma unroll
          for (int ks = 0; ks < SW / 4; ++ks)
            gl[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * ks], pb[4 * ks], gl[q], 0, 0, 0);
        }
      }
      __syncthreads();
      stamp(15);
      if constexpr (kFuseKg) {
        // ---- kernel gradients through KX while the G_KX strip is on chip (nothing leaves it any more)
        strip_kgrad_zx<DC>(Bs, Zt, Pt, zacc, n0, nc, s, inv_l2);
      } else {
        // ---- G_KX^T rows of the strip -> global (the only intermediate that leaves the chip): the kernel
        // gradient pass after the loop reads G_KX[k][n] as GKXT[n][k], contiguous in k
        for (int idx = threadIdx.x; idx < Mp * nc; idx += NT) {
          const int n = idx / Mp, k = idx - n * Mp;
          GKXT[(size_t)(n0 + n) * Mp + k] = Bs[k * RS + n];
        }
      }
      __syncthreads();
      stamp(18);
    }

    // ---- after the strips: scalars, G_m, ELBO value (last step only)
    const double g_c = block_sum(gc_part);
    const double gv_sum = block_sum(gvs_part);
    if (threadIdx.x < Mp) f.vec[V_GM][threadIdx.x] = gm_acc;
    if (last) {
      const double e_sum = block_sum(e_tot);
      double kl_part = 0.0;
      for (int idx = threadIdx.x; idx < M * M; idx += NT) {
        const int i = idx / M, j = idx - i * M;
        if (j <= i) {
          const double v = LS[(size_t)i * Mp + j];
          kl_part += v * v;
          if (i == j) kl_part -= log(v * v);
        }
      }
      for (int i = threadIdx.x; i < M; i += NT) kl_part += vm[i] * vm[i];
      const double kl = 0.5 * (block_sum(kl_part) - Nd);
      last_loss = -(e_sum / Nd - kl / Nd);
    }
    stamp(6);

    // ---- Adam on LS straight from the register tiles;  G_L tiles -> global for the tail products
    const double b1 = 0.9, b2 = 0.999, aeps = 1e-8;
    const double bc1 = 1.0 - pow(b1, (double)step), bc2s = sqrt(1.0 - pow(b2, (double)step));
    const double step_size = opt.lr / bc1;
#if GAPRO_NT < 320
    // Software-pipelined over the wave's tiles: the loads of tile q+1 are in flight while tile q is updated.  That
    // only works in STRAIGHT-LINE code: s_waitcnt counts memory operations in issue order, and behind any join of
    // two paths (a tile guard, a per-element "if active") the compiler can only wait for vmcnt(0), i.e. for every
    // store of the previous tile to reach memory before the next tile's loads are even consumed.  So: one
    // instantiation per number of rounds (uniform over the workgroup), no guard inside it.  A slot past the wave's
    // last tile runs on the wave's first tile with its stores redirected to a buffer that is dead here (Pm, written
    // in full by the first tail product); inactive elements (upper half of a diagonal tile, padded rows) are loaded
    // and stored back unchanged instead of being skipped.  Bit-identical to the guarded form below.
    auto adam_ls = [&](auto rtag) {
      constexpr int R = decltype(rtag)::value;
      double lsv[4], m1v[4], m2v[4], lsn[4], m1n[4], m2n[4];
      auto slot_tile = [&](int q, int* ti, int* tj) {
        const int t = wave + NW * q;
        const bool valid = t < nt_acc;
        lower_tile(valid ? t : wave, ti, tj);
        return valid;
      };
      auto load_tile = [&](int q, double (&l)[4], double (&a1)[4], double (&a2)[4]) {
        int ti, tj;
        slot_tile(q, &ti, &tj);
        const int j = 16 * tj + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const size_t o = (size_t)(16 * ti + lq + 4 * r) * Mp + j;
          l[r] = LS[o];
          a1[r] = MLS[o];
          a2[r] = VLS[o];
        }
      };
      load_tile(0, lsv, m1v, m2v);
#pragma unroll
      for (int q = 0; q < R; ++q) {
        if (q + 1 < R) load_tile(q + 1, lsn, m1n, m2n);
        int ti, tj;
        const bool valid = slot_tile(q, &ti, &tj);
        gd* wLS = uni_ptr(valid ? LS : dead);
        gd* wMLS = uni_ptr(valid ? MLS : dead);
        gd* wVLS = uni_ptr(valid ? VLS : dead);
        gd* wPmT = uni_ptr(valid ? Pm : dead);
        gd* wLST = uni_ptr(valid ? LST : dead);
        const int j = 16 * tj + lr;
        d4 newv, pv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * ti + lq + 4 * r;
          const size_t o = (size_t)i * Mp + j;
          const bool act = j <= i && i < M;
          const double l = act ? lsv[r] : 1.0;
          const double g = gls[q][r] + (l - (i == j ? 1.0 / l : 0.0)) / Nd;
          const double m1 = b1 * m1v[r] + (1.0 - b1) * g;
          const double m2 = b2 * m2v[r] + (1.0 - b2) * g * g;
          const double lnew = l - step_size * m1 / (sqrt(m2) / bc2s + aeps);
          wMLS[o] = act ? m1 : m1v[r];
          wVLS[o] = act ? m2 : m2v[r];
          wLS[o] = act ? lnew : lsv[r];
          newv[r] = act ? lnew : 0.0;
          pv[r] = (j < i) ? gl[q][r] : (j == i ? 0.5 * gl[q][r] : 0.0);
        }
        store_tile(newv, nullptr, wLST, Mp, 16 * ti, 16 * tj, tile);
        store_tile(pv, nullptr, wPmT, Mp, 16 * ti, 16 * tj, tile);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          lsv[r] = lsn[r];
          m1v[r] = m1n[r];
          m2v[r] = m2n[r];
        }
      }
    };
    if (wave < nt_acc) {
      switch ((nt_acc + NW - 1) / NW) {  // M_p <= 64 (the small-fit route) with four waves: at most three rounds
        case 1: adam_ls(std::integral_constant<int, 1>()); break;
        case 2: adam_ls(std::integral_constant<int, 2>()); break;
        default: adam_ls(std::integral_constant<int, 3>()); break;
      }
    }
#else
    // 512 threads: the same straight-line code, but as a function of its own (adam_ls_tiles).  Inlined here it makes
    // this phase faster and the strip phases slower (-1..-4 % overall, whatever the instantiation count); as a
    // function it is +3 % fits/s at M = 80, 96 and -0.5..-0.8 % at M = 112, 128 (the stores still in flight delay
    // the first tail product).  Keeping the guarded loop beside it for four and five rounds costs 2..3 % everywhere.
    if (wave < nt_acc) {
      switch ((nt_acc + NW - 1) / NW) {  // 64 < M_p <= 128 with eight waves: two to five rounds
        case 1:
        case 2: adam_ls_tiles<2>(gls[0], gls[1], gls[2], gls[3], gls[4], Nd, step_size, bc2s, tile); break;
        case 3: adam_ls_tiles<3>(gls[0], gls[1], gls[2], gls[3], gls[4], Nd, step_size, bc2s, tile); break;
        case 4: adam_ls_tiles<4>(gls[0], gls[1], gls[2], gls[3], gls[4], Nd, step_size, bc2s, tile); break;
        default: adam_ls_tiles<5>(gls[0], gls[1], gls[2], gls[3], gls[4], Nd, step_size, bc2s, tile); break;
      }
    }
    // Pm^T = Phi(L^T G_L)^T tiles -> global for the tail products
#pragma unroll
    for (int q = 0; q < kAccTiles; ++q) {
      const int t = wave + NW * q;
      if (t < nt_acc) {
        int ti, tj;
        lower_tile(t, &ti, &tj);
        const int j = 16 * tj + lr;
        d4 pv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * ti + lq + 4 * r;
          pv[r] = (j < i) ? gl[q][r] : (j == i ? 0.5 * gl[q][r] : 0.0);
        }
        store_tile(pv, nullptr, Pm, Mp, 16 * ti, 16 * tj, tile);
      }
    }
#endif
    __syncthreads();
    stamp(8);

    // ---- tail: G_Kzz = LI^T Pm LI through global memory, Pm = Phi(L^T G_L) from the register tiles above; two TN
    // products, associated as LI^T (Pm LI): W = Pm LI is lower (M^3 / 3), S = LI^T W costs 2 M^3 / 3
    auto tail = [&](auto tu_tag) {
      constexpr int TU = decltype(tu_tag)::value;
      constexpr int TS = 16 * TU;
      const int mt = Mp / 16;  // gemm_tn's extents: 16 x 16 tiles (TU = 1) or half tiles (TU = 2)
      gd* PmT = Pm;   // Pm^T: the upper and diagonal tiles are written, W reads exactly those
      gd* Wm = T1T;   // W: the lower and diagonal tiles are written, S reads exactly those
      gemm_tn<TU, false, 4>(mt, mt, true, PmT, f.mat[B_LI], Mp, nullptr,
                         [=](int i0, int j0, int* lo, int* hi) { *lo = j0; *hi = i0 + TS < Mp ? i0 + TS : Mp; },
                         [=](int i0, int j0, const d4& v) {
                           const int ln = threadIdx.x & 63, c = ln & 15, g4 = ln >> 4;
#pragma unroll
                           for (int r = 0; r < 4; ++r) {
                             const int i = i0 + g4 + 4 * r, j = j0 + c;
                             Wm[(size_t)i * Mp + j] = (j <= i) ? v[r] : 0.0;
                           }
                         });
      __syncthreads();
      gemm_tn<TU, false, 4, ORD_SHELLS>(mt, mt, false, f.mat[B_LI], Wm, Mp, nullptr,
                         [=](int i0, int j0, int* lo, int* hi) { *lo = i0 > j0 ? i0 : j0; *hi = Mp; },
                         [=](int i, int j, const d4& v) { store_tile(v, Gb, GTb, Mp, i, j, tile); });
      __syncthreads();
    };
    if (Mp >= 64)
      tail(std::integral_constant<int, 2>());
    else
      tail(std::integral_constant<int, 1>());
    stamp(13);
    double g_s, g_l;
    if constexpr (kFuseKg) {
      const double gs_zx = block_sum((int)threadIdx.x < M ? zacc[threadIdx.x * 8 + 6] : 0.0);
      const double gl_zx = block_sum((int)threadIdx.x < M ? zacc[threadIdx.x * 8 + 7] : 0.0);
      kernel_grads_adam_z<DMAX, false, 8, DC>(Zt, Pt, Gb, GTb, GKXT, s, inv_l2, step_size, bc2s, scratch, &g_s, &g_l,
                                              zacc);
      g_s += gs_zx;
      g_l += gl_zx;
    } else {
      kernel_grads_adam_z<DMAX, true, (DMAX <= 8 ? 8 : 2), DC>(Zt, Pt, Gb, GTb, GKXT, s, inv_l2, step_size, bc2s, scratch,
                                                           &g_s, &g_l);
    }
    g_s += gv_sum;
    g_l /= (ell * ell * ell);
    stamp(14);

    auto adam_upd = [&](double p, double& m1, double& m2, double g) {
      m1 = b1 * m1 + (1.0 - b1) * g;
      m2 = b2 * m2 + (1.0 - b2) * g * g;
      return p - step_size * m1 / (sqrt(m2) / bc2s + aeps);
    };
    for (int i = threadIdx.x; i < M; i += NT) {
      const double g = f.vec[V_GM][i] + vm[i] / Nd;
      f.vec[V_GM][i] = g;
      double m1 = f.vec[V_MM][i], m2 = f.vec[V_VM][i];
      vm[i] = adam_upd(vm[i], m1, m2, g);
      f.vec[V_MM][i] = m1;
      f.vec[V_VM][i] = m2;
    }
    if (threadIdx.x == 0) {
      double m1, m2;
      m1 = f.scal[S_MC]; m2 = f.scal[S_VC];
      sh.c = adam_upd(sh.c, m1, m2, g_c);
      f.scal[S_MC] = m1; f.scal[S_VC] = m2;
      m1 = f.scal[S_MRS]; m2 = f.scal[S_VRS];
      sh.rho_s = adam_upd(sh.rho_s, m1, m2, g_s * sigmoid(sh.rho_s));
      f.scal[S_MRS] = m1; f.scal[S_VRS] = m2;
      m1 = f.scal[S_MRL]; m2 = f.scal[S_VRL];
      sh.rho_l = adam_upd(sh.rho_l, m1, m2, g_l * sigmoid(sh.rho_l));
      f.scal[S_MRL] = m1; f.scal[S_VRL] = m2;
    }
    __syncthreads();
    stamp(16);
  }

  // ------------------------------- prediction ------------------------------
  refresh_hypers();
  if (!(opt.eval_stale_chol && opt.training_iter > 0)) factorize();
  const double s = sh.s, inv_l2 = sh.inv_l2, c = sh.c;
  for (int i = threadIdx.x; i < Mp; i += NT) m_s[i] = vm[i];
  for (int t0 = 0; t0 < T; t0 += Mp) {
    const int ncx = (T - t0) < Mp ? (T - t0) : Mp;
    __syncthreads();
    stage_points_t(Pt, f.Xt + (size_t)t0 * D, ncx, D, Mp);
    __syncthreads();
    for (int n0 = 0; n0 < ncx; n0 += SW) {
      const int nc = (ncx - n0) < SW ? (ncx - n0) : SW;
      strip_forward(Pt, n0, nc, s, inv_l2);
      if ((int)threadIdx.x < nc) {
        const int n = threadIdx.x;
        const double mu = mu_s[n] + c;
        const double var = fmax(var_s[n], opt.min_variance);
        const double p = 0.5 * erfc(-(mu / sqrt(1.0 + var)) * 0.70710678118654752440);
        const float pf = (float)p;                       // pred_probs            :432
        const bool lab = pf >= 0.5f;                     // pred_labels           :433
        const long long o = desc.out_offset + t0 + n0 + n;
        o_probs[o] = pf;
        o_probs_new[o] = lab ? pf : 1.0f - pf;           // pred_probs_new        :438
        o_labels[o] = lab ? 1 : 0;
        o_mu[o] = (float)mu;                             // pred_mu               :435
        o_var[o] = (float)var;                           // pred_variance         :436
        if ((!isfinite(mu) || !isfinite(var)) && sh.status == GAPRO_OK) sh.status = GAPRO_ERR_NOT_FINITE;  // first error wins
      }
      __syncthreads();
    }
  }
  stamp(17);
#ifdef GAPRO_PROFILE
  if (threadIdx.x == 0)
{
      for (int i = 0; i < kProfSlots; ++i) f.scal[24 + i] = (double)sh.prof[i];
      unsigned xcc, hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      f.scal[24 + 25] = (double)sh.t_start;  // timeline of the launch: tools/fit_timeline.py
      f.scal[24 + 26] = (double)wall_clock64();
      f.scal[24 + 27] = (double)(((xcc & 15u) << 16) | (hwid & 0xFFFFu));
    }
#endif
  if (threadIdx.x == 0) {
    f.scal[S_C] = sh.c;
    f.scal[S_RS] = sh.rho_s;
    f.scal[S_RL] = sh.rho_l;
    f.scal[S_LOSS] = last_loss;
    *loss_out = last_loss;
  }
}

// one workgroup per CU: the register-resident gradient tiles need the full 256-VGPR budget
template <int DMAX, int DC>
__global__ __launch_bounds__(NT, 2) void k_svgp_fit_strip(int n_fits, int D, const float* __restrict__ feats_spp,
                                                        const int* __restrict__ idx,
                                                        const gapro_fit_desc* __restrict__ descs,
                                                        const double* __restrict__ init_mean, gapro_fit_options opt,
                                                        double* __restrict__ ws, float* __restrict__ o_probs,
                                                        float* __restrict__ o_probs_new,
                                                        unsigned char* __restrict__ o_labels, float* __restrict__ o_mu,
                                                        float* __restrict__ o_var, int* __restrict__ o_status,
                                                        double* __restrict__ o_loss, unsigned* ticket) {
  extern __shared__ double dyn_lds[];
  const int fit = claim_fit(ticket);
  if (fit >= n_fits) return;
  const gapro_fit_desc desc = descs[fit];
  const int Mp = gapro_pad_m(desc.m1 + desc.m2, D);
  ldsd* Zt = (ldsd*)dyn_lds;
  ldsd* Pt = Zt + D * Mp;
  ldsd* region = Pt + D * Mp;
  fit_setup(desc, D, feats_spp, idx, init_mean, ws, Zt, Pt);
  double* loss_slot = &o_loss[desc.slot];
  fit_body_strip<DMAX, DC>(opt, Zt, Pt, region, desc, o_probs, o_probs_new, o_labels, o_mu, o_var, loss_slot);
  fit_epilogue(desc, opt, o_status, o_loss);
}

#ifdef GAPRO_DEBUG_TU
// ---- product engines side by side (debug entry, tools/product_bench.py) --------------------------------
// Every workgroup owns three M_p x M_p matrices (P, Q, C) of a slab and computes C = P^T Q `reps` times with one of the
// staged kernel's product engines, plain-store epilogue: engine 0 = gemm_tn with 32 x 32 wave tiles, 1 = 64 x 64 wave
// tiles, 2 = the workgroup-tiled form (gemm_wg on whole 128 x 128 tiles + per-wave strips at the edge).  shape 0: full contraction range; 1: the
// range [0, i0 + 16) of a lower-triangular P (the A = L^-1 K product's); 2: lower-triangular output, full range.
template <int ENGINE>
__global__ __launch_bounds__(NT, 2) void k_product_bench(int Mp, int reps, int shape, double* __restrict__ slab) {
  extern __shared__ double dyn_lds[];
  ldsd* scratch = (ldsd*)dyn_lds;
  gd* P = (gd*)slab + (size_t)blockIdx.x * 3 * Mp * Mp;
  gd* Q = P + (size_t)Mp * Mp;
  gd* Cm = Q + (size_t)Mp * Mp;
  if (threadIdx.x == 0) {
    g_sh.f.M = Mp;
    g_sh.f.M1 = Mp / 2;
    g_sh.f.Mp = Mp;
  }
  __syncthreads();
  const int shp = shape;
  constexpr int TSZ = ENGINE == 0 ? 32 : ENGINE == 1 ? 64 : 16;  // the engine's own tile: kr is asked per tile
  auto kr = [=](int i0, int j0, int* lo, int* hi) {
    // shapes 3 .. 5 (the caller zeroes the matching triangles of P and Q, so that any superset of a range gives the
    // same bits): 3 = [max(i0, j0), Mp) (T1), 4 = [j0, Mp) (B, G), 5 = [i0, Mp) (G_KX, Pm; with a lower output)
    *lo = shp == 3 ? (i0 > j0 ? i0 : j0) : shp == 4 ? j0 : shp == 5 ? i0 : 0;
    *hi = shp == 1 ? i0 + TSZ : Mp;
  };
  auto epi = [=](int i0, int j0, const d4& v) {
    const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) Cm[(size_t)(i0 + lq + 4 * r) * Mp + j0 + lr] = v[r];
  };
  for (int r = 0; r < reps; ++r) {
    if (ENGINE == 0) gemm_tn<2, false, 2, ORD_ROWS_DESC, true>(Mp / 16, Mp / 16, shp == 2 || shp == 5, P, Q, Mp, nullptr, kr, epi);
    else if (ENGINE == 1) gemm_tn<4, false, 2, ORD_ROWS_DESC, true>(Mp / 32, Mp / 32, shp == 2 || shp == 5, P, Q, Mp, nullptr, kr, epi);
    else if (ENGINE == 2) product<4, 1, false, ORD_ROWS_DESC>(Mp / 16, Mp / 16, shp == 2 || shp == 5, P, Q, Mp, nullptr, kr, epi, scratch);
    else return;  // (engine 3 was a two-team form of engine 2: no faster, taken out again -- DESIGN 6.0)
    __syncthreads();
  }
}

// ---- MFMA layout self-test (debug entry, used by tests/test_fit_gpu.py) ------------------------------
__global__ void k_mfma_selftest(const double* __restrict__ P, const double* __restrict__ Q, double* __restrict__ C,
                                int K) {
  // C[16][16] = sum_k P[k][i] Q[k][j], ld = 16
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k < K; k += 4)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(P[(k + lq) * 16 + lr], Q[(k + lq) * 16 + lr], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[(lq + 4 * r) * 16 + lr] = acc[r];
  if (K < 0) {  // never taken: keeps the four-block form of mfma64.h (an experiment of round 3) compiling (debug TU)
    d4 t = (d4){0.0, 0.0, 0.0, 0.0};
    gapro_mfma::mma16(P[lr], Q[lr], t);
    t = gapro_mfma::unrotate(t);
    C[lane] = t[0];
  }
}

#endif  // GAPRO_DEBUG_TU

}  // namespace

#ifdef GAPRO_SMALL_TU
// ---- small-fit translation unit (svgp_fit_small.hip builds this file with GAPRO_NT = 256) -----------------
// M_p <= 64 has at most 8 tiles per strip product: with 8 waves each wave has one tile and the CU idles through every
// memory round trip of the fit it hosts.  Here a fit gets 4 waves (256 VGPRs each, no tighter register budget than
// the 512-thread kernel) and a CU hosts TWO fits.
extern "C" int gapro_launch_fit_strip_small(void* stream, int n_fits, int n_wg, unsigned* d_ticket, int feat_dim,
                                            size_t lds_bytes, const float* d_feats_spp, const int32_t* d_idx,
                                            const gapro_fit_desc* d_descs, const double* d_init_mean,
                                            const gapro_fit_options* opt, double* d_workspace, float* d_probs,
                                            float* d_probs_new, uint8_t* d_labels, float* d_mu, float* d_var,
                                            int32_t* d_fit_status, double* d_fit_loss) {
  auto kern = feat_dim == 6 ? k_svgp_fit_strip<6, 6> : feat_dim == 32 ? k_svgp_fit_strip<32, 32> : k_svgp_fit_strip<32, 0>;
  if (lds_bytes > 48 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
    return GAPRO_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(n_wg), dim3(NT), lds_bytes, (hipStream_t)stream, n_fits, feat_dim, d_feats_spp, d_idx,
                     d_descs, d_init_mean, *opt, d_workspace, d_probs, d_probs_new, d_labels, d_mu, d_var, d_fit_status,
                     d_fit_loss, d_ticket);
  return hipGetLastError() == hipSuccess ? GAPRO_OK : GAPRO_ERR_HIP;
}
// LDS bytes of a small fit in THIS translation unit's layout (NT-dependent reduction scratch)
extern "C" long long gapro_fit_strip_small_lds_bytes(int m, int feat_dim) { return strip_lds_bytes(m, feat_dim); }
#elif defined(GAPRO_DEBUG_TU)
// ---- debug translation unit (svgp_fit_debug.hip -> libgapro_hip_debug.so; include/gapro_hip_debug.h) ------------
__global__ void k_stream_calib(long long n, const double* __restrict__ src, double* __restrict__ dst, int mode) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double v = src[i];
    if (mode == 1) dst[i] = v;
    else acc += v;
  }
  if (mode == 0) {
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&dst[blockIdx.x], acc);
  }
}

extern "C" {

// Debug: the staged kernel's product engines side by side (k_product_bench); d_slab: n_wg * 3 * mp * mp doubles,
// filled by the caller.  Returns the launch's milliseconds (HIP events on `stream`, blocking) in *out_ms.
int gapro_debug_product_bench(gapro_ctx* ctx, void* stream_, int32_t engine, int32_t shape, int32_t mp, int32_t reps,
                              int32_t n_wg, double* d_slab, float* out_ms) {
  if (!ctx || !d_slab || !out_ms || engine < 0 || engine > 2 || shape < 0 || shape > 5 || mp < 128 || mp % 32 ||
      reps <= 0 || n_wg <= 0)
    return GAPRO_ERR_BAD_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int lds = 8 * kWgRingDoubles + 1024;
  hipEvent_t e0, e1;
  GAPRO_HIP_CHECK(ctx, hipEventCreate(&e0));
  GAPRO_HIP_CHECK(ctx, hipEventCreate(&e1));
  auto launch = [&](int n) -> int {
#define GAPRO_PB(E)                                                                                              \
  do {                                                                                                           \
    GAPRO_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_product_bench<E>,                                    \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds));                 \
    hipLaunchKernelGGL(k_product_bench<E>, dim3(n_wg), dim3(NT), (size_t)lds, stream, (int)mp, n, (int)shape, d_slab); \
  } while (0)
    if (engine == 0) GAPRO_PB(0);
    else if (engine == 1) GAPRO_PB(1);
    else GAPRO_PB(2);
#undef GAPRO_PB
    return GAPRO_OK;
  };
  if (launch(1) != GAPRO_OK) return GAPRO_ERR_HIP;
  GAPRO_HIP_CHECK(ctx, hipEventRecord(e0, stream));
  if (launch(reps) != GAPRO_OK) return GAPRO_ERR_HIP;
  GAPRO_HIP_CHECK(ctx, hipEventRecord(e1, stream));
  GAPRO_HIP_CHECK(ctx, hipEventSynchronize(e1));
  GAPRO_HIP_CHECK(ctx, hipEventElapsedTime(out_ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  GAPRO_LAUNCH_CHECK(ctx);
  return GAPRO_OK;
}

// Debug: C = P^T Q for 16-column operands with K rows (K % 4 == 0); checks the MFMA lane maps.
int gapro_debug_mfma_tn(gapro_ctx* ctx, void* stream_, const double* d_P, const double* d_Q, double* d_C, int32_t K) {
  if (!ctx || !d_P || !d_Q || !d_C || K <= 0 || (K & 3)) return GAPRO_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_mfma_selftest, dim3(1), dim3(64), 0, (hipStream_t)stream_, d_P, d_Q, d_C, (int)K);
  GAPRO_LAUNCH_CHECK(ctx);
  return GAPRO_OK;
}

// Debug: streaming kernels with a known byte count in this library's own access pattern (one double per
// lane, grid-stride), used to calibrate the FETCH_SIZE / WRITE_SIZE counters.  mode 0: read n doubles and
// write one partial sum per workgroup; mode 1: copy n doubles.
// Debug: the fit kernels' own special functions (fit_math.h) evaluated elementwise, for tests/test_fit_gpu.py
__global__ void k_fit_math(long long n, const double* __restrict__ x, double* __restrict__ out, int which) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double lp, r;
  switch (which) {
    case 0: out[i] = gapro_fit_math::erfcx_tab(x[i]); break;
    case 1: out[i] = gapro_fit_math::exp_neg(x[i]); break;
    case 2: out[i] = ndtr_ratio(x[i]); break;
    default:
      log_ndtr_ratio(x[i], &lp, &r);
      out[i] = which == 3 ? lp : r;
  }
}
int gapro_debug_fit_math(gapro_ctx* ctx, void* stream_, int64_t n, const double* d_x, double* d_out, int32_t which) {
  if (!ctx || !d_x || !d_out || n <= 0 || which < 0 || which > 4) return GAPRO_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_fit_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, (long long)n, d_x,
                     d_out, (int)which);
  GAPRO_LAUNCH_CHECK(ctx);
  return GAPRO_OK;
}

int gapro_debug_stream(gapro_ctx* ctx, void* stream_, int64_t n, const double* d_src, double* d_dst, int32_t mode) {
  if (!ctx || !d_src || !d_dst || n <= 0 || mode < 0 || mode > 1) return GAPRO_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_stream_calib, dim3(4096), dim3(256), 0, (hipStream_t)stream_, (long long)n, d_src, d_dst,
                     (int)mode);
  GAPRO_LAUNCH_CHECK(ctx);
  return GAPRO_OK;
}

}  // extern "C"
#else
extern "C" int gapro_launch_fit_strip_small(void* stream, int n_fits, int n_wg, unsigned* d_ticket, int feat_dim,
                                            size_t lds_bytes, const float* d_feats_spp, const int32_t* d_idx,
                                            const gapro_fit_desc* d_descs, const double* d_init_mean,
                                            const gapro_fit_options* opt, double* d_workspace, float* d_probs,
                                            float* d_probs_new, uint8_t* d_labels, float* d_mu, float* d_var,
                                            int32_t* d_fit_status, double* d_fit_loss);
extern "C" long long gapro_fit_strip_small_lds_bytes(int m, int feat_dim);

// Conditioning figure of every fit of a launch (round 6): every kernel leaves the inverses of the 16 x 16 diagonal blocks of
// its LAST Cholesky factor in the fit's workspace (f.dinv: the triangular solves read them; the wave-per-fit kernel, which
// keeps L^-1 on chip, stores the diagonal on its way out), and the diagonal of L^-1 is 1 / L_jj.  One wave per fit:
// cond[slot] = (max_j L_jj / min_j L_jj)^2 over the M real rows -- a lower bound of cond_2(K_ZZ + jitter I) that costs a
// few hundred loads per fit.  Fits whose sigma^2 no float64 implementation reproduces to 1e-4 are exactly
// the ones where this figure is large; the caller gets it next to the status instead of finding out from an oracle.
__global__ __launch_bounds__(256) void k_fit_cond(int n_fits, int D, const gapro_fit_desc* __restrict__ descs,
                                                  const double* __restrict__ ws, double* __restrict__ out) {
  const int fit = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (fit >= n_fits) return;
  const gapro_fit_desc d = descs[fit];
  const int M = d.m1 + d.m2;
  const Layout lay = make_layout(M, d.t, D);
  const double* dinv = ws + d.ws_offset + lay.dinv;
  double rmin = 1e300, rmax = 0.0;
  for (int j = lane; j < M; j += 64) {
    const double r = fabs(dinv[(size_t)(j >> 4) * 256 + 17 * (j & 15)]);
    rmin = fmin(rmin, r);
    rmax = fmax(rmax, r);
  }
  for (int o = 32; o > 0; o >>= 1) {
    rmin = fmin(rmin, __shfl_xor(rmin, o, 64));
    rmax = fmax(rmax, __shfl_xor(rmax, o, 64));
  }
  if (lane == 0) {
    const double q = rmax / rmin;  // = max L_jj / min L_jj
    out[d.slot] = q * q;
  }
}

// 0 = strip-streaming kernel, 1 = LDS-staged kernel, 2 = generic kernel, 3 = strip-streaming kernel of the small-fit
// translation unit (M_p <= 64: 256 threads per fit, two fits per CU), 4 = cluster kernel (one fit over several
// workgroups), 5 = the wave-per-fit kernel (svgp_fit_wave.hip: M_p <= 48 at feat_dim 6).  flags:
// gapro_fit_options.reserved debug bits (bit 0: never the strip kernels, bit 2: no small-fit kernel, bit 3: no cluster
// kernel, bit 4: the cluster kernel for every fit it can take, M_p >= 64 and M_p % 32 == 0, bit 20: no wave kernel).
constexpr int kSmallFitMp = 64;
constexpr int kNoWaveFlag = 1 << 20;
static int fit_route(int m, int feat_dim, int flags) {
  if (!(flags & kNoWaveFlag) && gapro_pad_m(m, feat_dim) <= gapro_fit_wave_max_mp(feat_dim)) return 5;
  // large fits: spread over several CUs (svgp_fit_cluster.hip); debug bit 3 keeps them on one workgroup
  if (!(flags & 8) && feat_dim <= 32 && gapro_cluster_size(gapro_pad_m(m, feat_dim), (flags & 16) != 0) > 0) return 4;
  if (!(flags & 1) && strip_ok(m, feat_dim)) {
    const bool small = gapro_pad_m(m, feat_dim) <= kSmallFitMp && 2 * gapro_fit_strip_small_lds_bytes(m, feat_dim) + 16384 <= 160 * 1024;
    return (small && !(flags & 4)) ? 3 : 0;
  }
  if (staged_ok(m, feat_dim)) return 1;
  // neither LDS-resident kernel takes it (deep features: Z and X of M_p > 192 points at D = 32 do not fit beside the
  // Cholesky block column): the cluster kernel keeps the points in global memory and runs such a fit on one workgroup
  if (!(flags & 8) && feat_dim <= 32 && gapro_cluster_size(gapro_pad_m(m, feat_dim), true) > 0) return 4;
  return 2;
}

extern "C" {

int64_t gapro_fit_workspace_doubles(int32_t m, int32_t t, int32_t feat_dim) {
  if (m <= 0 || t < 0 || feat_dim <= 0) return 0;
  return make_layout(m, t, feat_dim).total;
}

int64_t gapro_fit_plan_workspace(gapro_fit_desc* h_descs, int32_t n_fits, int32_t feat_dim) {
  if (!h_descs || n_fits < 0) return 0;
  int64_t off = 0;
  for (int i = 0; i < n_fits; ++i) {
    h_descs[i].ws_offset = off;
    off += gapro_fit_workspace_doubles(h_descs[i].m1 + h_descs[i].m2, h_descs[i].t, feat_dim);
  }
  return off * (int64_t)sizeof(double);
}

int gapro_fit_workspace_layout(int32_t m, int32_t t, int32_t feat_dim, int64_t* out8) {
  if (!out8 || m <= 0 || feat_dim <= 0) return GAPRO_ERR_BAD_ARG;
  const Layout L = make_layout(m, t, feat_dim);
  out8[0] = L.Mp; out8[1] = L.mat; out8[2] = L.vec; out8[3] = L.xz; out8[4] = L.xt; out8[5] = L.dinv;
  out8[6] = L.scal; out8[7] = L.total;
  return GAPRO_OK;
}

int gapro_svgp_fit_batch(gapro_ctx* ctx, void* stream_, int32_t n_fits, int32_t feat_dim, const float* d_feats_spp,
                         const int32_t* d_idx, const gapro_fit_desc* h_descs, gapro_fit_desc* d_descs,
                         const double* d_init_mean, const gapro_fit_options* opt, double* d_workspace,
                         size_t workspace_bytes, float* d_probs, float* d_probs_new, uint8_t* d_labels, float* d_mu,
                         float* d_var, int32_t* d_fit_status, double* d_fit_loss) {
  return gapro_svgp_fit_batch_ex(ctx, stream_, n_fits, feat_dim, d_feats_spp, d_idx, h_descs, d_descs, d_init_mean, opt,
                                 d_workspace, workspace_bytes, d_probs, d_probs_new, d_labels, d_mu, d_var, d_fit_status,
                                 d_fit_loss, nullptr);
}

int gapro_svgp_fit_batch_ex(gapro_ctx* ctx, void* stream_, int32_t n_fits, int32_t feat_dim, const float* d_feats_spp,
                            const int32_t* d_idx, const gapro_fit_desc* h_descs, gapro_fit_desc* d_descs,
                            const double* d_init_mean, const gapro_fit_options* opt, double* d_workspace,
                            size_t workspace_bytes, float* d_probs, float* d_probs_new, uint8_t* d_labels, float* d_mu,
                            float* d_var, int32_t* d_fit_status, double* d_fit_loss, double* d_fit_cond) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  if (n_fits == 0) return GAPRO_OK;
  if (n_fits < 0 || feat_dim <= 0 || !d_feats_spp || !d_idx || !h_descs || !d_descs || !opt || !d_workspace ||
      !d_probs || !d_probs_new || !d_labels || !d_mu || !d_var || !d_fit_status || !d_fit_loss || workspace_bytes == 0)
    return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_svgp_fit_batch: bad argument");
  if (opt->training_iter < 0 || !(opt->lr > 0.0) || !(opt->jitter >= 0.0) || opt->psd_retries < 0 ||
      opt->psd_retries > 8 || !(opt->psd_jitter >= 0.0))
    return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_svgp_fit_batch: bad options");
  hipStream_t stream = (hipStream_t)stream_;
  // Routing (gapro_fit_route): strip-streaming kernel, LDS-staged kernel, generic kernel (working set beyond
  // LDS).  Every group is sorted longest processing time first (cost ~ M^3): workgroups are dispatched in
  // block order, so the expensive fits start first and the tail of a launch stays short.
  std::vector<gapro_fit_desc> strip, small, staged, large, clus, wave[3];  // wave[nb - 1]: M_p = 16 nb
  strip.reserve(n_fits);
  small.reserve(n_fits);
  long long need = 0, max_lds = 0, max_lds_strip = 0, max_lds_small = 0;
  const int route_flags = opt->reserved;
  for (int i = 0; i < n_fits; ++i) {
    gapro_fit_desc d = h_descs[i];
    const int m = d.m1 + d.m2;
    if (d.m1 <= 0 || d.m2 <= 0 || d.t < 0)
      return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_svgp_fit_batch: fit %d has an empty side", i);
    d.slot = i;
    need = std::max<long long>(need, (d.ws_offset + gapro_fit_workspace_doubles(m, d.t, feat_dim)) * 8LL);
    const int route = fit_route(m, feat_dim, route_flags);
    if (route == 0) {
      strip.push_back(d);
      max_lds_strip = std::max(max_lds_strip, strip_lds_bytes(m, feat_dim));
    } else if (route == 3) {
      small.push_back(d);
      max_lds_small = std::max(max_lds_small, gapro_fit_strip_small_lds_bytes(m, feat_dim));
    } else if (route == 1) {
      staged.push_back(d);
      max_lds = std::max(max_lds, staged_lds_bytes(m, feat_dim));
    } else if (route == 4) {
      clus.push_back(d);
    } else if (route == 5) {
      wave[gapro_pad_m(m, feat_dim) / 16 - 1].push_back(d);
    } else {
      large.push_back(d);
      // beyond the cluster kernel's cap (kClusterMaxMp: its merge inverse keeps one LDS record per pair of panels) a
      // fit runs on ONE workgroup of the generic kernel: correct, but minutes per fit -- never silently
      if (feat_dim <= 32 && gapro_pad_m(m, feat_dim) > gapro_fit::kClusterMaxMp) {
        static bool warned = false;
        if (!warned) {
          warned = true;
          fprintf(stderr, "libgapro_hip: a GP fit with M = %d inducing points exceeds the multi-workgroup kernel's cap "
                          "(M_p <= %d): it runs on one workgroup of the generic kernel (minutes per fit)\n", m,
                  gapro_fit::kClusterMaxMp);
        }
      }
    }
  }
  if ((size_t)need > workspace_bytes)
    return gapro_fail(ctx, GAPRO_ERR_WORKSPACE, "gapro_svgp_fit_batch: workspace too small (%lld > %zu)", need,
                      workspace_bytes);
  auto by_cost = [](const gapro_fit_desc& a, const gapro_fit_desc& b) { return a.m1 + a.m2 > b.m1 + b.m2; };
  std::stable_sort(strip.begin(), strip.end(), by_cost);
  std::stable_sort(small.begin(), small.end(), by_cost);
  std::stable_sort(staged.begin(), staged.end(), by_cost);
  std::stable_sort(large.begin(), large.end(), by_cost);
  std::stable_sort(clus.begin(), clus.end(), by_cost);
  for (auto& w : wave) std::stable_sort(w.begin(), w.end(), by_cost);
  // the staged fits whose LDS fits a CU twice and the larger ones are two launches (see below)
  const long long kTwice = 72 * 1024;
  size_t nbig = 0, nkmaj = 0;  // sorted by M, the LDS need grows with M: [0, nkmaj) M_p > kKminMaxMp, [0, nbig) "big"
  long long lds_big = 0, lds_rest = 0, lds_kmaj = 0;
  for (const gapro_fit_desc& d : staged) {
    const long long b = staged_lds_bytes(d.m1 + d.m2, feat_dim);
    // (the product forms of a fit are a function of its M_p, k_svgp_fit's KMIN: fits beyond kKminMaxMp are a launch of
    // their own -- with D = 6 exactly the fits that need more than kTwice)
    const bool kmaj = gapro_pad_m(d.m1 + d.m2, feat_dim) > kKminMaxMp;
    if (kmaj) {
      ++nkmaj;
      lds_kmaj = std::max(lds_kmaj, b);
    }
    if (kmaj || (b > kTwice && !(route_flags & 128))) {
      ++nbig;
      if (!kmaj) lds_big = std::max(lds_big, b);
    } else {
      lds_rest = std::max(lds_rest, b);
    }
  }
  // Workgroup b of a launch runs on XCD b % 8 (one slice of 32 CUs; the cluster kernel's same-XCD barrier relies on the
  // same fact and checks it), so in longest-first order XCD 0 would get the largest fit of every group of eight and XCD
  // 7 the smallest, in every kernel: at the end of a launch four XCDs sat idle for ~50 ms while the others still had
  // their share of strip fits (tools/fit_timeline.py).  Every second group of eight is reversed.
  auto serpentine = [](std::vector<gapro_fit_desc>& v, size_t lo, size_t hi) {
    for (size_t g0 = lo + 8; g0 + 8 <= hi; g0 += 16) std::reverse(v.begin() + g0, v.begin() + g0 + 8);
  };
  // Default since round 4: ticket counters (claim_fit) -- the fits stay in plain longest-first order and are taken by
  // the workgroups in the order in which they start; debug bit 18 restores the static mapping fit = blockIdx.x
  const bool tickets = !(route_flags & 262144) && ctx->d_tickets;
  if (!(route_flags & 512) && !tickets) {
    serpentine(strip, 0, strip.size());
    serpentine(small, 0, small.size());
    serpentine(staged, 0, nbig);
    serpentine(staged, nbig, staged.size());
  }
  std::vector<gapro_fit_desc> all(large);
  all.insert(all.end(), staged.begin(), staged.end());
  all.insert(all.end(), strip.begin(), strip.end());
  all.insert(all.end(), small.begin(), small.end());
  const size_t clus_base = all.size();
  all.insert(all.end(), clus.begin(), clus.end());
  size_t wave_base[3];
  for (int k = 2; k >= 0; --k) {  // the largest first
    wave_base[k] = all.size();
    all.insert(all.end(), wave[k].begin(), wave[k].end());
  }
  if (!clus.empty()) {  // staging of the cluster kernel: block table + one barrier counter line per fit
    const size_t need_stage = gapro_cluster_stage_bytes((int)clus.size());
    if (2 * need_stage > ctx->cl_stage_bytes) {  // cl_stage_bytes counts BOTH halves; a launch uses one of them
      if (ctx->h_cl_stage) (void)hipHostFree(ctx->h_cl_stage);
      if (ctx->d_cl_stage) (void)hipFree(ctx->d_cl_stage);
      ctx->h_cl_stage = ctx->d_cl_stage = nullptr;
      ctx->cl_stage_bytes = 0;
      GAPRO_HIP_CHECK(ctx, hipHostMalloc(&ctx->h_cl_stage, 2 * need_stage, hipHostMallocDefault));
      GAPRO_HIP_CHECK(ctx, hipMalloc(&ctx->d_cl_stage, 2 * need_stage));
      ctx->cl_stage_bytes = 2 * need_stage;
    }
    if (clus.size() > ctx->cl_ctl_fits) {
      if (ctx->d_cl_ctl) (void)hipFree(ctx->d_cl_ctl);
      ctx->d_cl_ctl = nullptr;
      ctx->cl_ctl_fits = 0;
      GAPRO_HIP_CHECK(ctx, hipMalloc((void**)&ctx->d_cl_ctl, 2 * 2 * clus.size() * 128));  // two halves
      ctx->cl_ctl_fits = 2 * clus.size();
    }
  }
  // ticket counters of this launch: [0] staged beyond kKminMaxMp, [1] staged "big", [2] the other staged fits, [3] strip,
  // [4] small, [5..7] cluster kernel diagnostics, [8..10] wave kernels.  The sets rotate, so that a set is zeroed again
  // only kTicketSets launches later
  unsigned* tk = nullptr;
  if (tickets) {
    tk = ctx->d_tickets + (size_t)(ctx->ticket_seq++ % gapro_ctx::kTicketSets) * gapro_ctx::kTicketsPerSet;
    GAPRO_HIP_CHECK(ctx, hipMemsetAsync(tk, 0, gapro_ctx::kTicketsPerSet * sizeof(unsigned), stream));
  }
  // cluster kernel: block table and barrier counters go up on `stream` too, so that after the synchronisation below the
  // kernel is the first thing its stream has to do
  int cl_blocks = 0, cl_members = 0, cl_par = 0;
  char* cl_d_half = nullptr;
  unsigned* cl_ctl_half = nullptr;
  if (!clus.empty()) {
    std::vector<int> fi(clus.size()), fmp(clus.size()), fg(clus.size());
    for (size_t k = 0; k < clus.size(); ++k) {
      fi[k] = (int)(clus_base + k);
      fmp[k] = gapro_pad_m(clus[k].m1 + clus[k].m2, feat_dim);
      fg[k] = gapro_cluster_size(fmp[k], true);
    }
    // the block table is built in one half of the context's pinned buffer, the halves alternating per launch: the
    // copy of launch i - 2 has long been consumed (callers collect a launch before they issue the one after next),
    // so the host never waits for the previous cluster kernel here
    // (the device copies alternate the same way, so that two launches issued from different streams -- debug bit 1
    // puts the kernels on the caller's stream -- never share a block table or a barrier counter)
    const size_t par = ctx->cl_parity & 1;
    cl_par = (int)par;
    // this half's previous user may still be running on another stream (overlapping launches, several caller streams):
    // the upload and the counter reset below wait for it
    if (ctx->clus_half_used[par]) GAPRO_HIP_CHECK(ctx, hipStreamWaitEvent(stream, ctx->ev_clus_half[par], 0));
    char* h_half = (char*)ctx->h_cl_stage + par * (ctx->cl_stage_bytes / 2);
    cl_d_half = (char*)ctx->d_cl_stage + par * (ctx->cl_stage_bytes / 2);
    cl_ctl_half = ctx->d_cl_ctl + par * ctx->cl_ctl_fits * 32;
    ctx->cl_parity++;
    const int rc = gapro_prepare_fit_cluster(stream, (int)clus.size(), fi.data(), fmp.data(), fg.data(), h_half, cl_d_half,
                                             cl_ctl_half, &cl_blocks, &cl_members);
    if (rc != GAPRO_OK) return gapro_fail(ctx, rc, "gapro_svgp_fit_batch: cluster block table upload failed");
  }
  // grid of a ticketed kernel: twice its fits, so that an XCD (a fixed eighth of the grid) can run up to twice its share
  auto grid_of = [&](size_t n) { return (int)(tickets ? 2 * n : n); };
  GAPRO_HIP_CHECK(ctx, hipMemcpyAsync(d_descs, all.data(), all.size() * sizeof(gapro_fit_desc), hipMemcpyHostToDevice,
                                      stream));
  GAPRO_HIP_CHECK(ctx, hipStreamSynchronize(stream));  // `all` is pageable host memory that dies with this call
  // The kernels go to the context's two fit streams so that the staged kernel (a few large fits, which leave
  // most CUs idle) and the strip kernel run side by side.  No fork event is needed: `stream` has just been
  // synchronised, so everything the kernels read is complete (and an event recorded here completes together
  // with the NEXT dispatch of `stream` under this runtime, which would serialise the kernels again).  Both
  // are joined back into `stream` with events.
  bool own = !(route_flags & 2);  // debug bit 1
  for (int k = 0; k < gapro_ctx::kFitStreams; ++k) own = own && ctx->fit_stream[k];
  hipStream_t s_staged = own ? ctx->fit_stream[0] : stream;
  hipStream_t s_strip = own ? ctx->fit_stream[1] : stream;
  // (round 4, tried: the small-fit kernel behind the strip kernel on ITS stream, so that the two-per-CU small fits fill
  // the end of the launch instead of competing with the whole-CU strip fits for every CU half that frees up: 341.9 /
  // 336.5 scenes/s against 337.7 / 341.6, same box, alternating -- nothing)
  hipStream_t s_small = own ? ctx->fit_stream[2] : stream;
  hipStream_t s_clus = own ? ctx->fit_stream[3] : stream;
  hipStream_t s_staged2 = own ? ctx->fit_stream[4] : stream;
  gapro_fit_timing* tm = ctx->armed_timing;
  ctx->armed_timing = nullptr;
  if (tm) {
    tm->tickets = tk;
    tm->used[0] = !large.empty() || !staged.empty();
    tm->used[1] = !strip.empty();
    tm->used[2] = !small.empty();
    tm->used[3] = !clus.empty();
    tm->used[4] = !wave[0].empty() || !wave[1].empty() || !wave[2].empty();
  }
  bool gate_staged2 = false;
  if (!clus.empty()) {  // first: the largest fits of the launch, each over several CUs
    if (tm) GAPRO_HIP_CHECK(ctx, hipEventRecord(tm->ev[6], s_clus));
    // (Round 4, tried: holding the other kernels back until the cluster's members are resident.  They are within 1 .. 15
    // us anyway, and every cluster sits on one XCD (gapro_fit_timing_cluster_info); what makes the same cluster kernel
    // take 240 ms in one run of a step and 500 ms in the next is what runs beside it -- LABNOTES R4.4.)
    const int rc = gapro_launch_fit_cluster(s_clus, cl_blocks, feat_dim, cl_d_half, cl_ctl_half, tk ? tk + 5 : nullptr,
                                            d_feats_spp, d_idx, d_descs, d_init_mean, *opt, d_workspace, d_probs,
                                            d_probs_new, d_labels, d_mu, d_var, d_fit_status, d_fit_loss);
    if (rc != GAPRO_OK) return gapro_fail(ctx, rc, "gapro_svgp_fit_batch: cluster kernel launch failed");
    if (tm) GAPRO_HIP_CHECK(ctx, hipEventRecord(tm->ev[7], s_clus));
    GAPRO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_clus_half[cl_par], s_clus));
    ctx->clus_half_used[cl_par] = true;
    // gate of the two-per-CU staged launch (below): policy 2 = always behind the cluster kernel (default), 1 = only when
    // the cluster's members fit the GPU at once, 0 = never (GAPRO_STAGED_GATE; debug bit 19 = never)
    static int gate_policy = -1;
    if (gate_policy < 0) {
      const char* e = getenv("GAPRO_STAGED_GATE");
      gate_policy = e ? atoi(e) : 2;
    }
    gate_staged2 = own && !(route_flags & 524288) &&
                   (gate_policy == 2 || (gate_policy == 1 && cl_members <= ctx->n_cu));
    if (gate_staged2) GAPRO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_gate, s_clus));
  }
  if (tm && tm->used[0]) GAPRO_HIP_CHECK(ctx, hipEventRecord(tm->ev[0], s_staged));
  if (!large.empty())
    gapro_launch_fit_large(s_staged, (int)large.size(), feat_dim, d_feats_spp, d_idx, d_descs, d_init_mean, *opt,
                           d_workspace, d_probs, d_probs_new, d_labels, d_mu, d_var, d_fit_status, d_fit_loss);
  if (!staged.empty()) {
    // The dynamic LDS of a launch is its largest fit's, and beyond ~72 KiB (M_p > 256 at D = 6) a CU holds ONE
    // workgroup whatever the registers allow: a launch with one such fit ran every staged fit one per CU.  So the
    // staged fits go out as two launches side by side: those whose LDS fits a CU twice in the `<4>` build (128 VGPRs,
    // two workgroups per CU: 17 % more fits/s at M <= 256 than one per CU), the larger ones in the `<2>` build (the
    // whole register file, no spills).  Either part with fewer fits than CUs takes `<2>` as well.
    auto launch = [&](hipStream_t st, size_t first, size_t count, long long lds, bool kmaj, int tslot) -> int {
      const bool one_per_cu = (int)count <= ctx->n_cu || lds > kTwice || (route_flags & 64);
      auto kern = kmaj ? k_svgp_fit<2, false> : one_per_cu ? k_svgp_fit<2, true> : k_svgp_fit<kWavesPerSimd, true>;
      if (lds > 48 * 1024)
        GAPRO_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(kern, dim3(grid_of(count)), dim3(NT), (size_t)lds, st, (int)count, (int)feat_dim, d_feats_spp,
                         d_idx, d_descs + large.size() + first, d_init_mean, *opt, d_workspace, d_probs, d_probs_new,
                         d_labels, d_mu, d_var, d_fit_status, d_fit_loss, tk ? tk + tslot : nullptr);
      return GAPRO_OK;
    };
    if (nkmaj > 0) {
      const int rc = launch(s_staged, 0, nkmaj, lds_kmaj, true, 0);
      if (rc != GAPRO_OK) return rc;
    }
    if (nbig > nkmaj) {  // M_p <= kKminMaxMp with an LDS need beyond kTwice (wide features): behind the former
      const int rc = launch(s_staged, nkmaj, nbig - nkmaj, lds_big, false, 1);
      if (rc != GAPRO_OK) return rc;
    }
    if (nbig < staged.size()) {
      hipStream_t st = nbig > 0 ? s_staged2 : s_staged;
      // The two-per-CU staged fits are the launch's memory-bound class (3.8 FLOP/B: with the chip to themselves they
      // sit on the HBM roof), the cluster fits its latency-bound one (a chain of cluster barriers and small products
      // per Adam step).  Side by side from the start, the same cluster kernel took 230 .. 270 ms or 500 .. 800 ms
      // depending on which kernels the runtime happened to put on one hardware queue (LABNOTES R4.4); so the order is
      // explicit: this launch starts when the cluster kernel has ended -- small and strip fits, light on memory, take
      // the CUs beside the clusters meanwhile.  Same box, 2 x 8 steps: 361 .. 372 scenes/s with 4 or 8 hardware queues
      // (GPU_MAX_HW_QUEUES), the cluster kernel at its undisturbed time in every step; without the gate 368 / 372 with
      // the default 4 queues (the lucky mapping) and 351 .. 362 with 8.
      if (gate_staged2) GAPRO_HIP_CHECK(ctx, hipStreamWaitEvent(st, ctx->ev_gate, 0));
      const int rc = launch(st, nbig, staged.size() - nbig, lds_rest, false, 2);
      if (rc != GAPRO_OK) return rc;
      if (st != s_staged) {  // everything staged is finished once s_staged has passed this point
        GAPRO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_join[4], st));
        GAPRO_HIP_CHECK(ctx, hipStreamWaitEvent(s_staged, ctx->ev_join[4], 0));
      }
    }
  }
  if (tm && tm->used[0]) GAPRO_HIP_CHECK(ctx, hipEventRecord(tm->ev[1], s_staged));
  if (!strip.empty()) {
    auto kern = feat_dim == 6 ? k_svgp_fit_strip<6, 6> : feat_dim == 32 ? k_svgp_fit_strip<32, 32> : k_svgp_fit_strip<32, 0>;
    if (max_lds_strip > 48 * 1024)
      GAPRO_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)max_lds_strip));
    if (tm) GAPRO_HIP_CHECK(ctx, hipEventRecord(tm->ev[2], s_strip));
    hipLaunchKernelGGL(kern, dim3(grid_of(strip.size())), dim3(NT), (size_t)max_lds_strip, s_strip, (int)strip.size(),
                       (int)feat_dim, d_feats_spp, d_idx, d_descs + large.size() + staged.size(), d_init_mean, *opt,
                       d_workspace, d_probs, d_probs_new, d_labels, d_mu, d_var, d_fit_status, d_fit_loss,
                       tk ? tk + 3 : nullptr);
    if (tm) GAPRO_HIP_CHECK(ctx, hipEventRecord(tm->ev[3], s_strip));
  }
  if (!small.empty()) {
    // the small fits start behind the cluster kernel as well (GAPRO_SMALL_GATE=0: at once): beside the clusters run
    // the whole-CU strip fits and the one-per-CU staged fits, and every CU half the two-per-CU staged fits leave later
    // goes to a small fit, light on memory, instead of a second memory-bound one -- 367.8 / 367.4 / 362.6 scenes/s
    // against 364.0 / 363.2 / 356.1, alternating on one box
    static int gate_small = -1;
    if (gate_small < 0) gate_small = getenv("GAPRO_SMALL_GATE") ? atoi(getenv("GAPRO_SMALL_GATE")) : 1;
    if (gate_small && gate_staged2) GAPRO_HIP_CHECK(ctx, hipStreamWaitEvent(s_small, ctx->ev_gate, 0));
    if (tm) GAPRO_HIP_CHECK(ctx, hipEventRecord(tm->ev[4], s_small));
    const int rc = gapro_launch_fit_strip_small(s_small, (int)small.size(), grid_of(small.size()), tk ? tk + 4 : nullptr,
                                                feat_dim, (size_t)max_lds_small, d_feats_spp,
                                                d_idx, d_descs + large.size() + staged.size() + strip.size(),
                                                d_init_mean, opt, d_workspace, d_probs, d_probs_new, d_labels, d_mu,
                                                d_var, d_fit_status, d_fit_loss);
    if (rc != GAPRO_OK) return gapro_fail(ctx, rc, "gapro_svgp_fit_batch: small-fit kernel launch failed");
    if (tm) GAPRO_HIP_CHECK(ctx, hipEventRecord(tm->ev[5], s_small));
  }
  if (!wave[0].empty() || !wave[1].empty() || !wave[2].empty()) {
    // One wave per fit (svgp_fit_wave.hip), one kernel per block count, side by side on their own streams; the waves
    // of a kernel take its fits by ticket until none is left.  They touch no memory between their first and last step,
    // so nothing holds them back: they take whatever wave slots and LDS the other kernels leave.
    hipStream_t s_wave[3] = {own ? ctx->fit_stream[7] : stream, own ? ctx->fit_stream[6] : stream,
                             own ? ctx->fit_stream[5] : stream};
    const int first = !wave[2].empty() ? 2 : !wave[1].empty() ? 1 : 0;
    if (tm) GAPRO_HIP_CHECK(ctx, hipEventRecord(tm->ev[8], s_wave[first]));
    for (int k = 2; k >= 0; --k) {
      if (wave[k].empty()) continue;
      const int slots = ctx->n_cu * gapro_fit_wave_per_cu(k + 1, feat_dim);
      const int n_wg = (int)std::min<size_t>(wave[k].size(), (size_t)std::max(slots, 1));
      const int rc = gapro_launch_fit_wave(s_wave[k], k + 1, (int)wave[k].size(), n_wg, tk ? tk + 8 + k : nullptr, feat_dim,
                                           d_feats_spp, d_idx, d_descs + wave_base[k], d_init_mean, *opt, d_workspace,
                                           d_probs, d_probs_new, d_labels, d_mu, d_var, d_fit_status, d_fit_loss);
      if (rc != GAPRO_OK) return gapro_fail(ctx, rc, "gapro_svgp_fit_batch: wave kernel launch failed");
      if (own && k != first) {  // everything wave-per-fit is finished once s_wave[first] has passed this point
        GAPRO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_join[5 + k], s_wave[k]));
        GAPRO_HIP_CHECK(ctx, hipStreamWaitEvent(s_wave[first], ctx->ev_join[5 + k], 0));
      }
    }
    if (tm) GAPRO_HIP_CHECK(ctx, hipEventRecord(tm->ev[9], s_wave[first]));
    if (own) {
      GAPRO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_join[5 + first], s_wave[first]));
      GAPRO_HIP_CHECK(ctx, hipStreamWaitEvent(stream, ctx->ev_join[5 + first], 0));
    }
  }
  if (own) {
    if (!large.empty() || !staged.empty()) {
      GAPRO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_join[0], s_staged));
      GAPRO_HIP_CHECK(ctx, hipStreamWaitEvent(stream, ctx->ev_join[0], 0));
    }
    if (!strip.empty()) {
      GAPRO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_join[1], s_strip));
      GAPRO_HIP_CHECK(ctx, hipStreamWaitEvent(stream, ctx->ev_join[1], 0));
    }
    if (!small.empty()) {
      GAPRO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_join[2], s_small));
      GAPRO_HIP_CHECK(ctx, hipStreamWaitEvent(stream, ctx->ev_join[2], 0));
    }
    if (!clus.empty()) {
      GAPRO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_join[3], s_clus));
      GAPRO_HIP_CHECK(ctx, hipStreamWaitEvent(stream, ctx->ev_join[3], 0));
    }
  }
  // conditioning figures, behind every kernel of the launch (they have been joined into `stream` above)
  if (d_fit_cond)
    hipLaunchKernelGGL(k_fit_cond, dim3((n_fits + 3) / 4), dim3(256), 0, stream, (int)n_fits, (int)feat_dim, d_descs,
                       d_workspace, d_fit_cond);
  GAPRO_LAUNCH_CHECK(ctx);
  return GAPRO_OK;
}

int gapro_fit_route(int32_t m, int32_t feat_dim) { return fit_route(m, feat_dim, 0); }

int gapro_fit_padded_m(int32_t m, int32_t feat_dim) { return gapro_pad_m(m, feat_dim); }

int gapro_fit_timing_create(gapro_ctx* ctx, gapro_fit_timing** out) {
  if (!ctx || !out) return GAPRO_ERR_BAD_ARG;
  gapro_fit_timing* t = new (std::nothrow) gapro_fit_timing();
  if (!t) return GAPRO_ERR_OOM;
  for (int i = 0; i < 2 * gapro_fit_timing::kKernels; ++i)
    if (hipEventCreate(&t->ev[i]) != hipSuccess) {
      gapro_fit_timing_destroy(t);
      return gapro_fail(ctx, GAPRO_ERR_HIP, "gapro_fit_timing_create: hipEventCreate failed");
    }
  *out = t;
  return GAPRO_OK;
}

void gapro_fit_timing_destroy(gapro_fit_timing* t) {
  if (!t) return;
  for (int i = 0; i < 2 * gapro_fit_timing::kKernels; ++i)
    if (t->ev[i]) (void)hipEventDestroy(t->ev[i]);
  delete t;
}

int gapro_fit_timing_arm(gapro_ctx* ctx, gapro_fit_timing* t) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  ctx->armed_timing = t;
  if (t) t->used[0] = t->used[1] = t->used[2] = t->used[3] = t->used[4] = false;
  return GAPRO_OK;
}

static int fit_timing_read(gapro_ctx* ctx, gapro_fit_timing* t, float* out_ms5, float* out_wave_ms) {
  constexpr int NK = gapro_fit_timing::kKernels;
  for (int i = 0; i < 5; ++i) out_ms5[i] = 0.f;
  if (out_wave_ms) *out_wave_ms = 0.f;
  const int slot[NK] = {0, 1, 3, 4, -1};  // staged, strip, small-fit strip, cluster, wave-per-fit
  float start[NK] = {0.f, 0.f, 0.f, 0.f, 0.f}, end[NK] = {0.f, 0.f, 0.f, 0.f, 0.f};
  int ref = -1;
  for (int k = 0; k < NK; ++k) {
    if (!t->used[k]) continue;
    float ms = 0.f;
    GAPRO_HIP_CHECK(ctx, hipEventSynchronize(t->ev[2 * k + 1]));
    GAPRO_HIP_CHECK(ctx, hipEventElapsedTime(&ms, t->ev[2 * k], t->ev[2 * k + 1]));
    if (slot[k] >= 0) out_ms5[slot[k]] = ms;
    else if (out_wave_ms) *out_wave_ms = ms;
    if (ref < 0) ref = k;
    float off = 0.f;  // start of kernel k relative to the first used kernel's start
    if (k != ref) GAPRO_HIP_CHECK(ctx, hipEventElapsedTime(&off, t->ev[2 * ref], t->ev[2 * k]));
    start[k] = off;
    end[k] = off + ms;
  }
  if (ref >= 0) {
    float lo = 0.f, hi = 0.f;
    for (int k = 0; k < NK; ++k)
      if (t->used[k]) {
        lo = start[k] < lo ? start[k] : lo;
        hi = end[k] > hi ? end[k] : hi;
      }
    out_ms5[2] = hi - lo;
  }
  return GAPRO_OK;
}

int gapro_fit_timing_read(gapro_ctx* ctx, gapro_fit_timing* t, float* out_ms5) {
  if (!ctx || !t || !out_ms5) return GAPRO_ERR_BAD_ARG;
  return fit_timing_read(ctx, t, out_ms5, nullptr);
}

int gapro_fit_timing_read_wave(gapro_ctx* ctx, gapro_fit_timing* t, float* out_ms) {
  if (!ctx || !t || !out_ms) return GAPRO_ERR_BAD_ARG;
  float ms5[5];
  return fit_timing_read(ctx, t, ms5, out_ms);
}

int gapro_fit_timing_cluster_info(gapro_ctx* ctx, gapro_fit_timing* t, int32_t* out3) {
  if (!ctx || !t || !out3) return GAPRO_ERR_BAD_ARG;
  out3[0] = out3[1] = out3[2] = 0;
  if (!t->tickets || !t->used[3]) return GAPRO_OK;
  GAPRO_HIP_CHECK(ctx, hipEventSynchronize(t->ev[7]));
  unsigned v[3] = {0, 0, 0};
  GAPRO_HIP_CHECK(ctx, hipMemcpy(v, t->tickets + 5, sizeof(v), hipMemcpyDeviceToHost));
  out3[0] = (int32_t)v[2];
  out3[1] = (int32_t)v[1];
  out3[2] = (int32_t)v[0];
  return GAPRO_OK;
}

int gapro_fit_timing_offsets(gapro_ctx* ctx, gapro_fit_timing* ref, gapro_fit_timing* t, float* out_ms2) {
  if (!ctx || !ref || !t || !out_ms2) return GAPRO_ERR_BAD_ARG;
  out_ms2[0] = out_ms2[1] = 0.f;
  int rk = -1;
  for (int k = 0; k < gapro_fit_timing::kKernels && rk < 0; ++k)
    if (ref->used[k]) rk = k;
  if (rk < 0) return GAPRO_OK;
  GAPRO_HIP_CHECK(ctx, hipEventSynchronize(ref->ev[2 * rk]));
  bool any = false;
  float lo = 0.f, hi = 0.f;
  for (int k = 0; k < gapro_fit_timing::kKernels; ++k) {
    if (!t->used[k]) continue;
    float a = 0.f, b = 0.f;
    GAPRO_HIP_CHECK(ctx, hipEventSynchronize(t->ev[2 * k + 1]));
    GAPRO_HIP_CHECK(ctx, hipEventElapsedTime(&a, ref->ev[2 * rk], t->ev[2 * k]));
    GAPRO_HIP_CHECK(ctx, hipEventElapsedTime(&b, ref->ev[2 * rk], t->ev[2 * k + 1]));
    lo = (!any || a < lo) ? a : lo;
    hi = (!any || b > hi) ? b : hi;
    any = true;
  }
  out_ms2[0] = lo;
  out_ms2[1] = hi;
  return GAPRO_OK;
}

}  // extern "C"
#endif  // GAPRO_SMALL_TU
