// One WAVE per GP fit (gfx950): the small fits of a launch, M_p = 16 NB <= 48, with the whole Adam loop on one wavefront.
//
// Why: a fit of M <= 48 points has 0.3 .. 1 MFLOP per Adam step -- microseconds of matrix work --
// and the workgroup-per-fit kernels of svgp_fit.hip spend 30 .. 100 us on it: ~25 barrier-separated phases per strip
// whose operands make a round trip through the HBM workspace, one wave factoring the 16 x 16 diagonal blocks while the
// others wait.  Here a fit is ONE wave: no workgroup barrier exists, no intermediate ever leaves the CU, and a CU
// hosts eight fits (NB <= 2: 256 registers per lane, two waves per SIMD) or four (NB = 3: the whole register file)
// whose serial chains (diagonal blocks, likelihood) overlap on the four SIMDs.
//
// Same arithmetic as svgp_fit.hip (reference gapro/gaussian_process_utils.py:11-25, 382-445 and the gpytorch objects
// behind it; oracle/svgp_oracle.py): whitened SVGP classifier, Z = X initially, 50 Adam steps on
// {Z, m, tril(L_S), c, rho_s, rho_l}, 20-point Gauss-Hermite Bernoulli likelihood, psd_safe_cholesky retries, then
// mu, sigma^2, Phi(mu / sqrt(1 + sigma^2)) for the test points.  float64 throughout.
//
// Data layout.  Every M_p x M_p matrix lives in REGISTERS as 16 x 16 tiles in the accumulator layout of
// v_mfma_f64_16x16x4_f64: lane l = (lq = l >> 4, lr = l & 15) holds, in register r of a tile, the element
// [lq + 4 r][lr].  A tile in that layout is at the same time a valid A/B operand of the next product in the TN form
//   C[i][j] += sum_k P[k][i] Q[k][j]:   MFMA r of a block takes a = P_tile[r], b = Q_tile[r]  (k = 4 r + lq)
// so products chain with no data movement at all, and a product's TRANSPOSE is the same MFMAs with the operands
// swapped (C^T = Q^T P).  Everything the step needs in both orientations (A and A^T, B and B^T, G_A and G_A^T, the
// inverse factor and its transpose, S and S^T) is therefore computed twice on the matrix cores -- they are idle
// otherwise -- instead of being transposed through memory.  Only L_S lives in LDS (row-major tiles, so that L_S^T is
// the same bytes read with the other index pattern), beside the points (Z^T, X^T: [d][i]), the Adam moments of Z and a
// few vectors.  Per step and NB = 2: ~350 MFMAs, 44 exp, 12 log Phi evaluations per lane, two 16 x 16 diagonal-block
// factorisations; no global-memory access between the first and the last step.
//
// Per Adam step (U = L^T upper tiles, LI = L^-1, LIT = LI^T):
//   U        left-looking block Cholesky of K_ZZ + jitter I in registers; diagonal blocks factored and inverted by the
//            readlane chain of svgp_fit.hip's diag_factor_invert (through one LDS tile for the row-per-lane layout)
//   LI, LIT  block forward substitution, both orientations from the same partial sums
//   KX; A = LI KX, At;  B = L_S^T A (transient: column sums only);  mu, var;  likelihood -> g_mu, g_v
//   B, Bt again;  G_LS = tril(At^T GBt);  G_A = m g_mu^T + L_S GB - 2 A diag(g_v), G_At;  Adam on L_S
//   Pm^T = Phi(-G_A A^T)^T;  G_KX = LI^T G_A;  W = Pm LI;  S = LI^T W, St;  G_Kzz = (S + St) / 2
//   kernel gradients: W_zx = G_KX o KX, W_zz = G_Kzz o K_ZZ elementwise on n-major tiles;
//   G_Z[k] = -sum_n (2 W_zz[k][n] (Z_k - Z_n) + W_zx[k][n] (Z_k - X_n)) / l^2 in the difference form (as a product
//   against [Z | 1] / [X | 1] it cancels digits away);  Adam on Z, m, c, rho_s, rho_l
#include <math.h>

#include "common.h"
#include "fit_layout.h"
#include "fit_math.h"

namespace {
using namespace gapro_fit;
using namespace gapro_fit_math;
using gapro_mfma::d4;
typedef __attribute__((address_space(3))) double ldsd;

constexpr int kTS = 16 * 17;  // doubles of one LDS tile (row stride 17)

// LDS of one fit (doubles)
template <int NB, int DC>
struct WaveLds {
  static constexpr int Mp = 16 * NB, NL = NB * (NB + 1) / 2;
  static constexpr int oZ = 0;                  // Z^T  [DC][Mp]
  static constexpr int oX = oZ + DC * Mp;       // X^T  [DC][Mp]
  static constexpr int oMZ = oX + DC * Mp;      // Adam moments of Z, [DC][Mp] like Z^T
  static constexpr int oVZ = oMZ + DC * Mp;
  static constexpr int oLS = oVZ + DC * Mp;     // tril(L_S): NL row-major tiles
  static constexpr int oLI = oLS + NL * kTS;    // LI = L^-1: NL row-major tiles
  static constexpr int oT0 = oLI + NL * kTS;    // two scratch tiles (diagonal blocks, transposes)
  static constexpr int oT1 = oT0 + kTS;
  static constexpr int oM = oT1 + kTS;          // variational mean m [Mp]
  static constexpr int oY = oM + Mp;            // labels y [Mp]
  static constexpr int oGmu = oY + Mp;          // g_mu [Mp]
  static constexpr int oGv = oGmu + Mp;         // g_v [Mp]
  static constexpr int oTmp = oGv + Mp;         // [Mp]
  static constexpr int oGh = oTmp + Mp;         // Gauss-Hermite nodes [10] and weights [10]
  static constexpr int oSc = oGh + 20;          // c, rho_s, rho_l [3] (their Adam moments: registers of lanes 61..63)
  static constexpr int oRs = oSc + 10;          // wide features: row sums of the kernel-gradient weights [Mp] ...
  static constexpr int oCen = oRs + Mp;         // ... and the centre the points are taken relative to [DC]
#ifdef GAPRO_PROFILE
  static constexpr int oProf = oCen + DC;       // phase clocks of the diagnostic build [16] + the last stamp
  static constexpr int total = (oProf + 18) / 2 * 2;
#else
  static constexpr int total = (oCen + DC + 1) / 2 * 2;
#endif
};

// LDS hand-over between the lanes of the one wave.  A wave's LDS operations execute in order, so all this has to do
// is keep the COMPILER from moving memory accesses across it.  (__builtin_amdgcn_fence(seq_cst, "wavefront", "local")
// does that too, but this compiler emits s_waitcnt vmcnt(0) for it: every hand-over then also waited for the Adam
// moments travelling to and from the workspace and for every register spill in flight -- 2.6 us per step at NB = 2.)
__device__ inline void wsync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
__device__ inline d4 zero4() { return (d4){0.0, 0.0, 0.0, 0.0}; }
// acc += P^T Q for one 16-row block of the contraction index (both operands as accumulator-layout tiles)
__device__ inline d4 tn(d4 acc, const d4& P, const d4& Q) {
#pragma unroll
  for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(P[r], Q[r], acc, 0, 0, 0);
  return acc;
}
__device__ inline d4 tn_neg(d4 acc, const d4& P, const d4& Q) {
#pragma unroll
  for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-P[r], Q[r], acc, 0, 0, 0);
  return acc;
}
__device__ inline int lt(int i, int j) { return i * (i + 1) / 2 + j; }  // index of lower tile (i, j), i >= j

// library calls that occur at several places of the step, each compiled once
__device__ __noinline__ double nl_exp(double x) { return rbf_exp(x); }
__device__ __noinline__ double nl_log(double x) { return log(x); }
__device__ __noinline__ double nl_softplus(double x) { return softplus(x); }
__device__ __noinline__ double nl_sigmoid(double x) { return sigmoid(x); }
// (its own function: inlined into the quadrature loop, the ~60 polynomial coefficients of erfcx are hoisted out of the
// loop into 120 registers that the caller of the loop would have to clear)
struct LpR {
  double lp, r;
};
__device__ __noinline__ LpR nl_log_ndtr_ratio(double z) {
  LpR o;
  log_ndtr_ratio(z, &o.lp, &o.r);
  return o;
}
// r(z) = phi(z) / Phi(z) alone: the 49 steps whose ELBO value nobody reads need no log Phi (same bits as the r above)
__device__ __noinline__ double nl_ndtr_ratio(double z) {
  const double t = lik_erfcx(fabs(z) * 0.70710678118654752440);
  const double e = rbf_exp(-0.5 * z * z);
  const bool neg = z < 0.0;
  return (neg ? 0.79788456080286535588 : e * 0.39894228040143267794) / (neg ? t : 1.0 - 0.5 * e * t);
}

// Factor the symmetric 16 x 16 block S = L L^T and invert L: Dinv = L^-1 and Dinv^T as accumulator-layout tiles.  The
// elimination is svgp_fit.hip's diag_factor_invert: lane r (mod 16) holds row r, pivots and multipliers travel by
// v_readlane; the inverse is a forward substitution, one column per lane, reading L as LDS broadcasts.  Returns
// bad: a pivot was not positive (psd_safe_cholesky's retry condition).
// The elementwise, transcendental-heavy pieces of a step (this one, rbf_tile, lik_column, adam_ls_tile) are functions
// of their own: inlined at every tile they made the NB = 2 kernel 94 KB of straight-line code (the instruction cache
// of a CU pair holds 64 KB) with 900 spilled registers; the MFMA skeleton around them stays unrolled.
struct DiagInv {
  d4 inv, invT;  // (the flag travels in the return value's own register: the struct stays within the register ABI)
};
__device__ __noinline__ DiagInv diag_factor(d4 S, ldsd* t0, ldsd* t1, int* bad_out) {
  const int lane = threadIdx.x & 63, r = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) t0[(lq + 4 * e) * 17 + r] = S[e];
  wsync();
  double rdiag[16];
  bool bad = false;
  {
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = t0[r * 17 + c];
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
    if (lane < 16) {
#pragma unroll
      for (int c = 0; c < 16; ++c) t1[r * 17 + c] = (c <= r) ? a[c] : 0.0;  // L
    }
  }
  wsync();
  {
    double x[16], b[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) b[rr] = (rr == r) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      x[q] = (q >= r) ? b[q] * rdiag[q] : 0.0;
#pragma unroll
      for (int rr = q + 1; rr < 16; ++rr) b[rr] = fma(-t1[rr * 17 + q], x[q], b[rr]);
    }
    if (lane < 16) {
#pragma unroll
      for (int c = 0; c < 16; ++c) t0[c * 17 + r] = x[c];  // Dinv[c][r]: lane r holds column r
    }
  }
  wsync();
  DiagInv o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o.inv[e] = t0[(lq + 4 * e) * 17 + r];
    o.invT[e] = t0[r * 17 + lq + 4 * e];
  }
  *bad_out = bad ? 1 : 0;
  wsync();
  return o;
}

// Squared distances d2 and e = exp(-d2 / (2 l^2)) between the four rows 16 rb + lq + 4 r of the points At and this
// lane's column 16 cb + lr of the points Bt ([d][i] in LDS, leading dimension Mp): one tile of an RBF kernel matrix
struct RbfTile {
  d4 d2, e;
};
template <int DC, int Mp>
__device__ __noinline__ RbfTile rbf_tile(const ldsd* At, int rb, const ldsd* Bt, int cb, double nh_inv_l2) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  double pc[DC];
#pragma unroll
  for (int d = 0; d < DC; ++d) pc[d] = Bt[d * Mp + 16 * cb + lr];
  RbfTile o;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double s2 = 0.0;
#pragma unroll
    for (int d = 0; d < DC; ++d) {
      const double t = At[d * Mp + 16 * rb + lq + 4 * r] - pc[d];
      s2 += t * t;
    }
    o.d2[r] = s2;
    o.e[r] = rbf_exp(nh_inv_l2 * s2);
  }
  return o;
}

// Gauss-Hermite sums of one training column over this lane's node pairs (lq, lq + 4, lq + 8 of the ten symmetric
// pairs; gh: nodes [10] | weights [10] in LDS): E = sum w log Phi(y f), dmu = sum w r, dvar = sum w t r
struct LikSums {
  double E, dmu, dvar;
};
__device__ __noinline__ LikSums lik_column(const ldsd* gh, double mu, double sd, double y, int on, int want_e) {
  const int lq = (threadIdx.x & 63) >> 4;
  LikSums o = {0.0, 0.0, 0.0};
  if (on) {
    // one evaluation per trip (node -t, then +t): two interleaved erfcx evaluations need ~200 registers, which the
    // caller would have to clear around every call
#pragma unroll 1
    for (int q2 = 2 * lq; q2 < 20; q2 += (q2 & 1) ? 7 : 1) {
      const int q = q2 >> 1;
      const double sg = (q2 & 1) ? 1.0 : -1.0;
      const double t = gh[q], w = gh[10 + q];
      const double z = y * (mu + sg * sd * t);
      double r;
      if (want_e) {
        const LpR v = nl_log_ndtr_ratio(z);
        o.E += w * v.lp;
        r = v.r;
      } else {
        r = nl_ndtr_ratio(z);
      }
      o.dmu += w * r;
      o.dvar += sg * (w * t * r);
    }
  }
  return o;
}

// torch.optim.Adam on one tile of tril(L_S) (LDS tile t, row-major) with the gradient tile g of the likelihood term;
// the KL term (l - 1 / l on the diagonal) / N joins here.  i0, j0: the tile's first row / column; moments in registers.
// (Inlined: a call boundary waits for every memory operation in flight, and the moments travel through global memory.)
struct AdamTile {
  d4 m1, m2;
};
template <bool DIAG>
__device__ inline AdamTile adam_ls_tile(ldsd* t, d4 g, d4 m1, d4 m2, int i0, int j0, int M, double Nd,
                                        double step_size, double ibc2s) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  AdamTile o;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = i0 + lq + 4 * r, col = j0 + lr;
    const bool act = col <= row && row < M;
    const double lv = t[(lq + 4 * r) * 17 + lr];
    const double l = act ? lv : 1.0;
    double kl = l;  // KL term of the ELBO: l - 1 / l on the diagonal (only diagonal tiles pay for the division)
    if (DIAG) kl -= row == col ? 1.0 / l : 0.0;
    const double gr = g[r] + kl / Nd;
    const double a1 = 0.9 * m1[r] + (1.0 - 0.9) * gr;
    const double a2 = 0.999 * m2[r] + (1.0 - 0.999) * gr * gr;
    const double lnew = l - step_size * a1 / (sqrt(a2) * ibc2s + 1e-8);
    o.m1[r] = act ? a1 : m1[r];
    o.m2[r] = act ? a2 : m2[r];
    if (act) t[(lq + 4 * r) * 17 + lr] = lnew;
  }
  return o;
}

// transpose of an accumulator-layout tile through one LDS tile
__device__ inline d4 transpose_tile(const d4& v, ldsd* t) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) t[(lq + 4 * e) * 17 + lr] = v[e];
  wsync();
  d4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = t[lr * 17 + lq + 4 * e];
  wsync();
  return o;
}

template <int NB, int DC>
__device__ inline void fit_wave(ldsd* L, const gapro_fit_desc& desc, const gapro_fit_options& opt,
                                const float* __restrict__ feats_spp, const int* __restrict__ idx,
                                const double* __restrict__ init_mean, double* __restrict__ ws,
                                float* __restrict__ o_probs, float* __restrict__ o_probs_new,
                                unsigned char* __restrict__ o_labels, float* __restrict__ o_mu,
                                float* __restrict__ o_var, int* __restrict__ o_status, double* __restrict__ o_loss) {
  typedef WaveLds<NB, DC> W;
  constexpr int Mp = W::Mp, NL = W::NL;
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const int M = desc.m1 + desc.m2, T = desc.t;
  const double Nd = uni_d((double)M);
  ldsd* Zt = L + W::oZ;
  ldsd* Xt = L + W::oX;
  ldsd* mZ = L + W::oMZ;
  ldsd* vZ = L + W::oVZ;
  ldsd* LS = L + W::oLS;
  ldsd* LIm = L + W::oLI;
  ldsd* t0 = L + W::oT0;
  ldsd* t1 = L + W::oT1;
  ldsd* vm = L + W::oM;
  ldsd* vy = L + W::oY;
  ldsd* vgmu = L + W::oGmu;
  ldsd* vgv = L + W::oGv;
  ldsd* vtmp = L + W::oTmp;
  ldsd* gh = L + W::oGh;
  ldsd* sc = L + W::oSc;
  ldsd* vrs = L + W::oRs;
  ldsd* cen = L + W::oCen;
  // Wide features (D = 32): G_Z as the product  rowsum_k (Z_k - c) - W (P - c)  on the matrix cores, relative to a centre c
  // (training point 0) -- the per-dimension register accumulators of the difference form are 64 NB registers at D = 32
  // (887 spilled registers at NB = 2, and the fit slower than on the workgroup kernel).  The plain product form,
  // rowsum_k Z_k - W P, cancels digits in proportion to |Z| / |Z_k - P_n| (the trap of the narrow path's comment: 5e-6 in
  // mu at M = 3); with the centre taken out what is left is the points' spread over the kernel's reach, one digit or
  // two of sixteen.
  constexpr bool kWide = DC > 8;
  constexpr int NDB = (DC + 15) / 16;  // 16-column blocks of the feature dimension
  const Layout lay = make_layout(M, T, DC);
  double* wbase = ws + desc.ws_offset;
  int status = GAPRO_OK;
#ifdef GAPRO_PROFILE
  const unsigned long long t_start = wall_clock64();
  typedef __attribute__((address_space(3))) unsigned long long ldsu64;
  ldsu64* prof = (ldsu64*)(L + W::oProf);
  if (lane < 17) prof[lane] = lane == 16 ? t_start : 0ull;
  auto stamp = [&](int id) {  // phase clocks (tools/bench_fit.py --profile): time since the previous stamp goes to slot id
    if (lane == 0) {
      const unsigned long long t = wall_clock64();
      prof[id] += t - prof[16];
      prof[16] = t;
    }
  };
#else
  auto stamp = [&](int) {};
#endif

  // ---- setup: parameters as gpytorch initialises them (gaussian_process_utils.py:386-403, :14) ----------------
  wsync();  // the previous fit of this wave is done with the LDS
  for (int e = lane; e < 4 * DC * Mp; e += 64) L[W::oZ + e] = 0.0;  // Z^T, X^T, Adam moments of Z
  for (int e = lane; e < NL * kTS; e += 64) LS[e] = 0.0;
  wsync();
  const int* my_idx = idx + desc.idx_offset;
  for (int e = lane; e < M * DC; e += 64) {
    const int i = e / DC, d = e - i * DC;
    const double v = (double)feats_spp[(size_t)my_idx[i] * DC + d];  // train_x = cat(b1_feats, b2_feats)  :395
    Zt[d * Mp + i] = v;  // inducing points initialised to train_x  (:14)
    Xt[d * Mp + i] = v;
  }
  if (lane < Mp) {
    const int i = lane;
    vy[i] = i < desc.m1 ? -1.0 : (i < M ? 1.0 : 0.0);  // train_y  :396-398
    vm[i] = (i < M && init_mean) ? init_mean[desc.idx_offset + i] : 0.0;
    vgmu[i] = 0.0;
    vgv[i] = 0.0;
    if (i < M) LS[lt(i >> 4, i >> 4) * kTS + (i & 15) * 18] = 1.0;  // chol_variational_covar = I
  }
  if (lane < 10) {
    gh[lane] = kGhT[lane];
    gh[10 + lane] = kGhW[lane];
  }
  if (lane < 3) sc[lane] = 0.0;
  if (kWide) {
    wsync();
    for (int d = lane; d < DC; d += 64) cen[d] = Xt[d * Mp];
    if (lane < Mp) vrs[lane] = 0.0;
  }
  double mm1 = 0.0, mm2 = 0.0;  // Adam moments of m[lane]
  // Adam moments of tril(L_S): the one piece of state that is touched once per step and nowhere else, so it lives in
  // the fit's workspace slab (B_MLS, B_VLS) in tile order [tile][r][lane] -- 512-byte rows, read at the start of the
  // Adam phase and written back from it -- instead of holding 16 NL registers through the likelihood calls
  double* gMLS = wbase + lay.mat + (long long)B_MLS * lay.Mp * lay.Mp;
  double* gVLS = wbase + lay.mat + (long long)B_VLS * lay.Mp * lay.Mp;
#pragma unroll
  for (int e = 0; e < 4 * NL; ++e) {
    gMLS[e * 64 + lane] = 0.0;
    gVLS[e * 64 + lane] = 0.0;
  }
  wsync();

  // ---- tiles ---------------------------------------------------------------------------------------------------
  // tile (kb, ib) of K_ZZ + jit I; the padded tail is an identity block
  auto kzz_tile = [&](int kb, int ib, double s, double inv_l2, double jit) {
    const RbfTile k = rbf_tile<DC, Mp>(Zt, kb, Zt, ib, -0.5 * inv_l2);
    const int col = 16 * ib + lr;
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * kb + lq + 4 * r;
      const double kv = s * k.e[r];
      v[r] = (row < M && col < M) ? (row == col ? kv + jit : kv) : (row == col ? 1.0 : 0.0);
    }
    return v;
  };
  // row-major LDS tile t as an operand: the tile itself (C layout) or its transpose (the other index pattern)
  auto ld_tile = [&](const ldsd* t) {
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = t[(lq + 4 * r) * 17 + lr];
    return v;
  };
  auto ld_tile_t = [&](const ldsd* t) {
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = t[lr * 17 + lq + 4 * r];
    return v;
  };
  auto st_tile = [&](ldsd* t, const d4& v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) t[(lq + 4 * r) * 17 + lr] = v[r];
  };
  auto ls_tile = [&](int i, int j) { return ld_tile(LS + lt(i, j) * kTS); };     // L_S tile (i, j), i >= j
  auto lst_tile = [&](int j, int i) { return ld_tile_t(LS + lt(i, j) * kTS); };  // (L_S^T) tile (j, i), j <= i
  auto li_tile = [&](int i, int k) { return ld_tile(LIm + lt(i, k) * kTS); };    // LI tile (i, k), i >= k
  auto lit_tile = [&](int k, int i) { return ld_tile_t(LIm + lt(i, k) * kTS); }; // (LI^T) tile (k, i), k <= i
  auto red_lq = [&](double v) { return sum_rows(v); };  // sum over the four row groups of a column

  // ---- factorisation: U = L^T (off-diagonal tiles), Dinv_k, Dinv_k^T -> LI = L^-1 as row-major LDS tiles (LI and
  // LI^T are then the same bytes read with either index pattern, like L_S)
  auto factorize = [&](double s, double inv_l2) {
    d4 U[NL];  // U tile (k, i), k < i, at lt(i, k)
    d4 Dinv[NB], DinvT[NB];
    double extra = 0.0;
    for (int attempt = 0;; ++attempt) {
      bool bad = false;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        d4 S[NB];
#pragma unroll
        for (int ib = kb; ib < NB; ++ib) {
          d4 acc = kzz_tile(kb, ib, s, inv_l2, opt.jitter + extra);
#pragma unroll
          for (int q = 0; q < kb; ++q) acc = tn_neg(acc, U[lt(kb, q)], U[lt(ib, q)]);
          S[ib] = acc;
        }
        int bad_k = 0;
        const DiagInv di = diag_factor(S[kb], t0, t1, &bad_k);
        Dinv[kb] = di.inv;
        DinvT[kb] = di.invT;
        bad |= bad_k != 0;
#pragma unroll
        for (int ib = kb + 1; ib < NB; ++ib) U[lt(ib, kb)] = tn(zero4(), DinvT[kb], S[ib]);
      }
      if (!bad) break;
      if (attempt >= opt.psd_retries) {
        if (status == GAPRO_OK) status = GAPRO_ERR_CHOLESKY;
        break;
      }
      extra = opt.psd_jitter;
      for (int e = 0; e < attempt; ++e) extra *= 10.0;  // psd_jitter 10^attempt (powers of ten are exact here)
    }
    // LI_kk = Dinv_k;  LI_ik = -Dinv_i sum_{j = k}^{i - 1} L_ij LI_jk   (i > k)
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      d4 col[NB];  // block column k of LI
      col[k] = Dinv[k];
      st_tile(LIm + lt(k, k) * kTS, Dinv[k]);
#pragma unroll
      for (int i = k + 1; i < NB; ++i) {
        d4 acc = zero4();
#pragma unroll
        for (int j = k; j < i; ++j) acc = tn(acc, U[lt(i, j)], col[j]);
        col[i] = tn_neg(zero4(), DinvT[i], acc);
        st_tile(LIm + lt(i, k) * kTS, col[i]);
      }
    }
    wsync();
  };

  // ---- forward pass of 16 columns given as KX tiles: A = LI KX (NB tiles) and, if wanted, its transpose
  auto forward_a = [&](const d4 (&KXc)[NB], d4 (&Ac)[NB], d4 (*Atc)[NB]) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      d4 a = zero4(), at = zero4();
#pragma unroll
      for (int k = 0; k <= i; ++k) {
        const d4 l = lit_tile(k, i);
        a = tn(a, l, KXc[k]);
        if (Atc) at = tn(at, KXc[k], l);
      }
      Ac[i] = a;
      if (Atc) (*Atc)[i] = at;
    }
  };
  // column sums mu, var of 16 columns from their A tiles (B = L_S^T A is transient)
  auto colsums = [&](const d4 (&Ac)[NB], double s, double* mu, double* var) {
    double pm = 0.0, pv = 0.0;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      d4 b = zero4();
#pragma unroll
      for (int k = i; k < NB; ++k) b = tn(b, ls_tile(k, i), Ac[k]);  // B = L_S^T A
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double a = Ac[i][r];
        pm += vm[16 * i + lq + 4 * r] * a;
        pv += b[r] * b[r] - a * a;
      }
    }
    *mu = red_lq(pm);
    *var = s + opt.jitter + red_lq(pv);
  };

  double last_loss = 0.0;
  const double b1 = 0.9, b2 = 0.999, aeps = 1e-8;
  // torch.optim.Adam; ibc2s = 1 / sqrt(1 - beta2^step), one division per step instead of one per element (the
  // denominator differs from sqrt(v) / sqrt(bc2) + eps in its last bit at most)
  auto adam = [&](double p, double& m1, double& m2, double g, double step_size, double ibc2s) {
    m1 = b1 * m1 + (1.0 - b1) * g;
    m2 = b2 * m2 + (1.0 - b2) * g * g;
    return p - step_size * m1 / (sqrt(m2) * ibc2s + aeps);
  };

  double b1p = 1.0, b2p = 1.0;  // beta^step as running products (a few ulp from pow(): far inside the oracle tolerance)
  for (int step = 1; step <= opt.training_iter; ++step) {
    b1p = uni_d(b1p * b1);
    b2p = uni_d(b2p * b2);
    // (wave-uniform scalars go through uni_d: in vector registers the allocator spilled them to scratch memory and
    // reloaded them element by element inside the Adam updates)
    const double c = uni_d(sc[0]), rho_s = uni_d(sc[1]), rho_l = uni_d(sc[2]);
    const double sp = nl_softplus(lane == 63 ? rho_l : rho_s);  // one call: lane 63 evaluates the length scale's
    const double s = lane_bcast(sp, 0), ell = lane_bcast(sp, 63), inv_l2 = uni_d(1.0 / (ell * ell));
    const bool last = step == opt.training_iter;
    const double bc1 = 1.0 - b1p, bc2s = uni_d(1.0 / sqrt(1.0 - b2p));  // (bc2s: the RECIPROCAL root, see adam)
    const double step_size = uni_d(opt.lr / bc1);
    stamp(0);
    factorize(s, inv_l2);
    stamp(1);
    if (last && opt.eval_stale_chol) {  // prediction with the factor of the last training step
      double* U = wbase + lay.mat + (long long)B_U * lay.Mp * lay.Mp;
      for (int e = lane; e < NL * kTS; e += 64) U[e] = LIm[e];
    }

    // ---- the training points in blocks of 16 columns: everything between KX and the gradient sums is column-wise
    // (the strip idea of svgp_fit.hip), so only the M x M sums G_LS, Pm^T and R live across the blocks
    d4 GLS[NL], PmT[NL];
    double gm_part[NB];
    // G_Z[k][d] = -sum_n W[k][n] (Z_k[d] - P_n[d]) / l^2 in the DIFFERENCE form, like svgp_fit.hip: as a product
    // (rowsum_k Z_k - W P) it loses digits to cancellation, and on tiny fits (M = 3) Adam turns that into 5e-6 in mu.
    // This lane accumulates column k = 16 kb + lr over the rows n of transposed tiles (n-major: W^T, or the symmetric
    // W_zz as it is); the four row groups are summed at the end.
    double gz[NB][kWide ? 1 : DC];
    // wide features: T = W_zx (X - c) + 2 W_zz (Z - c) as tiles (point block k, feature block db), row sums of the weights
    d4 GZT[NB][kWide ? NDB : 1];
    double rs_part[NB];
    // tile (n, db) of the points P ([d][i] in LDS) relative to the centre: rows = points 16 n + lq + 4 r, column = feature
    auto pc_tile = [&](const ldsd* Pt_, int n, int db) {
      const int d = 16 * db + lr;
      const double cd = d < DC ? cen[d] : 0.0;
      d4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = d < DC ? Pt_[d * Mp + 16 * n + lq + 4 * r] - cd : 0.0;
      return v;
    };
#pragma unroll
    for (int t = 0; t < NL; ++t) {
      GLS[t] = zero4();
      PmT[t] = zero4();
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      gm_part[k] = 0.0;
      rs_part[k] = 0.0;
#pragma unroll
      for (int d = 0; d < (kWide ? 1 : DC); ++d) gz[k][d] = 0.0;
#pragma unroll
      for (int db = 0; db < (kWide ? NDB : 1); ++db) GZT[k][db] = zero4();
    }
    double e_part = 0.0, gc_part = 0.0, gvs_part = 0.0, wsum = 0.0, gl = 0.0;
#pragma unroll 1
    for (int n = 0; n < NB; ++n) {
      const int col = 16 * n + lr;
      const bool on = col < M;
      // KX, A = LI KX, At = A^T
      d4 KX[NB], A[NB], At[NB];
      d4 KXD[kWide ? NB : 1];  // wide features: the squared distances of the KX tiles (the narrow path recomputes them)
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const RbfTile kx = rbf_tile<DC, Mp>(Zt, k, Xt, n, -0.5 * inv_l2);
#pragma unroll
        for (int r = 0; r < 4; ++r) KX[k][r] = (16 * k + lq + 4 * r < M && on) ? s * kx.e[r] : 0.0;
        if (kWide) KXD[k] = kx.d2;
      }
      stamp(2);
      forward_a(KX, A, nullptr);
      // mu, var; likelihood gradients (ten Gauss-Hermite pairs per column over the four row groups of its lanes)
      double mu_raw, vraw;
      colsums(A, s, &mu_raw, &vraw);
      const double mu = mu_raw + c;
      const bool clamped = vraw < opt.min_variance;
      const double var = clamped ? opt.min_variance : vraw;
      const double sd = sqrt(2.0 * var);
      const double y = vy[col];
      stamp(3);
      const LikSums ls = lik_column(gh, mu, sd, y, on ? 1 : 0, last ? 1 : 0);
      const double E = red_lq(ls.E), dmu = red_lq(ls.dmu), dvar = red_lq(ls.dvar);
      const double ipi = 0.56418958354775628695;  // 1 / sqrt(pi)
      const double g1 = on ? -(ipi * dmu * y) / Nd : 0.0;                      // g_mu of this lane's column
      const double g2 = (on && !clamped) ? -(ipi * dvar * y / sd) / Nd : 0.0;  // g_v
      if (lq == 0) {
        vgmu[col] = g1;
        vgv[col] = g2;
        if (on) {
          e_part += ipi * E;
          gc_part += g1;
          gvs_part += g2;
        }
      }
      wsync();
      stamp(4);
      forward_a(KX, A, &At);     // (A again, and A^T: fewer tiles live across the likelihood call)
      double gmu_r[4], gv_r[4];  // the same two vectors indexed by the ROWS of the transposed tiles
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gmu_r[r] = vgmu[16 * n + lq + 4 * r];
        gv_r[r] = vgv[16 * n + lq + 4 * r];
      }
      // G_LS += At^T GBt with GBt = 2 diag(g_v) B^T, one tile of GBt at a time;  G_m += At^T g_mu
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        d4 bt = zero4();
#pragma unroll
        for (int k = j; k < NB; ++k) bt = tn(bt, A[k], ls_tile(k, j));
#pragma unroll
        for (int r = 0; r < 4; ++r) bt[r] *= 2.0 * gv_r[r];
#pragma unroll
        for (int i = j; i < NB; ++i) GLS[lt(i, j)] = tn(GLS[lt(i, j)], At[i], bt);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) gm_part[i] += At[i][r] * gmu_r[r];
      stamp(5);
      // GB = 2 B diag(g_v);  G_A = m g_mu^T + L_S GB - 2 A diag(g_v)
      d4 GB[NB], GA[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        d4 b = zero4();
#pragma unroll
        for (int k = i; k < NB; ++k) b = tn(b, ls_tile(k, i), A[k]);
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r] *= 2.0 * g2;
        GB[i] = b;
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        d4 g = zero4(), gt = zero4();
#pragma unroll
        for (int k = 0; k <= i; ++k) {
          const d4 l = lst_tile(k, i);
          g = tn(g, l, GB[k]);
          gt = tn(gt, GB[k], l);
        }
        const double m_c = vm[16 * i + lr];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          g[r] += vm[16 * i + lq + 4 * r] * g1 - 2.0 * A[i][r] * g2;
          gt[r] += gmu_r[r] * m_c - 2.0 * gv_r[r] * At[i][r];  // G_A^T tile (n, i)
        }
        GA[i] = g;
        // Pm^T (j, i) -= At (n, j)^T G_A^T (n, i):  Pm = Phi(-G_A A^T)
#pragma unroll
        for (int j = 0; j <= i; ++j) PmT[lt(i, j)] = tn_neg(PmT[lt(i, j)], At[j], gt);
      }
      stamp(6);
      // kernel gradients through KX: W_zx = G_KX o KX with G_KX = LI^T G_A, transposed through LDS to n-major
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        d4 g = zero4();
#pragma unroll
        for (int i = k; i < NB; ++i) g = tn(g, li_tile(i, k), GA[i]);
#pragma unroll
        for (int r = 0; r < 4; ++r) g[r] *= KX[k][r];
        const d4 wt = transpose_tile(g, t0);  // rows n = 16 n + lq + 4 r, column k = 16 k + lr
        if constexpr (kWide) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            wsum += g[r];
            gl += g[r] * KXD[k][r];
            rs_part[k] += wt[r];  // this lane's rows of column k: summed over the row groups before the update
          }
#pragma unroll
          for (int db = 0; db < NDB; ++db) GZT[k][db] = tn(GZT[k][db], wt, pc_tile(Xt, n, db));
        } else {
          double zk[DC];
#pragma unroll
          for (int d = 0; d < DC; ++d) zk[d] = Zt[d * Mp + 16 * k + lr];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double s2 = 0.0;
#pragma unroll
            for (int d = 0; d < DC; ++d) {
              const double t = zk[d] - Xt[d * Mp + 16 * n + lq + 4 * r];
              s2 += t * t;
              gz[k][d] += wt[r] * t;
            }
            wsum += wt[r];
            gl += wt[r] * s2;
          }
        }
      }
      stamp(7);
    }
    // the Adam moments of tril(L_S) are requested now and arrive behind the sums below
    d4 MLS[NL], VLS[NL];
#pragma unroll
    for (int t = 0; t < NL; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        MLS[t][r] = gMLS[(4 * t + r) * 64 + lane];
        VLS[t][r] = gVLS[(4 * t + r) * 64 + lane];
      }
    const double g_c = uni_d(wave_sum(gc_part));
    const double gv_sum = uni_d(wave_sum(gvs_part));
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const double p = red_lq(gm_part[i]);
      if (lq == 0) vtmp[16 * i + lr] = p;  // G_m without the KL term
    }

    // ---- ELBO value of the last step (parameters before their update)
    if (last) {
      const double e_sum = wave_sum(e_part);
      double kl_part = 0.0;
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
          const d4 l = ls_tile(i, j);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * i + lq + 4 * r, cj = 16 * j + lr;
            if (cj <= row && row < M) {
              kl_part += l[r] * l[r];
              if (row == cj) kl_part -= nl_log(l[r] * l[r]);
            }
          }
        }
      if (lane < M) kl_part += vm[lane] * vm[lane];
      const double kl = 0.5 * (wave_sum(kl_part) - Nd);
      last_loss = -(e_sum / Nd - kl / Nd);
    }
    wsync();  // every read of L_S of this step is done
    stamp(8);

    // ---- Adam on tril(L_S)
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        const int t = lt(i, j);
        const AdamTile a = i == j ? adam_ls_tile<true>(LS + t * kTS, GLS[t], MLS[t], VLS[t], 16 * i, 16 * j, M, Nd, step_size, bc2s)
                                  : adam_ls_tile<false>(LS + t * kTS, GLS[t], MLS[t], VLS[t], 16 * i, 16 * j, M, Nd, step_size, bc2s);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gMLS[(4 * t + r) * 64 + lane] = a.m1[r];
          gVLS[(4 * t + r) * 64 + lane] = a.m2[r];
        }
      }

    stamp(9);
    // ---- G_Kzz = LI^T Pm LI, symmetrised:  Pm = Phi(.) (strictly lower + half the diagonal),  W = Pm LI (lower),
    // S = LI^T W and S^T = W^T LI;  W_zz = sym(G_Kzz) o K_ZZ (symmetric: every tile is n-major as it is)
    double gs = wsum / s;
    {
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int a = lq + 4 * r;  // element (a, b = lr) of the diagonal tile of Pm^T is Pm[b][a]: lower means a < b
          PmT[lt(i, i)][r] = a < lr ? PmT[lt(i, i)][r] : (a == lr ? 0.5 * PmT[lt(i, i)][r] : 0.0);
        }
      d4 Wm[NL];
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
          d4 w = zero4();
#pragma unroll
          for (int k = j; k <= i; ++k) w = tn(w, PmT[lt(i, k)], li_tile(k, j));
          Wm[lt(i, j)] = w;
        }
#pragma unroll
      for (int n = 0; n < NB; ++n) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          d4 sv = zero4(), st = zero4();  // tile (n, k) of S and of S^T
#pragma unroll
          for (int q = (n > k ? n : k); q < NB; ++q) {
            sv = tn(sv, li_tile(q, n), Wm[lt(q, k)]);
            st = tn(st, Wm[lt(q, n)], li_tile(q, k));
          }
          const RbfTile kz = rbf_tile<DC, Mp>(Zt, n, Zt, k, -0.5 * inv_l2);
          const int cj = 16 * k + lr;
          if constexpr (kWide) {
            d4 w2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * n + lq + 4 * r;
              const bool in = row < M && cj < M;
              const double gsym = in ? 0.5 * (sv[r] + st[r]) : 0.0;
              const double w = gsym * s * kz.e[r];
              gs += gsym * kz.e[r];
              gl += w * kz.d2[r];
              w2[r] = 2.0 * w;
              rs_part[k] += w2[r];
            }
#pragma unroll
            for (int db = 0; db < NDB; ++db) GZT[k][db] = tn(GZT[k][db], w2, pc_tile(Zt, n, db));
          } else {
          double zk[DC];
#pragma unroll
          for (int d = 0; d < DC; ++d) zk[d] = Zt[d * Mp + cj];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * n + lq + 4 * r;
            const bool in = row < M && cj < M;
            const double gsym = in ? 0.5 * (sv[r] + st[r]) : 0.0;
            const double w = gsym * s * kz.e[r];
            gs += gsym * kz.e[r];
            gl += w * kz.d2[r];
#pragma unroll
            for (int d = 0; d < DC; ++d) gz[k][d] += 2.0 * w * (zk[d] - Zt[d * Mp + row]);
          }
          }
        }
      }
    }
    gs = uni_d(wave_sum(gs) + gv_sum);
    gl = uni_d(wave_sum(gl) / (ell * ell * ell));
    stamp(10);

    // ---- Adam on Z: the four row groups of column k are summed; row group lq then updates d = lq, lq + 4, ...
    wsync();  // every read of Z of this step is done
    if constexpr (kWide) {
      // G_Z[k][d] = -(rowsum_k (Z_k[d] - c[d]) - T[k][d]) / l^2: the row sums travel through LDS (summed per column lane,
      // needed per row of the tiles); every lane then owns the elements (point 16 k + lq + 4 r, feature 16 db + lr)
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const double v = red_lq(rs_part[k]);
        if (lq == 0) vrs[16 * k + lr] = v;
      }
      wsync();
#pragma unroll
      for (int k = 0; k < NB; ++k)
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int pt = 16 * k + lq + 4 * r, d = 16 * db + lr;
            const bool act = d < DC && pt < M;
            const int o = act ? d * Mp + pt : 0;
            const double zc = Zt[o] - cen[act ? d : 0];
            const double grad = -inv_l2 * (vrs[act ? pt : 0] * zc - GZT[k][db][r]);
            double m1 = mZ[o], m2 = vZ[o];
            const double zn = adam(Zt[o], m1, m2, grad, step_size, bc2s);
            if (act) {
              Zt[o] = zn;
              mZ[o] = m1;
              vZ[o] = m2;
            }
          }
    } else {
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      double mine[(DC + 3) / 4];
#pragma unroll
      for (int d = 0; d < DC; ++d) {
        const double v = red_lq(gz[k][d]);
        if ((d & 3) == lq) mine[d >> 2] = v;
      }
      // (no branch around the arithmetic: the updates of a lane are independent chains of divisions and square roots,
      // and only in one basic block does the compiler interleave them)
      const int pt = 16 * k + lr;
#pragma unroll
      for (int e = 0; e < (DC + 3) / 4; ++e) {
        const int d = lq + 4 * e;
        const bool act = d < DC && pt < M;
        const int o = act ? d * Mp + pt : 0;
        const double grad = -inv_l2 * mine[e];
        double m1 = mZ[o], m2 = vZ[o];
        const double zn = adam(Zt[o], m1, m2, grad, step_size, bc2s);
        if (act) {
          Zt[o] = zn;
          mZ[o] = m1;
          vZ[o] = m2;
        }
      }
    }
    }

    // ---- Adam on m, c, rho_s, rho_l: ONE update per lane -- lanes < M hold m (M <= 48), lanes 61, 62, 63 the three
    // scalars; their moments are the same two registers (mm1, mm2) in every lane
    wsync();
    {
      const double sg = nl_sigmoid(lane == 63 ? rho_l : rho_s);  // softplus' for both raw hyperparameters in one call
      const bool is_m = lane < M;
      const double pv = is_m ? vm[lane] : (lane == 61 ? c : lane == 62 ? rho_s : rho_l);
      const double g = is_m ? vtmp[is_m ? lane : 0] + pv / Nd : (lane == 61 ? g_c : (lane == 62 ? gs : gl) * sg);
      const double pn = adam(pv, mm1, mm2, g, step_size, bc2s);
      wsync();
      if (is_m) vm[lane] = pn;
      if (lane >= 61) sc[lane - 61] = pn;
    }
    wsync();
    stamp(11);
  }

  // ------------------------------- prediction (gaussian_process_utils.py:426-438) ------------------------------
  {
    const double c = uni_d(sc[0]);
    const double sp = nl_softplus(lane == 63 ? sc[2] : sc[1]);
    const double s = lane_bcast(sp, 0), ell = lane_bcast(sp, 63), inv_l2 = uni_d(1.0 / (ell * ell));
    if (opt.eval_stale_chol && opt.training_iter > 0) {
      const double* U = wbase + lay.mat + (long long)B_U * lay.Mp * lay.Mp;
      for (int e = lane; e < NL * kTS; e += 64) LIm[e] = U[e];
      wsync();
    } else {
      factorize(s, inv_l2);
    }
#pragma unroll 1
    for (int t0c = 0; t0c < T; t0c += 16) {
      const int nc = (T - t0c) < 16 ? (T - t0c) : 16;
      double xt[DC];
      {
        const int row = my_idx[M + t0c + (lr < nc ? lr : 0)];  // intersect_feats  :386
#pragma unroll
        for (int d = 0; d < DC; ++d) xt[d] = (double)feats_spp[(size_t)row * DC + d];
      }
      d4 Ac[NB];
      {
        d4 KXc[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * k + lq + 4 * r;
            double s2 = 0.0;
#pragma unroll
            for (int d = 0; d < DC; ++d) {
              const double t = Zt[d * Mp + row] - xt[d];
              s2 += t * t;
            }
            KXc[k][r] = (row < M && lr < nc) ? s * nl_exp(-0.5 * inv_l2 * s2) : 0.0;
          }
        forward_a(KXc, Ac, nullptr);
      }
      double mu_raw, vraw;
      colsums(Ac, s, &mu_raw, &vraw);
      if (lq == 0 && lr < nc) {
        const double mu = mu_raw + c;
        const double var = fmax(vraw, opt.min_variance);
        const double p = 0.5 * erfc(-(mu / sqrt(1.0 + var)) * 0.70710678118654752440);
        const float pf = (float)p;                       // pred_probs            :432
        const bool lab = pf >= 0.5f;                     // pred_labels           :433
        const long long o = desc.out_offset + t0c + lr;
        o_probs[o] = pf;
        o_probs_new[o] = lab ? pf : 1.0f - pf;           // pred_probs_new        :438
        o_labels[o] = lab ? 1 : 0;
        o_mu[o] = (float)mu;                             // pred_mu               :435
        o_var[o] = (float)var;                           // pred_variance         :436
        if ((!isfinite(mu) || !isfinite(var)) && status == GAPRO_OK) status = GAPRO_ERR_NOT_FINITE;
      }
    }
  }
  stamp(12);
  // a status raised by any lane (the prediction's lanes differ) reaches lane 0: errors are negative, so the minimum
  // over the wave is the most severe code
  for (int o = 32; o > 0; o >>= 1) {
    const int other = __shfl_xor(status, o, 64);  // (once per fit: the LDS crossbar will do)
    status = other < status ? other : status;
  }
  // the diagonal of L^-1 (= 1 / L_jj of the last factorisation) where the other kernels keep their Dinv blocks: the
  // launch's conditioning figures are read from there (k_fit_cond, svgp_fit.hip)
  if (lane < M) {
    const int kb = lane >> 4, j = lane & 15;
    (wbase + lay.dinv)[(size_t)kb * 256 + 17 * j] = LIm[lt(kb, kb) * kTS + j * 17 + j];
  }
  if (lane == 0) {
    if (status == GAPRO_OK && !isfinite(last_loss) && opt.training_iter > 0) status = GAPRO_ERR_NOT_FINITE;
    o_status[desc.slot] = status;
    o_loss[desc.slot] = last_loss;
    double* scal = wbase + lay.scal;
    scal[S_C] = sc[0];
    scal[S_RS] = sc[1];
    scal[S_RL] = sc[2];
    scal[S_LOSS] = last_loss;
    scal[S_STATUS] = (double)status;
#ifdef GAPRO_PROFILE
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    for (int i = 0; i < 25; ++i) scal[24 + i] = i < 16 ? (double)prof[i] : 0.0;
    scal[24 + 25] = (double)t_start;  // timeline of the launch: tools/fit_timeline.py
    scal[24 + 26] = (double)wall_clock64();
    scal[24 + 27] = (double)(((xcc & 15u) << 16) | (hwid & 0xFFFFu));
#endif
  }
}

// WPE = waves per SIMD the register budget is sized for (2: 256 registers per lane, 1: the whole file)
template <int NB, int DC, int WPE>
__global__ __launch_bounds__(64, WPE) void k_svgp_fit_wave(int n_fits, const float* __restrict__ feats_spp,
                                                         const int* __restrict__ idx,
                                                         const gapro_fit_desc* __restrict__ descs,
                                                         const double* __restrict__ init_mean, gapro_fit_options opt,
                                                         double* __restrict__ ws, float* __restrict__ o_probs,
                                                         float* __restrict__ o_probs_new,
                                                         unsigned char* __restrict__ o_labels, float* __restrict__ o_mu,
                                                         float* __restrict__ o_var, int* __restrict__ o_status,
                                                         double* __restrict__ o_loss, unsigned* ticket) {
  extern __shared__ double dyn_lds[];
  ldsd* L = (ldsd*)dyn_lds;
  // A wave takes fits until none is left (by ticket: longest first, in the order in which the waves become free;
  // without a ticket counter: grid-stride), so a launch costs one dispatch per wave slot, not one per fit.
  for (int it = 0;; ++it) {
    int fit;
    if (ticket) {
      int t = 0;
      if ((threadIdx.x & 63) == 0) t = (int)atomicAdd(ticket, 1u);
      fit = __builtin_amdgcn_readfirstlane(t);
    } else {
      fit = blockIdx.x + it * gridDim.x;
    }
    if (fit >= n_fits) break;
    const gapro_fit_desc desc = descs[fit];
    fit_wave<NB, DC>(L, desc, opt, feats_spp, idx, init_mean, ws, o_probs, o_probs_new, o_labels, o_mu, o_var, o_status,
                     o_loss);
  }
}

}  // namespace

// Largest padded size the wave kernel takes at this feature width (0: none).  NB = 3 runs one wave per SIMD.
// Deep features (D = 32, the --use_deepfeat workflow, gen_ps.py:48-53; round 6): M_p <= 32 -- the points of three row
// blocks and their Adam moments (4 x 32 x 48 doubles) leave room for one fit per CU.
int gapro_fit_wave_max_mp(int feat_dim) { return feat_dim == 6 ? 48 : feat_dim == 32 ? 32 : 0; }

size_t gapro_fit_wave_lds_bytes(int nb, int feat_dim) {
  if (feat_dim == 32) return nb == 1 ? 8 * (size_t)WaveLds<1, 32>::total : nb == 2 ? 8 * (size_t)WaveLds<2, 32>::total : 0;
  if (feat_dim != 6) return 0;
  return 8 * (size_t)(nb == 1 ? WaveLds<1, 6>::total : nb == 2 ? WaveLds<2, 6>::total : WaveLds<3, 6>::total);
}

// fits per CU of one instantiation (register budget and LDS)
int gapro_fit_wave_per_cu(int nb, int feat_dim) {
  const size_t lds = gapro_fit_wave_lds_bytes(nb, feat_dim);
  if (!lds) return 0;
  // D = 32: the per-dimension register arrays (G_Z accumulators, a column's coordinates) take the whole register file
  const int by_lds = (int)((160 * 1024) / lds), by_reg = (nb <= 1 && feat_dim == 6) ? 8 : 4;
  return by_lds < by_reg ? by_lds : by_reg;
}

// n_fits descriptors of one block count NB (descs sorted longest first by the caller); n_wg waves are launched
int gapro_launch_fit_wave(hipStream_t stream, int nb, int n_fits, int n_wg, unsigned* d_ticket, int feat_dim,
                          const float* d_feats_spp, const int* d_idx, const gapro_fit_desc* d_descs,
                          const double* d_init_mean, const gapro_fit_options& opt, double* d_workspace, float* d_probs,
                          float* d_probs_new, unsigned char* d_labels, float* d_mu, float* d_var, int* d_fit_status,
                          double* d_fit_loss) {
  if ((feat_dim != 6 && feat_dim != 32) || nb < 1 || nb > (feat_dim == 6 ? 3 : 2)) return GAPRO_ERR_BAD_ARG;
  const size_t lds = gapro_fit_wave_lds_bytes(nb, feat_dim);
#define GAPRO_WAVE_LAUNCH(NBV, DV, WPEV)                                                                          \
  do {                                                                                                            \
    auto kern = k_svgp_fit_wave<NBV, DV, WPEV>;                                                                   \
    if (lds > 48 * 1024 &&                                                                                        \
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return GAPRO_ERR_HIP;                                                                                       \
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(64), lds, stream, n_fits, d_feats_spp, d_idx, d_descs, d_init_mean, opt, \
                       d_workspace, d_probs, d_probs_new, d_labels, d_mu, d_var, d_fit_status, d_fit_loss, d_ticket); \
  } while (0)
  if (feat_dim == 32) {
    if (nb == 1) GAPRO_WAVE_LAUNCH(1, 32, 1);
    else GAPRO_WAVE_LAUNCH(2, 32, 1);
  } else if (nb == 1) GAPRO_WAVE_LAUNCH(1, 6, 2);
  else if (nb == 2) GAPRO_WAVE_LAUNCH(2, 6, 1);
  else GAPRO_WAVE_LAUNCH(3, 6, 1);
#undef GAPRO_WAVE_LAUNCH
  return hipGetLastError() == hipSuccess ? GAPRO_OK : GAPRO_ERR_HIP;
}
