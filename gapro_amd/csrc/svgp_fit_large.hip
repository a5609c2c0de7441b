// Generic (any M) variant of the batched SVGP fit: every matrix lives in the global-memory workspace,
// nothing is staged in LDS.  It is the first correct kernel of round 1, kept as the fallback for fits
// whose working set does not fit the LDS-staged kernel of svgp_fit.hip (M > 512 or very wide features).
// Same arithmetic, same workspace layout, same results to rounding.
//
// Replaces reference gapro/gaussian_process_utils.py:382-445 (fit_gp_spp) and the gpytorch objects
// it builds (:11-25): CholeskyVariationalDistribution + whitened VariationalStrategy with learned
// inducing locations, ConstantMean, ScaleKernel(RBFKernel), BernoulliLikelihood (20-point
// Gauss-Hermite), VariationalELBO, Adam(lr=0.1) x training_iter, then prediction.  There is no
// autograd on the device: the backward pass is the hand-derived one of SURVEY.md Appendix B.5,
// restated and checked against torch autograd in oracle/svgp_oracle.py.
//
// Arithmetic: float64 throughout (the reference runs its Cholesky/solves in float64 and the rest in
// float32; float64 everywhere is a superset and is what makes "variances within 1e-4" testable
// against a float64 ground truth -- DESIGN.md "Precision").  Every M x M x M contraction is a
// TN-form MFMA product (v_mfma_f64_16x16x4_f64):  C[i][j] = sum_k P[k][i] * Q[k][j]  with both
// operands row-major in k, so that fragment loads are 128-byte row segments; matrices that are
// needed in both orientations are written in both by the producing epilogue.
//
//   forward : Kzz -> L (blocked left-looking Cholesky) -> LI = L^-1 (block-column parallel)
//             KX = k(Z, X);  A = LI KX;  B = LS^T A;  mu = A^T m + c;  var = s + eps + |B|^2 - |A|^2
//             E = Gauss-Hermite( log Phi(y f) );  loss = -(sum E - KL) / N
//   backward: G_m, G_c, G_LS = tril(A G_B^T) + KL',  G_A = m g_mu^T + LS G_B - 2 A diag(g_v)
//             G_KX = LI^T G_A;  G_L = -tril(G_KX A^T);  G_Kzz = LI^T Phi(L^T G_L) LI (symmetrised)
//             G_s, G_l, G_Z through the RBF kernel;  softplus' = sigmoid
//   Adam    : torch.optim.Adam defaults (beta 0.9/0.999, eps 1e-8), lr 0.1
#include <math.h>

#include "common.h"
#include "fit_layout.h"

namespace {
using namespace gapro_fit;

constexpr int NT = 512;       // threads per fit
constexpr int NW = NT / 64;   // waves per fit
constexpr int NGH = 20;       // Gauss-Hermite nodes (gpytorch settings.num_gauss_hermite_locs)

typedef double d4 __attribute__((ext_vector_type(4)));

// numpy.polynomial.hermite.hermgauss(20): positive nodes (ascending) and their weights; the rule is
// symmetric.  Printed with repr() from NumPy 2.2.
__constant__ double c_gh_t[10] = {0.24534070830090124, 0.7374737285453944, 1.234076215395323,  1.7385377121165861,
                                  2.2549740020892757,  2.7888060584281305, 3.3478545673832163, 3.944764040115625,
                                  4.603682449550744,   5.387480890011233};
__constant__ double c_gh_w[10] = {0.4622436696006101,     0.28667550536283415,    0.1090172060200233,
                                  0.024810520887463643,   0.0032437733422378567,  0.00022833863601635365,
                                  7.80255647853206e-06,   1.0860693707692782e-07, 4.3993409922731747e-10,
                                  2.2293936455341447e-13};


// ---- workspace layout (doubles) -------------------------------------------------------------------
// (enums, Layout and make_layout: fit_layout.h, shared by every fit kernel)


struct Fit {
  int M, T, D, Mp;
  double* mat[B_COUNT];
  double* vec[V_COUNT];
  double *X, *Z, *mZ, *vZ, *gZ, *Xt, *dinv, *dinvT, *scal;
};

// ---- small helpers ---------------------------------------------------------------------------------
__device__ inline double softplus(double x) { return log1p(exp(-fabs(x))) + fmax(x, 0.0); }
__device__ inline double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }

// log Phi(z) and r(z) = phi(z)/Phi(z), both tails stable (same branches as oracle/svgp_oracle.py).
__device__ inline void log_ndtr_ratio(double z, double* lp, double* r) {
  const double rs2 = 0.70710678118654752440;
  if (z < 0.0) {
    const double ex = erfcx(-z * rs2);
    *lp = log(0.5 * ex) - 0.5 * z * z;
    *r = 0.79788456080286535588 / ex;  // sqrt(2/pi) / erfcx
  } else {
    const double tail = 0.5 * erfc(z * rs2);
    *lp = log1p(-tail);
    *r = exp(-0.5 * z * z) * 0.39894228040143267794 / (1.0 - tail);
  }
}

__device__ inline double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Deterministic block sum (fixed tree), result broadcast to every thread.
__device__ inline double block_sum(double v, double* sh /* >= NW doubles */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < NW; ++w) t += sh[w];
  return t;
}

// ---- TN-form MFMA product ---------------------------------------------------------------------------
//   C[i][j] = sum_{k in [klo,khi)} P[k][i] * (Q[k][j] * qscale(k)),  ld = leading dimension of P and Q
// Each wave owns (16 TU) x (16 TU) output tiles, round-robin.  `lower_only` enumerates tiles with
// ti >= tj.  kr(i0, j0, &klo, &khi) restricts the contraction range (multiples of 4) to where the
// triangular operands are non-zero.  epi(i, j, value) stores the result (and any transposed copy).
// MFMA f64 16x16x4 lane maps (cdna_hip_programming.md section 3): A[i = l & 15][k = l >> 4],
// B[k = l >> 4][j = l & 15], C/D register r -> row (l >> 4) + 4 r, col l & 15.
template <int TU, typename KRange, typename QScale, typename Epi>
__device__ __noinline__ void gemm_tn(int mo_tiles, int no_tiles, bool lower_only, const double* __restrict__ P,
                               const double* __restrict__ Q, int ld, KRange kr, QScale qs, Epi epi) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lr = lane & 15, lq = lane >> 4;
  constexpr int TS = 16 * TU;
  const int ntiles = lower_only ? mo_tiles * (mo_tiles + 1) / 2 : mo_tiles * no_tiles;
  for (int t = wave; t < ntiles; t += NW) {
    int ti, tj;
    if (lower_only) {
      ti = 0;
      while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
      tj = t - ti * (ti + 1) / 2;
    } else {
      ti = t / no_tiles;
      tj = t - ti * no_tiles;
    }
    const int i0 = ti * TS, j0 = tj * TS;
    int klo, khi;
    kr(i0, j0, &klo, &khi);
    d4 acc[TU][TU];
#pragma unroll
    for (int u = 0; u < TU; ++u)
#pragma unroll
      for (int v = 0; v < TU; ++v) acc[u][v] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k = klo; k < khi; k += 4) {
      const double* prow = P + (size_t)(k + lq) * ld + i0 + lr;
      const double* qrow = Q + (size_t)(k + lq) * ld + j0 + lr;
      const double sc = qs(k + lq);
      double a[TU], b[TU];
#pragma unroll
      for (int u = 0; u < TU; ++u) a[u] = prow[16 * u];
#pragma unroll
      for (int v = 0; v < TU; ++v) b[v] = qrow[16 * v] * sc;
#pragma unroll
      for (int u = 0; u < TU; ++u)
#pragma unroll
        for (int v = 0; v < TU; ++v) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[v], acc[u][v], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < TU; ++u)
#pragma unroll
      for (int v = 0; v < TU; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) epi(i0 + 16 * u + lq + 4 * r, j0 + 16 * v + lr, acc[u][v][r]);
  }
}

struct NoScale {
  __device__ double operator()(int) const { return 1.0; }
};

// ---- Cholesky of the padded Kzz (in B_L, lower) -> L, LT; Dinv blocks -------------------------------
// Left-looking, 16-wide panels.  Panel kb: (1) S = Kzz[:,kb] - L[:, <kb] L[kb, <kb]^T  (MFMA, TN via
// LT), (2) wave 0 factors the 16x16 diagonal block and inverts it, (3) the panel below is
// S * Dinv^T.  The padded tail (index >= M) is an identity block.
__device__ __noinline__ void cholesky_blocked(const Fit& f, double* sh_d /* 16x17 */, double* sh_dinv /* 16x17 */,
                                 int* sh_status) {
  const int Mp = f.Mp, nb = Mp / 16;
  double* L = f.mat[B_L];
  double* LT = f.mat[B_LT];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lr = lane & 15, lq = lane >> 4;
  for (int kb = 0; kb < nb; ++kb) {
    // (1) update block column kb
    for (int ib = kb + wave; ib < nb; ib += NW) {
      d4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = L[(size_t)(16 * ib + lq + 4 * r) * Mp + 16 * kb + lr];
#pragma unroll 4
      for (int q = 0; q < 16 * kb; q += 4) {
        const double a = LT[(size_t)(q + lq) * Mp + 16 * ib + lr];
        const double b = LT[(size_t)(q + lq) * Mp + 16 * kb + lr];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a, b, acc, 0, 0, 0);
      }
      if (ib == kb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sh_d[(lq + 4 * r) * 17 + lr] = acc[r];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) L[(size_t)(16 * ib + lq + 4 * r) * Mp + 16 * kb + lr] = acc[r];
      }
    }
    __syncthreads();
    // (2) diagonal block: unblocked Cholesky + inverse, one wave
    if (wave == 0) {
      for (int j = 0; j < 16; ++j) {
        double d = sh_d[j * 17 + j];
        if (!(d > 0.0)) {  // not positive definite (or NaN): flag it, keep going with a tiny pivot
          if (lane == 0) *sh_status = 1;  // chol_bad: see factorize (psd_safe_cholesky retries)
          d = 1e-30;
        }
        d = sqrt(d);
        __builtin_amdgcn_wave_barrier();
        if (lane > j && lane < 16) sh_d[lane * 17 + j] /= d;
        if (lane == 0) sh_d[j * 17 + j] = d;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // trailing rank-1 update of the lower triangle: 256 (r, c) slots over 64 lanes
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = lane + 64 * e;
          const int r = idx >> 4, c = idx & 15;
          if (r > j && c > j && c <= r) sh_d[r * 17 + c] -= sh_d[r * 17 + j] * sh_d[c * 17 + j];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      // inverse of the lower-triangular block, one column per lane (forward substitution)
      if (lane < 16) {
        const int c = lane;
        double x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
          for (int q = 0; q < r; ++q) s -= (q >= c) ? sh_d[r * 17 + q] * x[q] : 0.0;
          x[r] = (r >= c) ? s / sh_d[r * 17 + r] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sh_dinv[r * 17 + c] = x[r];
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // write the diagonal block (lower, zero upper) to L and LT, and Dinv / Dinv^T to the workspace
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = lane + 64 * e;
        const int r = idx >> 4, c = idx & 15;
        const double v = (c <= r) ? sh_d[r * 17 + c] : 0.0;
        L[(size_t)(16 * kb + r) * Mp + 16 * kb + c] = v;
        LT[(size_t)(16 * kb + c) * Mp + 16 * kb + r] = v;
        const double di = sh_dinv[r * 17 + c];
        f.dinv[(size_t)kb * 256 + r * 16 + c] = di;
        f.dinvT[(size_t)kb * 256 + c * 16 + r] = di;
      }
    }
    __syncthreads();
    // (3) panel below the diagonal block: L[i][16kb + c] = sum_{q <= c} S[i][16kb + q] Dinv[c][q]
    const int rows_below = Mp - 16 * (kb + 1);
    for (int idx = threadIdx.x; idx < rows_below * 16; idx += NT) {
      const int i = 16 * (kb + 1) + (idx >> 4), c = idx & 15;
      const double* srow = L + (size_t)i * Mp + 16 * kb;
      double sv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) sv[q] = srow[q];
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) s += (q <= c) ? sv[q] * sh_dinv[c * 17 + q] : 0.0;
      __builtin_amdgcn_wave_barrier();  // all 16 lanes of a row have read S before any of them overwrites it
      L[(size_t)i * Mp + 16 * kb + c] = s;
      LT[(size_t)(16 * kb + c) * Mp + i] = s;
    }
    __syncthreads();
  }
}

// ---- LI = L^-1 (lower) and U = LI^T, one 16-wide block column per wave ----------------------------
//   LI_kk = Dinv_k;   LI_ik = -Dinv_i * sum_{j=k}^{i-1} L_ij LI_jk   (i > k)
// Block columns are independent; inside one, block rows are sequential but need no workgroup
// barrier (a wave re-reads only blocks it wrote itself, after a workgroup-scope fence).
__device__ __noinline__ void tri_inverse(const Fit& f) {
  const int Mp = f.Mp, nb = Mp / 16;
  const double* LT = f.mat[B_LT];
  double* LI = f.mat[B_LI];
  double* U = f.mat[B_U];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lr = lane & 15, lq = lane >> 4;
  for (int k = wave; k < nb; k += NW) {
    // diagonal block (blocks above it stay zero: the buffer is zero-initialised and never written there)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = lq + 4 * r;
      const double v = f.dinv[(size_t)k * 256 + rr * 16 + lr];
      LI[(size_t)(16 * k + rr) * Mp + 16 * k + lr] = v;
      U[(size_t)(16 * k + lr) * Mp + 16 * k + rr] = v;
    }
    for (int i = k + 1; i < nb; ++i) {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
      for (int j = k; j < i; ++j) {
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
          const double a = LT[(size_t)(16 * j + q + lq) * Mp + 16 * i + lr];  // L[16i + lr][16j + q + lq]
          const double b = LI[(size_t)(16 * j + q + lq) * Mp + 16 * k + lr];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
      }
      // out = -Dinv_i * acc : register r of acc holds rows 4r + lq, exactly the B operand of k-step r
      d4 out = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double a = -f.dinvT[(size_t)i * 256 + (4 * s + lq) * 16 + lr];  // -Dinv_i[lr][4s + lq]
        out = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[s], out, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = lq + 4 * r;
        LI[(size_t)(16 * i + rr) * Mp + 16 * k + lr] = out[r];
        U[(size_t)(16 * k + lr) * Mp + 16 * i + rr] = out[r];
      }
    }
  }
}

// ---- kernel matrices --------------------------------------------------------------------------------
__device__ inline double sqdist(const double* a, const double* b, int D) {
  double s = 0.0;
  for (int d = 0; d < D; ++d) {
    const double t = a[d] - b[d];
    s += t * t;
  }
  return s;
}

// Kzz + jitter*I (lower incl. diagonal, zero strict upper, identity on the padded tail) into B_L.
__device__ __noinline__ void build_kzz(const Fit& f, double s, double inv_l2, double jitter) {
  const int Mp = f.Mp, M = f.M, D = f.D;
  double* L = f.mat[B_L];
  for (int idx = threadIdx.x; idx < Mp * Mp; idx += NT) {
    const int i = idx / Mp, j = idx - i * Mp;
    double v = 0.0;
    if (i < M && j <= i) {
      v = s * exp(-0.5 * inv_l2 * sqdist(f.Z + (size_t)i * D, f.Z + (size_t)j * D, D));
      if (i == j) v += jitter;
    } else if (i >= M && i == j) {
      v = 1.0;
    }
    L[idx] = v;
  }
}

// KX[k][n] = s exp(-|Z_k - P_n|^2 / (2 l^2)) for n < ncols (points P), zero elsewhere.
__device__ __noinline__ void build_kx(const Fit& f, const double* pts, int ncols, double s, double inv_l2) {
  const int Mp = f.Mp, M = f.M, D = f.D;
  double* KX = f.mat[B_KX];
  for (int idx = threadIdx.x; idx < Mp * Mp; idx += NT) {
    const int k = idx / Mp, n = idx - k * Mp;
    double v = 0.0;
    if (k < M && n < ncols) v = s * exp(-0.5 * inv_l2 * sqdist(f.Z + (size_t)k * D, pts + (size_t)n * D, D));
    KX[idx] = v;
  }
}

// out[c] = sum_r w[r] * Mtx[r][c]  for c < Mp (deterministic: fixed row partition, fixed order)
__device__ __noinline__ void weighted_colsum(const double* Mtx, const double* w, int Mp, double* out, double* sh_part) {
  const int G = NT / Mp > 0 ? NT / Mp : 1;  // row groups (Mp <= NT) or 1
  __syncthreads();
  for (int c0 = 0; c0 < Mp; c0 += NT) {
    const int c = c0 + (threadIdx.x % (Mp < NT ? Mp : NT));
    const int g = Mp < NT ? threadIdx.x / Mp : 0;
    if (g < G && c < Mp) {
      double s = 0.0;
      for (int r = g; r < Mp; r += G) s += w[r] * Mtx[(size_t)r * Mp + c];
      sh_part[g * (Mp < NT ? Mp : NT) + (c - c0)] = s;
    }
    __syncthreads();
    if (threadIdx.x < (Mp < NT ? Mp : NT) && c0 + (int)threadIdx.x < Mp) {
      double s = 0.0;
      for (int g2 = 0; g2 < G; ++g2) s += sh_part[g2 * (Mp < NT ? Mp : NT) + threadIdx.x];
      out[c0 + threadIdx.x] = s;
    }
    __syncthreads();
  }
}

// var[n] = s + jitter + sum_i (BM[i][n]^2 - A[i][n]^2), clamped at min_variance (gradient 0 if clamped)
__device__ __noinline__ void column_variance(const Fit& f, double s, double jitter, double* sh_part) {
  const int Mp = f.Mp;
  const double* A = f.mat[B_A];
  const double* BM = f.mat[B_BM];
  const int W = Mp < NT ? Mp : NT;
  const int G = NT / Mp > 0 ? NT / Mp : 1;
  __syncthreads();
  for (int c0 = 0; c0 < Mp; c0 += NT) {
    const int c = c0 + (threadIdx.x % W);
    const int g = Mp < NT ? threadIdx.x / Mp : 0;
    if (g < G && c < Mp) {
      double acc = 0.0;
      for (int r = g; r < Mp; r += G) {
        const double b = BM[(size_t)r * Mp + c], a = A[(size_t)r * Mp + c];
        acc += b * b - a * a;
      }
      sh_part[g * W + (c - c0)] = acc;
    }
    __syncthreads();
    if (threadIdx.x < W && c0 + (int)threadIdx.x < Mp) {
      double acc = 0.0;
      for (int g2 = 0; g2 < G; ++g2) acc += sh_part[g2 * W + threadIdx.x];
      f.vec[V_VAR][c0 + threadIdx.x] = s + jitter + acc;
    }
    __syncthreads();
  }
}

// A = LI * KX (+ AT), then BMT = A^T LS (+ BM), over `ncol_tiles` 16TU-wide column tiles
template <int TU>
__device__ __noinline__ void forward_products(const Fit& f, int ncols) {
  const int Mp = f.Mp;
  constexpr int TS = 16 * TU;
  const int mt = Mp / TS, nt = (ncols + TS - 1) / TS;
  double* A = f.mat[B_A];
  double* AT = f.mat[B_AT];
  double* BM = f.mat[B_BM];
  double* BMT = f.mat[B_BMT];
  // A[i][n] = sum_k U[k][i] KX[k][n],  U[k][i] = LI[i][k] = 0 for k > i
  gemm_tn<TU>(mt, nt, false, f.mat[B_U], f.mat[B_KX], Mp,
              [=](int i0, int, int* lo, int* hi) { *lo = 0; *hi = i0 + TS; }, NoScale(),
              [=](int i, int n, double v) { A[(size_t)i * Mp + n] = v; AT[(size_t)n * Mp + i] = v; });
  __syncthreads();
  // BMT[n][j] = sum_i A[i][n] LS[i][j],  LS[i][j] = 0 for i < j
  gemm_tn<TU>(nt, mt, false, A, f.mat[B_LS], Mp,
              [=](int, int j0, int* lo, int* hi) { *lo = j0; *hi = Mp; }, NoScale(),
              [=](int n, int j, double v) { BMT[(size_t)n * Mp + j] = v; BM[(size_t)j * Mp + n] = v; });
  __syncthreads();
}

#ifdef GAPRO_PROFILE
constexpr int kProfSlots = 20;
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
  double part[NT];
  double c, rho_s, rho_l, s, ell, inv_l2;
  int status;
  int chol_bad;
};

template <int TU>
__device__ void fit_body(const Fit& f, const gapro_fit_options& opt, Shared& sh, const gapro_fit_desc& desc,
                         float* __restrict__ o_probs, float* __restrict__ o_probs_new,
                         unsigned char* __restrict__ o_labels, float* __restrict__ o_mu, float* __restrict__ o_var,
                         double* loss_out) {
  const int M = f.M, Mp = f.Mp, D = f.D, T = f.T;
  constexpr int TS = 16 * TU;
  const int mt = Mp / TS;
  const double Nd = (double)M;  // num_data = train_y.numel() (gaussian_process_utils.py:414)
  const double jitter = opt.jitter;
  double* LS = f.mat[B_LS];
  double* LST = f.mat[B_LST];
  double* GLS = f.mat[B_GLS];
  double* A = f.mat[B_A];
  double* AT = f.mat[B_AT];
  double* BM = f.mat[B_BM];
  double* BMT = f.mat[B_BMT];
  double* GA = f.mat[B_GA];
  double* GKX = f.mat[B_GKX];
  double* GKXT = f.mat[B_GKXT];
  double* KX = f.mat[B_KX];
  double* vm = f.vec[V_M];
  double* gmu = f.vec[V_GMU];
  double* gv = f.vec[V_GV];
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
    // psd_safe_cholesky of gpytorch: repeat on K + psd_jitter 10^i I (i < psd_retries) before giving up; see the
    // comment at GAPRO_PSD_SAFE_CHOLESKY in svgp_fit.hip
    double extra = 0.0;
    for (int attempt = 0;; ++attempt) {
      build_kzz(f, sh.s, sh.inv_l2, jitter + extra);
      __syncthreads();
      stamp(0);
      cholesky_blocked(f, sh.dblk, sh.dinv, &sh.chol_bad);
      __syncthreads();
      const int bad = sh.chol_bad;
      __syncthreads();
      if (!bad) break;
      if (threadIdx.x == 0) {
        sh.chol_bad = 0;
        if (attempt >= opt.psd_retries) sh.status = GAPRO_ERR_CHOLESKY;
      }
      __syncthreads();
      if (attempt >= opt.psd_retries) break;
      extra = opt.psd_jitter * pow(10.0, (double)attempt);
    }
    stamp(1);
    tri_inverse(f);
    __syncthreads();
    stamp(2);
  };

  for (int step = 1; step <= opt.training_iter; ++step) {
    refresh_hypers();
    const double s = sh.s, ell = sh.ell, inv_l2 = sh.inv_l2, c = sh.c;
    // ------------------------------- forward -------------------------------
    factorize();
    build_kx(f, f.X, M, s, inv_l2);
    __syncthreads();
    stamp(3);
    forward_products<TU>(f, M);
    stamp(4);
    weighted_colsum(A, vm, Mp, f.vec[V_MU], sh.part);  // mu (without c)
    column_variance(f, s, jitter, sh.part);
    // quadrature: E_n, dE/dmu, dE/dvar  (BernoulliLikelihood.expected_log_prob, 20-point Gauss-Hermite)
    double e_part = 0.0, gmu_part = 0.0, gv_part = 0.0;
    for (int n = threadIdx.x; n < Mp; n += NT) {
      double g1 = 0.0, g2 = 0.0;
      if (n < M) {
        const double mu = f.vec[V_MU][n] + c;
        const double vraw = f.vec[V_VAR][n];
        const bool clamped = vraw < opt.min_variance;
        const double var = clamped ? opt.min_variance : vraw;
        const double sd = sqrt(2.0 * var);
        const double y = f.vec[V_Y][n];
        double E = 0.0, dmu = 0.0, dvar = 0.0;
        for (int q = 0; q < NGH / 2; ++q) {
          const double t = c_gh_t[q], w = c_gh_w[q];
          double lp, r;
          log_ndtr_ratio(y * (mu - sd * t), &lp, &r);
          E += w * lp; dmu += w * r; dvar -= w * t * r;
          log_ndtr_ratio(y * (mu + sd * t), &lp, &r);
          E += w * lp; dmu += w * r; dvar += w * t * r;
        }
        const double ipi = 0.56418958354775628695;  // 1/sqrt(pi)
        e_part += ipi * E;
        g1 = -(ipi * dmu * y) / Nd;
        g2 = clamped ? 0.0 : -(ipi * dvar * y / sd) / Nd;
      }
      gmu[n] = g1;
      gv[n] = g2;
      gmu_part += g1;
      gv_part += g2;
    }
    const double e_sum = block_sum(e_part, sh.red);
    const double g_c = block_sum(gmu_part, sh.red);
    const double gv_sum = block_sum(gv_part, sh.red);
    // KL(q(u) || N(0, I)) = 0.5 (|LS|_F^2 + |m|^2 - M - sum log LS_jj^2)
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
    const double kl = 0.5 * (block_sum(kl_part, sh.red) - Nd);
    last_loss = -(e_sum / Nd - kl / Nd);
    stamp(6);

    // ------------------------------- backward ------------------------------
    // G_m = A g_mu + m / N  (through AT, coalesced)
    weighted_colsum(AT, gmu, Mp, f.vec[V_GM], sh.part);
    for (int i = threadIdx.x; i < M; i += NT) f.vec[V_GM][i] += vm[i] / Nd;
    __syncthreads();
    // G_A[i][n] = 2 g_v[n] sum_j LS[i][j] BM[j][n] + m[i] g_mu[n] - 2 A[i][n] g_v[n]
    gemm_tn<TU>(mt, mt, false, LST, BM, Mp, [=](int i0, int, int* lo, int* hi) { *lo = 0; *hi = i0 + TS; },
                NoScale(), [=](int i, int n, double v) {
                  GA[(size_t)i * Mp + n] = 2.0 * gv[n] * v + vm[i] * gmu[n] - 2.0 * A[(size_t)i * Mp + n] * gv[n];
                });
    stamp(7);
    // G_LS[i][j] = sum_n A[i][n] 2 g_v[n] BM[j][n] (lower) + KL' ; n runs over the M train columns
    gemm_tn<TU>(mt, mt, true, AT, BMT, Mp, [=](int, int, int* lo, int* hi) { *lo = 0; *hi = Mp; },
                [=](int k) { return 2.0 * gv[k]; },
                [=](int i, int j, double v) {
                  double g = 0.0;
                  if (j <= i && i < M) {
                    const double l = LS[(size_t)i * Mp + j];
                    g = v + (l - (i == j ? 1.0 / l : 0.0)) / Nd;
                  }
                  GLS[(size_t)i * Mp + j] = g;
                });
    __syncthreads();
    stamp(8);
    // G_KX = LI^T G_A   (P = LI[k][i], non-zero for k >= i)
    gemm_tn<TU>(mt, mt, false, f.mat[B_LI], GA, Mp, [=](int i0, int, int* lo, int* hi) { *lo = i0; *hi = Mp; },
                NoScale(), [=](int i, int n, double v) { GKX[(size_t)i * Mp + n] = v; GKXT[(size_t)n * Mp + i] = v; });
    __syncthreads();
    stamp(9);
    // G_L = -tril(G_KX A^T)  -> BM buffer (lower tiles; strict upper of diagonal tiles zeroed)
    double* GL = BM;
    gemm_tn<TU>(mt, mt, true, GKXT, AT, Mp, [=](int, int, int* lo, int* hi) { *lo = 0; *hi = Mp; }, NoScale(),
                [=](int i, int j, double v) { GL[(size_t)i * Mp + j] = (j <= i) ? -v : 0.0; });
    __syncthreads();
    stamp(10);
    // G_Kzz (unsymmetrised) = L^-T Pm L^-1, Pm = Phi(tril(L^T G_L)), associated as L^-T (Pm L^-1) like the MFMA kernels
    // (svgp_fit.hip): W = Pm L^-1 is lower (M^3 / 3), S = L^-T W costs 2 M^3 / 3.
    // Pm^T -> GA buffer   (k >= i0 on lower tiles)
    double* PmT = GA;
    gemm_tn<TU>(mt, mt, true, f.mat[B_L], GL, Mp, [=](int i0, int, int* lo, int* hi) { *lo = i0; *hi = Mp; },
                NoScale(),
                [=](int i, int j, double v) { PmT[(size_t)j * Mp + i] = (j < i) ? v : (j == i ? 0.5 * v : 0.0); });
    __syncthreads();
    stamp(11);
    // W = Pm L^-1 (lower tiles, zeros above the diagonal inside them) -> BMT buffer   (j0 <= k < i0 + tile)
    double* Wm = BMT;
    gemm_tn<TU>(mt, mt, true, PmT, f.mat[B_LI], Mp,
                [=](int i0, int j0, int* lo, int* hi) { *lo = j0; *hi = i0 + TS; }, NoScale(),
                [=](int i, int j, double v) { Wm[(size_t)i * Mp + j] = (j <= i) ? v : 0.0; });
    __syncthreads();
    stamp(12);
    // S = L^-T W -> G in the BM buffer, G^T in the GKXT buffer   (k >= max(i0, j0): lower tiles of W only)
    double* G = BM;
    double* GT = GKXT;
    gemm_tn<TU>(mt, mt, false, f.mat[B_LI], Wm, Mp,
                [=](int i0, int j0, int* lo, int* hi) { *lo = i0 > j0 ? i0 : j0; *hi = Mp; }, NoScale(),
                [=](int i, int j, double v) { G[(size_t)i * Mp + j] = v; GT[(size_t)j * Mp + i] = v; });
    __syncthreads();
    stamp(13);
    // kernel weights: Wzz = sym(G) o (s Ezz) -> BM buffer in place, Wzx = G_KX o KX -> GKX in place;
    // scalar sums for d/ds and d/dl
    double gs_part = 0.0, gl_part = 0.0;
    for (int idx = threadIdx.x; idx < M * M; idx += NT) {
      const int i = idx / M, j = idx - i * M;
      const size_t o = (size_t)i * Mp + j;
      const double d2 = sqdist(f.Z + (size_t)i * D, f.Z + (size_t)j * D, D);
      const double e = exp(-0.5 * inv_l2 * d2);
      const double gsym = 0.5 * (G[o] + GT[o]);
      const double w = gsym * s * e;
      gs_part += gsym * e;
      gl_part += w * d2;
      G[o] = w;
      const double d2x = sqdist(f.Z + (size_t)i * D, f.X + (size_t)j * D, D);
      const double kx = KX[o];
      const double wx = GKX[o] * kx;
      gs_part += GKX[o] * kx / s;
      gl_part += wx * d2x;
      GKX[o] = wx;
    }
    const double g_s = block_sum(gs_part, sh.red) + gv_sum;
    const double g_l = block_sum(gl_part, sh.red) / (ell * ell * ell);
    stamp(14);
    // G_Z[i][d] = -(1/l^2) ( sum_j 2 Wzz[i][j] (Z_i - Z_j)[d] + sum_n Wzx[i][n] (Z_i - X_n)[d] )
    for (int idx = threadIdx.x; idx < M * D; idx += NT) {
      const int i = idx / D, d = idx - i * D;
      const double zi = f.Z[(size_t)i * D + d];
      const double* wz = G + (size_t)i * Mp;
      const double* wx = GKX + (size_t)i * Mp;
      double acc = 0.0;
      for (int j = 0; j < M; ++j)
        acc += 2.0 * wz[j] * (zi - f.Z[(size_t)j * D + d]) + wx[j] * (zi - f.X[(size_t)j * D + d]);
      f.gZ[idx] = -inv_l2 * acc;
    }
    __syncthreads();
    stamp(15);

    // ------------------------------- Adam ----------------------------------
    const double b1 = 0.9, b2 = 0.999, aeps = 1e-8;
    const double bc1 = 1.0 - pow(b1, (double)step), bc2s = sqrt(1.0 - pow(b2, (double)step));
    const double step_size = opt.lr / bc1;
    auto adam = [&](double& p, double& m1, double& m2, double g) {
      m1 = b1 * m1 + (1.0 - b1) * g;
      m2 = b2 * m2 + (1.0 - b2) * g * g;
      p -= step_size * m1 / (sqrt(m2) / bc2s + aeps);
    };
    for (int idx = threadIdx.x; idx < M * D; idx += NT) adam(f.Z[idx], f.mZ[idx], f.vZ[idx], f.gZ[idx]);
    for (int i = threadIdx.x; i < M; i += NT) adam(vm[i], f.vec[V_MM][i], f.vec[V_VM][i], f.vec[V_GM][i]);
    for (int idx = threadIdx.x; idx < M * M; idx += NT) {
      const int i = idx / M, j = idx - i * M;
      if (j <= i) {
        const size_t o = (size_t)i * Mp + j;
        adam(LS[o], f.mat[B_MLS][o], f.mat[B_VLS][o], GLS[o]);
        LST[(size_t)j * Mp + i] = LS[o];
      }
    }
    if (threadIdx.x == 0) {
      adam(sh.c, f.scal[S_MC], f.scal[S_VC], g_c);
      adam(sh.rho_s, f.scal[S_MRS], f.scal[S_VRS], g_s * sigmoid(sh.rho_s));
      adam(sh.rho_l, f.scal[S_MRL], f.scal[S_VRL], g_l * sigmoid(sh.rho_l));
    }
    __syncthreads();
    stamp(16);
  }

  // ------------------------------- prediction ------------------------------
  refresh_hypers();
  if (!(opt.eval_stale_chol && opt.training_iter > 0)) factorize();
  const double s = sh.s, inv_l2 = sh.inv_l2, c = sh.c;
  for (int t0 = 0; t0 < T; t0 += Mp) {
    const int nc = (T - t0) < Mp ? (T - t0) : Mp;
    build_kx(f, f.Xt + (size_t)t0 * D, nc, s, inv_l2);
    __syncthreads();
    forward_products<TU>(f, nc);
    weighted_colsum(A, vm, Mp, f.vec[V_MU], sh.part);
    column_variance(f, s, jitter, sh.part);
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

__global__ __launch_bounds__(NT) void k_svgp_fit_large(int n_fits, int D, const float* __restrict__ feats_spp,
                                                 const int* __restrict__ idx, const gapro_fit_desc* __restrict__ descs,
                                                 const double* __restrict__ init_mean, gapro_fit_options opt,
                                                 double* __restrict__ ws, float* __restrict__ o_probs,
                                                 float* __restrict__ o_probs_new, unsigned char* __restrict__ o_labels,
                                                 float* __restrict__ o_mu, float* __restrict__ o_var,
                                                 int* __restrict__ o_status, double* __restrict__ o_loss) {
  __shared__ Shared sh;
  const int fit = blockIdx.x;
  if (fit >= n_fits) return;
  const gapro_fit_desc desc = descs[fit];
  Fit& f = sh.f;
  if (threadIdx.x == 0) {
    f.M = desc.m1 + desc.m2;
    f.T = desc.t;
    f.D = D;
  }
  const Layout lay = make_layout(desc.m1 + desc.m2, desc.t, D);
  double* base = ws + desc.ws_offset;
  if (threadIdx.x == 0) {
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
  }
  __syncthreads();

  // ---- initialisation (gaussian_process_utils.py:386-403; gpytorch parameter inits) ----
  for (long long i = threadIdx.x; i < lay.total; i += NT) base[i] = 0.0;
  __syncthreads();
  const int M = f.M, Mp = f.Mp;
  const int* my_idx = idx + desc.idx_offset;
  for (int e = threadIdx.x; e < M * D; e += NT) {
    const int i = e / D, d = e - i * D;
    const double v = (double)feats_spp[(size_t)my_idx[i] * D + d];  // train_x = cat(b1_feats, b2_feats)  :395
    f.X[e] = v;
    f.Z[e] = v;  // inducing points initialised to train_x  (:14)
  }
  for (int e = threadIdx.x; e < f.T * D; e += NT) {
    const int i = e / D, d = e - i * D;
    f.Xt[e] = (double)feats_spp[(size_t)my_idx[M + i] * D + d];  // intersect_feats  :386
  }
  for (int i = threadIdx.x; i < M; i += NT) {
    f.vec[V_Y][i] = i < desc.m1 ? -1.0 : 1.0;  // train_y  :396-398
    f.vec[V_M][i] = init_mean ? init_mean[desc.idx_offset + i] : 0.0;
    f.mat[B_LS][(size_t)i * Mp + i] = 1.0;  // chol_variational_covar = I
    f.mat[B_LST][(size_t)i * Mp + i] = 1.0;
  }
  if (threadIdx.x == 0) {
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
  __syncthreads();
  // 32 x 32 wave tiles need M_p to be a multiple of 32: fit_body<2> counts tiles as M_p / 32 and would leave the last 16
  // rows / columns of an odd multiple of 16 uncomputed (D > 32 reaches this kernel with M_p = 144, 176, ...)
  if (Mp >= 128 && Mp % 32 == 0)
    fit_body<2>(f, opt, sh, desc, o_probs, o_probs_new, o_labels, o_mu, o_var, &o_loss[desc.slot]);
  else
    fit_body<1>(f, opt, sh, desc, o_probs, o_probs_new, o_labels, o_mu, o_var, &o_loss[desc.slot]);
  __syncthreads();
  if (threadIdx.x == 0) {
    int st = sh.status;
    if (st == GAPRO_OK && !isfinite(o_loss[desc.slot]) && opt.training_iter > 0) st = GAPRO_ERR_NOT_FINITE;
    o_status[desc.slot] = st;
    f.scal[S_STATUS] = (double)st;
  }
}

}  // namespace

// Internal launcher used by gapro_svgp_fit_batch (svgp_fit.hip) for fits that exceed the LDS-staged kernel.
void gapro_launch_fit_large(hipStream_t stream, int n_fits, int feat_dim, const float* d_feats_spp, const int* d_idx,
                            const gapro_fit_desc* d_descs, const double* d_init_mean, const gapro_fit_options& opt,
                            double* d_workspace, float* d_probs, float* d_probs_new, unsigned char* d_labels,
                            float* d_mu, float* d_var, int* d_fit_status, double* d_fit_loss) {
  hipLaunchKernelGGL(k_svgp_fit_large, dim3(n_fits), dim3(NT), 0, stream, n_fits, feat_dim, d_feats_spp, d_idx, d_descs,
                     d_init_mean, opt, d_workspace, d_probs, d_probs_new, d_labels, d_mu, d_var, d_fit_status,
                     d_fit_loss);
}
