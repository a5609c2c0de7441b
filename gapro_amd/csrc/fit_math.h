// Scalar pieces of the SVGP fit shared by the fit kernels of libgapro_hip.so: Gauss-Hermite rule, link functions,
// wave-level helpers.  (svgp_fit.hip, svgp_fit_cluster.hip and svgp_fit_large.hip predate this header and still carry
// their own copies of these few lines; svgp_fit_wave.hip is the first kernel written against it.)
#pragma once
#include <math.h>

#include "common.h"

namespace gapro_fit_math {

// numpy.polynomial.hermite.hermgauss(20): positive nodes (ascending) and their weights; the rule is symmetric.
// Printed with repr() from NumPy 2.2.  (gpytorch settings.num_gauss_hermite_locs = 20)
constexpr double kGhT[10] = {0.24534070830090124, 0.7374737285453944, 1.234076215395323,  1.7385377121165861,
                             2.2549740020892757,  2.7888060584281305, 3.3478545673832163, 3.944764040115625,
                             4.603682449550744,   5.387480890011233};
constexpr double kGhW[10] = {0.4622436696006101,     0.28667550536283415,    0.1090172060200233,
                             0.024810520887463643,   0.0032437733422378567,  0.00022833863601635365,
                             7.80255647853206e-06,   1.0860693707692782e-07, 4.3993409922731747e-10,
                             2.2293936455341447e-13};

__device__ inline double softplus(double x) { return log1p(exp(-fabs(x))) + fmax(x, 0.0); }
__device__ inline double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }

// log Phi(z) and r(z) = phi(z) / Phi(z), both tails stable; both signs share t = erfcx(|z| / sqrt 2) and
// e = exp(-z^2 / 2) (see the derivation at its twin in svgp_fit.hip):
//   z <  0:  log Phi = log(t / 2) - z^2 / 2     r = sqrt(2 / pi) / t
//   z >= 0:  log Phi = log(1 - e t / 2)         r = e / (sqrt(2 pi) Phi)
__device__ inline void log_ndtr_ratio(double z, double* lp, double* r) {
  const double rs2 = 0.70710678118654752440;
  const double t = erfcx(fabs(z) * rs2);
  const double hz2 = 0.5 * z * z;
  const double e = exp(-hz2);
  const bool neg = z < 0.0;
  const double phi_pos = 1.0 - 0.5 * e * t;  // Phi(z) for z >= 0
  *lp = log(neg ? 0.5 * t : phi_pos) - (neg ? hz2 : 0.0);
  *r = neg ? 0.79788456080286535588 / t : e * 0.39894228040143267794 / phi_pos;
}

// value of `v` in lane `lane` (wave-uniform, compile-time after unrolling): v_readlane, no LDS crossbar
__device__ inline double lane_bcast(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

__device__ inline double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace gapro_fit_math
