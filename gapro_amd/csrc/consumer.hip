// Consumer-side label ops: what ISBNet / SPFormer do with the generated
// pseudo labels on the training side of the .pth boundary.
//   * custom_scatter_mean of prob / mu / var to superpoints   ISBNet/isbnet/model/model_utils.py:600-613,
//                                                             isbnet.py:387-389
//   * probability-weighted BCE-with-logits                    ISBNet/isbnet/model/criterion.py:287-288
//   * KL-to-GP auxiliary loss                                 ISBNet/isbnet/model/criterion.py:435-463
// Each is one or two streaming passes: forward value and the gradients w.r.t. the network outputs come out of
// the same launches (the reductions are float64 sums of float32 terms, so the values do not depend on the
// launch shape beyond float64 rounding).
#include "common.h"

namespace {

constexpr int kThreads = 256;

inline int grid_for(long long n, int cap = 1024) {
  long long g = (n + kThreads - 1) / kThreads;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

__device__ inline double wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-reduce K doubles and add them to out[0..K) with one atomic per workgroup and value
template <int K>
__device__ inline void block_accumulate(double (&v)[K], double* out) {
  __shared__ double sh[K][kThreads / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double s = wave_sum_d(v[k]);
    if (lane == 0) sh[k][w] = s;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double s = 0.0;
    for (int j = 0; j < kThreads / 64; ++j) s += sh[threadIdx.x][j];
    atomicAdd(&out[threadIdx.x], s);
  }
}

// ---- custom_scatter_mean of three label channels ------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_pool3_sum(long long n, const long long* __restrict__ idx,
                                                        const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ c, double* __restrict__ sums,
                                                        int* __restrict__ counts) {
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const long long s = idx[i];
    atomicAdd(&sums[3 * s], (double)a[i]);
    atomicAdd(&sums[3 * s + 1], (double)b[i]);
    atomicAdd(&sums[3 * s + 2], (double)c[i]);
    atomicAdd(&counts[s], 1);
  }
}
__global__ __launch_bounds__(kThreads) void k_pool3_mean(int n_out, const double* __restrict__ sums,
                                                         const int* __restrict__ counts, float* __restrict__ oa,
                                                         float* __restrict__ ob, float* __restrict__ oc) {
  const int s = blockIdx.x * kThreads + threadIdx.x;
  if (s >= n_out) return;
  const double cnt = counts[s] > 0 ? (double)counts[s] : 1.0;  // torch_scatter clamps the count at 1
  oa[s] = (float)(sums[3 * s] / cnt);
  ob[s] = (float)(sums[3 * s + 1] / cnt);
  oc[s] = (float)(sums[3 * s + 2] / cnt);
}

// ---- probability-weighted BCE with logits ---------------------------------------------------------------
// loss = sum_{g,p} bce(x[g][p], y[g][p]) w[p] / sum_p w[p] / (G + 1e-6)          criterion.py:287-288
// acc[0] = sum bce * w, acc[1] = sum w
__global__ __launch_bounds__(kThreads) void k_wbce_reduce(int G, long long P, const float* __restrict__ x,
                                                          const float* __restrict__ y, const float* __restrict__ w,
                                                          double* __restrict__ acc) {
  double v[2] = {0.0, 0.0};
  const long long n = (long long)G * P, stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const long long p = i % P;
    const float xi = x[i], yi = y[i];
    // F.binary_cross_entropy_with_logits: max(x, 0) - x y + log(1 + exp(-|x|))
    const float l = fmaxf(xi, 0.f) - xi * yi + log1pf(expf(-fabsf(xi)));
    v[0] += (double)(l * w[p]);
    if (i < P) v[1] += (double)w[i];
  }
  block_accumulate<2>(v, acc);
}
__global__ __launch_bounds__(kThreads) void k_wbce_grad(int G, long long P, const float* __restrict__ x,
                                                        const float* __restrict__ y, const float* __restrict__ w,
                                                        const double* __restrict__ acc, float grad_out,
                                                        float* __restrict__ loss, float* __restrict__ gx) {
  const double sw = acc[1], denom = sw * ((double)G + 1e-6);
  if (blockIdx.x == 0 && threadIdx.x == 0) *loss = (float)(acc[0] / sw / ((double)G + 1e-6));
  if (!gx) return;
  const long long n = (long long)G * P, stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const double xi = (double)x[i];
    const double sg = 1.0 / (1.0 + exp(-xi));  // float64: sigmoid(x) - y cancels for confident predictions
    gx[i] = (float)((double)grad_out * (sg - (double)y[i]) * (double)w[i % P] / denom);
  }
}

// ---- KL-to-GP loss -----------------------------------------------------------------------------------------
// acc[0..3] = sum / count of the var <= eps branch, sum / count of the var > eps branch   criterion.py:445-461
__device__ inline int kl_branch(float mu_l, float var_l, float eps) {
  if (mu_l == -100.f || var_l == -100.f) return 0;
  return var_l <= eps ? 1 : 2;
}
__global__ __launch_bounds__(kThreads) void k_kl_reduce(long long n, const float* __restrict__ mu_l,
                                                        const float* __restrict__ var_l, const float* __restrict__ mu_p,
                                                        const float* __restrict__ lv_p, float eps,
                                                        double* __restrict__ acc) {
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const int br = kl_branch(mu_l[i], var_l[i], eps);
    if (br == 1) {
      const float e = expf(lv_p[i]) - 1.f, d = mu_p[i] - mu_l[i];
      v[0] += (double)(e * e + d * d);
      v[1] += 1.0;
    } else if (br == 2) {
      const float d = mu_p[i] - mu_l[i], vl = var_l[i], lv = lv_p[i];
      v[2] += (double)((lv - logf(vl)) + (d * d + vl * vl) * expf(-2.f * lv) - 0.5f);
      v[3] += 1.0;
    }
  }
  block_accumulate<4>(v, acc);
}
__global__ __launch_bounds__(kThreads) void k_kl_grad(long long n, const float* __restrict__ mu_l,
                                                      const float* __restrict__ var_l, const float* __restrict__ mu_p,
                                                      const float* __restrict__ lv_p, float eps, float weight,
                                                      const double* __restrict__ acc, float grad_out,
                                                      float* __restrict__ loss, float* __restrict__ g_mu,
                                                      float* __restrict__ g_lv) {
  const double c0 = acc[1], c1 = acc[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double l = 0.0;
    if (c0 > 0.0) l += acc[0] / (c0 + 1e-4) * (double)weight;
    if (c1 > 0.0) l += acc[2] / (c1 + 1e-4) * (double)weight;
    *loss = (float)l;
  }
  if (!g_mu || !g_lv) return;
  const double s0 = (double)grad_out * (double)weight / (c0 + 1e-4), s1 = (double)grad_out * (double)weight / (c1 + 1e-4);
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const int br = kl_branch(mu_l[i], var_l[i], eps);
    float gm = 0.f, gl = 0.f;
    if (br == 1) {
      const double ex = exp((double)lv_p[i]);
      gm = (float)(s0 * 2.0 * ((double)mu_p[i] - (double)mu_l[i]));
      gl = (float)(s0 * 2.0 * (ex - 1.0) * ex);
    } else if (br == 2) {
      const double d = (double)mu_p[i] - (double)mu_l[i], vl = (double)var_l[i], e2 = exp(-2.0 * (double)lv_p[i]);
      gm = (float)(s1 * 2.0 * d * e2);
      gl = (float)(s1 * (1.0 - 2.0 * (d * d + vl * vl) * e2));
    }
    g_mu[i] = gm;
    g_lv[i] = gl;
  }
}

}  // namespace

extern "C" {

int gapro_label_pool_mean(gapro_ctx* ctx, void* stream_, int64_t n_points, int32_t n_out, const int64_t* d_index,
                          const float* d_prob, const float* d_mu, const float* d_var, double* d_sums_ws,
                          int32_t* d_counts_ws, float* d_out_prob, float* d_out_mu, float* d_out_var) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  if (n_points <= 0 || n_out <= 0 || !d_index || !d_prob || !d_mu || !d_var || !d_sums_ws || !d_counts_ws ||
      !d_out_prob || !d_out_mu || !d_out_var)
    return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_label_pool_mean: bad argument");
  hipStream_t stream = (hipStream_t)stream_;
  GAPRO_HIP_CHECK(ctx, hipMemsetAsync(d_sums_ws, 0, (size_t)n_out * 3 * sizeof(double), stream));
  GAPRO_HIP_CHECK(ctx, hipMemsetAsync(d_counts_ws, 0, (size_t)n_out * sizeof(int32_t), stream));
  hipLaunchKernelGGL(k_pool3_sum, dim3(grid_for(n_points, 2048)), dim3(kThreads), 0, stream, (long long)n_points,
                     (const long long*)d_index, d_prob, d_mu, d_var, d_sums_ws, d_counts_ws);
  hipLaunchKernelGGL(k_pool3_mean, dim3((n_out + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, (int)n_out,
                     d_sums_ws, d_counts_ws, d_out_prob, d_out_mu, d_out_var);
  GAPRO_LAUNCH_CHECK(ctx);
  return GAPRO_OK;
}

int gapro_weighted_bce_with_logits(gapro_ctx* ctx, void* stream_, int32_t n_rows, int64_t n_cols, const float* d_logits,
                                   const float* d_targets, const float* d_weights, float grad_out, double* d_acc2,
                                   float* d_loss, float* d_grad_logits) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  if (n_rows <= 0 || n_cols <= 0 || !d_logits || !d_targets || !d_weights || !d_acc2 || !d_loss)
    return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_weighted_bce_with_logits: bad argument");
  hipStream_t stream = (hipStream_t)stream_;
  const long long n = (long long)n_rows * n_cols;
  GAPRO_HIP_CHECK(ctx, hipMemsetAsync(d_acc2, 0, 2 * sizeof(double), stream));
  hipLaunchKernelGGL(k_wbce_reduce, dim3(grid_for(n)), dim3(kThreads), 0, stream, (int)n_rows, (long long)n_cols,
                     d_logits, d_targets, d_weights, d_acc2);
  hipLaunchKernelGGL(k_wbce_grad, dim3(d_grad_logits ? grid_for(n) : 1), dim3(kThreads), 0, stream, (int)n_rows,
                     (long long)n_cols, d_logits, d_targets, d_weights, (const double*)d_acc2, grad_out, d_loss,
                     d_grad_logits);
  GAPRO_LAUNCH_CHECK(ctx);
  return GAPRO_OK;
}

int gapro_kl_gp_loss(gapro_ctx* ctx, void* stream_, int64_t n, const float* d_mu_labels, const float* d_var_labels,
                     const float* d_mu_pred, const float* d_logvar_pred, float epsilon, float weight, float grad_out,
                     double* d_acc4, float* d_loss, float* d_grad_mu, float* d_grad_logvar) {
  if (!ctx) return GAPRO_ERR_BAD_ARG;
  if (n <= 0 || !d_mu_labels || !d_var_labels || !d_mu_pred || !d_logvar_pred || !d_acc4 || !d_loss)
    return gapro_fail(ctx, GAPRO_ERR_BAD_ARG, "gapro_kl_gp_loss: bad argument");
  hipStream_t stream = (hipStream_t)stream_;
  GAPRO_HIP_CHECK(ctx, hipMemsetAsync(d_acc4, 0, 4 * sizeof(double), stream));
  hipLaunchKernelGGL(k_kl_reduce, dim3(grid_for(n)), dim3(kThreads), 0, stream, (long long)n, d_mu_labels, d_var_labels,
                     d_mu_pred, d_logvar_pred, epsilon, d_acc4);
  const bool grads = d_grad_mu && d_grad_logvar;
  hipLaunchKernelGGL(k_kl_grad, dim3(grads ? grid_for(n) : 1), dim3(kThreads), 0, stream, (long long)n, d_mu_labels,
                     d_var_labels, d_mu_pred, d_logvar_pred, epsilon, weight, (const double*)d_acc4, grad_out, d_loss,
                     d_grad_mu, d_grad_logvar);
  GAPRO_LAUNCH_CHECK(ctx);
  return GAPRO_OK;
}

}  // extern "C"
