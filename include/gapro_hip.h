/*
 * gapro_hip.h -- C ABI of libgapro_hip.so: the MI355X (gfx950) implementation of GaPro's
 * Gaussian-Process pseudo-label generator.
 *
 * The reference has no FFI for this path: it is plain Python on torch + gpytorch +
 * torch_scatter (paths relative to the reference checkout):
 *   gapro/gen_ps_utils.py:293-482            gen_pseudo_label_gaussian_process
 *   gapro/gaussian_process_utils.py:382-445  fit_gp_spp
 * Each entry point below names the reference lines it replaces.  A maintainer binds them
 * with ctypes (gapro_amd/_lib.py is that binding).
 *
 * Conventions
 *   - every function returns a gapro_status (0 = ok); nothing throws across the ABI;
 *     gapro_last_error(ctx) gives the message of the last failure on that context;
 *   - pointers named d_* are DEVICE pointers (hipMalloc'd or torch CUDA tensors), h_* are host
 *     pointers; the caller allocates every buffer, the library owns only gapro_ctx and
 *     gapro_schedule handles;
 *   - `stream` is a hipStream_t passed as void* (0 = null stream); device entry points only
 *     enqueue work unless documented "blocking";
 *   - one ctx per (process, device); a ctx is not thread-safe, distinct ctxs are.
 */
#ifndef GAPRO_HIP_H
#define GAPRO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAPRO_VERSION 200 /* 0.2.0 */

typedef enum {
  GAPRO_OK = 0,
  GAPRO_ERR_BAD_ARG = -1,
  GAPRO_ERR_OOM = -2,
  GAPRO_ERR_HIP = -3,
  GAPRO_ERR_NOT_FINITE = -4,   /* a fit produced NaN/Inf */
  GAPRO_ERR_CHOLESKY = -5,     /* K_ZZ + jitter*I not positive definite */
  GAPRO_ERR_SPP_RANGE = -6,    /* superpoint id range exceeds the rank-table capacity */
  GAPRO_ERR_WORKSPACE = -7,    /* workspace too small */
  GAPRO_ERR_TIMEOUT = -8,      /* a fit spread over several workgroups gave up at a cluster barrier: a member was
                                * not resident within GAPRO_CLUSTER_BARRIER_TIMEOUT_MS (default 5000); per-fit
                                * status like GAPRO_ERR_CHOLESKY, the launch's other fits are unaffected.  The outputs
                                * (and the workspace) of a fit with this status are UNDEFINED: its members computed on
                                * unsynchronised data after the barrier gave up.  Transient by nature: callers retry the
                                * fit with the cluster kernel switched off (gapro_fit_options.reserved | 8), as
                                * gapro_amd.pipeline.Pipeline does */
  GAPRO_ERR_IO = -9,           /* gapro_pth_*: open / read / write failed, or a damaged file */
  GAPRO_ERR_UNSUPPORTED = -10  /* gapro_pth_*: a well-formed file this reader does not handle (compressed member,
                                * tensor storages, object / big-endian / Fortran arrays, ...): fall back to torch.load */
} gapro_status;

typedef struct gapro_ctx gapro_ctx;
typedef struct gapro_schedule gapro_schedule;

int gapro_version(void);
int gapro_ctx_create(int device, gapro_ctx** out);
void gapro_ctx_destroy(gapro_ctx* ctx);
const char* gapro_last_error(const gapro_ctx* ctx);

/* ------------------------------------------------------------------------------------------
 * Scene partition (device).  Replaces gen_ps_utils.py:312-326 and :347-363.
 * ---------------------------------------------------------------------------------------- */

/* Scene statistics, produced on the device by gapro_partition_prepare and copied to the host. */
typedef struct {
  double coord_min[3];   /* torch.min(coords_float, dim=0)        gen_ps_utils.py:317 */
  double coord_max[3];   /* torch.max(coords_float, dim=0)        gen_ps_utils.py:318 */
  int64_t spp_min;       /* smallest / largest superpoint id                          */
  int64_t spp_max;
  float feat_absmax;     /* max |mask_feats| (sets the fixed-point scale of the pooled sums) */
  int32_t fixed_shift;   /* pooled feature sums are exact int64 sums of rint(x * 2^fixed_shift) */
  int32_t n_spps;        /* len(torch.unique(spp))                gen_ps_utils.py:312-313 */
  int32_t status;        /* gapro_status raised on the device (e.g. GAPRO_ERR_SPP_RANGE) */
} gapro_scene_header;

/* Bytes of device workspace gapro_partition_prepare needs for `spp_range_cap` distinct id slots. */
size_t gapro_partition_prepare_workspace_bytes(int64_t n_points, int64_t spp_range_cap);

/* BLOCKING.  coords min/max, |feats| max, and the dense rank of every point's superpoint id
 * (the `return_inverse` output of torch.unique, gen_ps_utils.py:312).
 *   d_coords f64[N,3], d_feats f32[N,D], d_spp i64[N]  ->  d_spp_inv i32[N], *h_header        */
int gapro_partition_prepare(gapro_ctx* ctx, void* stream, int64_t n_points, int32_t feat_dim,
                            const double* d_coords, const float* d_feats, const int64_t* d_spp,
                            int64_t spp_range_cap, void* d_workspace, size_t workspace_bytes,
                            int32_t* d_spp_inv, gapro_scene_header* h_header);

/* Same, enqueue only: the header lands in `h_header_pinned` (page-locked host memory) once the stream
 * reaches that point; the caller synchronises and checks header->status itself.  Lets a batch of
 * scenes share one synchronisation. */
int gapro_partition_prepare_async(gapro_ctx* ctx, void* stream, int64_t n_points, int32_t feat_dim,
                                  const double* d_coords, const float* d_feats, const int64_t* d_spp,
                                  int64_t spp_range_cap, void* d_workspace, size_t workspace_bytes,
                                  int32_t* d_spp_inv, gapro_scene_header* h_header_pinned);

/* Fused point-in-box membership + superpoint pooling (gen_ps_utils.py:349-363) in one pass over
 * the points.  Box corners are the float64 `boxes` of gen_ps_utils.py:329-341 (float32-rounded
 * instance/wall corners held in float64, then the float64 floor box); the +-0.005 margin is
 * applied inside, in float64.
 *   in : d_coords f64[N,3], d_feats f32[N,D], d_spp_inv i32[N], d_boxes f64[B,6]
 *   tmp: d_feat_sum i64[S,D]  (zeroed by the call)
 *   out: d_occ_count i32[S,B], d_point_count i32[S], d_feats_spp f32[S,D],
 *        d_occ_bits u64[S, ceil(B/64)]  (bit b of row s = bb_occupancy_spp[s,b]),
 *        d_n_bbs i32[S]                 (n_bbs_per_spp, gen_ps_utils.py:363)
 * `thresh_spp_occu` is compared as float32, as torch does.             */
int gapro_partition_pool(gapro_ctx* ctx, void* stream, int64_t n_points, int32_t feat_dim,
                         int32_t n_boxes, int32_t n_spps, int32_t fixed_shift, float thresh_spp_occu,
                         const double* d_coords, const float* d_feats, const int32_t* d_spp_inv,
                         const double* d_boxes, int64_t* d_feat_sum, int32_t* d_occ_count,
                         int32_t* d_point_count, float* d_feats_spp, uint64_t* d_occ_bits,
                         int32_t* d_n_bbs);

/* Superpoint -> point broadcast of the three point-length outputs (gen_ps_utils.py:478-480). */
int gapro_broadcast_labels(gapro_ctx* ctx, void* stream, int64_t n_points, const int32_t* d_spp_inv,
                           const int32_t* d_sem_spp, const int32_t* d_inst_spp, const float* d_prob_spp,
                           int32_t* d_sem, int32_t* d_inst, float* d_prob);

/* ---- Batched forms: every scene of a batch in ONE launch per kernel (grid.y = scene). ----------
 * A 64-scene batch through the per-scene calls above is ~1200 launches of 3-5 us kernels; the batched
 * calls are ~15.  One gapro_scene_task per scene holds the device pointers of that scene; the three
 * calls read the fields of their stage (the caller fills n_spps / fixed_shift from the headers between
 * prepare and pool).  `h_tasks` is copied to `d_tasks` (device, n_scenes tasks) on the stream and must
 * stay valid until the stream has executed that copy. */
typedef struct {
  /* every stage */
  int64_t n_points;
  const double* coords;       /* f64[N,3] */
  const float* feats;         /* f32[N,D] */
  const int64_t* spp;         /* i64[N]   */
  int32_t* spp_inv;           /* i32[N]   out of prepare, in of pool / broadcast */
  /* prepare */
  void* prepare_ws;           /* >= gapro_partition_prepare_workspace_bytes(N, spp_range_cap) */
  int64_t spp_range_cap;
  /* pool */
  const double* boxes;        /* f64[B,6] */
  int32_t n_boxes, n_spps, fixed_shift;
  float thresh_spp_occu;
  int64_t* feat_sum;          /* i64[S,D] tmp */
  int32_t* occ_count;         /* i32[S,B] */
  int32_t* point_count;       /* i32[S]   */
  float* feats_spp;           /* f32[S,D] */
  uint64_t* occ_bits;         /* u64[S,ceil(B/64)] */
  int32_t* n_bbs;             /* i32[S]   */
  /* broadcast */
  const int32_t* sem_spp;     /* i32[S] */
  const int32_t* inst_spp;    /* i32[S] */
  const float* prob_spp;      /* f32[S] */
  int32_t* sem;               /* i32[N] */
  int32_t* inst;              /* i32[N] */
  float* prob;                /* f32[N] */
} gapro_scene_task;

/* gapro_partition_prepare_async for n_scenes scenes; the headers land in h_headers_pinned[n_scenes]
 * (page-locked) through d_headers[n_scenes] (device) once the stream reaches that point. */
int gapro_partition_prepare_batch(gapro_ctx* ctx, void* stream, int32_t n_scenes, int32_t feat_dim,
                                  const gapro_scene_task* h_tasks, gapro_scene_task* d_tasks,
                                  gapro_scene_header* d_headers, gapro_scene_header* h_headers_pinned);
/* gapro_partition_pool for n_scenes scenes (zeroes the tallies itself). */
int gapro_partition_pool_batch(gapro_ctx* ctx, void* stream, int32_t n_scenes, int32_t feat_dim,
                               const gapro_scene_task* h_tasks, gapro_scene_task* d_tasks);
/* gapro_broadcast_labels for n_scenes scenes. */
int gapro_broadcast_labels_batch(gapro_ctx* ctx, void* stream, int32_t n_scenes,
                                 const gapro_scene_task* h_tasks, gapro_scene_task* d_tasks);

/* ------------------------------------------------------------------------------------------
 * Label-side kernels on either end of the path (not needed by a caller of
 * gen_pseudo_label_gaussian_process itself).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t instance_num;  /* int(instance_label.max()) + 1            gen_ps_utils.py:200 */
  int32_t n_boxes;       /* non-empty instance ids = rows of the outputs              */
  int32_t status;        /* GAPRO_ERR_BAD_ARG: an id >= max_instances was met         */
  int32_t reserved;
} gapro_instance_header;

size_t gapro_instance_info_workspace_bytes(int32_t max_instances);
/* getInstanceInfo (gen_ps_utils.py:195-239) in one pass over the points: per non-empty GT instance id, in
 * ascending id order, the axis-aligned box [min xyz | max xyz] (f64), the class = semantic label of the
 * instance's first point (minus 2 unless -100 when scannet_class_shift != 0, :236-237) and the volume
 * prod(clip(max - min, 0)) (f64).  Labels come as the float64 arrays the ScanNet .pth files hold.
 *   in : d_coords f64[N,3], d_instance_label f64[N], d_semantic_label f64[N]
 *   out: d_box f64[<=max_instances,6], d_cls f64[..], d_volume f64[..], d_corners f32[N,6] or NULL
 *        (corners_label, :203,219-220), header (device + page-locked host copy, enqueue only)     */
int gapro_instance_info(gapro_ctx* ctx, void* stream, int64_t n_points, const double* d_coords,
                        const double* d_instance_label, const double* d_semantic_label, int32_t max_instances,
                        int32_t scannet_class_shift, void* d_workspace, size_t workspace_bytes, double* d_box,
                        double* d_cls, double* d_volume, float* d_corners, gapro_instance_header* d_header,
                        gapro_instance_header* h_header_pinned);

typedef struct {
  int32_t n_gt;          /* instance_label.max() + 1                  eval_ps_labels.py:101 */
  int32_t n_ps;          /* ps_instance_label.max() + 1               :109 */
  int32_t status;
  int32_t reserved;
} gapro_eval_header;

size_t gapro_eval_workspace_bytes(int32_t max_gt, int32_t max_ps);
/* get_miou_scene (eval_ps_labels.py:100-147, cal_iou :35-42): for every GT instance id g < n_gt the largest
 * IoU = inter / (|gt| + |ps| - inter + 1e-4) (float32, the reference's operation order) over the pseudo
 * instances whose class (label of their first point) equals the GT instance's, and that class (-1 for an
 * empty id; the caller keeps the rows with class >= 0, :139).  Labels are int64, as the reference passes.
 *   out: d_max_iou f32[max_gt], d_gt_cls f32[max_gt] (first n_gt entries), header */
int gapro_eval_miou(gapro_ctx* ctx, void* stream, int64_t n_points, const int64_t* d_semantic_label,
                    const int64_t* d_instance_label, const int64_t* d_ps_semantic_label,
                    const int64_t* d_ps_instance_label, int32_t max_gt, int32_t max_ps, void* d_workspace,
                    size_t workspace_bytes, float* d_max_iou, float* d_gt_cls, gapro_eval_header* d_header,
                    gapro_eval_header* h_header_pinned);
/* get_scene_sem_conf (eval_ps_labels.py:150-172): conf i64[C,C], rows = GT class, columns = pseudo class,
 * over the points with GT != -100; a pseudo label of -100 counts as a wrong class. */
int gapro_eval_sem_confusion(gapro_ctx* ctx, void* stream, int64_t n_points, const int64_t* d_semantic_label,
                             const int64_t* d_ps_semantic_label, int32_t num_classes, int64_t* d_conf);

/* Heuristic labelers: gen_pseudo_label (gen_ps_utils.py:485-569; rule 0 = "volume",
 * 1 = "dist", 2 = "none") and gen_pseudo_label_box2mask (:242-290; rule 3).  Membership in the INSTANCE boxes
 * (float32 box, 0.005 margin applied in float32, compared in float64), the rule for points inside several boxes,
 * then (align != 0, the scannetv2 branch) the superpoint vote spp_align_label (:99-123) with the >= 0.7
 * occupancy mask (not for box2mask).  "dist" reproduces the reference's indexing of the coordinate array by the
 * rank among the multi-box points (:525).  d_spp_inv = dense superpoint ranks from gapro_partition_prepare.
 *   out: d_sem i32[N] (class, instance_classes for background, -100), d_inst i32[N] (box index or -100) */
size_t gapro_label_heuristic_workspace_bytes(int64_t n_points, int32_t n_spps, int32_t n_boxes);
int gapro_label_heuristic(gapro_ctx* ctx, void* stream, int64_t n_points, const double* d_coords,
                          const int32_t* d_spp_inv, int32_t n_spps, int32_t n_boxes, const float* d_box,
                          const float* d_volume, const int64_t* d_cls, int32_t rule, int32_t align,
                          int32_t instance_classes, void* d_workspace, size_t workspace_bytes, int32_t* d_sem,
                          int32_t* d_inst);

/* ------------------------------------------------------------------------------------------
 * Consumer-side label ops: what the training code does with the generated labels.
 * Forward values and the gradients w.r.t. the network outputs come out of the same call; reductions are
 * float64 sums of float32 terms.  `grad_out` scales the gradients (pass 1 and multiply later to stay async).
 * ---------------------------------------------------------------------------------------- */
/* custom_scatter_mean (ISBNet/isbnet/model/model_utils.py:600-613) of the three label channels at once
 * (isbnet.py:387-389): out[s] = mean of the points with index s (count clamped at 1, as torch_scatter does).
 *   d_index i64[N] in [0, n_out), d_sums_ws f64[3 n_out] and d_counts_ws i32[n_out] are scratch. */
int gapro_label_pool_mean(gapro_ctx* ctx, void* stream, int64_t n_points, int32_t n_out, const int64_t* d_index,
                          const float* d_prob, const float* d_mu, const float* d_var, double* d_sums_ws,
                          int32_t* d_counts_ws, float* d_out_prob, float* d_out_mu, float* d_out_var);
/* Probability-weighted BCE with logits (ISBNet/isbnet/model/criterion.py:287-288):
 *   loss = sum_{g,p} bce(x[g][p], y[g][p]) w[p] / sum_p w[p] / (G + 1e-6),  x, y f32[G,P] row-major, w f32[P];
 *   d_grad_logits f32[G,P] or NULL; d_acc2 f64[2] scratch. */
int gapro_weighted_bce_with_logits(gapro_ctx* ctx, void* stream, int32_t n_rows, int64_t n_cols, const float* d_logits,
                                   const float* d_targets, const float* d_weights, float grad_out, double* d_acc2,
                                   float* d_loss, float* d_grad_logits);
/* KL-to-GP auxiliary loss (ISBNet/isbnet/model/criterion.py:435-463): labels of -100 are ignored; GP variances
 * <= epsilon use the (exp(logvar) - 1)^2 + (mu - mu_l)^2 branch, the others the Gaussian KL expression; each
 * branch is averaged over its own count (+1e-4) and scaled by `weight`.  Gradients w.r.t. mu_pred / logvar_pred
 * (f32[n], or both NULL); d_acc4 f64[4] scratch. */
int gapro_kl_gp_loss(gapro_ctx* ctx, void* stream, int64_t n, const float* d_mu_labels, const float* d_var_labels,
                     const float* d_mu_pred, const float* d_logvar_pred, float epsilon, float weight, float grad_out,
                     double* d_acc4, float* d_loss, float* d_grad_mu, float* d_grad_logvar);

/* ------------------------------------------------------------------------------------------
 * Static pair schedule and merge (host).  Replaces the control flow of gen_ps_utils.py:365-476.
 * Which pairs are fitted and on which superpoints depends only on (boxes, bb_occupancy_spp),
 * never on GP outputs, so the whole schedule is enumerated before any fit runs.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t n_events;       /* containment verdicts + GP fits, in reference order */
  int32_t n_fits;
  int64_t n_event_idx;    /* total length of all intersect index lists */
  int64_t n_fit_idx;      /* total length of all [b1_inds | b2_inds | intersect_inds] lists */
  int64_t n_fit_out;      /* total number of test superpoints over all fits */
  int32_t max_m;          /* largest m1+m2 of any fit */
  int32_t max_t;          /* largest |intersect_inds| of any fit */
} gapro_schedule_counts;

/* One GP fit of the batch.  Index lists live in one int32 array laid out
 * [b1_inds (m1) | b2_inds (m2) | intersect_inds (t)] at idx_offset; values are ROWS of the
 * feats_spp array handed to gapro_svgp_fit_batch (superpoint index + feats_row_base). */
typedef struct {
  int32_t m1, m2, t;
  int32_t b1, b2;        /* the two boxes (informational) */
  int32_t scene;         /* caller tag (informational) */
  int32_t slot;          /* row of d_fit_status / d_fit_loss this fit reports to (set by the library) */
  int32_t reserved;
  int64_t idx_offset;    /* into the index array */
  int64_t out_offset;    /* into the per-test-superpoint outputs */
  int64_t ws_offset;     /* into the workspace, in doubles (filled by gapro_fit_plan_workspace) */
} gapro_fit_desc;

/* h_boxes f64[B,6], h_occ_bits u64[S,ceil(B/64)], h_n_bbs i32[S] (outputs of gapro_partition_pool). */
int gapro_schedule_build(int32_t n_spps, int32_t n_boxes, const double* h_boxes,
                         const uint64_t* h_occ_bits, const int32_t* h_n_bbs, gapro_schedule** out);
void gapro_schedule_free(gapro_schedule* s);
int gapro_schedule_get_counts(const gapro_schedule* s, gapro_schedule_counts* out);
/* Export the fits: h_descs[n_fits], h_idx i32[n_fit_idx].  `feats_row_base` is added to every
 * index, `idx_base`/`out_base` to the offsets, `scene` is copied into the descs (for batching
 * several scenes into one gapro_svgp_fit_batch call). */
int gapro_schedule_export_fits(const gapro_schedule* s, int32_t feats_row_base, int64_t idx_base,
                               int64_t out_base, int32_t scene, gapro_fit_desc* h_descs, int32_t* h_idx);
/* Export the events for inspection/tests: kind (0 contain, 1 fit), b1, b2, winner (contain) or
 * fit id (fit), offsets[n_events+1] into h_event_idx (superpoint indices of the intersection). */
int gapro_schedule_export_events(const gapro_schedule* s, uint8_t* h_kind, int32_t* h_b1, int32_t* h_b2,
                                 int32_t* h_aux, int64_t* h_offsets, int32_t* h_event_idx);

/* Merge + fallback + label tables (gen_ps_utils.py:365-383, :411-423, :438-476).
 * Fit outputs are this schedule's slices (out_offset relative to 0): probs_new f32, labels u8,
 * mu f32, var f32, each [n_fit_out].
 *   in : h_boxes_cls i64[B], h_boxes_volume f64[B], n_fg_instances, instance_classes
 *   out: h_sem_spp i32[S], h_inst_spp i32[S], h_prob_spp f32[S], h_mu_spp f32[S], h_var_spp f32[S] */
int gapro_schedule_merge(const gapro_schedule* s, const float* h_probs_new, const uint8_t* h_labels,
                         const float* h_mu, const float* h_var, const int64_t* h_boxes_cls,
                         const double* h_boxes_volume, int32_t n_fg_instances, int32_t instance_classes,
                         int32_t* h_sem_spp, int32_t* h_inst_spp, float* h_prob_spp, float* h_mu_spp,
                         float* h_var_spp);

/* ------------------------------------------------------------------------------------------
 * Batched variational-GP fit (device).  Replaces gaussian_process_utils.py:382-445 and the
 * gpytorch objects it builds (GPClassificationModel :11-25, BernoulliLikelihood, VariationalELBO,
 * Adam lr 0.1, 50 steps).  One workgroup trains one fit for all `training_iter` steps inside a
 * single launch; every fit of every scene in the batch runs concurrently.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t training_iter;     /* 50   gaussian_process_utils.py:382,416 */
  double lr;                 /* 0.1  gaussian_process_utils.py:410 */
  double jitter;             /* 1e-4 gpytorch variational_cholesky_jitter (float32 default) */
  double min_variance;       /* 1e-6 gpytorch settings.min_variance */
  int32_t eval_stale_chol;   /* 0 = refactor K_ZZ with the trained parameters for prediction (default);
                                1 = reuse the factor of the last training step */
  int32_t reserved;          /* 0; debug bits: 1 = never route a fit to the strip-streaming kernels,
                              * 2 = launch the fit kernels on the caller's stream (not the fit streams),
                              * 4 = no small-fit kernel (M_p <= 64 runs the 512-thread strip kernel),
                              * 8 = no cluster kernel (large fits stay on one workgroup),
                              * 16 = the cluster kernel for every fit it can take (M_p >= 64, M_p % 32 == 0),
                              * 32 = 64 x 64 wave tiles in the cluster kernel, 64 = the full-register build for every
                              * staged launch, 128 = the staged fits as ONE launch (not split by their LDS need),
                              * 256 = cluster barriers always with the L2 write-back, 512 = plain longest-first order
                              * (no serpentine over the XCDs), 8192 = workgroup-tiled products through LDS in the staged
                              * kernel at every M_p > 128 (default: M_p = 256, 384 only; bit-identical either way),
                              * 16384 = with 8192: in the two-per-CU build only, 131072 = never (round 2's products),
                              * 32768 = TEST: the last member of every cluster never arrives (cluster barrier timeout),
                              * 262144 = workgroup b of a fit kernel runs fit b of its list (default: the workgroups
                              * take the fits in the order in which they start, claim_fit in csrc/svgp_fit.hip),
                              * 524288 = the two-per-CU staged launch is not held back behind the cluster kernel,
                              * 1048576 = no wave-per-fit kernel (M_p <= 48 runs the small-fit strip kernel)
                              * -- A/B switches of tools/bench_fit.py / fit_timeline.py */
  int32_t psd_retries;       /* 3    gpytorch settings.cholesky_max_tries: a factorisation that meets a non-positive
                              *      pivot is repeated on K + psd_jitter 10^i I, i < psd_retries (psd_safe_cholesky,
                              *      reached from gaussian_process_utils.py:417); 0 = fail at once */
  int32_t precision;         /* GAPRO_PRECISION_*: arithmetic of the fit (0 = float64 throughout, the default) */
  double psd_jitter;         /* 1e-8 gpytorch settings.cholesky_jitter for float64 (K_ZZ is factored in double) */
} gapro_fit_options;

/* gapro_fit_options.precision */
enum {
  GAPRO_PRECISION_F64 = 0,   /* everything in float64 (a superset of the reference's split) */
  GAPRO_PRECISION_MIXED = 1  /* the reference's own split: parameters, kernel matrices, A, B, variances and their
                              * gradients in float32 (v_mfma_f32), float64 for the Cholesky factor, the L^-1
                              * products and their backward (gpytorch _cholesky_factor / torch autograd).
                              * Implemented by the cluster kernel (route 4); fits routed to the other kernels run
                              * in float64, a superset of the split */
};

void gapro_fit_options_default(gapro_fit_options* opt);

/* Workspace doubles one fit with m = m1+m2 inducing points, t test points, feat_dim d needs. */
int64_t gapro_fit_workspace_doubles(int32_t m, int32_t t, int32_t feat_dim);
/* Fill ws_offset of every desc; returns the total workspace size in BYTES. */
int64_t gapro_fit_plan_workspace(gapro_fit_desc* h_descs, int32_t n_fits, int32_t feat_dim);

/* d_feats_spp f32[rows,D]; d_idx i32; h_descs gapro_fit_desc[n_fits] on the HOST (the library orders
 * the launch longest-fit-first and picks the kernel variant per fit) and d_descs, a device buffer of
 * n_fits descriptors the library fills (the call synchronises the stream once for that copy, then
 * only enqueues);
 * d_init_mean f64 (optional, may be NULL = zeros): initial variational mean of fit i at
 *   d_init_mean[idx_offset ... + m] (gpytorch adds 1e-3*randn here; zeros make runs reproducible);
 * outputs per test superpoint at out_offset: d_probs f32, d_probs_new f32, d_labels u8,
 *   d_mu f32, d_var f32  (pred_probs, pred_probs_new, pred_labels, pred_mu, pred_variance);
 * d_fit_status i32[n_fits] (gapro_status per fit, in h_descs order), d_fit_loss f64[n_fits] (last
 *   ELBO loss, in h_descs order). */
int gapro_svgp_fit_batch(gapro_ctx* ctx, void* stream, int32_t n_fits, int32_t feat_dim,
                         const float* d_feats_spp, const int32_t* d_idx, const gapro_fit_desc* h_descs,
                         gapro_fit_desc* d_descs, const double* d_init_mean, const gapro_fit_options* opt,
                         double* d_workspace,
                         size_t workspace_bytes, float* d_probs, float* d_probs_new, uint8_t* d_labels,
                         float* d_mu, float* d_var, int32_t* d_fit_status, double* d_fit_loss);
/* The same launch with one more optional per-fit output (round 6): d_fit_cond[n_fits] (double, device; NULL = not wanted)
 * receives a conditioning figure of each fit's LAST Cholesky factorisation, (max_j L_jj / min_j L_jj)^2 over the fit's M
 * rows -- a lower bound of cond_2(K_ZZ + jitter I), read from the diagonal-block inverses the kernels keep anyway.  A
 * diagnostic, NOT a predictor of which fits are numerically soft: on the S3DIS-shaped test scene the two fits whose
 * sigma^2 no float64 implementation reproduces to 1e-4 rank 38th and 55th of 66 by this figure (and 35th / 61st by the
 * true cond_2 at the initial hyper-parameters); what identifies them is a perturbation probe -- the fits once more with
 * the jitter on K_ZZ's diagonal scaled by (1 + 1e-11) (Pipeline.reproducibility_probe).  Status, outputs and arithmetic
 * are those of gapro_svgp_fit_batch. */
int gapro_svgp_fit_batch_ex(gapro_ctx* ctx, void* stream, int32_t n_fits, int32_t feat_dim,
                            const float* d_feats_spp, const int32_t* d_idx, const gapro_fit_desc* h_descs,
                            gapro_fit_desc* d_descs, const double* d_init_mean, const gapro_fit_options* opt,
                            double* d_workspace,
                            size_t workspace_bytes, float* d_probs, float* d_probs_new, uint8_t* d_labels,
                            float* d_mu, float* d_var, int32_t* d_fit_status, double* d_fit_loss, double* d_fit_cond);

/* Which kernel gapro_svgp_fit_batch routes a fit of m = m1 + m2 inducing points to: 0 = strip-streaming
 * kernel (64 < M_p <= 128), 1 = LDS-staged kernel (128 < M_p < 512 while Z and X fit the LDS: M_p <= 192 at
 * feat_dim 32), 2 = generic kernel (feat_dim > 32), 3 = the small-fit strip kernel (M_p <= 64: 256 threads per fit,
 * two fits per CU), 4 = the cluster kernel (M_p >= 512: one fit spread over 4..32 workgroups with cluster barriers;
 * also, on one workgroup, every fit that fits neither LDS kernel), 5 = the wave-per-fit kernel (M_p <= 48 at
 * feat_dim 6: one wavefront trains one fit, eight / four fits per CU, nothing leaves the CU between the first and the
 * last Adam step).  M_p = m padded to the MFMA tile. */
int gapro_fit_route(int32_t m, int32_t feat_dim);
/* Padded size M_p of a fit's M x M matrices (a multiple of 16; of 32 where the kernel that takes the fit needs it):
 * a function of (M, D) only, never of the routing options.  The workspace layout is built on it. */
int gapro_fit_padded_m(int32_t m, int32_t feat_dim);

/* Optional device-side timing of one fit launch (bench.py's roofline figure).  gapro_svgp_fit_batch runs
 * its kernels on streams the context owns (cluster, staged, strip and small-fit strip kernel side by side), so
 * events on the caller's stream do not bracket them.  An armed timing object makes the NEXT gapro_svgp_fit_batch
 * record HIP events around each kernel on the stream it is launched on.  gapro_fit_timing_read blocks until
 * that launch has finished: out_ms5 = {staged (+ generic) kernel ms, strip kernel ms, first start -> last end
 * ms, small-fit strip kernel ms, cluster kernel ms}; a kernel that was not launched reads 0. */
typedef struct gapro_fit_timing gapro_fit_timing;
int gapro_fit_timing_create(gapro_ctx* ctx, gapro_fit_timing** out);
void gapro_fit_timing_destroy(gapro_fit_timing* t);
int gapro_fit_timing_arm(gapro_ctx* ctx, gapro_fit_timing* t);
int gapro_fit_timing_read(gapro_ctx* ctx, gapro_fit_timing* t, float* out_ms5);
/* Milliseconds of the wave-per-fit kernels of the timed launch (route 5; first start -> last end of its up to three
 * kernels; 0 when the launch had none).  They are part of gapro_fit_timing_read's span.  Blocks like it. */
int gapro_fit_timing_read_wave(gapro_ctx* ctx, gapro_fit_timing* t, float* out_ms);
/* Diagnostics of the cluster kernel of the timed launch (blocks until it has finished; read it before 64 further
 * launches of this context): out3 = {clusters of more than one workgroup, those whose members did NOT all run on one
 * XCD (their barriers carry the L2 write-back), member workgroups}. */
int gapro_fit_timing_cluster_info(gapro_ctx* ctx, gapro_fit_timing* t, int32_t* out3);
/* Where launch t lies on the time axis of launch ref: out_ms2 = {first kernel start, last kernel end} of t in ms after
 * the start event of ref's first kernel.  Consecutive launches of a pipeline overlap (the next one is enqueued while
 * the tail of the previous one runs, and a start event fires when its stream reaches it, not when the kernel gets
 * CUs): bench.py counts the overlapped time once when it averages launch durations.  Blocks until t has finished. */
int gapro_fit_timing_offsets(gapro_ctx* ctx, gapro_fit_timing* ref, gapro_fit_timing* t, float* out_ms2);

/* ------------------------------------------------------------------------------------------
 * Scene / label files (host; no device, no ctx, no Python objects -- callable from any thread without the GIL).
 * Replaces the `torch.load` of gen_ps.py:45-46 (scene tuple written by ISBNet/dataset/scannetv2/prepare_data_inst.py:104,
 * superpoint ids written by prepare_superpoint.py:27) and the `torch.save` of gen_ps.py:132.
 *
 * A torch.save()d NumPy array / tuple of NumPy arrays is a stored zip whose data.pkl (pickle protocol 2) holds every
 * array buffer as latin-1 text re-encoded as UTF-8; unpickling it costs ~25 ms of a core per ScanNet scene under the GIL.
 * gapro_pth_open maps the file and walks the pickle; gapro_pth_read transcodes one array's payload back to bytes
 * straight into the caller's buffer (e.g. pinned staging memory).  GAPRO_ERR_UNSUPPORTED = well-formed but not handled
 * here: the caller falls back to torch.load.  Errors: gapro_pth_last_error() (thread-local text).
 * ---------------------------------------------------------------------------------------- */
typedef struct gapro_pth_file gapro_pth_file;
typedef struct gapro_pth_array {
  int32_t kind;      /* NumPy dtype kind character: 'f', 'i', 'u' or 'b' (bool) */
  int32_t itemsize;  /* bytes per element: 1, 2, 4 or 8 (little-endian) */
  int32_t ndim;      /* 0..4 */
  int32_t encoded;   /* read: 1 = stored as UTF-8 text (protocol 2), 0 = raw bytes (protocol >= 3); ignored on write */
  int64_t shape[4];  /* C order; entries beyond ndim are 1 */
  int64_t nbytes;    /* prod(shape) * itemsize */
} gapro_pth_array;
int gapro_pth_open(const char* path, gapro_pth_file** out);
/* number of arrays (1 for a bare array), and whether the top-level object is a tuple / list */
int gapro_pth_count(const gapro_pth_file* f);
int gapro_pth_is_sequence(const gapro_pth_file* f);
int gapro_pth_info(const gapro_pth_file* f, int32_t index, gapro_pth_array* out);
/* decode array `index` into h_dst; dst_bytes must equal its nbytes */
int gapro_pth_read(const gapro_pth_file* f, int32_t index, void* h_dst, int64_t dst_bytes);
void gapro_pth_close(gapro_pth_file* f);
/* Write n_arrays host arrays as one torch.load()-able file (tuple when as_tuple, else the single bare array):
 * stored zip, protocol-2 pickle naming numpy.core.multiarray (importable by NumPy 1.x and 2.x); written to a
 * temporary name in the same directory and renamed (atomic). */
int gapro_pth_write(const char* path, int32_t n_arrays, const gapro_pth_array* descs, const void* const* h_data,
                    int32_t as_tuple);
const char* gapro_pth_last_error(void);
/* which UTF-8 -> byte transcoder this process uses: "avx512" (VBMI2), "bmi2" or "scalar" (GAPRO_PTH_DECODER pins one;
 * a tier the CPU lacks falls back to the next) */
const char* gapro_pth_decoder(void);
/* the writer's tiers: byte -> UTF-8 transcoder "avx512" / "bmi2" / "scalar" (GAPRO_PTH_ENCODER pins one) and zip CRC-32
 * "clmul" (PCLMULQDQ folding) / "table" (slice-by-8; GAPRO_PTH_CRC pins one) */
const char* gapro_pth_encoder(void);
const char* gapro_pth_crc(void);
/* the two writer primitives on their own (tests hold every tier to the scalar one): CRC-32 of n bytes; latin-1 -> UTF-8
 * of n bytes into dst (dst_cap >= 2 n + 64), returns the encoded length or a negative gapro_status */
uint32_t gapro_pth_crc32(const void* data, int64_t n);
int64_t gapro_pth_encode_latin1(const void* src, int64_t n, void* dst, int64_t dst_cap);
/* The reference's default features (gen_ps.py:55: np.concatenate([xyz, rgb], -1) of the UN-aligned coordinates, uploaded
 * as float32 at :84): h_feats[n][6] = float32 of [xyz | rgb], one pass on the host. */
int gapro_scene_default_feats(const double* h_xyz, const double* h_rgb, int64_t n_points, float* h_feats);
/* getInstanceInfo (gen_ps_utils.py:195-239) on the HOST in one pass, without the corner labels: boxes
 * h_box[n_boxes][6] = (min xyz, max xyz), class h_cls (ScanNet: -2 unless -100), volume h_vol, indexed by the rank among
 * the non-empty instance ids; *instance_num = max id + 1.  cap = rows the output arrays hold; GAPRO_ERR_BAD_ARG with
 * *instance_num set when it is too small.  For loader threads: the device form (gapro_instance_info) is a kernel and
 * would wait for a running fit launch to drain. */
int gapro_scene_instance_boxes(const double* h_xyz, const double* h_inst, const double* h_sem, int64_t n_points,
                               int32_t scannet, int32_t cap, double* h_box, double* h_cls, double* h_vol,
                               int32_t* n_boxes, int32_t* instance_num);

/* ------------------------------------------------------------------------------------------
 * Batch feeder of the gen_ps driver (host threads of the library's own; no Python objects, no GIL; csrc/feeder.hip).
 * Replaces the per-scene host half of gen_ps.py:36-132: torch.load of the scene tuple and the superpoint ids (:45-46),
 * the default features from the UN-aligned points (:55), the axis alignment (:58-69), getInstanceInfo (:71-77), the
 * upload (:83-87), and on the way out the device -> host copies and torch.save of the 5-tuple (:126-132).
 *
 *   submit    scene / superpoint / alignment (/ feature) file paths, in the order the scenes are wanted
 *   poll      wait until the next scenes of that order are loaded into pinned memory; how many, and the device bytes
 *   upload    one asynchronous copy per scene into the caller's device slab on the feed's own stream; records out
 *   batch_wait  make a stream wait for a batch's copies
 *   export    label files: device -> host behind the caller's event, gapro_pth_write; export_wait collects them
 *
 * device < 0: host-only mode (gen_ps --dry_run): pageable memory, no copies; `upload` hands out host images and
 * `export` takes host pointers.  Scenes the native reader does not handle come back with status GAPRO_ERR_UNSUPPORTED
 * (the caller reads them its own way); a scene without instances has n_instances == 0.
 * ---------------------------------------------------------------------------------------- */
typedef struct gapro_feed gapro_feed;
typedef struct gapro_feed_scene {
  int32_t status;        /* gapro_status of the load */
  int32_t n_points;      /* N */
  int32_t n_instances;   /* instance boxes found (getInstanceInfo's non-empty ids) */
  int32_t feat_dim;      /* D: 6 (xyz + rgb) or the width of the feature file */
  /* byte offsets of the scene's arrays in the device slab of its upload (256-byte aligned):
   * coords f64[N,3] (aligned), feats f32[N,D], spp i64[N], sem f64[N], inst f64[N] (the file's label arrays) */
  int64_t off_coords, off_feats, off_spp, off_sem, off_inst;
  const double* inst_box;  /* f64[n_instances,6] (min xyz, max xyz), inst_cls f64[n_instances], inst_vol f64[n_instances]: */
  const double* inst_cls;  /* host memory of the feed, valid until the next gapro_feed_upload                            */
  const double* inst_vol;
  void* host_image;      /* host-only mode: base address the offsets are relative to (valid until release_batch) */
} gapro_feed_scene;
typedef struct gapro_feed_out {
  const void* d_sem;     /* i32[n_points] */
  const void* d_inst;    /* i32[n_points] */
  const void* d_prob;    /* f32[n_points] */
  const void* d_mu;      /* f32[n_mu] */
  const void* d_var;     /* f32[n_mu] */
  int64_t n_points, n_mu;
  const char* path;      /* label file to write (atomically) */
} gapro_feed_out;
/* n_threads loader / writer threads; budget_bytes caps the pinned memory of scenes loaded but not yet uploaded (plus
 * label files being written); default_feat_dim = 6 */
int gapro_feed_create(int32_t device, int32_t n_threads, int64_t budget_bytes, int32_t default_feat_dim,
                      gapro_feed** out);
void gapro_feed_destroy(gapro_feed* f);
/* Stop the threads and leave everything else to the end of the process: unpinning the staging pool of a worker takes
 * ~0.6 s (8 GB), which a process that is about to exit need not spend.  The handle must not be used afterwards. */
void gapro_feed_detach(gapro_feed* f);
const char* gapro_feed_last_error(const gapro_feed* f);
/* feat_paths may be NULL (default features), and so may its entries */
int gapro_feed_submit(gapro_feed* f, int32_t n, const char* const* scene_paths, const char* const* spp_paths,
                      const char* const* align_paths, const char* const* feat_paths);
/* no further submit: polls stop waiting for scenes that will never come */
int gapro_feed_close(gapro_feed* f);
/* Block until min_ready scenes at the head of the order are loaded (fewer when the feed is closed and runs out, or
 * when the byte budget holds fewer: the call then returns what IS loaded as soon as no loader can make progress
 * without the caller taking it -- it never waits for a count the budget cannot hold), or timeout_ms passed (< 0: no
 * limit); *n_ready = loaded scenes in a row from the head (<= max_scenes), *slab_bytes the device bytes their images
 * need. */
int gapro_feed_poll(gapro_feed* f, int32_t min_ready, int32_t max_scenes, int32_t timeout_ms, int32_t* n_ready,
                    int64_t* slab_bytes);
/* The next n loaded scenes: copies into d_slab (slab_bytes >= what poll reported for them), out[n], *batch_id.  The
 * copies run on the feed's own stream behind everything queued so far on slab_stream (hipStream_t; NULL = the default
 * stream): the stream the slab was allocated on when it comes from a stream-ordered / caching allocator.  Sizes and
 * states are validated before anything changes; after a HIP failure (GAPRO_ERR_HIP) the feed refuses further calls. */
int gapro_feed_upload(gapro_feed* f, int32_t n, void* d_slab, int64_t slab_bytes, void* slab_stream, gapro_feed_scene* out,
                      int64_t* batch_id);
int gapro_feed_batch_wait(gapro_feed* f, int64_t batch_id, void* stream);
/* host-only mode: the images of a batch are no longer needed (device mode: recycles completed batches) */
int gapro_feed_release_batch(gapro_feed* f, int64_t batch_id);
/* Queue n label files.  ready_event (hipEvent_t or NULL): recorded by the caller behind the kernels that produce the
 * arrays; they must stay valid until gapro_feed_export_wait has counted the scene. */
int gapro_feed_export(gapro_feed* f, int32_t n, const gapro_feed_out* items, void* ready_event);
/* wait until the first `until_done` label files in submission order (< 0: all queued so far) are written or failed;
 * *n_done = files finished as a CONTIGUOUS PREFIX of the submission order: the arrays of export k may be released
 * when n_done > k, whatever later exports have finished already */
int gapro_feed_export_wait(gapro_feed* f, int64_t until_done, int32_t timeout_ms, int64_t* n_done, int64_t* n_failed);
int gapro_feed_export_error(gapro_feed* f, int32_t index, char* buf, int32_t cap);
/* where the feeder threads' time went, summed over threads: out8 = {s inside the staging allocations, blocks, bytes,
 * s inside scene loads, scenes, s inside label writes, files, s from create to the first loaded scene} */
int gapro_feed_stats(gapro_feed* f, double* out8);

/* ------------------------------------------------------------------------------------------
 * Device memory, streams, events owned by the library (round 6; csrc/devmem.hip).  SURVEY.md 8b: "library owns an
 * opaque ctx: device, stream, workspace arena".  A caller without torch (the gen_ps workers: gapro_amd/devmem.py) gets
 * everything the path needs from here; the torch-tensor API shims keep using torch's allocator and streams.
 * The reference has no counterpart: its tensors come from torch (gen_ps.py:79-89).
 * ---------------------------------------------------------------------------------------- */
/* Caching arena with stream-ordered reuse: a block is allocated FOR a stream (hipStream_t; NULL = the default stream)
 * and, once freed, is only handed out again for that stream, so gapro_dev_free never waits for the device.  A block
 * used on a second stream must be ordered by events before it is freed. */
int gapro_dev_alloc(gapro_ctx* ctx, size_t bytes, void* stream, void** out);
int gapro_dev_free(gapro_ctx* ctx, void* p);
int gapro_dev_trim(gapro_ctx* ctx);   /* give every cached block back to the driver (synchronises the device) */
int gapro_dev_stats(gapro_ctx* ctx, int64_t* reserved_bytes, int64_t* in_use_bytes, int64_t* device_free_bytes,
                    int64_t* device_total_bytes);   /* any pointer may be NULL */
int gapro_host_alloc(gapro_ctx* ctx, size_t bytes, void** out);   /* page-locked host memory */
int gapro_host_free(gapro_ctx* ctx, void* p);
int gapro_stream_create(gapro_ctx* ctx, void** out);              /* non-blocking stream on the context's device */
int gapro_stream_destroy(gapro_ctx* ctx, void* stream);
int gapro_stream_sync(gapro_ctx* ctx, void* stream);
int gapro_device_sync(gapro_ctx* ctx);
int gapro_event_create(gapro_ctx* ctx, int32_t timing, void** out);
int gapro_event_destroy(gapro_ctx* ctx, void* ev);
int gapro_event_record(gapro_ctx* ctx, void* ev, void* stream);
int gapro_stream_wait_event(gapro_ctx* ctx, void* stream, void* ev);
int gapro_event_sync(gapro_ctx* ctx, void* ev);
int gapro_event_query(gapro_ctx* ctx, void* ev);                  /* 1 complete, 0 not yet, < 0 gapro_status */
int gapro_event_elapsed_ms(gapro_ctx* ctx, void* ev_start, void* ev_end, float* out_ms);
/* kind: 0 host -> device, 1 device -> host, 2 device -> device */
int gapro_memcpy_async(gapro_ctx* ctx, void* dst, const void* src, size_t bytes, int32_t kind, void* stream);
int gapro_memset_async(gapro_ctx* ctx, void* dst, int32_t value, size_t bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Inspection (tests): where a fit's trained parameters live in its workspace.  The measurement / self-test entry points
 * (gapro_debug_*) are NOT part of this library: include/gapro_hip_debug.h, libgapro_hip_debug.so.
 * ---------------------------------------------------------------------------------------- */
/* Offsets (doubles, relative to a fit's ws_offset) of the fit's workspace regions, so tests can
 * inspect trained parameters: out8 = {Mp, matrices, vectors, X/Z block, test points, Dinv blocks,
 * scalars, total}.  Matrices are Mp x Mp row-major in the order LS, LS^T, Adam m/v of LS, G_LS, L,
 * L^T, L^-1, L^-T, KX, A, A^T, B, B^T, G_A, G_KX, G_KX^T; scalars: c, rho_s, rho_l, ... , loss.
 * Which of the intermediate slots a fit fills depends on its kernel (the single-workgroup MFMA kernels keep L^T but
 * not L, and up to M_p = 256 no A^T / B^T): trained parameters -- L_S, m (vector 1), Z, the scalars -- always. */
int gapro_fit_workspace_layout(int32_t m, int32_t t, int32_t feat_dim, int64_t* out8);

#ifdef __cplusplus
}
#endif
#endif /* GAPRO_HIP_H */
