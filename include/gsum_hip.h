/*
 * gsum_hip.h -- C ABI of libgsum_hip.so: the MI355X (gfx950) replacement for the compiled third-party routines behind
 * buqeye/gsum's GP hot path.  The reference is pure Python; its "FFI" for this path is the numpy / scipy / scikit-learn
 * operator interface (paths relative to the reference repository, gsum/models.py unless noted):
 *     kernel(X[, Y])                        :708, 822-824, 958-960       -> gsum_kernel_build[_dev]
 *     numpy.linalg.cholesky(A)              :711, 809, 969               -> gsum_potrf_lower
 *     scipy.linalg.cho_solve((L, True), B)  :479 (solve_sqrt), 432-439, 831, 836, 1032 -> gsum_forward_*, gsum_cho_solve
 *     np.log(np.diag(L)).sum(), einsum      :1015, 1035                  -> folded into gsum_forward_gram / the fused path
 * and the library is bound with ctypes (gsum_amd/_lib.py; INTEGRATION.md shows the stub a gsum maintainer would add).
 *
 * Conventions: matrices are fp64, row-major (numpy C order); host buffers are caller-owned and borrowed for the call only;
 * device objects are opaque handles owned by the library.  Return value 0 = success, < 0 = API / runtime error (message:
 * gsum_last_error).  A matrix that is not positive definite is NOT an error: *info > 0 (LAPACK's 1-based index of the first
 * bad pivot) reports it -- what numpy.linalg.cholesky turns into LinAlgError.  A gsum_ctx is bound to ONE GPU and is not
 * thread-safe (a gsum_group holds one per GPU and drives them from threads of its own); every function is synchronous at return.
 * Diagnostics, probes and schedule experiments are NOT part of this contract: include/gsum_hip_debug.h, built only into
 * libgsum_hip_lab.so (-DGSUM_LAB).
 */
#ifndef GSUM_HIP_H
#define GSUM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GSUM_MAX_D 8          /* max input dimension of X                      */
#define GSUM_MAX_RHS 16       /* max right-hand-side columns of the fused path */
#define GSUM_NB 128           /* Cholesky block size (informational)           */

typedef struct gsum_ctx gsum_ctx;
typedef struct gsum_mat gsum_mat;   /* device square matrix / Cholesky factor */
enum { GSUM_RBF = 0, GSUM_MATERN52 = 1, GSUM_MATERN32 = 2, GSUM_MATERN12 = 3,
       GSUM_RQ = 4, GSUM_EXPSINE = 5, GSUM_MATERN_INF = 6, /* RationalQuadratic, ExpSineSquared, Matern(nu = inf): tree leaves only */
       GSUM_DOT = 7 /* DotProduct (tree leaf; sigma_0 in length_scale[0]): x . y + sigma_0^2, the one leaf whose diagonal is not 1 */ };
/* A scikit-learn kernel (the reference accepts any: models.py:146-147, 686-688, 958-960).  Arithmetic follows
 * sklearn/gaussian_process/kernels.py: RBF 1556-1565, Matern 1711-1738, RationalQuadratic 1874-1903, WhiteKernel 1401-1414, Sum 858-866,
 * Product 956-966, ExpSineSquared 2032-2064, Exponentiation 1126-1150, DotProduct 2166-2186.  n_ops == 0: the flattened form amplitude * base(X / length_scale) + additive_const (+ white_noise on the
 * one-argument diagonal) -- the reference's own kernels, the templated fast kernels.  n_ops > 0: a Sum / Product tree as a postfix
 * program in scikit-learn's evaluation order: GSUM_OP_LEAF + l pushes leaf[l](x, y) (exactly 1 on the one-argument diagonal; DotProduct: x . x + sigma_0^2),
 * GSUM_OP_CONST + c pushes cval[c], GSUM_OP_WHITE + c pushes cval[c] on that diagonal and 0 elsewhere, ADD / MUL combine two values,
 * GSUM_OP_POW + c raises the top value to the power cval[c] (Exponentiation: the exponent is no hyperparameter). */
#define GSUM_MAX_LEAVES 4
#define GSUM_MAX_OPS 24
enum { GSUM_OP_ADD = 1, GSUM_OP_MUL = 2, GSUM_OP_LEAF = 16, GSUM_OP_CONST = 32, GSUM_OP_WHITE = 64, GSUM_OP_POW = 128 };
typedef struct {
    int32_t family;                    /* GSUM_RBF ... GSUM_DOT */
    int32_t anisotropic;               /* 0: length_scale[0] for every dimension */
    double length_scale[GSUM_MAX_D];
    double alpha;                      /* RationalQuadratic: scale mixture alpha; ExpSineSquared: periodicity (unused otherwise) */
} gsum_kernel_leaf;
typedef struct {
    int32_t family;                    /* GSUM_RBF ... GSUM_MATERN12 */
    int32_t anisotropic;               /* 0: length_scale[0] for every dimension */
    double length_scale[GSUM_MAX_D];
    double amplitude;                  /* ConstantKernel factor (1.0 if absent)  */
    double additive_const;             /* ConstantKernel summand (0.0 if absent) */
    double white_noise;                /* WhiteKernel noise_level (0.0 if absent)*/
    int32_t n_ops, n_leaves;           /* 0, 0: the flattened form above is the kernel */
    int32_t op[GSUM_MAX_OPS];
    double cval[GSUM_MAX_OPS];
    gsum_kernel_leaf leaf[GSUM_MAX_LEAVES];
} gsum_kernel_desc;

/* Coefficient covariance -> partial-sum covariance (TruncationProcess.cov, models.py:1343-1354; helpers.py:149-182):
 * cov_ij = factor * ref_i ref_j * S(ratio_i ratio_j) * kernel_ij,  S(x) = (x^start - x^(end+1)) / (1 - x) - sum of x^e over the
 * excluded orders e in [start, end];  end < 0: the infinite sum. */
#define GSUM_MAX_EXCLUDED 16
typedef struct {
    int32_t start, end;
    int32_t n_excluded;
    int32_t excluded[GSUM_MAX_EXCLUDED];
    double factor;
} gsum_series_scale;

/* One free log-hyperparameter theta_p, in scikit-learn's order (k1 before k2, a leaf's attributes alphabetical): what
 * d kernel(X) / d theta_p is.  AMPLITUDE: the ConstantKernel factor of the stationary term; LENGTH_ISO / LENGTH_DIM: the (dim-th)
 * length scale; WHITE / ADDITIVE: a WhiteKernel / additive ConstantKernel term whose own value is `weight`. */
#define GSUM_MAX_GRAD 12
enum { GSUM_GRAD_AMPLITUDE = 0, GSUM_GRAD_LENGTH_ISO = 1, GSUM_GRAD_LENGTH_DIM = 2, GSUM_GRAD_WHITE = 3, GSUM_GRAD_ADDITIVE = 4,
       /* tree form (n_ops > 0); `dim` = slot index for CONST / WHITE, leaf * 16 + dimension for the leaf parameters: */
       GSUM_GRAD_TREE_CONST = 16, GSUM_GRAD_TREE_WHITE = 17, GSUM_GRAD_TREE_LENGTH_ISO = 18, GSUM_GRAD_TREE_LENGTH_DIM = 19,
       GSUM_GRAD_TREE_ALPHA = 20 /* a leaf's second parameter: RationalQuadratic alpha, ExpSineSquared periodicity */ };
typedef struct {
    int32_t code;
    int32_t dim;
    double weight;
} gsum_grad_param;

/* ---- context -------------------------------------------------------------------------------------------------------- */
int gsum_init(int device, gsum_ctx** out);
void gsum_destroy(gsum_ctx* ctx);                       /* call before the process exits */
const char* gsum_last_error(gsum_ctx* ctx);             /* NULL ctx: error of a failed gsum_init */
/* Options of the contract (all leave results bit-identical; < 0 for an unknown name):
 *   "release_scratch" (any value)  free the grown work buffers and workspace matrices now
 *   "profile_gemm"    0 | N        HIP events around every launch of every N-th single evaluation, and of every batch call
 *                                  (gsum_kernel_profile reads them)
 *   "small_path", "medium_path" 0 | 1, "medium_min_batch"   n <= 128 / 128 < n <= 4096: whole evaluations in one workgroup when a
 *                                  call carries enough of them (<= 0: the measured break-even)
 *   "wave_groups" 1..4, "wave_size" 1..24   layout of a batch: groups x evaluations per group in flight (default 3 x 8; each
 *                                  evaluation in flight owns a workspace matrix, 0.55 GB at n = 8192; a call of at least
 *                                  4 x groups x size evaluations runs two cohorts per group: twice as many in flight)
 *   "chain_persist"   -1 | 0 | 1   schedule of ONE factorisation: persistent chain kernel from order 768 up / never / whenever
 *                                  the order allows;  "lookahead" 0 | 1  look-ahead in the host-enqueued schedule
 *   "pivot_guard_ulps" 0..1024     a pivot p <= guard * eps * A_jj counts as not positive (default 2; process-wide)
 * gsum_get_option reads these back, plus "wave_streams" (streams the last batch call used: groups + 1), "chain_probe",
 * "chain_aborts" (give-ups of the persistent-chain schedule; the fused path re-runs itself, see GSUM_ERR_CHAIN_ABORT), and
 * "pipes_ok" / "pipe_overlap_permille" / "pipe_heals": gsum_init runs a 100-us kernel on every pair of the context's four streams and
 * replaces a stream that takes turns with another (two streams on one hardware queue: seen for the second context of a process) by a
 * new one, up to eight times ("pipe_heals"); 1 = every pair overlaps (>= half its length; the smallest overlap in 1/1000), 0 = some
 * pair still takes turns (e.g. a tool serialises dispatches): results are unaffected, a batch runs 3-6 % and a single factorisation
 * 30-70 % slower. */
int gsum_set_option(gsum_ctx* ctx, const char* name, int64_t value);
int64_t gsum_get_option(gsum_ctx* ctx, const char* name);

/* ---- operator level (one reference call each) ------------------------------------------------------------------------ */
/* kernel(X) / kernel(X, Y) -> host array.  Replaces models.py:708 (corr_), 822 (R_on), 824 (R_nn), 599 (cov).  Y == NULL:
 * one-argument form (n x n, unit diagonal forced, WhiteKernel noise and diag_add on the diagonal); else n x m, no diagonal terms. */
int gsum_kernel_build(gsum_ctx* ctx, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d,
                      const double* Y, int64_t m, double diag_add, double* out);
/* ... scaled on the device before it leaves, out_ij = sc.factor * ref_i ref_j S(ratio_i ratio_j) * kernel_ij: TruncationProcess.cov
 * (models.py:1343-1348) without its four n x m host temporaries.  Y == NULL: ref_y, ratio_y are ignored. */
int gsum_kernel_build_series(gsum_ctx* ctx, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d, const double* Y, int64_t m,
                             double diag_add, const gsum_series_scale* sc, const double* ref_x, const double* ratio_x,
                             const double* ref_y, const double* ratio_y, double* out);
/* kernel(X) + diag_add * I kept on the device, ready to factorise.  Replaces models.py:958-963, 708 + 711.
 * Orders: 1 .. 46000 (validated to 40960; a padded matrix of 2^31 elements or more is refused with an error, never attempted). */
int gsum_kernel_build_dev(gsum_ctx* ctx, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d,
                          double diag_add, gsum_mat** out);
/* upload a caller-built symmetric matrix (only its lower triangle is read) */
int gsum_mat_from_host(gsum_ctx* ctx, const double* A, int64_t n, gsum_mat** out);
/* numpy.linalg.cholesky(A), in place on the device.  Replaces models.py:711, 809, 969.  *info = LAPACK dpotrf's info.
 * GSUM_ERR_CHAIN_ABORT: the persistent-chain schedule of a single factorisation timed out (streams of the process did not run
 * side by side); the matrix is destroyed and the schedule switched off for the context -- rebuild the matrix and call again. */
#define GSUM_ERR_CHAIN_ABORT (-3)
int gsum_potrf_lower(gsum_ctx* ctx, gsum_mat* A, int64_t* info);
/* W = L^-1 RHS (forward substitution only), G = W^T W (k x k), sum_log_diag = sum_i log L_ii.  Replaces the cho_solve calls
 * at models.py:432, 438, 439, 1032 (+269, 217), the reductions at :433, 1015, 1035 and the N x N temporary at :441-442. */
int gsum_forward_gram(gsum_ctx* ctx, gsum_mat* L, const double* RHS, int64_t n, int32_t k, double* G, double* sum_log_diag);
/* W = L^-1 RHS (n x k host): the forward half of cho_solve (models.py:479; with gsum_predict_terms: the solve at :831) */
int gsum_forward_solve(gsum_ctx* ctx, gsum_mat* L, const double* RHS, int64_t n, int32_t k, double* W);
/* X = cho_solve((L, True), B): both triangular solves of solve_sqrt (models.py:460-479); B, X n x k host, k <= GSUM_MAX_RHS */
int gsum_cho_solve(gsum_ctx* ctx, gsum_mat* L, const double* B, int64_t n, int32_t k, double* X);
/* out = L Z (n x k): draws from N(mean, L L^T) without the n x n SVD of models.py:869-876, datasets.py:69-70 */
int gsum_tri_multiply(gsum_ctx* ctx, gsum_mat* L, const double* Z, int64_t n, int32_t k, double* out);
/* Predictive pieces for m new points Xs (models.py:822-836) from the factor of kernel(X) + nugget: V = L^-1 kernel(X, Xs);
 * colsumsq[j] = sum_i V_ij^2;  VtW = V^T (L^-1 RHS) (m x k; NULL / k = 0 to skip);  cov_out (or NULL) = V^T V (m x m). */
int gsum_predict_terms(gsum_ctx* ctx, gsum_mat* L, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d,
                       const double* Xs, int64_t m, const double* RHS, int32_t k, double* colsumsq, double* VtW, double* cov_out);
/* the same under SURVEY.md section 8b's name: colsumsq (m) and, optionally, VtW (m x GSUM_MAX_RHS, columns >= k zero) for the k right-hand
 * sides whose forward solve the factor already holds (the last gsum_forward_gram / gsum_forward_solve on it): predict after fit without
 * handing the residuals over again (models.py:822-836) */
int gsum_predict_var(gsum_ctx* ctx, gsum_mat* L, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d,
                     const double* Xs, int64_t m, double* colsumsq, double* VtW);
/* in-place series scaling of an unfactored device matrix (ref, ratio: n host values): K_oo of models.py:1443, 1466 */
int gsum_mat_scale_series(gsum_ctx* ctx, gsum_mat* A, const gsum_series_scale* sc, const double* ref, const double* ratio);
/* gsum_predict_terms with kernel(X, Xs) scaled like cov(X, Xs, start, end) first: the pieces of models.py:1449-1452, 1470-1473 */
int gsum_predict_terms_series(gsum_ctx* ctx, gsum_mat* L, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d,
                              const double* Xs, int64_t m, const double* RHS, int32_t k, const gsum_series_scale* sc,
                              const double* ref_x, const double* ratio_x, const double* ref_s, const double* ratio_s,
                              double* colsumsq, double* VtW, double* cov_out);
/* copy out: the full symmetric matrix (before potrf) or L with a zeroed upper triangle (after) */
int gsum_mat_to_host(gsum_ctx* ctx, const gsum_mat* A, double* out);
int64_t gsum_mat_n(const gsum_mat* A);
void gsum_mat_free(gsum_ctx* ctx, gsum_mat* A);

/* ---- fused hot path: K build -> jittered Cholesky -> Gram / log-det, per kernel ---------------------------------------- */
/* One full evaluation per descriptor, the work of one ConjugateGaussianProcess.log_marginal_likelihood call (models.py:958-1039):
 *   R = kernel_i(X) + nugget I;  L = chol(R);  W = L^-1 RHS;  G_i = W^T W;  sld_i = sum log diag L.
 * G_out n_kernels x k x k, sld_out n_kernels, info_out n_kernels (potrf info; G / sld undefined where > 0). */
int gsum_lml_batch(gsum_ctx* ctx, const gsum_kernel_desc* kernels, int32_t n_kernels, const double* X, int64_t n, int32_t d,
                   const double* RHS, int32_t k, double nugget, double* G_out, double* sld_out, int64_t* info_out);
/* The same with inputs resident in HBM: gsum_set_inputs uploads X and RHS once, gsum_lml_resident evaluates descriptors against
 * them (what bench.py times).  Three or more evaluations advance in groups, ONE launch per kernel class and outer step carrying
 * all members of a group, on groups + 1 streams; results equal one-at-a-time runs bit for bit.  n <= 128,
 * and 128 < n <= 4096 with many evaluations per call, run whole evaluations in one workgroup each (flattened descriptors). */
int gsum_set_inputs(gsum_ctx* ctx, const double* X, int64_t n, int32_t d, const double* RHS, int32_t k);
int gsum_resident_shape(gsum_ctx* ctx, int64_t* n, int32_t* d, int32_t* k);      /* zeros before the first gsum_set_inputs */
int gsum_lml_resident(gsum_ctx* ctx, const gsum_kernel_desc* kernels, int32_t n_kernels, double nugget,
                      double* G_out, double* sld_out, int64_t* info_out);
/* Several right-hand-side sets resident at once (RHS_sets: n_sets x n x k, C order), evaluation i reading set set_of[i]: a whole
 * (ratio, theta) likelihood surface -- one set of coefficient curves per ratio row, docs/notebooks/correlated_EFT_publication.ipynb:
 * 1444-1459 -- is then ONE call, its rounds back to back on the device instead of one call per row with the host in between. */
int gsum_set_inputs_sets(gsum_ctx* ctx, const double* X, int64_t n, int32_t d, const double* RHS_sets, int32_t n_sets, int32_t k);
int gsum_lml_resident_sets(gsum_ctx* ctx, const gsum_kernel_desc* kernels, const int32_t* set_of, int32_t n_kernels, double nugget,
                           double* G_out, double* sld_out, int64_t* info_out);
/* Value AND gradient pieces of one evaluation (log_marginal_likelihood(theta, eval_gradient=True)).  Replaces
 * kernel(X, eval_gradient=True) (the n x n x p array at models.py:958, 1204), cho_solve(L, eye(N)) (:1044, 1266) and the einsum
 * contractions at :229, 276, 453-454, 1049.  With R = kernel(X) + nugget I, V = R^-1 RHS: G, sld, info as above;
 * trace_out[p] = tr(R^-1 dR_p);  H_out[p] (k x k) = V^T dR_p V.  dR_p is never materialised. */
int gsum_lml_grad(gsum_ctx* ctx, const gsum_kernel_desc* desc, const gsum_grad_param* params, int32_t n_params,
                  const double* X, int64_t n, int32_t d, const double* RHS, int32_t k, double nugget, double* G_out,
                  double* sld_out, int64_t* info_out, double* trace_out, double* H_out);
/* ... for n_desc kernels of ONE hyperparameter structure (the restarts of a multi-start fit, models.py:641-662): params holds
 * n_desc x n_params entries; outputs stacked: G (n_desc, k, k), sld, info (n_desc), trace (n_desc, P), H (n_desc, P, k, k). */
int gsum_lml_grad_batch(gsum_ctx* ctx, const gsum_kernel_desc* descs, int32_t n_desc, const gsum_grad_param* params,
                        int32_t n_params, const double* X, int64_t n, int32_t d, const double* RHS, int32_t k, double nugget,
                        double* G_out, double* sld_out, int64_t* info_out, double* trace_out, double* H_out);

/* ---- multi-GPU: the grid partition inside the ABI ------------------------------------------------------------------------
 * One process (or one context) per GPU; grid points are independent, so the list is block-partitioned over `world` ranks with
 * no data-path collective.  gsum_shard_range is THE partition: rank r owns [min(total, r c), min(total, (r + 1) c)),
 * c = ceil(total / world).  gsum_lml_resident_shard evaluates this rank's descriptors into THEIR positions of full-length
 * arrays, so that a host holding its own RCCL communicator all-gathers in place:
 *     ncclAllGather(sld + lo, sld, c, ncclDouble, comm, stream)          (tests/c_host/shard_host_rccl.c)
 * A host with its own communicator gathers like that; the group calls below carry the same step inside the library.  Replaces the serial loop docs/notebooks/correlated_EFT_publication.ipynb:1457-1459. */
int gsum_shard_range(int64_t total, int32_t rank, int32_t world, int64_t* lo, int64_t* hi);
int gsum_lml_resident_shard(gsum_ctx* ctx, const gsum_kernel_desc* kernels, int32_t n_kernels, int32_t rank, int32_t world,
                            double nugget, double* G_out, double* sld_out, int64_t* info_out, int64_t* lo, int64_t* hi);

/* ---- multi-GPU from ONE caller thread: a group of contexts (gsum_init over a device list; 8e) ------------------
 * The reference's user is a single Python process (the notebook above; gsum/models.py:1485-1507), so the node's GPUs are also reachable
 * without one process per GPU: gsum_init_multi opens one gsum_ctx per listed device (device_ids == NULL: devices 0 .. n_devices - 1;
 * n_devices <= 0: every visible GPU), and the calls below cut the descriptor list with gsum_shard_range, run each device's block on a
 * host thread of its own inside the library and write results straight into the caller's full-length arrays -- equal, bit for bit, to
 * the one-device call.  flags = GSUM_GATHER_RCCL adds the path's one exchange step on the devices: the packed result rows (G, sld, info)
 * are staged to each device's gather buffer at their positions, exchanged with one in-place ncclAllGather per device
 * (ncclCommInitAll over the group's devices, opened on the first such call; librccl.so is loaded then, not linked), read back from
 * rank 0 and checked byte for byte against every other rank's copy.  gsum_group_allgather is that step for any row-partitioned host
 * array (rows x width doubles; rank r's rows are gsum_shard_range(rows, r, world)).  gsum_group_ctx lends a member context for the
 * operator-level calls (predict: new points sharded over devices); a member is used by one thread at a time.
 * gsum_group_get: "rccl" (0 not opened, 1 open, -1 unavailable), "rccl_gathers", "devices_used" (devices with work in the last scan). */
typedef struct gsum_group gsum_group;
#define GSUM_GATHER_RCCL 1
int gsum_init_multi(int n_devices, const int* device_ids, gsum_group** out);
/* ... or over contexts the caller opened (one per GPU) and keeps: gsum_group_destroy then leaves them alive (destroy the group first) */
int gsum_group_adopt(int n_ctx, gsum_ctx* const* ctxs, gsum_group** out);
void gsum_group_destroy(gsum_group* group);
int32_t gsum_group_size(const gsum_group* group);
gsum_ctx* gsum_group_ctx(gsum_group* group, int32_t i);                 /* borrowed: destroyed with the group */
const char* gsum_group_last_error(gsum_group* group);                   /* NULL group: error of a failed gsum_init_multi */
int64_t gsum_group_get(gsum_group* group, const char* name);
int gsum_group_set_inputs(gsum_group* group, const double* X, int64_t n, int32_t d, const double* RHS, int32_t k);   /* every device */
int gsum_group_lml_resident(gsum_group* group, const gsum_kernel_desc* kernels, int32_t n_kernels, double nugget,
                            double* G_out, double* sld_out, int64_t* info_out, int32_t flags);
/* ... with several right-hand-side sets (gsum_set_inputs_sets): every device keeps all sets, the descriptors are partitioned */
int gsum_group_set_inputs_sets(gsum_group* group, const double* X, int64_t n, int32_t d, const double* RHS_sets, int32_t n_sets, int32_t k);
int gsum_group_lml_resident_sets(gsum_group* group, const gsum_kernel_desc* kernels, const int32_t* set_of, int32_t n_kernels, double nugget,
                                 double* G_out, double* sld_out, int64_t* info_out, int32_t flags);
/* gsum_lml_batch over the group's devices: X and RHS uploaded to every device (0.5 MB at n = 8192), descriptors block-partitioned */
int gsum_lml_batch_multi(gsum_group* group, const gsum_kernel_desc* kernels, int32_t n_kernels, const double* X, int64_t n, int32_t d,
                         const double* RHS, int32_t k, double nugget, double* G_out, double* sld_out, int64_t* info_out, int32_t flags);
int gsum_group_allgather(gsum_group* group, double* buf, int64_t rows, int64_t width);

/* ---- measurement -------------------------------------------------------------------------------------------------------- */
/* HIP-event times (ms) of the last single fused evaluation: K build, Cholesky (incl. fused solve), read-out + D2H, total */
int gsum_timers(gsum_ctx* ctx, double* ms, int32_t n);
/* With "profile_gemm" on: summed HIP-event durations (ms), algorithmic flops and launch counts since the last call, five classes:
 * [0] kernel build, [1] diagonal blocks, [2] panel solves and near updates, [3] the bulk trailing update, [4] the rest; each launch
 * is timed on the stream it runs on (the bulk class of a batch runs on one stream: its sum is wall time).  Resets the record. */
int gsum_kernel_profile(gsum_ctx* ctx, double* ms5, double* flops5, int64_t* launches5);
#ifdef __cplusplus
}
#endif
#endif /* GSUM_HIP_H */
