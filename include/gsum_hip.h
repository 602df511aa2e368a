/*
 * gsum_hip.h — C ABI of libgsum_hip.so: the MI355X (gfx950) replacement for the
 * compiled third-party routines that buqeye/gsum's GP hot path calls.
 *
 * Every entry point names the reference call site(s) it replaces
 * (paths relative to the reference repository, gsum/models.py unless noted).
 * The reference is pure Python; its "FFI" for this path is the numpy / scipy /
 * scikit-learn operator interface:
 *     kernel(X[, Y])                       models.py:708, 822-824, 958-960
 *     numpy.linalg.cholesky(A)             models.py:711, 809, 969
 *     scipy.linalg.cho_solve((L, True), B) models.py:479 (solve_sqrt), 432-439, 831, 836, 1032
 *     np.log(np.diag(L)).sum(), einsum     models.py:1015, 1035
 * and this library is bound with ctypes (gsum_amd/_lib.py; INTEGRATION.md shows
 * the stub a gsum maintainer would add).
 *
 * Conventions
 *   - all matrices are fp64, row-major (numpy C order); host buffers are
 *     caller-owned and borrowed for the duration of the call only;
 *   - device objects are opaque handles owned by the library;
 *   - return value: 0 = success, <0 = API/runtime error (message via
 *     gsum_last_error).  A non-positive-definite matrix is NOT an error: the
 *     LAPACK-style *info > 0 (1-based index of the first bad pivot) reports it,
 *     exactly what numpy.linalg.cholesky turns into LinAlgError;
 *   - a gsum_ctx is bound to ONE GPU and is not thread-safe; every function is
 *     synchronous at return (results on the host are valid).
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

enum { GSUM_RBF = 0, GSUM_MATERN52 = 1, GSUM_MATERN32 = 2, GSUM_MATERN12 = 3 };

/* Flattened scikit-learn kernel tree  amplitude * base(X/length_scale) + additive_const (+ white on the
 * one-argument diagonal).  Arithmetic follows sklearn/gaussian_process/kernels.py: RBF 1556-1565,
 * Matern 1711-1738, WhiteKernel 1401-1414, Sum 858-866, Product 956-966. */
typedef struct {
    int32_t family;                    /* GSUM_RBF ... */
    int32_t anisotropic;               /* 0: length_scale[0] for every dimension */
    double length_scale[GSUM_MAX_D];
    double amplitude;                  /* ConstantKernel factor (1.0 if absent)  */
    double additive_const;             /* ConstantKernel summand (0.0 if absent) */
    double white_noise;                /* WhiteKernel noise_level (0.0 if absent)*/
} gsum_kernel_desc;

/* Scaling of a coefficient covariance into a partial-sum covariance (TruncationProcess.cov, models.py:1343-1354,
 * with helpers.py:149-182):  cov_ij = factor * ref_i ref_j * S(ratio_i ratio_j) * kernel_ij,
 *   S(x) = (x^start - x^(end+1)) / (1 - x) - sum over excluded orders e with start <= e <= end of x^e.
 * end < 0 stands for the infinite sum (x^(end+1) taken as 0, as numpy gives for |x| < 1 and end = inf). */
#define GSUM_MAX_EXCLUDED 16
typedef struct {
    int32_t start, end;
    int32_t n_excluded;
    int32_t excluded[GSUM_MAX_EXCLUDED];
    double factor;
} gsum_series_scale;

/* One free log-hyperparameter theta_p of the kernel tree, as scikit-learn orders them (k1 before k2, attributes
 * in alphabetical order): what d kernel(X) / d theta_p is.  AMPLITUDE: a ConstantKernel factor of the stationary
 * term (gradient = that whole term); LENGTH_ISO / LENGTH_DIM: the (dim-th) length scale; WHITE / ADDITIVE: one
 * WhiteKernel / additive ConstantKernel term whose own value is `weight` (gradient = weight on the diagonal /
 * everywhere). */
#define GSUM_MAX_GRAD 12
enum { GSUM_GRAD_AMPLITUDE = 0, GSUM_GRAD_LENGTH_ISO = 1, GSUM_GRAD_LENGTH_DIM = 2, GSUM_GRAD_WHITE = 3, GSUM_GRAD_ADDITIVE = 4 };
typedef struct {
    int32_t code;
    int32_t dim;
    double weight;
} gsum_grad_param;

/* ---- context ---------------------------------------------------------------------------------- */
int gsum_init(int device, gsum_ctx** out);
/* Call before the process exits: a context may own streams created with a CU mask ("reserve_cus"), and a process that
 * exits with one alive makes rocprofv3 crash in its finaliser (the Python binding registers an atexit for this). */
void gsum_destroy(gsum_ctx* ctx);
const char* gsum_last_error(gsum_ctx* ctx);            /* NULL ctx: error of a failed gsum_init   */
/* knobs: "lookahead", "build_lower_only", "profile_gemm", "diag_stamps" (0/1), "batch_slots" (1..24; default 16 with
 * GPU_MAX_HW_QUEUES >= 16 in the environment, else 12 / 8 / 3 -- keep it at 16 or below in a process that also owns an
 * RCCL communicator: the device time-slices user compute queues beyond 24), "diag_algo" / "build_algo" (2 = the round-2
 * kernels, 1 = the round-1 ones, kept for A/B), "bulk_lds_pad" (bytes of LDS the bulk kernel requests in the look-ahead
 * schedule of a factorisation, to leave room for chain workgroups; default 80 KB = two bulk workgroups per CU, 0 = three),
 * "small_path" (0/1: fused single-workgroup evaluation for n <= 128), "medium_path" (0/1) and "medium_min_batch"
 * (128 < n <= 4096: calls with at least that many evaluations -- <= 0: auto, max(4, n^1.45 / 985) -- run
 * one workgroup per evaluation, 256 in flight), "stagger" (-1 auto, 0 off, n: de-phase co-resident workgroups by n x 2048 cycles),
 * "reserve_cus" (0 = off, the default; 1..8: CUs per XCD the bulk stream's CU mask leaves to the panel chain while
 * a look-ahead factorisation runs; -1: 2 from order 6144 up), "lazy_far" (0/1: batches at n >= 8192 update the far trailing region every other panel with K = 512),
 * "chain_prefetch" (1, default: four operand chunks in flight in the 32x128 chain GEMM tile; 0: one), and schedule
 * variants that give bit-identical results ("Chain experiments"): "chain_fused"
 * (two diagonal blocks per launch + both panels of the rows below in one: 1 always, 0 never, -1 = default = in batches
 * only, where it is 1.8 % faster; alone it is slower), "chain_window" (look-ahead on a window of rows,
 * the rest of each panel on a second stream), "la_depth2" (1, default: bulk update in two launches, the chain waits for the first), "la_split" (rows; 0 = off, default:
 * split the look-ahead update while the trailing matrix is at least that tall),
 * "release_scratch" (any value: free the grown work buffers and the per-slot workspace matrices now).
 * <0 for an unknown name. */
int gsum_set_option(gsum_ctx* ctx, const char* name, int64_t value);
/* read back: "batch_slots", "lookahead", "bulk_cfg", and the queue-concurrency probe that runs before the first batch
 * wanting more than 4 evaluations in flight -- "queue_probe_streams" (0: not run yet), "queue_probe_concurrency_x100"
 * (streams x spin time / elapsed, x 100), "queue_probe_fell_back" (the slot count that was refused: the batch then runs
 * 3 in flight, the optimum on the runtime's default 4 hardware queues).  -1 for an unknown name. */
int64_t gsum_get_option(gsum_ctx* ctx, const char* name);

/* ---- operator level (one reference call each) ------------------------------------------------- */

/* kernel(X) / kernel(X, Y) -> host array.   replaces models.py:708 (corr_), 822 (R_on), 824 (R_nn),
 * 599 (cov).  Y == NULL: one-argument form (n x n, unit diagonal forced, WhiteKernel noise and
 * diag_add on the diagonal).  Y != NULL: cross form (n x m, no diagonal terms; kernels.py:1413-1414). */
int gsum_kernel_build(gsum_ctx* ctx, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d,
                      const double* Y, int64_t m, double diag_add, double* out);

/* kernel(X) + diag_add*I kept on the device, ready to factorise.   replaces models.py:958-963, 708+711. */
int gsum_kernel_build_dev(gsum_ctx* ctx, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d,
                          double diag_add, gsum_mat** out);

/* upload a caller-built symmetric matrix (only its lower triangle is read). */
int gsum_mat_from_host(gsum_ctx* ctx, const double* A, int64_t n, gsum_mat** out);

/* numpy.linalg.cholesky(A), in place on the device.   replaces models.py:711, 809, 969.
 * *info = 0 or the LAPACK dpotrf info (>0: leading minor of that order is not positive definite).
 * Returns GSUM_ERR_CHAIN_ABORT when the persistent-chain schedule of a single factorisation timed out (its streams did not run
 * side by side, e.g. under a tool that serialises dispatches): the matrix is destroyed, the schedule is switched off for the
 * context -- rebuild the matrix and call again (the Python binding's HipContext.factorize does exactly that, once). */
#define GSUM_ERR_CHAIN_ABORT (-3)
int gsum_potrf_lower(gsum_ctx* ctx, gsum_mat* A, int64_t* info);

/* W = L^-1 RHS (forward substitution only), G = W^T W (k x k), sum_log_diag = sum_i log L_ii.
 * replaces the cho_solve calls at models.py:432, 438, 439, 1032 (+269, 217), the reductions at
 * :433, :1015, :1035 and the dense N x N Woodbury temporary at :441-442. */
int gsum_forward_gram(gsum_ctx* ctx, gsum_mat* L, const double* RHS, int64_t n, int32_t k,
                      double* G, double* sum_log_diag);

/* W = L^-1 RHS (n x k host, k <= GSUM_MAX_RHS): the forward half of scipy.linalg.cho_solve
 * (models.py:479); with gsum_predict_terms it replaces the solve at models.py:831. */
int gsum_forward_solve(gsum_ctx* ctx, gsum_mat* L, const double* RHS, int64_t n, int32_t k, double* W);

/* X = scipy.linalg.cho_solve((L, True), B) = L^-T (L^-1 B) for a factorised matrix: both triangular solves of
 * solve_sqrt(sqrt_R, y, 'cholesky') (models.py:460-479; call sites :217, 269, 432, 438, 439, 1032).  B, X: n x k host,
 * row-major, k <= GSUM_MAX_RHS (the product classes never need it -- they read the Gram matrix of the forward half --
 * but a caller-side solve_sqrt binds to it). */
int gsum_cho_solve(gsum_ctx* ctx, gsum_mat* L, const double* B, int64_t n, int32_t k, double* X);

/* out = L Z (n x k, k <= GSUM_MAX_RHS) for a factorised matrix: the transform y = mean + L z that turns standard
 * normal draws into draws from N(mean, L L^T).  Replaces the n x n SVD / eigendecomposition inside
 * rng.multivariate_normal (models.py:869-876) and scipy.stats.multivariate_normal.rvs (datasets.py:69-70). */
int gsum_tri_multiply(gsum_ctx* ctx, gsum_mat* L, const double* Z, int64_t n, int32_t k, double* out);

/* Predictive pieces for m new points Xs (models.py:822-836), from the factor of
 * kernel(X)+nugget:  V = L^-1 kernel(X, Xs);  colsumsq[j] = sum_i V_ij^2;
 * VtW = V^T Whalf (m x k) with Whalf = L^-1 RHS (RHS n x k host; NULL/k=0 to skip).
 * If cov_out != NULL it receives V^T V (m x m), the reduction term of models.py:836. */
int gsum_predict_terms(gsum_ctx* ctx, gsum_mat* L, const gsum_kernel_desc* desc, const double* X, int64_t n,
                       int32_t d, const double* Xs, int64_t m, const double* RHS, int32_t k,
                       double* colsumsq, double* VtW, double* cov_out);

/* In-place series scaling of an unfactored device matrix (ref, ratio: n host values each): the conditioning matrix
 * K_oo = cov(Xc, Xc, start, end) of TruncationProcess.predict (models.py:1443, 1466) from kernel(Xc, Xc). */
int gsum_mat_scale_series(gsum_ctx* ctx, gsum_mat* A, const gsum_series_scale* sc, const double* ref, const double* ratio);

/* gsum_predict_terms with the cross matrix kernel(X, Xs) scaled like cov(X, Xs, start, end) first (ref_x / ratio_x:
 * n values, ref_s / ratio_s: m values): with L = chol(K_oo) its outputs are the pieces of models.py:1449-1452 and
 * :1470-1473 -- K_no alpha = VtW, K_no K_oo^-1 K_on = cov_out, its diagonal = colsumsq. */
int gsum_predict_terms_series(gsum_ctx* ctx, gsum_mat* L, const gsum_kernel_desc* desc, const double* X, int64_t n,
                              int32_t d, const double* Xs, int64_t m, const double* RHS, int32_t k,
                              const gsum_series_scale* sc, const double* ref_x, const double* ratio_x,
                              const double* ref_s, const double* ratio_s, double* colsumsq, double* VtW,
                              double* cov_out);

/* copy out: the full symmetric matrix (before potrf) or L with a zeroed upper triangle (after). */
int gsum_mat_to_host(gsum_ctx* ctx, const gsum_mat* A, double* out);
int64_t gsum_mat_n(const gsum_mat* A);
void gsum_mat_free(gsum_ctx* ctx, gsum_mat* A);

/* ---- fused hot path: K build -> jittered Cholesky -> Gram / log-det, per kernel --------------- */

/* One full evaluation per kernel descriptor, exactly the work of one
 * ConjugateGaussianProcess.log_marginal_likelihood call (models.py:958-1039):
 *   R = kernel_i(X) + nugget*I;  L = chol(R);  W = L^-1 RHS;  G_i = W^T W;  sld_i = sum log diag L.
 * G_out: n_kernels x k x k, sld_out: n_kernels, info_out: n_kernels (potrf info; G/sld undefined if >0). */
int gsum_lml_batch(gsum_ctx* ctx, const gsum_kernel_desc* kernels, int32_t n_kernels, const double* X, int64_t n,
                   int32_t d, const double* RHS, int32_t k, double nugget, double* G_out, double* sld_out,
                   int64_t* info_out);

/* Value AND gradient pieces of one evaluation: everything ConjugateGaussianProcess / ConjugateStudentProcess
 * .log_marginal_likelihood(theta, eval_gradient=True) needs beyond O(k^2) host algebra.  Replaces
 * kernel(X, eval_gradient=True) (the n x n x p array at models.py:958, 1204), cho_solve(L, eye(N)) (:1044, 1266)
 * and the einsum contractions at :229, 276, 453-454, 1049.  With R = kernel(X) + nugget I, V = R^-1 RHS:
 *   G_out (k x k), sld_out, info_out as gsum_lml_batch;
 *   trace_out[p] = tr(R^-1 dR_p);  H_out[p] (k x k) = V^T dR_p V,   dR_p = d kernel(X) / d theta_p.
 * R^-1 is formed on the device as U U^T with U = L^-T (2 n^3 / 3 flops on the MFMA GEMM kernels); dR_p is never
 * materialised. */
int gsum_lml_grad(gsum_ctx* ctx, const gsum_kernel_desc* desc, const gsum_grad_param* params, int32_t n_params,
                  const double* X, int64_t n, int32_t d, const double* RHS, int32_t k, double nugget, double* G_out,
                  double* sld_out, int64_t* info_out, double* trace_out, double* H_out);
/* The same for n_desc kernels of ONE hyperparameter structure on one set of inputs: the current points of all restarts of a
 * multi-start fit (gsum/models.py:641-662 runs them one after the other), or a grid of gradients.  Independent evaluations,
 * pipelined over the library's slots (up to 8 in flight; each owns a workspace matrix, U = L^-T and R^-1: 3 n^2 doubles).
 * params holds n_desc x n_params entries (codes and dims equal across kernels, the weights -- hyperparameter values -- per kernel).
 * Outputs are gsum_lml_grad's, stacked: G (n_desc, k, k), sld (n_desc), info (n_desc), trace (n_desc, P), H (n_desc, P, k, k). */
int gsum_lml_grad_batch(gsum_ctx* ctx, const gsum_kernel_desc* descs, int32_t n_desc, const gsum_grad_param* params,
                        int32_t n_params, const double* X, int64_t n, int32_t d, const double* RHS, int32_t k, double nugget,
                        double* G_out, double* sld_out, int64_t* info_out, double* trace_out, double* H_out);

/* Same, with inputs already resident in HBM: gsum_set_inputs uploads X and RHS once,
 * gsum_lml_resident evaluates descriptors against them (what bench.py times).  The evaluations of one
 * call are independent, so up to "batch_slots" of them (16 with GPU_MAX_HW_QUEUES >= 16 in the environment when the
 * HIP runtime initialises, 12 / 8 with 12 / 8, else 3) are kept in flight on separate streams and
 * workspaces (288 GB of HBM holds hundreds of 0.5 GB matrices): the latency-bound panel chain of one
 * factorisation overlaps the bulk GEMMs of the others.  Results are identical to one-at-a-time runs.
 * n <= 128 (the reference's own problem sizes) takes a fused path: one workgroup per evaluation builds K,
 * factors it, solves and reduces, hundreds of evaluations per launch.  128 < n <= 4096 with many evaluations per
 * call does the same with the matrix of each evaluation in its own HBM scratch (one CU per evaluation). */
int gsum_set_inputs(gsum_ctx* ctx, const double* X, int64_t n, int32_t d, const double* RHS, int32_t k);
/* shape of the resident inputs (zeros before the first gsum_set_inputs).  Only gsum_set_inputs writes them: every other
 * entry point, gsum_lml_batch and gsum_lml_grad included, uploads into buffers of its own. */
int gsum_resident_shape(gsum_ctx* ctx, int64_t* n, int32_t* d, int32_t* k);
int gsum_lml_resident(gsum_ctx* ctx, const gsum_kernel_desc* kernels, int32_t n_kernels, double nugget,
                      double* G_out, double* sld_out, int64_t* info_out);

/* ---- multi-GPU: the grid partition inside the ABI ----------------------------------------------------------------
 * One process per GPU, one context each; grid points (and new points of predict) are independent, so the flattened
 * list is block-partitioned over `world` ranks with no data-path collective.  gsum_shard_range is THE partition (the
 * Python layer, gsum_amd/grid.py, calls it too): rank r owns [lo, hi) = [min(total, r c), min(total, (r + 1) c)) with
 * c = ceil(total / world).  gsum_lml_resident_shard evaluates this rank's descriptors only and writes them to THEIR
 * positions of full-length output arrays (the other entries are left untouched), so that a host holding its own RCCL
 * communicator all-gathers in place -- e.g. for the log-determinants, on buffers of world * c doubles:
 *     ncclAllGather(sld + lo, sld, c, ncclDouble, comm, stream)
 * (INTEGRATION.md).  The library itself opens no communicator: which ranks form the group and over which transport
 * is the host's decision (the reference's Python host uses torch.distributed, backend "nccl" = RCCL over xGMI).
 * Replaces: the nested Python loop over grid points, docs/notebooks/correlated_EFT_publication.ipynb:1457-1459. */
int gsum_shard_range(int64_t total, int32_t rank, int32_t world, int64_t* lo, int64_t* hi);
int gsum_lml_resident_shard(gsum_ctx* ctx, const gsum_kernel_desc* kernels, int32_t n_kernels, int32_t rank, int32_t world,
                            double nugget, double* G_out, double* sld_out, int64_t* info_out, int64_t* lo, int64_t* hi);

/* ---- measurement ------------------------------------------------------------------------------ */
/* HIP-event times (ms) of the last fused evaluation on the library's own streams:
 * ms[0] K build, ms[1] Cholesky (incl. fused forward solve), ms[2] finalize + D2H, ms[3] total.
 * With option "diag_stamps" = 1 and n > 4: ms[4..8] = shader-cycle stamps of the last diagonal-block
 * kernel {prologue, column loop, block inverse, total} and its total in 100 MHz ticks. */
int gsum_timers(gsum_ctx* ctx, double* ms, int32_t n);
/* diagnostic: the 64 raw stamp words of the last diagonal-block kernel run with "diag_stamps" = 1 ([0..4] as above;
 * round-2 kernel: [7] start, [8 + 2j], [9 + 2j] wave 0 behind the two barriers of micro-block column j, [24 + j] cycles
 * of the pivot recurrence of micro-block j). */
int gsum_debug_diag_stamps(gsum_ctx* ctx, int64_t* out64);
/* diagnostic: per-outer-step realtime stamps of the last persistent-chain factorisation (options "chain_stamps" = 1,
 * "chain_persist"; gsum_potrf_lower / a single fused evaluation on a matrix whose order is a multiple of 256): out holds
 * 24 values per step in 100 MHz ticks relative to the first stamp (-1: not written); [16..23] = first start / last end of the
 * step's four host-enqueued launches (panel of the rows below the window, updates A, B, Far).  Indices: D role 0 step begins, 1 its
 * diagonal block is up to date, 2 first block's tables published, 3 block row k + 1 up to date, 4 L(k+1, k) published,
 * 5 sibling update done, 6 second block's tables published; P wave 0: 8 its rows are up to date, 9 first tables seen,
 * 10 sibling update done, 11 second tables seen, 12 rows published, 13 its first update task starts, 14 is done.
 * This is the timeline evidence for numpy.linalg.cholesky at gsum/models.py:711, 809, 969 (one factorisation alone). */
int gsum_debug_chain_stamps(gsum_ctx* ctx, double* out, int32_t max_steps, int32_t* steps);
/* With option "profile_gemm" = N > 0 every kernel launch of every N-th fused evaluation (the 1st, N+1-th, ... since the
 * option was set; operator-level calls: every launch) is bracketed by HIP events on the stream it is launched on
 * (~230 launches per evaluation at n = 8192: N = 2 costs ~9 % of batch throughput).  This returns the summed durations
 * (ms), the summed algorithmic flops and the launch count of the BULK trailing-update kernel since the last call, and
 * resets the record (gsum_kernel_profile: all classes). */
int gsum_gemm_profile(gsum_ctx* ctx, double* total_ms, double* total_flops, int64_t* launches);
/* The same record for every kernel class of the fused path, five entries each: [0] kernel-matrix build, [1] diagonal
 * blocks (k_potrf_diag), [2] panel GEMMs (TRSM against the block inverse, sibling and look-ahead columns, border rows),
 * [3] the bulk trailing update (what gsum_gemm_profile returns), [4] the rest (border set-up, diagonal save, read-out).
 * Durations are summed per launch, on the stream of the launch: with several evaluations in flight they overlap, so
 * the sum over classes is bounded by (evaluations in flight) x (wall time), not by the wall time.  Resets the record. */
int gsum_kernel_profile(gsum_ctx* ctx, double* ms5, double* flops5, int64_t* launches5);
/* fp64 MFMA issue-rate probe (v_mfma_f64_16x16x4_f64, operands in registers, waves_per_simd resident
 * waves on every SIMD, n_acc independent accumulators per wave; n_acc = 1 gives the dependent latency):
 * out3 = {achieved TFLOP/s, shader cycles per MFMA per wave, in-kernel clock GHz}. */
int gsum_probe_mfma_f64(gsum_ctx* ctx, int32_t iters, int32_t waves_per_simd, int32_t n_acc, double* out3);
/* HBM streaming-store probe: achieved GB/s writing `bytes` with 16-B stores. */
int gsum_probe_hbm_write(gsum_ctx* ctx, int64_t bytes, double* gbps);
/* diagnostic: run nblocks workgroups on a stream restricted by a CU mask (hipExtStreamCreateWithCUMask; nwords = 0:
 * unrestricted) and report where each ran: out[2b] = XCC id, out[2b+1] = the HW_ID register (CU, SH, SE fields). */
int gsum_probe_cu_mask(gsum_ctx* ctx, const uint32_t* mask, int32_t nwords, int32_t nblocks, int64_t* out);
/* microbenchmark of the MFMA tile kernel on device-resident pseudo-random operands (leading dimension
 * lda >= K for A and B, as inside the factorisation): out2 = {algorithmic TFLOP/s, microseconds per launch}.
 * tri != 0: SYRK form (B = A, lower tiles only, M == N, flops counted as M(M+1)K).
 * Option "bench_fill" = 1 zeroes the operands first (timing is value-independent, board power is not: tools/gpu_power_probe.py).
 * cfg = 99 is not a GEMM: the pure issue rate of v_mfma_f64_16x16x4_f64 on register operands -- M workgroups of N threads (a multiple
 * of 64, <= 512), K rounds of lda (4 or 8) independent MFMAs per wave, tri ignored: what the matrix pipes sustain on this card
 * (77.6 TFLOP/s measured, profiles/r03_mfma_peak.log), the ceiling the tile kernels are measured against. */
int gsum_bench_gemm_nt(gsum_ctx* ctx, int32_t cfg, int32_t tri, int64_t M, int64_t N, int64_t K, int64_t lda,
                       int32_t reps, double* out2);
/* diagnostic build of the 128x128-tile kernel with s_memtime stamps around its loop phases (never used by
 * the product path): out5 = mean shader cycles per wave in {prologue, global-load issue, fragment reads +
 * MFMAs, vmcnt wait + LDS stores, barrier} for one SYRK launch of order M, depth K. */
int gsum_debug_gemm_phases(gsum_ctx* ctx, int64_t M, int64_t K, int64_t lda, double* out5);
/* debug: C(MxN) = beta*C + sign * A(MxK) B(NxK)^T through the MFMA tile kernel (cfg 0: 128x128 tile,
 * 1: 32x128 tile, 2: 16x256 tile, 5: 128x128 tile with 8 waves, 6: the same with LDS-direct operand staging, 7: 128x64
 * tile with LDS-direct staging, three workgroups per CU; tri != 0: lower tiles only, needs M == N). */
int gsum_debug_gemm_nt(gsum_ctx* ctx, int32_t cfg, int32_t tri, double* C, const double* A, const double* B,
                       int64_t M, int64_t N, int64_t K, int32_t beta, double sign);

#ifdef __cplusplus
}
#endif
#endif /* GSUM_HIP_H */
