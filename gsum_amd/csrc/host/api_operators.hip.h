// C ABI, operator level: kernel build, potrf, triangular solves, predictive pieces, series scaling
// (part of gsum_capi.hip: included from there, in order -- one translation unit)
#pragma once
// Small inputs are remembered on the host: an objective evaluation of fit() hands the SAME points and right-hand sides over tens of times
// (models.py:634-640), and at the reference's own sizes the two copies and their synchronisation cost as much as the kernel.  A call whose
// bytes equal the remembered ones copies nothing (ctx->uploads_skipped counts them); every writer of a set refreshes or drops its memory.
#define GS_UPLOAD_REMEMBER (256 * 1024)
static int gs_upload_X(gsum_ctx* ctx, gs_inputs* I, const double* X, int64_t n, int d) {
    if (!X || n <= 0) GS_FAIL("X is NULL or empty");
    const size_t cnt = (size_t)n * d, bytes = cnt * sizeof(double);
    if (bytes <= GS_UPLOAD_REMEMBER && I->X && I->n == n && I->d == d && I->x_host.size() == cnt && !memcmp(I->x_host.data(), X, bytes)) {
        ++ctx->uploads_skipped;
        return 0;
    }
    if (gs_reserve(ctx, &I->X, &I->X_cap, bytes)) return -1;
    GS_CHECK(hipMemcpyAsync(I->X, X, bytes, hipMemcpyHostToDevice, ctx->cur->sm));
    ctx->upload_pending = true;
    if (bytes <= GS_UPLOAD_REMEMBER) I->x_host.assign(X, X + cnt); else I->x_host.clear();
    I->n = n;
    I->d = d;
    return 0;
}

static int gs_upload_Z(gsum_ctx* ctx, gs_inputs* I, const double* Z, int64_t n, int k, int n_sets = 1) {
    if (k < 0 || k > GSUM_MAX_RHS) GS_FAIL("k must be 0..GSUM_MAX_RHS");
    if (k > 0 && !Z) GS_FAIL("RHS is NULL");
    if (n_sets < 1 || n_sets > (1 << 20)) GS_FAIL("the number of right-hand-side sets must be 1..2^20");
    const size_t cnt = (size_t)n_sets * n * k, bytes = cnt * sizeof(double);
    if (k > 0 && bytes <= GS_UPLOAD_REMEMBER && I->Z && I->k == k && I->n_sets == n_sets && I->z_rows == n && I->z_host.size() == cnt &&
        !memcmp(I->z_host.data(), Z, bytes)) {
        ++ctx->uploads_skipped;
        return 0;
    }
    if (gs_reserve(ctx, &I->Z, &I->Z_cap, std::max<size_t>(8, bytes))) return -1;
    if (k > 0) {
        GS_CHECK(hipMemcpyAsync(I->Z, Z, bytes, hipMemcpyHostToDevice, ctx->cur->sm));
        ctx->upload_pending = true;
    }
    if (k > 0 && bytes <= GS_UPLOAD_REMEMBER) I->z_host.assign(Z, Z + cnt); else I->z_host.clear();
    I->z_rows = n;
    I->k = k;
    I->n_sets = n_sets;
    return 0;
}

static int gs_check_series(gsum_ctx* ctx, const gsum_series_scale* sc);

// kernel(X[, Y]) -> host, optionally scaled like TruncationProcess.cov on the device before it leaves (sc != NULL)
static int gs_kernel_build_host(gsum_ctx* ctx, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d, const double* Y, int64_t m,
                                double diag_add, const gsum_series_scale* sc, const double* ref_x, const double* ratio_x, const double* ref_y,
                                const double* ratio_y, double* out) {
    if (!ctx) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (gs_check_desc(ctx, desc, d)) return -2;
    if (!X || !out || n <= 0) GS_FAIL("bad argument");
    const bool cross = Y != nullptr;
    const int64_t cols = cross ? m : n;
    if (cols <= 0) GS_FAIL("bad argument");
    if (sc && (gs_check_series(ctx, sc) || !ref_x || !ratio_x || (cross && (!ref_y || !ratio_y)))) {
        if (ctx->err.empty()) ctx->err = "series scaling needs ref / ratio for both point sets";
        return -2;
    }
    const int64_t ldo = (cols + 1) / 2 * 2;
    const size_t xb = (size_t)n * d * sizeof(double), yb = cross ? (size_t)m * d * sizeof(double) : 0;
    const size_t ob = (size_t)n * ldo * sizeof(double), vb = sc ? (size_t)2 * (n + cols) * sizeof(double) : 0;
    const size_t off_y = (xb + 255) / 256 * 256, off_o = off_y + (yb + 255) / 256 * 256, off_v = off_o + (ob + 255) / 256 * 256;
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, off_v + vb)) return -1;
    char* base = (char*)ctx->scratch;
    double* dXl = (double*)base;
    double* dYl = (double*)(base + off_y);
    double* dO = (double*)(base + off_o);
    hipStream_t s = ctx->cur->sm;
    GS_CHECK(hipMemcpyAsync(dXl, X, xb, hipMemcpyHostToDevice, s));
    if (cross) GS_CHECK(hipMemcpyAsync(dYl, Y, yb, hipMemcpyHostToDevice, s));
    if (cross ? gs_launch_build<true>(ctx, s, dO, ldo, dXl, dYl, n, m, n, ldo, d, desc, 0.0, 0)
              : gs_launch_build<false>(ctx, s, dO, ldo, dXl, nullptr, n, n, n, ldo, d, desc, diag_add, 0))
        return -1;
    if (sc) {
        double* v = (double*)(base + off_v);
        double *d_ref_r = v, *d_rat_r = v + n, *d_ref_c = v + 2 * n, *d_rat_c = v + 2 * n + cols;
        GS_CHECK(hipMemcpyAsync(d_ref_r, ref_x, (size_t)n * 8, hipMemcpyHostToDevice, s));
        GS_CHECK(hipMemcpyAsync(d_rat_r, ratio_x, (size_t)n * 8, hipMemcpyHostToDevice, s));
        GS_CHECK(hipMemcpyAsync(d_ref_c, cross ? ref_y : ref_x, (size_t)cols * 8, hipMemcpyHostToDevice, s));
        GS_CHECK(hipMemcpyAsync(d_rat_c, cross ? ratio_y : ratio_x, (size_t)cols * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_scale_series, dim3((unsigned)((cols + 255) / 256), (unsigned)n), dim3(256), 0, s, dO, ldo, (int)n, (int)cols,
                           d_ref_r, d_rat_r, d_ref_c, d_rat_c, *sc);
        GS_CHECK(hipGetLastError());
    }
    GS_CHECK(hipMemcpy2DAsync(out, (size_t)cols * sizeof(double), dO, (size_t)ldo * sizeof(double),
                              (size_t)cols * sizeof(double), (size_t)n, hipMemcpyDeviceToHost, s));
    GS_CHECK(hipStreamSynchronize(s));
    return 0;
}

int gsum_kernel_build(gsum_ctx* ctx, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d,
                      const double* Y, int64_t m, double diag_add, double* out) {
    return gs_kernel_build_host(ctx, desc, X, n, d, Y, m, diag_add, nullptr, nullptr, nullptr, nullptr, nullptr, out);
}

int gsum_kernel_build_series(gsum_ctx* ctx, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d, const double* Y, int64_t m,
                             double diag_add, const gsum_series_scale* sc, const double* ref_x, const double* ratio_x,
                             const double* ref_y, const double* ratio_y, double* out) {
    if (!ctx || !sc) return -2;
    return gs_kernel_build_host(ctx, desc, X, n, d, Y, m, diag_add, sc, ref_x, ratio_x, ref_y, ratio_y, out);
}

int gsum_kernel_build_dev(gsum_ctx* ctx, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d,
                          double diag_add, gsum_mat** out) {
    if (!ctx || !out) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (gs_check_desc(ctx, desc, d)) return -2;
    if (gs_upload_X(ctx, &ctx->op, X, n, d)) return -1;
    gsum_mat* m = nullptr;
    if (gs_mat_alloc(ctx, n, &m)) return -1;
    if (gs_build_into(ctx, ctx->cur->sm, m, desc, ctx->op.X, d, diag_add, ctx->build_lower_only) ||
        gs_set_border(ctx, ctx->cur->sm, m, nullptr, 0)) {
        gs_mat_release(m);
        return -1;
    }
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    *out = m;
    return 0;
}

int gsum_mat_from_host(gsum_ctx* ctx, const double* Ah, int64_t n, gsum_mat** out) {
    if (!ctx || !out || !Ah) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    gsum_mat* m = nullptr;
    if (gs_mat_alloc(ctx, n, &m)) return -1;
    hipError_t e = hipMemcpy2DAsync(m->A, (size_t)m->ld * sizeof(double), Ah, (size_t)n * sizeof(double),
                                    (size_t)n * sizeof(double), (size_t)n, hipMemcpyHostToDevice, ctx->cur->sm);
    if (e == hipSuccess && m->np > n) {
        hipLaunchKernelGGL(k_pad_identity, dim3((unsigned)((m->np + 255) / 256), (unsigned)(m->np - n)), dim3(256), 0,
                           ctx->cur->sm, m->A, m->ld, (int)n, (int)m->np);
        e = hipGetLastError();
    }
    if (e != hipSuccess || gs_set_border(ctx, ctx->cur->sm, m, nullptr, 0)) {
        gs_mat_release(m);
        if (e != hipSuccess) ctx->err = std::string("upload failed: ") + hipGetErrorString(e);
        return -1;
    }
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    *out = m;
    return 0;
}

int gsum_potrf_lower(gsum_ctx* ctx, gsum_mat* A, int64_t* info) {
    if (!ctx || !A || !info) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (A->factored) GS_FAIL("matrix is already factorised");
    if (gs_potrf(ctx, A)) return -1;
    if (gs_finalize(ctx, A)) return -1;
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    *info = (int64_t)ctx->cur->hres[257];
    if (*info == GS_INFO_CHAIN_ABORT) {
        ctx->chain_persist = 0;
        ++ctx->chain_aborts;
        A->factored = false;
        ctx->err = "the persistent chain schedule timed out (streams of this process do not run side by side); the matrix is "
                   "destroyed -- rebuild it and factorise again: the schedule is now switched off (option chain_persist = 0)";
        return GSUM_ERR_CHAIN_ABORT;        // a runtime failure the caller can recover from: rebuild the matrix, factorise again
    }
    if (*info > A->n) *info = A->n;     // cannot happen (identity padding), kept as a guard
    A->factored = (*info == 0);
    return 0;
}

// Forward substitution on the border rows against an existing factor (right-looking, block by block):
//   W_c = Z_c L_cc^-T ;  Z[:, rest] -= W_c L[rest, c]^T ;  corner accumulates -W W^T.
static int gs_border_solve(gsum_ctx* ctx, gsum_mat* m) {
    const int64_t ld = m->ld, naug = m->np + GS_BORDER;
    double* A = m->A;
    double* Brow = A + m->np * ld;
    for (int k = 0; k < m->T; ++k) {
        const int64_t c0 = (int64_t)k * GS_NB, r0 = c0 + GS_NB;
        if (gs_trsm_rows(ctx, ctx->cur->sm, m, k, Brow + c0, ld, GS_BORDER)) return -1;
        if (gs_gemm(ctx, ctx->cur->sm, 2, Brow + r0, ld, Brow + c0, ld, A + r0 * ld + c0, ld, GS_BORDER, naug - r0, GS_NB, 0, 1,
                    -1.0))
            return -1;
    }
    return 0;
}

// border rows <- (L^-1 RHS)^T, corner <- -W^T W; skipped when the rows already hold the solve of the same RHS (predict is
// called again and again with the same training residual: T x 2 dependent launches saved per call)
static int gs_border_prepare(gsum_ctx* ctx, gsum_mat* L, const double* RHS, int64_t n, int k) {
    const size_t cnt = (size_t)n * k;
    if (L->solved_k == k && L->solved_rhs.size() == cnt && !memcmp(L->solved_rhs.data(), RHS, cnt * sizeof(double))) return 0;
    L->solved_k = -1;
    if (gs_upload_Z(ctx, &ctx->op, RHS, n, k)) return -1;
    if (gs_set_border(ctx, ctx->cur->sm, L, ctx->op.Z, k)) return -1;
    if (gs_border_solve(ctx, L)) return -1;
    L->solved_rhs.assign(RHS, RHS + cnt);
    L->solved_k = k;
    return 0;
}

int gsum_forward_gram(gsum_ctx* ctx, gsum_mat* L, const double* RHS, int64_t n, int32_t k, double* G,
                      double* sum_log_diag) {
    if (!ctx || !L || !G || !sum_log_diag) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (!L->factored) GS_FAIL("forward_gram needs a factorised matrix");
    if (n != L->n) GS_FAIL("RHS has the wrong number of rows");
    if (k < 1 || k > GSUM_MAX_RHS) GS_FAIL("k must be 1..GSUM_MAX_RHS");
    GS_CHECK(hipMemsetAsync(ctx->cur->dinfo, 0, sizeof(int), ctx->cur->sm));
    if (gs_border_prepare(ctx, L, RHS, n, k)) return -1;
    if (gs_finalize(ctx, L)) return -1;
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) G[i * k + j] = ctx->cur->hres[i * 16 + j];
    *sum_log_diag = ctx->cur->hres[256];
    return 0;
}

int gsum_forward_solve(gsum_ctx* ctx, gsum_mat* L, const double* RHS, int64_t n, int32_t k, double* W) {
    if (!ctx || !L || !W) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (!L->factored) GS_FAIL("forward_solve needs a factorised matrix");
    if (n != L->n) GS_FAIL("RHS has the wrong number of rows");
    if (k < 1 || k > GSUM_MAX_RHS) GS_FAIL("k must be 1..GSUM_MAX_RHS");
    if (gs_border_prepare(ctx, L, RHS, n, k)) return -1;
    std::vector<double> rows((size_t)k * n);
    GS_CHECK(hipMemcpy2DAsync(rows.data(), (size_t)n * sizeof(double), L->A + L->np * L->ld, (size_t)L->ld * sizeof(double),
                              (size_t)n * sizeof(double), (size_t)k, hipMemcpyDeviceToHost, ctx->cur->sm));
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < k; ++c) W[i * k + c] = rows[(size_t)c * n + i];
    return 0;
}

// scipy.linalg.cho_solve((L, True), B) = L^-T (L^-1 B): the forward half is gs_border_solve (border rows = W^T), the
// backward half runs right-looking from the last block column to the first, in place on the border rows:
//   X_c^T = W_c^T L_cc^-1 ;  W^T[:, cols < c0] -= X_c^T L[c rows, cols < c0]        (k_back_first / k_back_step)
int gsum_cho_solve(gsum_ctx* ctx, gsum_mat* L, const double* B, int64_t n, int32_t k, double* X) {
    if (!ctx || !L || !B || !X) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (!L->factored) GS_FAIL("cho_solve needs a factorised matrix");
    if (n != L->n) GS_FAIL("B has the wrong number of rows");
    if (k < 1 || k > GSUM_MAX_RHS) GS_FAIL("k must be 1..GSUM_MAX_RHS");
    hipStream_t s = ctx->cur->sm;
    if (gs_border_prepare(ctx, L, B, n, k)) return -1;
    L->solved_k = -1;                  // the back-substitution below overwrites the border rows in place
    if (gs_need_linv(ctx, s, L)) return -1;
    double* Brow = L->A + L->np * L->ld;
    const int T = L->T;
    hipLaunchKernelGGL(k_back_first, dim3(1), dim3(256), 0, s, Brow, L->ld, L->Linv + (size_t)(T - 1) * GS_NB * GS_NB, (T - 1) * GS_NB);
    GS_CHECK(hipGetLastError());
    for (int c = T - 1; c >= 1; --c) {
        hipLaunchKernelGGL(k_back_step, dim3((unsigned)c), dim3(256), 0, s, L->A, L->ld, Brow, L->Linv, c);
        GS_CHECK(hipGetLastError());
    }
    std::vector<double> rows((size_t)k * n);
    GS_CHECK(hipMemcpy2DAsync(rows.data(), (size_t)n * sizeof(double), Brow, (size_t)L->ld * sizeof(double),
                              (size_t)n * sizeof(double), (size_t)k, hipMemcpyDeviceToHost, s));
    GS_CHECK(hipStreamSynchronize(s));
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < k; ++c) X[i * k + c] = rows[(size_t)c * n + i];
    return 0;
}

int gsum_tri_multiply(gsum_ctx* ctx, gsum_mat* L, const double* Z, int64_t n, int32_t k, double* out) {
    if (!ctx || !L || !Z || !out) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (!L->factored) GS_FAIL("tri_multiply needs a factorised matrix");
    if (n != L->n) GS_FAIL("Z has the wrong number of rows");
    if (k < 1 || k > GSUM_MAX_RHS) GS_FAIL("k must be 1..GSUM_MAX_RHS");
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, (size_t)2 * n * 16 * 8)) return -1;
    double* dZ16 = ctx->scratch;
    double* dOut = dZ16 + (size_t)n * 16;
    hipStream_t s = ctx->cur->sm;
    std::vector<double> pad((size_t)n * 16, 0.0);
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < k; ++c) pad[(size_t)i * 16 + c] = Z[i * k + c];
    GS_CHECK(hipMemcpyAsync(dZ16, pad.data(), pad.size() * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_tri_multiply, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, L->A, L->ld, (int)n, dZ16, dOut);
    GS_CHECK(hipGetLastError());
    GS_CHECK(hipMemcpyAsync(pad.data(), dOut, pad.size() * 8, hipMemcpyDeviceToHost, s));
    GS_CHECK(hipStreamSynchronize(s));
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < k; ++c) out[i * k + c] = pad[(size_t)i * 16 + c];
    return 0;
}

// V^T = kernel(Xs, X) L^-T, one row per new point (m x np, row-major): the same right-looking sweep as
// the factorisation's panel step, with the rows of kernel(Xs, X) in the role of the rows below the panel.
static int gs_check_series(gsum_ctx* ctx, const gsum_series_scale* sc) {
    if (sc->start < 0 || (sc->end >= 0 && sc->end < sc->start)) GS_FAIL("series scale: end must be >= start >= 0");
    if (sc->n_excluded < 0 || sc->n_excluded > GSUM_MAX_EXCLUDED) GS_FAIL("series scale: too many excluded orders");
    return 0;
}

int gsum_mat_scale_series(gsum_ctx* ctx, gsum_mat* A, const gsum_series_scale* sc, const double* ref, const double* ratio) {
    if (!ctx || !A || !sc || !ref || !ratio) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (A->factored) GS_FAIL("scale_series needs an unfactored matrix");
    if (gs_check_series(ctx, sc)) return -2;
    const int64_t n = A->n;
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, (size_t)2 * n * 8)) return -1;
    double* dref = ctx->scratch;
    double* drat = dref + n;
    hipStream_t s = ctx->cur->sm;
    GS_CHECK(hipMemcpyAsync(dref, ref, (size_t)n * 8, hipMemcpyHostToDevice, s));
    GS_CHECK(hipMemcpyAsync(drat, ratio, (size_t)n * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_scale_series, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, s, A->A, A->ld, (int)n, (int)n,
                       dref, drat, dref, drat, *sc);
    GS_CHECK(hipGetLastError());
    GS_CHECK(hipStreamSynchronize(s));
    return 0;
}

static int gs_predict_terms(gsum_ctx* ctx, gsum_mat* L, const gsum_kernel_desc* desc, const double* X, int64_t n,
                            int32_t d, const double* Xs, int64_t m, const double* RHS, int32_t k,
                            const gsum_series_scale* sc, const double* ref_x, const double* ratio_x,
                            const double* ref_s, const double* ratio_s, double* colsumsq, double* VtW, double* cov_out);

int gsum_predict_terms(gsum_ctx* ctx, gsum_mat* L, const gsum_kernel_desc* desc, const double* X, int64_t n,
                       int32_t d, const double* Xs, int64_t m, const double* RHS, int32_t k,
                       double* colsumsq, double* VtW, double* cov_out) {
    return gs_predict_terms(ctx, L, desc, X, n, d, Xs, m, RHS, k, nullptr, nullptr, nullptr, nullptr, nullptr, colsumsq,
                            VtW, cov_out);
}

// SURVEY.md section 8b's name and signature for the predictive pieces: the right-hand sides are the ones the factor's border rows already hold
// (the last gsum_forward_gram / gsum_forward_solve / gsum_predict_terms on this factor).  The signature carries no k, so VtW is
// m x GSUM_MAX_RHS (row-major, columns >= k zero); VtW == NULL: the column sums of squares only
int gsum_predict_var(gsum_ctx* ctx, gsum_mat* L, const gsum_kernel_desc* desc, const double* X, int64_t n, int32_t d, const double* Xs,
                     int64_t m, double* colsumsq, double* VtW) {
    if (!ctx || !L) return -2;
    if (VtW && L->solved_k <= 0) GS_FAIL("gsum_predict_var: V^T W needs right-hand sides solved against this factor first (gsum_forward_gram)");
    const int k = VtW ? L->solved_k : 0;
    const std::vector<double> rhs = VtW ? L->solved_rhs : std::vector<double>();       // (a copy: gs_border_prepare compares against the original)
    std::vector<double> vw(VtW ? (size_t)m * k : 0);
    const int rc = gs_predict_terms(ctx, L, desc, X, n, d, Xs, m, k ? rhs.data() : nullptr, k, nullptr, nullptr, nullptr, nullptr, nullptr, colsumsq,
                                    k ? vw.data() : nullptr, nullptr);
    if (rc || !VtW) return rc;
    for (int64_t j = 0; j < m; ++j)
        for (int c = 0; c < GSUM_MAX_RHS; ++c) VtW[j * GSUM_MAX_RHS + c] = c < k ? vw[(size_t)j * k + c] : 0.0;
    return 0;
}

int gsum_predict_terms_series(gsum_ctx* ctx, gsum_mat* L, const gsum_kernel_desc* desc, const double* X, int64_t n,
                              int32_t d, const double* Xs, int64_t m, const double* RHS, int32_t k,
                              const gsum_series_scale* sc, const double* ref_x, const double* ratio_x,
                              const double* ref_s, const double* ratio_s, double* colsumsq, double* VtW,
                              double* cov_out) {
    if (!ctx || !sc || !ref_x || !ratio_x || !ref_s || !ratio_s) return -2;
    if (gs_check_series(ctx, sc)) return -2;
    return gs_predict_terms(ctx, L, desc, X, n, d, Xs, m, RHS, k, sc, ref_x, ratio_x, ref_s, ratio_s, colsumsq, VtW,
                            cov_out);
}

static int gs_predict_terms(gsum_ctx* ctx, gsum_mat* L, const gsum_kernel_desc* desc, const double* X, int64_t n,
                            int32_t d, const double* Xs, int64_t m, const double* RHS, int32_t k,
                            const gsum_series_scale* sc, const double* ref_x, const double* ratio_x,
                            const double* ref_s, const double* ratio_s, double* colsumsq, double* VtW, double* cov_out) {
    if (!ctx || !L || !X || !Xs || !colsumsq) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    if (gs_check_desc(ctx, desc, d)) return -2;
    if (!L->factored) GS_FAIL("predict_terms needs a factorised matrix");
    if (n != L->n || m <= 0) GS_FAIL("bad shapes");
    if (k < 0 || k > GSUM_MAX_RHS || (k > 0 && (!RHS || !VtW))) GS_FAIL("bad RHS / k");
    const int64_t np = L->np, ld = L->ld, ldb = np + GS_BORDER;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t o_xs = 0, o_bt = up((size_t)m * d * 8), o_vw = o_bt + up((size_t)m * ldb * 8),
                 o_ss = o_vw + up((size_t)m * 16 * 8), o_cv = o_ss + up((size_t)m * 8),
                 o_sc = o_cv + (cov_out ? up((size_t)m * m * 8) : 0),
                 total = o_sc + (sc ? up((size_t)2 * (n + m) * 8) : 0);
    if (gs_upload_X(ctx, &ctx->op, X, n, d)) return -1;
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, total)) return -1;
    char* base = (char*)ctx->scratch;
    double *dXs = (double*)(base + o_xs), *Bt = (double*)(base + o_bt), *dVW = (double*)(base + o_vw),
           *dSS = (double*)(base + o_ss), *dCov = (double*)(base + o_cv);
    GS_CHECK(hipMemcpyAsync(dXs, Xs, (size_t)m * d * 8, hipMemcpyHostToDevice, ctx->cur->sm));
    if (gs_launch_build<true>(ctx, ctx->cur->sm, Bt, ldb, dXs, ctx->op.X, m, n, m, np, d, desc, 0.0, 0)) return -1;
    if (sc) {
        // rows of Bt are the new points, columns the conditioning points
        double* v = (double*)(base + o_sc);
        double *d_ref_s = v, *d_rat_s = v + m, *d_ref_x = v + 2 * m, *d_rat_x = v + 2 * m + n;
        GS_CHECK(hipMemcpyAsync(d_ref_s, ref_s, (size_t)m * 8, hipMemcpyHostToDevice, ctx->cur->sm));
        GS_CHECK(hipMemcpyAsync(d_rat_s, ratio_s, (size_t)m * 8, hipMemcpyHostToDevice, ctx->cur->sm));
        GS_CHECK(hipMemcpyAsync(d_ref_x, ref_x, (size_t)n * 8, hipMemcpyHostToDevice, ctx->cur->sm));
        GS_CHECK(hipMemcpyAsync(d_rat_x, ratio_x, (size_t)n * 8, hipMemcpyHostToDevice, ctx->cur->sm));
        hipLaunchKernelGGL(k_scale_series, dim3((unsigned)((n + 255) / 256), (unsigned)m), dim3(256), 0, ctx->cur->sm, Bt, ldb, (int)m,
                           (int)n, d_ref_s, d_rat_s, d_ref_x, d_rat_x, *sc);
        GS_CHECK(hipGetLastError());
    }
    // LOOK-AHEAD SWEEP (round 5; m >= 1024 rows, large orders): pairs of block columns are steps, two steps a macro-step.  The context's chain
    // stream carries what the next step needs -- P(a) (both panels of pair a, k_panel256), N(a) (pair b's 256 columns, K = 256), P(b), then NB:
    // the next macro-step's 512 columns with both pairs at once (K = 512) --, the main stream ONE far launch per macro-step for everything right
    // of those (K = 512), which the chain stream only meets again a macro-step later: the 64 latency-bound panel launches of n = 16384 (17 % of
    // the one-stream sweep: the chip idles while 128 waves solve) run beside a far launch instead of between two.  Column block q receives the
    // far launches of all earlier macro-steps, then NB of the macro-step before its own, then N: ascending k per element, bit-identical.
    if (m >= 1024 && ctx->predict_lookahead && ctx->predict_lazy && ctx->predict_panel256 && np >= ctx->lazy_min_np && ctx->cur->sa && L->T >= 8) {
        const int T = L->T, S2 = (T + 1) / 2;
        hipStream_t sc = ctx->cur->sa, sf = ctx->cur->sm;
        auto col = [&](int st) { return std::min<int64_t>(2 * GS_NB * (int64_t)st, np); };
        if (gs_need_lsib(ctx, sf, L)) return -1;
        if (gs_potrf_events(ctx, ctx->cur, 2)) return -1;
        hipEvent_t evP = ctx->cur->evP[0], evF = ctx->cur->evP[1];
        GS_CHECK(hipEventRecord(ctx->cur->evFork, sf));
        GS_CHECK(hipStreamWaitEvent(sc, ctx->cur->evFork, 0));
        bool far_pending = false;
        // macro-steps of D pairs (option predict_depth; 2: K = 512 far launches, 4: K = 1024): inside one, pair a + i's near update N applies
        // all pairs of the macro-step so far to pair a + i + 1's 256 columns (K = 256 (i + 1)); NB and Far apply all D pairs
        const int D = std::max(2, std::min(8, ctx->predict_depth));
        for (int a = 0; a < S2; a += D) {
            const int64_t ca = col(a);
            int last = a;
            for (int q = 0; q < D && a + q < S2; ++q) {
                const int st = a + q;
                const int64_t cq = col(st);
                last = st;
                if (2 * st + 1 < T) { if (gs_panel256(ctx, sc, L, 2 * st, Bt + cq, ldb, m)) return -1; }
                else if (gs_trsm_rows(ctx, sc, L, 2 * st, Bt + cq, ldb, m)) return -1;
                if (q + 1 < D && st + 1 < S2) {            // N: the next pair's columns take every pair of the macro-step so far
                    const int64_t c1 = col(st + 1), c2 = col(st + 2);
                    if (gs_gemm(ctx, sc, 7, Bt + c1, ldb, Bt + ca, ldb, L->A + c1 * ld + ca, ld, m, c2 - c1, (int)(c1 - ca), 0, 1, -1.0)) return -1;
                }
            }
            const int64_t cn = col(last + 1), cf = col(last + 1 + D);      // the next macro-step's columns [cn, cf); far: [cf, np)
            if (cn >= np) break;
            GS_CHECK(hipEventRecord(evP, sc));                               // every pair of the macro-step is solved
            if (far_pending) GS_CHECK(hipStreamWaitEvent(sc, evF, 0));      // the previous far launch covered the columns NB is about to update
            if (gs_gemm(ctx, sc, 7, Bt + cn, ldb, Bt + ca, ldb, L->A + cn * ld + ca, ld, m, cf - cn, (int)(cn - ca), 0, 1, -1.0)) return -1;
            if (cf < np) {
                GS_CHECK(hipStreamWaitEvent(sf, evP, 0));
                if (gs_gemm(ctx, sf, GS_BULK, Bt + cf, ldb, Bt + ca, ldb, L->A + cf * ld + ca, ld, m, np - cf, (int)(cn - ca), 0, 1, -1.0)) return -1;
                GS_CHECK(hipEventRecord(evF, sf));
                far_pending = true;
            }
        }
        GS_CHECK(hipEventRecord(ctx->cur->evS, sc));
        GS_CHECK(hipStreamWaitEvent(sf, ctx->cur->evS, 0));
    } else {
    // V^T = kernel(Xs, X) L^-T by a right-looking sweep, two block columns per trailing update (K = 256) like the
    // factorisation: the trailing part of Bt is read and written once per 256 eliminated columns instead of once per 128
    // (at m = 2048, n = 16384 a K = 128 sweep moved 0.5 GB per step against 190 us of MFMA work).
    // The rows of Bt (the new points) never meet: from 1024 rows up the sweep runs as TWO independent half-sweeps on two of the
    // context's streams, so that the latency-bound panel launch of one half (64 of them on the critical path at n = 16384) runs
    // beside the trailing update of the other (round 5; one stream: 11.6 ms at n = 16384, m = 2048).
    const int sib_cfg = m >= 1024 ? GS_BULK : 1;
    const int n_half = (m >= 1024 && ctx->predict_split && ctx->cur->sa) ? 2 : 1;
    hipStream_t hs[2] = {ctx->cur->sm, ctx->cur->sa};
    if (n_half == 2) {
        if (gs_need_lsib(ctx, ctx->cur->sm, L)) return -1;                 // (before the fork: both halves read the images)
        GS_CHECK(hipEventRecord(ctx->cur->evFork, ctx->cur->sm));
        GS_CHECK(hipStreamWaitEvent(hs[1], ctx->cur->evFork, 0));
    }
    const int64_t m_lo = n_half == 2 ? (m / 2 + 127) / 128 * 128 : m;       // rows of the first half: whole 128-row tiles
    // The batch factorisation's grouping of trailing updates (gs_wave_bulk_plan) applied to this sweep: steps are pairs of block columns;
    // inside a macro-step of up to `predict_depth` pairs a step updates only the NEXT pair's 256 columns, with all pairs of the macro-step so
    // far at once (K = 256 (i + 1)); its last step applies all of them to everything to its right in ONE launch (K up to 1024) -- the
    // trailing part of Bt is read and written once per macro-step and the tile runs at its K = 1024 rate (65 TF/s alone against 58 at
    // K = 512, 47 at 256).  Round 3 paired (depth 2); round 5: depth 4.  Same products in the same ascending-k order per element.
    const int T = L->T, S2 = (T + 1) / 2;
    struct Step { bool near; int first; };
    std::vector<Step> plan((size_t)S2, Step{false, 0});
    {
        const int depth = (ctx->predict_lazy && m >= 1024 && np >= ctx->lazy_min_np) ? std::max(1, ctx->predict_depth) : 1;
        for (int a = 0; a < S2;) {
            int Lm = 1;
            while (Lm < depth && 2 * (a + Lm) + 1 < T && 2 * GS_NB * (int64_t)(a + Lm + 1) <= np) ++Lm;      // step a + Lm is a full pair
            if (Lm > 2 && np - 2 * GS_NB * (int64_t)(a + 1) < ctx->wave_deep_rows) Lm = 2;
            for (int q = 0; q < Lm; ++q) plan[(size_t)(a + q)] = Step{q < Lm - 1, a};
            a += Lm;
        }
    }
    for (int c = 0; c < T; c += 2) {
        const bool two = c + 1 < T;
        const int64_t c0 = (int64_t)c * GS_NB, c1 = c0 + GS_NB, r2 = two ? c1 + GS_NB : c1;
        const Step st = plan[(size_t)(c / 2)];
        const int64_t cp = 2 * GS_NB * (int64_t)st.first;          // the macro-step's first column: [cp, r2) are its panels so far
        for (int h = 0; h < n_half; ++h) {
            hipStream_t sth = hs[h];
            const int64_t row0 = h == 0 ? 0 : m_lo, mh = n_half == 1 ? m : (h == 0 ? m_lo : m - m_lo);
            double* Bh = Bt + row0 * ldb;
            if (mh <= 0) continue;
            if (two && ctx->predict_panel256) {
                // both panels of the pair and the sibling update between them in ONE launch, 16 rows per wave (k_panel256: the factorisation's
                // own panel step; rounds 1-4 ran k_panel, a K = 128 GEMM and k_panel again here: 3 dependent launches per pair, 896 launches
                // of ~12 us on the sweep's critical path at n = 16384)
                if (gs_need_lsib(ctx, sth, L)) return -1;
                if (gs_panel256(ctx, sth, L, c, Bh + c0, ldb, mh)) return -1;
            } else {
                if (gs_trsm_rows(ctx, sth, L, c, Bh + c0, ldb, mh)) return -1;
                if (two) {
                    if (gs_gemm(ctx, sth, sib_cfg, Bh + c1, ldb, Bh + c0, ldb, L->A + c1 * ld + c0, ld, mh, GS_NB, GS_NB, 0, 1, -1.0)) return -1;
                    if (gs_trsm_rows(ctx, sth, L, c + 1, Bh + c1, ldb, mh)) return -1;
                }
            }
            if (r2 >= np) continue;
            const int64_t ncols = st.near ? 2 * GS_NB : np - r2;
            if (gs_gemm(ctx, sth, GS_BULK, Bh + r2, ldb, Bh + cp, ldb, L->A + r2 * ld + cp, ld, mh, ncols, (int)(r2 - cp), 0, 1, -1.0)) return -1;
        }
    }
    if (n_half == 2) {
        GS_CHECK(hipEventRecord(ctx->cur->evS, hs[1]));
        GS_CHECK(hipStreamWaitEvent(ctx->cur->sm, ctx->cur->evS, 0));
    }
    }
    std::vector<double> vw;
    if (k > 0) {
        // row sums of squares and V^T W in ONE pass over V^T (k_rowsumsq_vw)
        if (gs_border_prepare(ctx, L, RHS, n, k)) return -1;
        hipLaunchKernelGGL(k_rowsumsq_vw, dim3((unsigned)((m + 4 * GS_VW_ROWS - 1) / (4 * GS_VW_ROWS))), dim3(256), 0, ctx->cur->sm, Bt, ldb, (int)m, (int)np,
                           L->A + np * ld, ld, dSS, dVW);
        GS_CHECK(hipGetLastError());
        vw.resize((size_t)m * 16);
        GS_CHECK(hipMemcpyAsync(vw.data(), dVW, (size_t)m * 16 * 8, hipMemcpyDeviceToHost, ctx->cur->sm));
    } else {
        hipLaunchKernelGGL(k_rowsumsq, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->cur->sm, Bt, ldb, (int)m, (int)np, dSS);
        GS_CHECK(hipGetLastError());
    }
    GS_CHECK(hipMemcpyAsync(colsumsq, dSS, (size_t)m * 8, hipMemcpyDeviceToHost, ctx->cur->sm));
    if (cov_out) {
        // V^T V is symmetric: lower tiles only (half the flops of the square product), then mirrored in place
        if (gs_gemm(ctx, ctx->cur->sm, GS_BULK, dCov, m, Bt, ldb, Bt, ldb, m, m, (int)np, 1, 0, 1.0)) return -1;
        hipLaunchKernelGGL(k_mirror_lower, dim3((unsigned)((m + 255) / 256), (unsigned)m), dim3(256), 0, ctx->cur->sm, dCov, m, (int)m);
        GS_CHECK(hipGetLastError());
        GS_CHECK(hipMemcpyAsync(cov_out, dCov, (size_t)m * m * 8, hipMemcpyDeviceToHost, ctx->cur->sm));
    }
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    for (int64_t j = 0; j < m && k > 0; ++j)
        for (int c = 0; c < k; ++c) VtW[j * k + c] = vw[(size_t)j * 16 + c];
    return 0;
}

int gsum_mat_to_host(gsum_ctx* ctx, const gsum_mat* A, double* out) {
    if (!ctx || !A || !out) return -2;
    GS_CHECK(hipSetDevice(ctx->device));
    const int64_t n = A->n;
    if (gs_reserve(ctx, &ctx->scratch, &ctx->scratch_cap, (size_t)n * n * sizeof(double))) return -1;
    hipLaunchKernelGGL(k_export, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, ctx->cur->sm, A->A, A->ld, (int)n,
                       ctx->scratch, A->factored ? 1 : 0);
    GS_CHECK(hipGetLastError());
    GS_CHECK(hipMemcpyAsync(out, ctx->scratch, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, ctx->cur->sm));
    GS_CHECK(hipStreamSynchronize(ctx->cur->sm));
    return 0;
}

int64_t gsum_mat_n(const gsum_mat* A) { return A ? A->n : -1; }

void gsum_mat_free(gsum_ctx* ctx, gsum_mat* A) {
    if (!A) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->cur->sm);
        if (ctx->cur->sp) (void)hipStreamSynchronize(ctx->cur->sp);
    }
    gs_mat_release(A);
}

