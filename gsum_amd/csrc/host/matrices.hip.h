// device matrices, descriptor checks, kernel-matrix build, border rows
// (part of gsum_capi.hip: included from there, in order -- one translation unit)
#pragma once
// ---- matrices ---------------------------------------------------------------------------------
// bulk update of a look-ahead schedule: the one launch class that asks for `bulk_lds_pad` bytes of LDS (two workgroups per
// CU, so that chain workgroups find room as soon as one retires); every other user of the bulk tile wants three per CU
static int gs_bulk_la(gsum_ctx* ctx, hipStream_t s, int cfg, double* C, int64_t ldc, const double* A, int64_t lda, const double* B,
                      int64_t ldb, int64_t M, int64_t N, int K, int tri, int beta, double sign) {
    ctx->bulk_pad_now = true;
    const int rc = gs_gemm(ctx, s, cfg, C, ldc, A, lda, B, ldb, M, N, K, tri, beta, sign);
    ctx->bulk_pad_now = false;
    return rc;
}

// Padded order of a matrix (identity padding: exact zeros in every product, so results do not depend on it).  128 is the block
// size; where the persistent-chain schedule applies the order goes to the next multiple of 256 instead -- an even number of block
// columns -- so that schedule serves every order, not only multiples of 256 (n = 7976: 8192 instead of 8064 rows, +1.6 % work for a
// factorisation that is 15 % faster).
static int64_t gs_padded_order(const gsum_ctx* ctx, int64_t n) {
    const int64_t p128 = (n + GS_NB - 1) / GS_NB * GS_NB, p256 = (n + 2 * GS_NB - 1) / (2 * GS_NB) * (2 * GS_NB);
    if (ctx->chain_persist != 0 && p256 >= ctx->chain_min_np) return p256;
    return p128;
}

// orders the factorisation paths accept (see gs_mat_alloc)
static int gs_check_order(gsum_ctx* ctx, int64_t n) {
    if (n <= 0 || (n + 2 * GS_NB) * (n + 2 * GS_NB + GS_BORDER) >= (int64_t)1 << 31) GS_FAIL("matrix order out of range (1 .. 46000)");
    return 0;
}

static int gs_mat_alloc(gsum_ctx* ctx, int64_t n, gsum_mat** out) {
    // validated up to n = 40960 (tools/gpu_large_order.py: build + factorisation + solve reproduce K[cols, cols] to 2e-15 at 24576 / 32768 / 40960,
    // 57-60 TF/s); above 46 000 the padded matrix has 2^31 elements or more, which no test has exercised: refused rather than trusted
    if (gs_check_order(ctx, n)) return -2;
    gsum_mat* m = new gsum_mat();
    m->n = n;
    m->np = gs_padded_order(ctx, n);
    m->ld = GS_LD(m->np);               // (see GS_LD_EXTRA: not np + 16)
    m->T = (int)(m->np / GS_NB);
    hipError_t e = hipMalloc((void**)&m->A, (size_t)(m->np + GS_BORDER) * m->ld * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&m->Linv, (size_t)m->T * GS_NB * GS_NB * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&m->Ltab, (size_t)m->T * GS_LTAB * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&m->Lsib, (size_t)(m->T / 2 + 1) * GS_LSIB * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&m->logdet, (size_t)m->T * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&m->diag0, (size_t)m->np * sizeof(double));
    if (e != hipSuccess) {
        for (double** q : {&m->A, &m->Linv, &m->Ltab, &m->Lsib, &m->logdet, &m->diag0}) {
            if (*q) (void)hipFree(*q);
            *q = nullptr;
        }
        delete m;
        ctx->err = std::string("hipMalloc(matrix) failed: ") + hipGetErrorString(e);
        return -1;
    }
    *out = m;
    return 0;
}

static void gs_mat_release(gsum_mat* m) {
    if (!m) return;
    if (m->A) (void)hipFree(m->A);
    if (m->Linv) (void)hipFree(m->Linv);
    if (m->Ltab) (void)hipFree(m->Ltab);
    if (m->Lsib) (void)hipFree(m->Lsib);
    if (m->logdet) (void)hipFree(m->logdet);
    if (m->diag0) (void)hipFree(m->diag0);
    if (m->cflags) (void)hipFree(m->cflags);
    if (m->cdump) (void)hipFree(m->cdump);
    if (m->cstamps) (void)hipFree(m->cstamps);
    delete m;
}

static int gs_check_desc(gsum_ctx* ctx, const gsum_kernel_desc* desc, int d) {
    if (!desc) GS_FAIL("kernel descriptor is NULL");
    if (d < 1 || d > GSUM_MAX_D) GS_FAIL("input dimension must be 1..GSUM_MAX_D");
    if (desc->n_ops == 0) {
        if (desc->family < GSUM_RBF || desc->family > GSUM_MATERN12) GS_FAIL("unknown kernel family");
        int nls = desc->anisotropic ? d : 1;
        for (int i = 0; i < nls; ++i)
            if (!(desc->length_scale[i] > 0.0)) GS_FAIL("length_scale must be positive");
        return 0;
    }
    // a tree: a well-formed postfix program over valid leaves
    if (desc->n_ops < 0 || desc->n_ops > GSUM_MAX_OPS || desc->n_leaves < 0 || desc->n_leaves > GSUM_MAX_LEAVES) GS_FAIL("kernel tree: bad op / leaf count");
    int depth = 0;
    for (int k = 0; k < desc->n_ops; ++k) {
        const int op = desc->op[k];
        if (op == GSUM_OP_ADD || op == GSUM_OP_MUL) {
            if (depth < 2) GS_FAIL("kernel tree: operator without two operands");
            --depth;
        } else if (op >= GSUM_OP_POW) {
            if (depth < 1 || op - GSUM_OP_POW >= GSUM_MAX_OPS) GS_FAIL("kernel tree: exponentiation without an operand or with a bad exponent slot");
        } else {
            const int idx = op >= GSUM_OP_WHITE ? op - GSUM_OP_WHITE : (op >= GSUM_OP_CONST ? op - GSUM_OP_CONST : op - GSUM_OP_LEAF);
            if (op < GSUM_OP_LEAF || idx < 0 || idx >= (op >= GSUM_OP_CONST ? GSUM_MAX_OPS : desc->n_leaves)) GS_FAIL("kernel tree: bad operand");
            if (++depth > 8) GS_FAIL("kernel tree: deeper than 8 pending operands");
        }
    }
    if (depth != 1) GS_FAIL("kernel tree: the program does not reduce to one value");
    for (int l = 0; l < desc->n_leaves; ++l) {
        const gsum_kernel_leaf& lf = desc->leaf[l];
        if (lf.family < GSUM_RBF || lf.family > GSUM_DOT) GS_FAIL("kernel tree: unknown leaf family");
        if (lf.family == GSUM_DOT) {
            if (lf.anisotropic || !(lf.length_scale[0] >= 0.0)) GS_FAIL("kernel tree: DotProduct needs sigma_0 >= 0");
            continue;
        }
        if (lf.family == GSUM_EXPSINE && (lf.anisotropic || !(lf.alpha > 0.0))) GS_FAIL("kernel tree: ExpSineSquared needs periodicity > 0 and an isotropic length scale");
        if (lf.family == GSUM_RQ && (lf.anisotropic || !(lf.alpha > 0.0))) GS_FAIL("kernel tree: RationalQuadratic needs alpha > 0 and an isotropic length scale");
        for (int i = 0; i < (lf.anisotropic ? d : 1); ++i)
            if (!(lf.length_scale[i] > 0.0)) GS_FAIL("length_scale must be positive");
    }
    return 0;
}

// Kernel-matrix build launcher: picks the template instance (family, one-dimensional fast path) of k_build2.
// tri != 0: lower 128-column tiles of a square padded matrix only.
template <bool CROSS>
static int gs_launch_build(gsum_ctx* ctx, hipStream_t s, double* out, int64_t ldo, const double* X, const double* Y, int64_t n,
                           int64_t m, int64_t prow, int64_t pcol, int d, const gsum_kernel_desc* desc, double diag_add, int tri) {
    const int64_t tr = (prow + GS_B2_ROWS - 1) / GS_B2_ROWS, tc = (pcol + 127) / 128, t128 = (prow + 127) / 128;
    const int64_t blocks = tri ? 4 * (t128 * (t128 + 1) / 2) : tr * tc;
    if (desc->n_ops > 0) {                      // a general Sum / Product tree: entry-by-entry evaluation (k_build_tree)
        hipLaunchKernelGGL((k_build_tree<CROSS>), dim3((unsigned)blocks), dim3(256), 0, s, out, ldo, X, Y, (int)n, (int)m, (int)prow,
                           (int)pcol, d, *desc, diag_add, tri);
        GS_CHECK(hipGetLastError());
        return 0;
    }
#define GS_B2_LAUNCH(FAM, D1)                                                                                              \
    hipLaunchKernelGGL((k_build2<CROSS, FAM, D1>), dim3((unsigned)blocks), dim3(256), 0, s, out, ldo, X, Y, (int)n, (int)m, \
                       (int)prow, (int)pcol, d, *desc, diag_add, tri)
    const bool d1 = d == 1;
    switch (desc->family) {
        case GSUM_RBF: if (d1) GS_B2_LAUNCH(GSUM_RBF, true); else GS_B2_LAUNCH(GSUM_RBF, false); break;
        case GSUM_MATERN52: if (d1) GS_B2_LAUNCH(GSUM_MATERN52, true); else GS_B2_LAUNCH(GSUM_MATERN52, false); break;
        case GSUM_MATERN32: if (d1) GS_B2_LAUNCH(GSUM_MATERN32, true); else GS_B2_LAUNCH(GSUM_MATERN32, false); break;
        default: if (d1) GS_B2_LAUNCH(GSUM_MATERN12, true); else GS_B2_LAUNCH(GSUM_MATERN12, false); break;
    }
#undef GS_B2_LAUNCH
    GS_CHECK(hipGetLastError());
    return 0;
}

// K1 into an augmented matrix (square, symmetric form).  X must already be on the device.
static int gs_build_into(gsum_ctx* ctx, hipStream_t s, gsum_mat* m, const gsum_kernel_desc* desc, const double* dX,
                         int d, double diag_add, int lower_only) {
    const int rec = gs_prof_begin(ctx, s, GS_PROF_BUILD, 0.0);
    const int rc = gs_launch_build<false>(ctx, s, m->A, m->ld, dX, nullptr, m->n, m->n, m->np, m->np, d, desc, diag_add, lower_only);
    gs_prof_end(ctx, s, rec);
    if (rc) return rc;
    m->factored = false;
    return 0;
}

static int gs_set_border(gsum_ctx* ctx, hipStream_t s, gsum_mat* m, const double* dZ, int k) {
    int64_t cols = m->np + GS_BORDER;
    const int rec = gs_prof_begin(ctx, s, GS_PROF_OTHER, 0.0);
    hipLaunchKernelGGL(k_set_border, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s, m->A, m->ld, (int)m->n,
                       (int)m->np, dZ, k);
    gs_prof_end(ctx, s, rec);
    GS_CHECK(hipGetLastError());
    return 0;
}

