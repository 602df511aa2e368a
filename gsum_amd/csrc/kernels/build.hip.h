// K1: the kernel-matrix build (numpy's exp restated, flattened descriptors, Sum / Product trees), border rows, small helpers
// (part of gsum_kernels.hip.h: included from there, in order; gfx950 only)
#pragma once
// ------------------------------------------------------------------------------------------------
// K1: kernel matrix build
// ------------------------------------------------------------------------------------------------
// sklearn arithmetic, restated (sklearn/gaussian_process/kernels.py):
//   RBF     :1556-1560  exp(-0.5 * sqeuclidean(X/ls)), diagonal forced to 1
//   Matern  :1711-1738  d = euclidean(X/ls); nu=.5: exp(-d); 1.5: t=d*sqrt3, (1+t)exp(-t);
//                       2.5: t=d*sqrt5, (1+t+t*t/3)exp(-t)
//   Product :956-966 (amplitude * base), Sum :858-866 (+ constant), White :1401-1414 (+ noise on diag)
// Floating-point contraction is off so that sums of squares round like the host code does.
//
// exp: numpy's float64 exp on AVX512 hosts (the reference's CPU path, and the GPU box's own host) is
// Intel SVML's __svml_exp8_ha.  It is NOT correctly rounded — e.g. exp(-0.125) comes out 1 ulp low —
// and on a uniform grid one such value fills a whole diagonal of K: that single ulp moves the S3
// log-likelihood by 6e-10.  gs_exp_np therefore restates the published structure of that
// routine operation for operation (Tang-style: N = floor_{1/16}(x log2 e), two-step Cody-Waite
// reduction, 16-entry 2^(j/16) table with tail, degree-6 polynomial, every step one IEEE fma), so
// kernel-matrix entries come out bit-identical to sklearn's.  Checked bit-for-bit against np.exp on 13k
// arguments by tests/test_exp_restatement.py (CPU) and on the device by tests/test_gpu_parity.py.
// |x| >= 707.7 (results below 4.6e-308 or overflow) takes SVML's scalar "rare" path on the host; here it
// falls through to the device library's exp: such entries are < 1e-307 against a unit diagonal.
__device__ __constant__ double gs_exp_th[16] = {
    0x1.0000000000000p+0, 0x1.0b5586cf9890fp+0, 0x1.172b83c7d517bp+0, 0x1.2387a6e756238p+0,
    0x1.306fe0a31b715p+0, 0x1.3dea64c123422p+0, 0x1.4bfdad5362a27p+0, 0x1.5ab07dd485429p+0,
    0x1.6a09e667f3bcdp+0, 0x1.7a11473eb0187p+0, 0x1.8ace5422aa0dbp+0, 0x1.9c49182a3f090p+0,
    0x1.ae89f995ad3adp+0, 0x1.c199bdd85529cp+0, 0x1.d5818dcfba487p+0, 0x1.ea4afa2a490dap+0};
__device__ __constant__ double gs_exp_tl[16] = {
    0x0.0p+0, 0x1.79aa65d837b6dp-54, -0x1.01b15eaa59348p-55, 0x1.68efde3a8a894p-54,
    0x1.34d754db0abb6p-55, 0x1.59f48a72a4c6dp-55, 0x1.690cebb7aafb0p-56, 0x1.063e1e21c5409p-54,
    -0x1.3b3efbf5e2228p-54, -0x1.b32dcb94da51dp-56, 0x1.db72fc1f0eab4p-55, 0x1.1affc2b91ce27p-56,
    0x1.c1a7792cb3387p-55, 0x1.36eae30af0cb3p-56, 0x1.4a385a63d07a7p-56, -0x1.ff7128fd391f0p-55};

__device__ __forceinline__ double gs_exp_np(double x) {
#pragma clang fp contract(off)
    if (!(fabs(x) < 0x1.61da04cbafe44p+9)) return exp(x);
    const double L2E = 0x1.71547652b82fep+0, SH = 0x1.8000000003ff0p+48;
    const double L2H = 0x1.62e42fefa39efp-1, L2L = 0x1.abc9e3b39803fp-56;
    // M = RZ(x*L2E + SH): round-toward-zero of a positive sum = floor on the 1/16 grid.  Emulated with a
    // round-to-nearest fma and an exact sign test of the residual.
    const double t = __builtin_fma(x, L2E, SH);
    const double nn = t - SH;
    const double dd = __builtin_fma(x, L2E, -nn);
    const double N = dd < 0.0 ? nn - 0.0625 : nn;
    const long long k16 = (long long)(N * 16.0);
    const int j = (int)(k16 & 15);
    double R = __builtin_fma(-N, L2H, x);
    R = __builtin_fma(-N, L2L, R);
    const double R2 = R * R;
    const double pA = __builtin_fma(0x1.7411836940c04p-10, R, 0x1.1101cbbc265c0p-7);
    const double pB = __builtin_fma(0x1.55557242d68fep-5, R, 0x1.5555553939732p-3);
    const double pC = __builtin_fma(0x1.000000000d008p-1, R, 0x1.fffffffffff70p-1);
    double pp = __builtin_fma(R2, pA, pB);
    pp = __builtin_fma(R2, pp, pC);
    const double q = __builtin_fma(pp, R, gs_exp_tl[j]);
    const double th = gs_exp_th[j];
    const double res = __builtin_fma(th, q, th);
    return ldexp(res, (int)(k16 >> 4));
}

__device__ __forceinline__ double gs_base_value(int family, double s) {
#pragma clang fp contract(off)
    if (family == GSUM_RBF) return gs_exp_np(-0.5 * s);
    double dist = sqrt(s);
    if (family == GSUM_MATERN52) {
        double t = dist * 2.23606797749979;      // math.sqrt(5)
        return (1.0 + t + (t * t) / 3.0) * gs_exp_np(-t);
    }
    if (family == GSUM_MATERN32) {
        double t = dist * 1.7320508075688772;    // math.sqrt(3)
        return (1.0 + t) * gs_exp_np(-t);
    }
    return gs_exp_np(-dist);
}


// ---- second version of the kernel build (round 2) ---------------------------------------------------------------
// The first version was bound by instruction issue, not by HBM: a runtime switch over the kernel family and a runtime
// loop over the input dimensions inside the per-entry loop, the 2^(j/16) table read through divergent global loads from
// constant memory, a 64-bit float -> integer conversion, and diagonal / padding tests on every entry: ~90 VALU
// instructions per entry, 89 us for the 34 M entries of the n = 8192 lower triangle (3.1 TB/s of stores).  Here the
// family and the one-dimensional case are template parameters, the table lives in LDS, the exponent goes through
// v_cvt_i32_f64, tiles that touch neither the diagonal nor the padding skip every test, and a workgroup takes 32 x 128
// entries (8320 tiles at n = 8192: 4 rounds of the 2048 resident workgroups instead of 1.02 with a one-tile tail).
// Same arithmetic, operation for operation (array_equal to scikit-learn is asserted on the device for every family).
__device__ __forceinline__ double gs_exp_np_t(double x, const double* th, const double* tl) {
#pragma clang fp contract(off)
    if (!(fabs(x) < 0x1.61da04cbafe44p+9)) return exp(x);
    const double L2E = 0x1.71547652b82fep+0, SH = 0x1.8000000003ff0p+48;
    const double L2H = 0x1.62e42fefa39efp-1, L2L = 0x1.abc9e3b39803fp-56;
    const double t = __builtin_fma(x, L2E, SH);
    const double nn = t - SH;
    const double dd = __builtin_fma(x, L2E, -nn);
    const double N = dd < 0.0 ? nn - 0.0625 : nn;
    const int k16 = (int)(N * 16.0);                      // |N| < 1022: exact in 32 bits
    const int j = k16 & 15;
    double R = __builtin_fma(-N, L2H, x);
    R = __builtin_fma(-N, L2L, R);
    const double R2 = R * R;
    const double pA = __builtin_fma(0x1.7411836940c04p-10, R, 0x1.1101cbbc265c0p-7);
    const double pB = __builtin_fma(0x1.55557242d68fep-5, R, 0x1.5555553939732p-3);
    const double pC = __builtin_fma(0x1.000000000d008p-1, R, 0x1.fffffffffff70p-1);
    double pp = __builtin_fma(R2, pA, pB);
    pp = __builtin_fma(R2, pp, pC);
    const double q = __builtin_fma(pp, R, tl[j]);
    const double thj = th[j];
    const double res = __builtin_fma(thj, q, thj);
    return ldexp(res, k16 >> 4);
}

template <int FAM>
__device__ __forceinline__ double gs_base_value_t(double s, const double* th, const double* tl) {
#pragma clang fp contract(off)
    if (FAM == GSUM_RBF) return gs_exp_np_t(-0.5 * s, th, tl);
    const double dist = sqrt(s);
    if (FAM == GSUM_MATERN52) {
        const double t = dist * 2.23606797749979;      // math.sqrt(5)
        return (1.0 + t + (t * t) / 3.0) * gs_exp_np_t(-t, th, tl);
    }
    if (FAM == GSUM_MATERN32) {
        const double t = dist * 1.7320508075688772;    // math.sqrt(3)
        return (1.0 + t) * gs_exp_np_t(-t, th, tl);
    }
    return gs_exp_np_t(-dist, th, tl);
}

// Branch-free form for the build kernel's inner loop.  Two argument ranges need no table arithmetic at all:
//   x < -745.2       exp(x) is exactly 0.0 in fp64 (below half the smallest denormal) -- on a grid with dx = 0.5 l that
//                    is every entry more than 39 length scales from the diagonal, i.e. most of a large matrix;
//   |x| < 707.7      the table algorithm (gs_exp_np_t's fast path).
// What is left (the band -745.2 <= x <= -707.7 where the result is a denormal, overflow, NaN) is flagged and recomputed
// by the caller with the library exp, wave-uniformly, so that the common paths carry no per-entry branch.
__device__ __forceinline__ double gs_exp_np_nobranch(double x, const double* th, const double* tl, bool& slow) {
#pragma clang fp contract(off)
    const bool far = x < -745.2;
    const bool inr = fabs(x) < 0x1.61da04cbafe44p+9;
    slow = !(far || inr);
    const double xs = inr ? x : 0.0;
    const double L2E = 0x1.71547652b82fep+0, SH = 0x1.8000000003ff0p+48;
    const double L2H = 0x1.62e42fefa39efp-1, L2L = 0x1.abc9e3b39803fp-56;
    const double t = __builtin_fma(xs, L2E, SH);
    const double nn = t - SH;
    const double dd = __builtin_fma(xs, L2E, -nn);
    const double N = dd < 0.0 ? nn - 0.0625 : nn;
    const int k16 = (int)(N * 16.0);
    const int j = k16 & 15;
    double R = __builtin_fma(-N, L2H, xs);
    R = __builtin_fma(-N, L2L, R);
    const double R2 = R * R;
    const double pA = __builtin_fma(0x1.7411836940c04p-10, R, 0x1.1101cbbc265c0p-7);
    const double pB = __builtin_fma(0x1.55557242d68fep-5, R, 0x1.5555553939732p-3);
    const double pC = __builtin_fma(0x1.000000000d008p-1, R, 0x1.fffffffffff70p-1);
    double pp = __builtin_fma(R2, pA, pB);
    pp = __builtin_fma(R2, pp, pC);
    const double q = __builtin_fma(pp, R, tl[j]);
    const double thj = th[j];
    const double res = ldexp(__builtin_fma(thj, q, thj), k16 >> 4);
    return far ? 0.0 : res;
}

// base value with the exp argument's class reported: far = the exponential is exactly zero
template <int FAM>
__device__ __forceinline__ double gs_base_value_nb(double s, const double* th, const double* tl, bool& slow) {
#pragma clang fp contract(off)
    if (FAM == GSUM_RBF) return gs_exp_np_nobranch(-0.5 * s, th, tl, slow);
    const double dist = sqrt(s);
    if (FAM == GSUM_MATERN52) {
        const double t = dist * 2.23606797749979;
        return (1.0 + t + (t * t) / 3.0) * gs_exp_np_nobranch(-t, th, tl, slow);
    }
    if (FAM == GSUM_MATERN32) {
        const double t = dist * 1.7320508075688772;
        return (1.0 + t) * gs_exp_np_nobranch(-t, th, tl, slow);
    }
    return gs_exp_np_nobranch(-dist, th, tl, slow);
}

// is the exponential of this squared scaled distance exactly zero?  (the argument of exp is -0.5 s, -sqrt(5 s), ...)
template <int FAM>
__device__ __forceinline__ bool gs_base_is_zero(double s) {
    if (FAM == GSUM_RBF) return s > 1490.5;                       // -0.5 s < -745.25
    if (FAM == GSUM_MATERN52) return s > 111100.0;                // sqrt(5 s) > 745.3
    if (FAM == GSUM_MATERN32) return s > 185200.0;                // sqrt(3 s) > 745.4
    return s > 555500.0;                                          // sqrt(s) > 745.3
}

// run-time family, exp tables in LDS (the fused one-workgroup kernels: the family is uniform over the workgroup, and a table
// in LDS costs an LDS read per entry where the __constant__ one of gs_exp_np costs a vector load from memory)
__device__ __forceinline__ double gs_base_value_f(int family, double s, const double* th, const double* tl) {
    if (family == GSUM_RBF) return gs_base_value_t<GSUM_RBF>(s, th, tl);
    if (family == GSUM_MATERN52) return gs_base_value_t<GSUM_MATERN52>(s, th, tl);
    if (family == GSUM_MATERN32) return gs_base_value_t<GSUM_MATERN32>(s, th, tl);
    return gs_base_value_t<GSUM_MATERN12>(s, th, tl);
}

// One 128 x 128 tile of the kernel matrix for the fused one-workgroup kernels (256 threads; ui / uj: the scaled coordinates
// of the tile's rows / columns in LDS; th / tl: the exp tables in LDS): k_build2's per-entry arithmetic and its short cuts
// -- family and the one-dimensional case as template parameters, tiles that touch neither the diagonal nor the padding
// without per-entry tests, two rows per pass -- instead of the round-1 loop (runtime family switch and dimension loop,
// diagonal and padding tests on every entry: ~90 VALU instructions per entry, 17-21 % of k_lml_medium at n = 512 ... 1024).
// Wave w takes rows w, w + 4, ... ; each lane two adjacent columns.  The diagonal's values also go to diag0.
template <int FAM, bool D1>
__device__ __forceinline__ void gs_build_tile128(double* A, int64_t ld, const double* ui, const double* uj, const double* th,
                                                 const double* tl, int bi, int bj, int n, int d, const gsum_kernel_desc& desc,
                                                 double diag_add, double* diag0, int w, int lane) {
#pragma clang fp contract(off)
    double vj0[D1 ? 1 : GSUM_MAX_D], vj1[D1 ? 1 : GSUM_MAX_D];
    if (D1) {
        vj0[0] = uj[2 * lane];
        vj1[0] = uj[2 * lane + 1];
    } else {
#pragma unroll
        for (int dd = 0; dd < GSUM_MAX_D; ++dd) {
            vj0[dd] = dd < d ? uj[(2 * lane) * d + dd] : 0.0;
            vj1[dd] = dd < d ? uj[(2 * lane + 1) * d + dd] : 0.0;
        }
    }
    const int r0 = bi * 128, c0 = bj * 128, gj0 = c0 + 2 * lane;
    const bool plain = bi != bj && r0 + 128 <= n && c0 + 128 <= n;
    const double amp = desc.amplitude, addc = desc.additive_const;
#pragma unroll 1
    for (int rp = w; rp < 128; rp += 8) {
        double s[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rr = rp + 4 * h;
            if (D1) {
                const double xi = ui[rr];
                const double e0 = xi - vj0[0], e1 = xi - vj1[0];
                s[h][0] = e0 * e0;
                s[h][1] = e1 * e1;
            } else {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int dd = 0; dd < GSUM_MAX_D; ++dd) {
                    if (dd < d) {
                        const double xi = ui[rr * d + dd];
                        const double e0 = xi - vj0[dd], e1 = xi - vj1[dd];
                        s0 = s0 + e0 * e0;
                        s1 = s1 + e1 * e1;
                    }
                }
                s[h][0] = s0;
                s[h][1] = s1;
            }
        }
        double v[2][2];
        if (plain) {
            const bool nz = !(gs_base_is_zero<FAM>(s[0][0]) && gs_base_is_zero<FAM>(s[0][1]) && gs_base_is_zero<FAM>(s[1][0]) &&
                              gs_base_is_zero<FAM>(s[1][1]));
            if (__builtin_amdgcn_ballot_w64(nz) == 0) {
                const double z = amp * 0.0 + addc;
                v[0][0] = v[0][1] = v[1][0] = v[1][1] = z;
            } else {
                bool slow[2][2];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int c = 0; c < 2; ++c) v[h][c] = amp * gs_base_value_nb<FAM>(s[h][c], th, tl, slow[h][c]) + addc;
                if (__builtin_amdgcn_ballot_w64(slow[0][0] || slow[0][1] || slow[1][0] || slow[1][1]) != 0) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            if (slow[h][c]) v[h][c] = amp * gs_base_value_t<FAM>(s[h][c], th, tl) + addc;
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int gi = r0 + rp + 4 * h, gj = gj0 + c;
                    double val;
                    if (gi >= n || gj >= n) {
                        val = (gi == gj) ? 1.0 : 0.0;                 // identity padding up to a multiple of 128
                    } else {
                        const bool dg = gi == gj;
                        const double b = dg ? 1.0 : gs_base_value_t<FAM>(s[h][c], th, tl);   // np.fill_diagonal(K, 1)
                        val = amp * b;
                        if (dg) val = val + desc.white_noise;
                        val = val + addc;
                        if (dg) val = val + diag_add;
                    }
                    if (gi == gj) diag0[gi] = val;
                    v[h][c] = val;
                }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const gs_d2 o = {v[h][0], v[h][1]};
            *reinterpret_cast<gs_d2*>(A + (int64_t)(r0 + rp + 4 * h) * ld + gj0) = o;
        }
    }
}

// family and dimension are uniform over the workgroup: one scalar branch per tile
__device__ __forceinline__ void gs_build_tile128_any(double* A, int64_t ld, const double* ui, const double* uj, const double* th,
                                                     const double* tl, int bi, int bj, int n, int d, const gsum_kernel_desc& desc,
                                                     double diag_add, double* diag0, int w, int lane) {
#define GS_BT(F)                                                                                                       \
    if (d == 1) gs_build_tile128<F, true>(A, ld, ui, uj, th, tl, bi, bj, n, d, desc, diag_add, diag0, w, lane);        \
    else gs_build_tile128<F, false>(A, ld, ui, uj, th, tl, bi, bj, n, d, desc, diag_add, diag0, w, lane)
    if (desc.family == GSUM_RBF) { GS_BT(GSUM_RBF); }
    else if (desc.family == GSUM_MATERN52) { GS_BT(GSUM_MATERN52); }
    else if (desc.family == GSUM_MATERN32) { GS_BT(GSUM_MATERN32); }
    else { GS_BT(GSUM_MATERN12); }
#undef GS_BT
}

#define GS_B2_ROWS 32
// One 32 x 128 tile per 256-thread workgroup: wave w takes rows 8 w .. 8 w + 7, two at a time; each lane owns two adjacent
// columns (one 16-B store per row, 1 KiB per wave-instruction).  Grid: CROSS or tri == 0: (prow / 32 rounded up) x (pcol /
// 128 rounded up) tiles, row-slice fastest; tri != 0: the 128-column tiles on or below the diagonal, four row slices each.
template <bool CROSS, int FAM, bool D1>
__global__ __launch_bounds__(256) void k_build2(double* out, int64_t ldo, const double* X, const double* Y, int n, int m,
                                                 int prow, int pcol, int d, gsum_kernel_desc desc, double diag_add, int tri) {
#pragma clang fp contract(off)
    __shared__ double ui[GS_B2_ROWS * GSUM_MAX_D];
    __shared__ double uj[128 * GSUM_MAX_D];
    __shared__ double tab[32];
    const int t = threadIdx.x;
    int bi, bj;                                        // 32-row slice index, 128-column tile index
    if (tri) {
        const int bid = blockIdx.x >> 2;
        int b128 = (int)((sqrt(8.0 * (double)bid + 1.0) - 1.0) * 0.5);
        while ((int64_t)(b128 + 1) * (b128 + 2) / 2 <= bid) ++b128;
        while ((int64_t)b128 * (b128 + 1) / 2 > bid) --b128;
        bj = bid - (int)((int64_t)b128 * (b128 + 1) / 2);
        bi = 4 * b128 + (blockIdx.x & 3);
    } else {
        const int tr = (prow + GS_B2_ROWS - 1) / GS_B2_ROWS;
        bi = blockIdx.x % tr;
        bj = blockIdx.x / tr;
    }
    const double* Yp = CROSS ? Y : X;
    const int ny = CROSS ? m : n;
    const int r0 = bi * GS_B2_ROWS, c0 = bj * 128;
    if (t < 16) tab[t] = gs_exp_th[t];
    else if (t < 32) tab[t] = gs_exp_tl[t - 16];
    for (int idx = t; idx < 128 * d; idx += 256) {
        const int r = idx / d, dd = idx - r * d;
        const double ls = desc.anisotropic ? desc.length_scale[dd] : desc.length_scale[0];
        const int gj = c0 + r;
        uj[idx] = gj < ny ? Yp[(int64_t)gj * d + dd] / ls : 0.0;
        if (r < GS_B2_ROWS) {
            const int gi = r0 + r;
            ui[idx] = gi < n ? X[(int64_t)gi * d + dd] / ls : 0.0;
        }
    }
    __syncthreads();
    const double* th = tab;
    const double* tl = tab + 16;
    const int lane = t & 63, w = t >> 6;
    const int gj0 = c0 + 2 * lane;
    if (gj0 >= pcol) return;
    double vj0[D1 ? 1 : GSUM_MAX_D], vj1[D1 ? 1 : GSUM_MAX_D];
    if (D1) {
        vj0[0] = uj[2 * lane];
        vj1[0] = uj[2 * lane + 1];
    } else {
#pragma unroll
        for (int dd = 0; dd < GSUM_MAX_D; ++dd) {
            vj0[dd] = dd < d ? uj[(2 * lane) * d + dd] : 0.0;
            vj1[dd] = dd < d ? uj[(2 * lane + 1) * d + dd] : 0.0;
        }
    }
    // a tile is plain when no entry needs a diagonal or padding rule: then value = amplitude * base + additive
    const bool plain = CROSS ? (r0 + GS_B2_ROWS <= n && c0 + 128 <= m)
                             : (r0 + GS_B2_ROWS <= n && c0 + 128 <= n && (c0 + 128 <= r0 || r0 + GS_B2_ROWS <= c0));
    const double amp = desc.amplitude, addc = desc.additive_const;
#pragma unroll 1
    for (int rp = 0; rp < 8; rp += 2) {
        double s[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rr = 8 * w + rp + h;
            if (D1) {
                const double xi = ui[rr];
                const double e0 = xi - vj0[0], e1 = xi - vj1[0];
                s[h][0] = e0 * e0;
                s[h][1] = e1 * e1;
            } else {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int dd = 0; dd < GSUM_MAX_D; ++dd) {
                    if (dd < d) {
                        const double xi = ui[rr * d + dd];
                        const double e0 = xi - vj0[dd], e1 = xi - vj1[dd];
                        s0 = s0 + e0 * e0;
                        s1 = s1 + e1 * e1;
                    }
                }
                s[h][0] = s0;
                s[h][1] = s1;
            }
        }
        double v[2][2];
        if (plain) {
            // wave-uniform short cut: every exponential of these 2 x 128 entries is exactly zero
            const bool nz = !(gs_base_is_zero<FAM>(s[0][0]) && gs_base_is_zero<FAM>(s[0][1]) && gs_base_is_zero<FAM>(s[1][0]) &&
                              gs_base_is_zero<FAM>(s[1][1]));
            if (__builtin_amdgcn_ballot_w64(nz) == 0) {
                const double z = amp * 0.0 + addc;
                v[0][0] = v[0][1] = v[1][0] = v[1][1] = z;
            } else {
                bool slow[2][2];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int c = 0; c < 2; ++c) v[h][c] = amp * gs_base_value_nb<FAM>(s[h][c], th, tl, slow[h][c]) + addc;
                if (__builtin_amdgcn_ballot_w64(slow[0][0] || slow[0][1] || slow[1][0] || slow[1][1]) != 0) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            if (slow[h][c]) v[h][c] = amp * gs_base_value_t<FAM>(s[h][c], th, tl) + addc;
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int gi = r0 + 8 * w + rp + h, gj = gj0 + c;
                    double val;
                    if (CROSS) {
                        val = (gi < n && gj < m) ? amp * gs_base_value_t<FAM>(s[h][c], th, tl) + addc : 0.0;
                    } else if (gi >= n || gj >= n) {
                        val = (gi == gj) ? 1.0 : 0.0;                 // identity padding up to a multiple of 128
                    } else {
                        const bool dg = gi == gj;
                        const double b = dg ? 1.0 : gs_base_value_t<FAM>(s[h][c], th, tl);   // np.fill_diagonal(K, 1)
                        val = amp * b;
                        if (dg) val = val + desc.white_noise;
                        val = val + addc;
                        if (dg) val = val + diag_add;
                    }
                    v[h][c] = val;
                }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int gi = r0 + 8 * w + rp + h;
            if (gi < prow) {
                const gs_d2 o = {v[h][0], v[h][1]};
                *reinterpret_cast<gs_d2*>(out + (int64_t)gi * ldo + gj0) = o;
            }
        }
    }
}

// ---- general kernel trees (gsum_kernel_desc with n_ops > 0) ----------------------------------------------------------------------
// The reference hands ANY scikit-learn kernel to its three call sites (gsum/models.py:708, 822-824, 958-960).  The flattened
// descriptor covers the family its own tests and notebooks use and runs the templated kernels above; everything else that is a
// Sum / Product / Exponentiation tree over stationary leaves (RBF, Matern 1/2 3/2 5/2 inf, RationalQuadratic, ExpSineSquared), ConstantKernel and WhiteKernel is
// evaluated entry by entry as a postfix program in scikit-learn's own evaluation order (Sum: k1 + k2, kernels.py:858-866;
// Product: k1 * k2, :956-966), so that `RBF + RBF` or `C * RBF + C * Matern` come out bit-identical to sklearn's matrix like
// the flattened family does; RationalQuadratic goes through pow() and is within an ulp or two of numpy's.
// The same walk with dual numbers gives d kernel / d log(parameter) for the gradient path (sklearn's K_gradient formulas per leaf:
// RBF :1567-1577, Matern :1740-1771, RationalQuadratic :1893-1901; product and sum rules for the operators).
#define GS_TREE_STACK 8

// want: 0 value only, 1 d / d log length_scale (isotropic), 2 d / d log length_scale[dim], 3 d / d log alpha
// cross: the TWO-argument form kernel(X, Y) (it differs from the one-argument form off the diagonal only where scikit-learn orders the arithmetic
// differently: ExpSineSquared)
__device__ __forceinline__ void gs_leaf_eval(const gsum_kernel_leaf& lf, const double* xi, const double* xj, int d, bool diag, int want,
                                             int dim, double& v, double& dv, bool cross = false) {
#pragma clang fp contract(off)
    dv = 0.0;
    if (lf.family == GSUM_DOT) {                 // kernels.py DotProduct.__call__: np.inner(X, Y) + sigma_0 ** 2, on and off the diagonal; K_gradient = 2 sigma_0 ** 2
        double s = 0.0;
        for (int m = 0; m < d; ++m) s = s + xi[m] * xj[m];
        const double s0 = lf.length_scale[0] * lf.length_scale[0];
        v = s + s0;
        if (want == 1) dv = 2.0 * s0;
        return;
    }
    if (diag) {                                  // np.fill_diagonal(K, 1) of the one-argument form; every leaf gradient is 0 there
        v = 1.0;
        return;
    }
    if (lf.family == GSUM_RQ) {                  // kernels.py:1886-1890: (1 + sqeuclidean(X) / (2 alpha ls^2)) ** -alpha
        double s = 0.0;
        for (int m = 0; m < d; ++m) {
            const double e = xi[m] - xj[m];
            s = s + e * e;
        }
        const double ls2 = lf.length_scale[0] * lf.length_scale[0];
        const double base = 1.0 + s / ((2.0 * lf.alpha) * ls2);
        v = pow(base, -lf.alpha);
        if (want == 1) dv = s * v / (ls2 * base);
        else if (want == 3) dv = v * (-lf.alpha * log(base) + s / ((2.0 * ls2) * base));
        return;
    }
    if (lf.family == GSUM_EXPSINE) {             // kernels.py ExpSineSquared.__call__: exp(-2 (sin(pi dists / p) / ls)^2), dists = euclidean(X)
        double s = 0.0;
        for (int m = 0; m < d; ++m) {
            const double e = xi[m] - xj[m];
            s = s + e * e;
        }
        const double ls = lf.length_scale[0];
        // (alpha holds the periodicity)  one argument: arg = pi * dists / p; two arguments: sin(pi / p * dists) -- at arguments of tens of radians
        // one ulp of the argument is tens of ulps of the value
        const double arg = cross ? (3.141592653589793 / lf.alpha) * sqrt(s) : 3.141592653589793 * sqrt(s) / lf.alpha;
        const double sn = sin(arg);
        const double q = sn / ls;
        v = gs_exp_np(-2.0 * (q * q));
        if (want == 1) dv = 4.0 / (ls * ls) * (sn * sn) * v;                 // d / d log length_scale
        else if (want == 3) dv = 4.0 * arg / (ls * ls) * cos(arg) * sn * v;  // d / d log periodicity
        return;
    }
    double s = 0.0, dsel = 0.0;                  // sqeuclidean(X / length_scale): divide first, like pdist on the scaled points
    for (int m = 0; m < d; ++m) {
        const double ls = lf.anisotropic ? lf.length_scale[m] : lf.length_scale[0];
        const double u = xi[m] / ls - xj[m] / ls;
        const double dmm = u * u;
        s = s + dmm;
        if (m == dim) dsel = dmm;
    }
    if (lf.family == GSUM_MATERN_INF) {          // Matern(nu = inf): exp(-dists^2 / 2) with dists = euclidean(X / length_scale) -- the root is taken and squared again
        const double dist = sqrt(s);
        v = gs_exp_np(-(dist * dist) / 2.0);
        if (want == 1 || want == 2) dv = (want == 1 ? s : dsel) * v;
        return;
    }
    v = gs_base_value(lf.family, s);
    if (want == 1 || want == 2) {
        const double dm = want == 1 ? s : dsel;
        if (lf.family == GSUM_RBF) {
            dv = v * dm;
        } else if (lf.family == GSUM_MATERN52) {
            const double tmp = sqrt(5.0 * s);
            dv = 5.0 / 3.0 * dm * (tmp + 1.0) * gs_exp_np(-tmp);
        } else if (lf.family == GSUM_MATERN32) {
            dv = 3.0 * dm * gs_exp_np(-sqrt(3.0 * s));
        } else {
            const double den = sqrt(s);
            dv = den != 0.0 ? v * (dm / den) : 0.0;
        }
    }
}

// value of the tree at (xi, xj); diag: the entry is on the diagonal of the ONE-argument form (leaves exactly 1, WhiteKernel on).
// pr != NULL: *dout = d value / d log(parameter pr) as well.
__device__ __forceinline__ double gs_tree_eval(const gsum_kernel_desc& t, const double* xi, const double* xj, int d, bool diag,
                                               const gsum_grad_param* pr, double* dout, bool cross = false) {
#pragma clang fp contract(off)
    double sv[GS_TREE_STACK], sd[GS_TREE_STACK];
    int sp = 0;
    const int code = pr ? pr->code : -1, pdim = pr ? pr->dim : 0;
    for (int k = 0; k < t.n_ops; ++k) {
        const int op = t.op[k];
        if (op >= GSUM_OP_POW) {                  // Exponentiation: K ** exponent, K_gradient *= exponent K ** (exponent - 1)
            const double e = t.cval[op - GSUM_OP_POW], a = sv[sp - 1];
            sv[sp - 1] = pow(a, e);
            sd[sp - 1] = sd[sp - 1] * (e * pow(a, e - 1.0));
        } else if (op >= GSUM_OP_WHITE) {
            const int c = op - GSUM_OP_WHITE;
            const double w = diag ? t.cval[c] : 0.0;
            sv[sp] = w;
            sd[sp] = (code == GSUM_GRAD_TREE_WHITE && pdim == c) ? w : 0.0;
            ++sp;
        } else if (op >= GSUM_OP_CONST) {
            const int c = op - GSUM_OP_CONST;
            sv[sp] = t.cval[c];
            sd[sp] = (code == GSUM_GRAD_TREE_CONST && pdim == c) ? t.cval[c] : 0.0;
            ++sp;
        } else if (op >= GSUM_OP_LEAF) {
            const int l = op - GSUM_OP_LEAF;
            int want = 0;
            if (code >= GSUM_GRAD_TREE_LENGTH_ISO && (pdim >> 4) == l)
                want = code == GSUM_GRAD_TREE_LENGTH_ISO ? 1 : (code == GSUM_GRAD_TREE_LENGTH_DIM ? 2 : 3);
            double v, dv;
            gs_leaf_eval(t.leaf[l], xi, xj, d, diag, want, pdim & 15, v, dv, cross);
            sv[sp] = v;
            sd[sp] = dv;
            ++sp;
        } else {
            const double b = sv[sp - 1], db = sd[sp - 1], a = sv[sp - 2], da = sd[sp - 2];
            sp -= 2;
            if (op == GSUM_OP_ADD) {
                sv[sp] = a + b;
                sd[sp] = da + db;
            } else {
                sv[sp] = a * b;
                sd[sp] = da * b + a * db;
            }
            ++sp;
        }
    }
    if (dout) *dout = sd[0];
    return sv[0];
}

// One 128 x 128 tile of a tree's matrix inside the one-workgroup-per-evaluation kernels (k_lml_small / k_lml_medium): the geometry
// and the diagonal / padding rules of gs_build_tile128's general branch (wave w takes rows w, w + 4, ...; a lane two adjacent
// columns; identity padding beyond n; the diagonal's values also to diag0), the values through gs_tree_eval on the points themselves
// (a tree's leaves divide by their own length scales) -- so the matrix equals k_build_tree's bit for bit.  `desc` stays where it
// is (global memory, uniform address: scalar loads), the program is walked with dynamic indices.
__device__ __forceinline__ void gs_build_tile128_tree(double* A, int64_t ld, const double* X, int bi, int bj, int n, int d,
                                                      const gsum_kernel_desc& desc, double diag_add, double* diag0, int w, int lane) {
#pragma clang fp contract(off)
    const int r0 = bi * 128, gj0 = bj * 128 + 2 * lane;
    double xj[2][GSUM_MAX_D];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int mm = 0; mm < GSUM_MAX_D; ++mm) xj[c][mm] = (mm < d && gj0 + c < n) ? X[(int64_t)(gj0 + c) * d + mm] : 0.0;
#pragma unroll 1
    for (int rp = w; rp < 128; rp += 4) {
        const int gi = r0 + rp;
        double xi[GSUM_MAX_D];
#pragma unroll
        for (int mm = 0; mm < GSUM_MAX_D; ++mm) xi[mm] = (mm < d && gi < n) ? X[(int64_t)gi * d + mm] : 0.0;
        double v[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int gj = gj0 + c;
            if (gi >= n || gj >= n) {
                v[c] = gi == gj ? 1.0 : 0.0;                       // identity padding
            } else {
                v[c] = gs_tree_eval(desc, xi, xj[c], d, gi == gj, nullptr, nullptr);
                if (gi == gj) v[c] = v[c] + diag_add;
            }
            if (gi == gj) diag0[gi] = v[c];
        }
        const gs_d2 o = {v[0], v[1]};
        *reinterpret_cast<gs_d2*>(A + (int64_t)gi * ld + gj0) = o;
    }
}

// kernel matrix of a tree: the tile geometry of k_build2 (32 x 128 tiles, a lane owns two adjacent columns), values through gs_tree_eval
template <bool CROSS>
__global__ __launch_bounds__(256) void k_build_tree(double* out, int64_t ldo, const double* X, const double* Y, int n, int m, int prow,
                                                     int pcol, int d, gsum_kernel_desc desc, double diag_add, int tri) {
#pragma clang fp contract(off)
    const int t = threadIdx.x;
    int bi, bj;
    if (tri) {
        const int bid = blockIdx.x >> 2;
        int b128 = (int)((sqrt(8.0 * (double)bid + 1.0) - 1.0) * 0.5);
        while ((int64_t)(b128 + 1) * (b128 + 2) / 2 <= bid) ++b128;
        while ((int64_t)b128 * (b128 + 1) / 2 > bid) --b128;
        bj = bid - (int)((int64_t)b128 * (b128 + 1) / 2);
        bi = 4 * b128 + (blockIdx.x & 3);
    } else {
        const int tr = (prow + GS_B2_ROWS - 1) / GS_B2_ROWS;
        bi = blockIdx.x % tr;
        bj = blockIdx.x / tr;
    }
    const double* Yp = CROSS ? Y : X;
    const int ny = CROSS ? m : n;
    const int lane = t & 63, w = t >> 6;
    const int gj0 = bj * 128 + 2 * lane;
    if (gj0 >= pcol) return;
    double xj[2][GSUM_MAX_D];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int mm = 0; mm < GSUM_MAX_D; ++mm) xj[c][mm] = (mm < d && gj0 + c < ny) ? Yp[(int64_t)(gj0 + c) * d + mm] : 0.0;
    for (int rr = 0; rr < 8; ++rr) {
        const int gi = bi * GS_B2_ROWS + 8 * w + rr;
        if (gi >= prow) continue;
        double xi[GSUM_MAX_D];
#pragma unroll
        for (int mm = 0; mm < GSUM_MAX_D; ++mm) xi[mm] = (mm < d && gi < n) ? X[(int64_t)gi * d + mm] : 0.0;
        double v[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int gj = gj0 + c;
            if (CROSS) {
                v[c] = (gi < n && gj < m) ? gs_tree_eval(desc, xi, xj[c], d, false, nullptr, nullptr, true) : 0.0;
            } else if (gi >= n || gj >= n) {
                v[c] = gi == gj ? 1.0 : 0.0;                       // identity padding
            } else {
                v[c] = gs_tree_eval(desc, xi, xj[c], d, gi == gj, nullptr, nullptr);
                if (gi == gj) v[c] = v[c] + diag_add;
            }
        }
        const gs_d2 o = {v[0], v[1]};
        *reinterpret_cast<gs_d2*>(out + (int64_t)gi * ldo + gj0) = o;
    }
}

// Border rows np..np+15 of the augmented matrix: row c = column c of RHS (n x k, row-major), zero
// beyond k / n, and a zero 16x16 corner.
__global__ __launch_bounds__(256) void k_set_border(double* A, int64_t ld, int n, int np, const double* Z, int k) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= np + GS_BORDER) return;
#pragma unroll
    for (int c = 0; c < GS_BORDER; ++c)
        A[(int64_t)(np + c) * ld + i] = (c < k && i < n) ? Z[(int64_t)i * k + c] : 0.0;
}

// diag0[i] = A[i][i] before the factorisation touches it.
__global__ __launch_bounds__(256) void k_save_diag(const double* A, int64_t ld, int np, double* diag0) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < np) diag0[i] = A[(int64_t)i * ld + i];
}

// Rows >= n of the padded square part become identity rows; used after a host upload.
__global__ __launch_bounds__(256) void k_pad_identity(double* A, int64_t ld, int n, int np) {
    int j = blockIdx.x * 256 + threadIdx.x;
    int i = n + blockIdx.y;
    if (j >= np || i >= np) return;
    A[(int64_t)i * ld + j] = (i == j) ? 1.0 : 0.0;
}

