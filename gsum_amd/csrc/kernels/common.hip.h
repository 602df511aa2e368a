// shared definitions: block size, pivot guard, vector types
// (part of gsum_kernels.hip.h: included from there, in order; gfx950 only)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "gsum_hip.h"

#define GS_NB 128
// pivot test: LAPACK's dpotf2 rule (p <= 0 or NaN -> info) with a guard of 2 eps: a pivot p of column j also counts as not positive
// when p <= gs_pivot_guard * A_jj (the original diagonal entry).  Rounds 1-2 shipped 8 eps.  Measured in round 3
// (tools/gpu_info_sweep.py, the duplicated-point cases of the test suite): with 0 an exactly singular matrix (n = 2048 Matern-5/2,
// one point duplicated, no nugget) factorises on the device with a pivot of +1e-17 where numpy.linalg.cholesky raises; with 4 eps
// and more the device refuses 2-D Matern-5/2 matrices that are singular to working precision and that LAPACK still factorises;
// 1 and 2 eps reproduce LAPACK's outcome on all of them.  Option "pivot_guard_ulps" (lab build: also GSUM_PIVOT_GUARD_ULPS).
__device__ double gs_pivot_guard = 2.0 * 2.220446049250313e-16;
#define GS_BORDER 16
// Row stride of a workspace matrix of padded order np = np + 16 border columns (+ GS_LD_EXTRA, a multiple of 16: rows stay 128-B aligned).
// Round 5 measured the stride (profiles/r05_lda_sweep.log): alone, the bulk tile at K = 1024, M = 7184 runs at 61.8 TF/s on np + 16 =
// 8208 (64 KiB + 128 B) against 63.8-64.6 on np + 32 ... np + 272, other orders do not care -- and the batch of 20 at n = 8192 ran at
// 330.6 evals/s on np + 80 against 330.3-331 on np + 16 (profiles/r05_bench_ld80.jsonl): no gain in situ, so the stride stays.
#define GS_LD_EXTRA 0
#define GS_LD(np) ((np) + GS_BORDER + GS_LD_EXTRA)
#define GS_KC 16                  // K chunk staged through LDS (16 doubles = one 128-B line per row)
#define GS_LSTR (GS_KC + 1)       // odd LDS row stride (17 doubles): the compiler pairs fragment reads into
                                  // ds_read2_b64, which banks mod 32 dwords -> rows 2 dwords apart, no conflicts

typedef double gs_d4 __attribute__((ext_vector_type(4)));
typedef double gs_d2 __attribute__((ext_vector_type(2)));

