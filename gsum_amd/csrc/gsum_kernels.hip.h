// Device kernels of libgsum_hip.so — written for gfx950 (MI355X, CDNA4) only.
//
//   k_build2       pairwise-distance + RBF/Matern kernel matrix   (HBM-store bound)
//   k_set_border   RHS^T -> border rows of the augmented matrix
//   k_potrf_diag[256]  128x128 diagonal block(s): Cholesky + substitution tables, micro-blocks in MFMA accumulator registers
//   k_gemm_ld3     C (+)= s * A * B^T on v_mfma_f64_16x16x4_f64, 128x64 tiles, operands staged global -> LDS directly,
//                  three workgroups per CU: the bulk trailing update   (fp64 MFMA bound); k_gemm_ld3g: one launch for a group of evaluations
//   k_gemm_nt      the same product with register staging: panel TRSM / sibling / border tiles (32x128, 16x256) and the
//                  earlier bulk tiles
//   k_lml_small, k_lml_medium   whole evaluations in one workgroup (n <= 128; 128 < n <= 4096 on HBM-resident matrices)
//   k_finalize     Gram / log-det read-out of the bordered factorisation
//   k_rowsumsq, k_scale_series, k_tri_multiply, k_grad_contract, k_grad_reduce*   prediction, series scaling, sampling and
//                  gradient contractions
//   probes         fp64 MFMA issue rate, HBM store rate, workgroup placement under a CU mask
//
// Data layout: the factorisation works on ONE augmented row-major fp64 matrix
//     [ K (np x np, lower triangle)  .            ]      np = n rounded up to 128 (identity padding)
//     [ RHS^T (16 x np)              -G (16 x 16) ]      leading dimension ld = np + 16
// A right-looking blocked Cholesky over the first np columns turns the border rows into
// W^T = (L^-1 RHS)^T and the corner into -W^T W, so the forward solve and the Gram reduction of the
// log-likelihood cost no extra pass over L.
//
// The kernels live in kernels/*.hip.h, included here in dependency order (one translation unit: gsum_capi.hip).
#pragma once
#include "kernels/common.hip.h"
#include "kernels/build.hip.h"
#include "kernels/diag.hip.h"
#include "kernels/panel.hip.h"
#include "kernels/chain.hip.h"
#include "kernels/gemm_nt.hip.h"
#include "kernels/fused.hip.h"
#include "kernels/tile.hip.h"
#include "kernels/solve.hip.h"
#include "kernels/grad.hip.h"
#include "kernels/probes.hip.h"
