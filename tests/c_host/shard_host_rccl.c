/* The multi-GPU recipe of INTEGRATION.md ("Multi-GPU from a C host") with a REAL RCCL communicator: one process, one gsum_ctx
 * and one RCCL rank per visible GPU (ncclCommInitAll), every rank evaluates its slice of a theta scan with
 * gsum_lml_resident_shard into its positions of padded full-length arrays, the slices are staged to the device and gathered IN
 * PLACE with exactly the three ncclAllGather calls of the recipe, and every rank's gathered arrays are compared bit for bit
 * with ONE unsharded gsum_lml_resident call.  World = the number of visible GPUs: 1 on a single-GPU test box (the collective
 * still runs through RCCL), 8 on a node -- no code change.  Plain C99 + the public headers of HIP and RCCL.
 * Round 5: the ranks' blocks run AT THE SAME TIME, one POSIX thread per device (every library call is synchronous, and a gsum_ctx belongs
 * to one thread at a time), and the same scan is repeated through the library's own group entry -- gsum_init_multi +
 * gsum_lml_batch_multi(..., GSUM_GATHER_RCCL): threads, partition and the in-place all-gather inside the library.
 *     shard_host_rccl [n] [n_theta]
 * Exit codes: 0 ok, 1 mismatch, 2 library / runtime error.
 * Replaces the reference's serial loop over grid points, docs/notebooks/correlated_EFT_publication.ipynb:1457-1459. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "gsum_hip.h"

#define MAXDEV 16
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 2; } } while (0)

static double frand(uint64_t* s) {
    *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17;
    return (double)(*s >> 11) / 9007199254740992.0 - 0.5;
}

/* one rank's block of the scan, on a thread of its own */
typedef struct {
    gsum_ctx* ctx; int rank, world, n_theta, d, k; int64_t n;
    const double *X, *Z; const gsum_kernel_desc* descs;
    double *G, *sld; int64_t *inf, lo, hi; int rc;
} rank_job;

static void* run_rank(void* arg) {
    rank_job* j = (rank_job*)arg;
    j->rc = hipSetDevice(j->rank) != hipSuccess;
    if (!j->rc) j->rc = gsum_set_inputs(j->ctx, j->X, j->n, j->d, j->Z, j->k);
    if (!j->rc) j->rc = gsum_lml_resident_shard(j->ctx, j->descs, j->n_theta, j->rank, j->world, 1e-10, j->G, j->sld, j->inf, &j->lo, &j->hi);
    return NULL;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 600;
    const int n_theta = argc > 2 ? atoi(argv[2]) : 11;
    const int d = 1, k = 4;
    int world = 0;
    HIPCHK(hipGetDeviceCount(&world));
    if (world < 1 || world > MAXDEV) { fprintf(stderr, "need 1..%d GPUs, found %d\n", MAXDEV, world); return 2; }
    ncclComm_t comm[MAXDEV];
    NCCLCHK(ncclCommInitAll(comm, world, NULL));                          /* ranks 0 .. world - 1 on devices 0 .. world - 1 */

    double* X = malloc(sizeof(double) * n * d);
    double* Z = malloc(sizeof(double) * n * k);
    uint64_t seed = 88172645463325252ull;
    for (int64_t i = 0; i < n; ++i) X[i] = 0.1 * (double)i;
    for (int64_t i = 0; i < n * k; ++i) Z[i] = (i % k == k - 1) ? 1.0 : frand(&seed);
    gsum_kernel_desc* descs = calloc((size_t)n_theta, sizeof(gsum_kernel_desc));
    for (int j = 0; j < n_theta; ++j) {
        descs[j].family = GSUM_RBF;
        descs[j].length_scale[0] = 0.15 + 0.01 * j;
        descs[j].amplitude = 1.0;
    }
    const int64_t c = (n_theta + world - 1) / world, padded = c * world;   /* = the block size gsum_shard_range uses */

    gsum_ctx* ctx[MAXDEV];
    hipStream_t stream[MAXDEV];
    double *G[MAXDEV], *sld[MAXDEV], *d_G[MAXDEV], *d_sld[MAXDEV];
    int64_t *inf[MAXDEV], *d_inf[MAXDEV], lo[MAXDEV];
    rank_job job[MAXDEV];
    pthread_t th[MAXDEV];
    for (int rank = 0; rank < world; ++rank) {
        HIPCHK(hipSetDevice(rank));
        if (gsum_init(rank, &ctx[rank]) != 0) { fprintf(stderr, "gsum_init(%d): %s\n", rank, gsum_last_error(NULL)); return 2; }
        HIPCHK(hipStreamCreate(&stream[rank]));
        G[rank] = malloc(sizeof(double) * padded * k * k);                 /* padded to world * c entries */
        sld[rank] = malloc(sizeof(double) * padded);
        inf[rank] = malloc(sizeof(int64_t) * padded);
        memset(G[rank], 0xff, sizeof(double) * padded * k * k);
        memset(sld[rank], 0xff, sizeof(double) * padded);
        memset(inf[rank], 0xff, sizeof(int64_t) * padded);
        HIPCHK(hipMalloc((void**)&d_G[rank], sizeof(double) * padded * k * k));
        HIPCHK(hipMalloc((void**)&d_sld[rank], sizeof(double) * padded));
        HIPCHK(hipMalloc((void**)&d_inf[rank], sizeof(int64_t) * padded));
        rank_job j = {ctx[rank], rank, world, n_theta, d, k, n, X, Z, descs, G[rank], sld[rank], inf[rank], 0, 0, 0};
        job[rank] = j;
    }
    /* every device works on its block at the same time */
    for (int rank = 0; rank < world; ++rank)
        if (pthread_create(&th[rank], NULL, run_rank, &job[rank]) != 0) { fprintf(stderr, "pthread_create failed\n"); return 2; }
    for (int rank = 0; rank < world; ++rank) pthread_join(th[rank], NULL);
    for (int rank = 0; rank < world; ++rank) {
        if (job[rank].rc != 0) { fprintf(stderr, "rank %d: %s\n", rank, gsum_last_error(ctx[rank])); return 2; }
        lo[rank] = job[rank].lo;
        HIPCHK(hipSetDevice(rank));
        /* host -> device staging of the three slices (whole padded arrays: only block `rank` holds results) */
        HIPCHK(hipMemcpyAsync(d_G[rank], G[rank], sizeof(double) * padded * k * k, hipMemcpyHostToDevice, stream[rank]));
        HIPCHK(hipMemcpyAsync(d_sld[rank], sld[rank], sizeof(double) * padded, hipMemcpyHostToDevice, stream[rank]));
        HIPCHK(hipMemcpyAsync(d_inf[rank], inf[rank], sizeof(int64_t) * padded, hipMemcpyHostToDevice, stream[rank]));
    }
    /* the recipe's collective, in place, one group for the ranks of this process */
    NCCLCHK(ncclGroupStart());
    for (int rank = 0; rank < world; ++rank) {
        NCCLCHK(ncclAllGather(d_sld[rank] + lo[rank], d_sld[rank], c, ncclDouble, comm[rank], stream[rank]));
        NCCLCHK(ncclAllGather(d_G[rank] + lo[rank] * k * k, d_G[rank], c * k * k, ncclDouble, comm[rank], stream[rank]));
        NCCLCHK(ncclAllGather(d_inf[rank] + lo[rank], d_inf[rank], c, ncclInt64, comm[rank], stream[rank]));
    }
    NCCLCHK(ncclGroupEnd());
    for (int rank = 0; rank < world; ++rank) {
        HIPCHK(hipSetDevice(rank));
        HIPCHK(hipMemcpyAsync(G[rank], d_G[rank], sizeof(double) * padded * k * k, hipMemcpyDeviceToHost, stream[rank]));
        HIPCHK(hipMemcpyAsync(sld[rank], d_sld[rank], sizeof(double) * padded, hipMemcpyDeviceToHost, stream[rank]));
        HIPCHK(hipMemcpyAsync(inf[rank], d_inf[rank], sizeof(int64_t) * padded, hipMemcpyDeviceToHost, stream[rank]));
        HIPCHK(hipStreamSynchronize(stream[rank]));
    }
    /* the unsharded answer, on rank 0's GPU */
    double* G1 = malloc(sizeof(double) * n_theta * k * k);
    double* s1 = malloc(sizeof(double) * n_theta);
    int64_t* i1 = malloc(sizeof(int64_t) * n_theta);
    HIPCHK(hipSetDevice(0));
    if (gsum_lml_resident(ctx[0], descs, n_theta, 1e-10, G1, s1, i1) != 0) { fprintf(stderr, "gsum_lml_resident: %s\n", gsum_last_error(ctx[0])); return 2; }
    int bad = 0;
    for (int rank = 0; rank < world; ++rank)
        bad |= memcmp(G[rank], G1, sizeof(double) * n_theta * k * k) != 0 || memcmp(sld[rank], s1, sizeof(double) * n_theta) != 0 ||
               memcmp(inf[rank], i1, sizeof(int64_t) * n_theta) != 0;
    for (int j = 0; j < n_theta; ++j) bad |= i1[j] != 0;
    printf("n=%lld n_theta=%d world=%d (RCCL, ncclCommInitAll, %d threads): gathered == unsharded on every rank: %s; sld[0] = %.17g\n",
           (long long)n, n_theta, world, world, bad ? "NO" : "yes", s1[0]);
    for (int rank = 0; rank < world; ++rank) {
        HIPCHK(hipSetDevice(rank));
        gsum_destroy(ctx[rank]);
        ncclCommDestroy(comm[rank]);
    }
    /* the same scan through the library's own group: contexts, threads, partition and the RCCL gather behind ONE call */
    gsum_group* grp = NULL;
    if (gsum_init_multi(0, NULL, &grp) != 0) { fprintf(stderr, "gsum_init_multi: %s\n", gsum_group_last_error(NULL)); return 2; }
    double* G2 = malloc(sizeof(double) * n_theta * k * k);
    double* s2 = malloc(sizeof(double) * n_theta);
    int64_t* i2 = malloc(sizeof(int64_t) * n_theta);
    if (gsum_lml_batch_multi(grp, descs, n_theta, X, n, d, Z, k, 1e-10, G2, s2, i2, GSUM_GATHER_RCCL) != 0) {
        fprintf(stderr, "gsum_lml_batch_multi: %s\n", gsum_group_last_error(grp));
        return 2;
    }
    const int bad2 = memcmp(G2, G1, sizeof(double) * n_theta * k * k) != 0 || memcmp(s2, s1, sizeof(double) * n_theta) != 0 ||
                     memcmp(i2, i1, sizeof(int64_t) * n_theta) != 0 || gsum_group_size(grp) != world ||
                     gsum_group_get(grp, "rccl_gathers") != 1;
    printf("gsum_lml_batch_multi over %d device(s), GSUM_GATHER_RCCL == unsharded, group entry: %s\n", (int)gsum_group_size(grp), bad2 ? "NO" : "yes");
    gsum_group_destroy(grp);
    return (bad || bad2) ? 1 : 0;
}
