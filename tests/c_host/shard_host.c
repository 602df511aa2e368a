/* A C host of libgsum_hip.so for the sharded likelihood scan ("Multi-GPU from a C host"), as a test:
 * plain C99, include/gsum_hip.h only.  It plays every rank of a world of `world` processes one after the other on one GPU --
 * each "rank" evaluates its slice (gsum_shard_range + gsum_lml_resident_shard) into ITS OWN padded buffers, exactly as a
 * rank of a real job would -- then performs what an in-place all-gather does (copy block r of rank r's buffers into the
 * result) and compares the stitched arrays with one unsharded gsum_lml_resident call: bit-identical or exit 1.
 *     shard_host <n> <n_theta> <world>
 * Exit codes: 0 ok, 1 mismatch, 2 library error.   Built by tests/test_host_logic.py (gcc -c: the header is valid C) and
 * built + run by tests/test_gpu_round3.py on the GPU box. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gsum_hip.h"

static double frand(uint64_t* s) {                 /* xorshift: deterministic inputs without libm */
    *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17;
    return (double)(*s >> 11) / 9007199254740992.0 - 0.5;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 600;
    const int n_theta = argc > 2 ? atoi(argv[2]) : 11, world = argc > 3 ? atoi(argv[3]) : 3;
    const int d = 1, k = 4;
    gsum_ctx* ctx = NULL;
    if (gsum_init(0, &ctx) != 0) {
        fprintf(stderr, "gsum_init: %s\n", gsum_last_error(NULL));
        return 2;
    }
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
    if (gsum_set_inputs(ctx, X, n, d, Z, k) != 0) {
        fprintf(stderr, "gsum_set_inputs: %s\n", gsum_last_error(ctx));
        return 2;
    }
    const int64_t c = (n_theta + world - 1) / world, padded = c * world;   /* = the block size gsum_shard_range uses */
    double* G = malloc(sizeof(double) * padded * k * k);                   /* the gathered result */
    double* sld = malloc(sizeof(double) * padded);
    int64_t* inf = malloc(sizeof(int64_t) * padded);
    for (int rank = 0; rank < world; ++rank) {
        double* Gr = malloc(sizeof(double) * padded * k * k);              /* this rank's own padded buffers */
        double* sr = malloc(sizeof(double) * padded);
        int64_t* ir = malloc(sizeof(int64_t) * padded);
        memset(Gr, 0xff, sizeof(double) * padded * k * k);
        memset(sr, 0xff, sizeof(double) * padded);
        memset(ir, 0xff, sizeof(int64_t) * padded);
        int64_t lo = -1, hi = -1, lo2 = -2, hi2 = -2;
        if (gsum_lml_resident_shard(ctx, descs, n_theta, rank, world, 1e-10, Gr, sr, ir, &lo, &hi) != 0) {
            fprintf(stderr, "gsum_lml_resident_shard: %s\n", gsum_last_error(ctx));
            return 2;
        }
        if (gsum_shard_range(n_theta, rank, world, &lo2, &hi2) != 0 || lo != lo2 || hi != hi2 || lo != (rank * c < n_theta ? rank * c : n_theta)) {
            fprintf(stderr, "rank %d: slice [%lld, %lld) does not match gsum_shard_range [%lld, %lld)\n", rank, (long long)lo,
                    (long long)hi, (long long)lo2, (long long)hi2);
            return 1;
        }
        /* what ncclAllGather(buf + lo, buf, c, ...) leaves in every rank's block `rank` */
        memcpy(G + (size_t)rank * c * k * k, Gr + (size_t)rank * c * k * k, sizeof(double) * c * k * k);
        memcpy(sld + (size_t)rank * c, sr + (size_t)rank * c, sizeof(double) * c);
        memcpy(inf + (size_t)rank * c, ir + (size_t)rank * c, sizeof(int64_t) * c);
        free(Gr); free(sr); free(ir);
    }
    double* G1 = malloc(sizeof(double) * n_theta * k * k);
    double* s1 = malloc(sizeof(double) * n_theta);
    int64_t* i1 = malloc(sizeof(int64_t) * n_theta);
    if (gsum_lml_resident(ctx, descs, n_theta, 1e-10, G1, s1, i1) != 0) {
        fprintf(stderr, "gsum_lml_resident: %s\n", gsum_last_error(ctx));
        return 2;
    }
    int bad = memcmp(G, G1, sizeof(double) * n_theta * k * k) != 0 || memcmp(sld, s1, sizeof(double) * n_theta) != 0 ||
              memcmp(inf, i1, sizeof(int64_t) * n_theta) != 0;
    for (int j = 0; j < n_theta; ++j) bad |= i1[j] != 0;
    printf("n=%lld n_theta=%d world=%d: gathered == unsharded: %s; sld[0] = %.17g\n", (long long)n, n_theta, world, bad ? "NO" : "yes",
           s1[0]);
    gsum_destroy(ctx);
    return bad ? 1 : 0;
}
