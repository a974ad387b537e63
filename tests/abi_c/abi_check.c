/* Plain-C consumer of the C ABI (include/nabo_knn.h): proves the header is valid C99, that every entry point
 * links, and -- when a GPU is present (argv[1] == "run") -- that nabo_knn / nabo_pairwise return what a
 * straightforward C restatement of nabo/_mapping.py:16-26 + the canonical order computes.
 *   gcc -std=c99 -Wall -Iinclude tests/abi_c/abi_check.c -Lnabo_amd -lnabo_knn -Wl,-rpath,$PWD/nabo_amd -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nabo_knn.h"

static double frand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (double)(*s >> 8) / 16777216.0 - 0.5; }

int main(int argc, char **argv)
{
    /* take the address of every declared entry point: an unresolved one fails the link */
    const void *syms[] = {(void *)nabo_version, (void *)nabo_last_error, (void *)nabo_device_count, (void *)nabo_knn,
                          (void *)nabo_pairwise, (void *)nabo_index_create, (void *)nabo_index_destroy, (void *)nabo_index_set_option, (void *)nabo_query_plan,
                          (void *)nabo_index_set_ref, (void *)nabo_index_set_mask, (void *)nabo_index_query,
                          (void *)nabo_index_query_async, (void *)nabo_index_query_wait,
                          (void *)nabo_index_query_candidates, (void *)nabo_index_last_stats, (void *)nabo_index_last_kernel, (void *)nabo_index_last_passes, (void *)nabo_index_last_row_pass, (void *)nabo_dev_mem_info, (void *)nabo_merge_topk,
                          (void *)nabo_snn_counts, (void *)nabo_pyset_order, (void *)nabo_component_labels, (void *)nabo_group_edges, (void *)nabo_score_null, (void *)nabo_score_null_edges, (void *)nabo_dev_malloc, (void *)nabo_dev_free,
                          (void *)nabo_memcpy_h2d, (void *)nabo_memcpy_d2h, (void *)nabo_dev_synchronize,
                          (void *)nabo_comm_unique_id, (void *)nabo_comm_create, (void *)nabo_comm_create_all,
                          (void *)nabo_comm_create_loopback, (void *)nabo_comm_destroy, (void *)nabo_comm_rank,
                          (void *)nabo_comm_world, (void *)nabo_comm_transport_ranks, (void *)nabo_comm_abort, (void *)nabo_comm_set_timeout, (void *)nabo_comm_set_ref_shards, (void *)nabo_comm_barrier, (void *)nabo_comm_allreduce_max_f64,
                          (void *)nabo_candidates_per_shard, (void *)nabo_sharded_query, (void *)nabo_sharded_last_stats,
                          (void *)nabo_knn_devices};
    printf("%s: %d entry points\n", nabo_version(), (int)(sizeof(syms) / sizeof(syms[0])));
    if (argc < 2 || strcmp(argv[1], "run") != 0) return 0;
    if (nabo_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 2; }

    enum { M = 37, N = 500, G = 13, K = 7 };
    static double X[M * G], Y[N * G], D[M * N], gd[M * K];
    static int64_t gi[M * K];
    unsigned seed = 12345u;
    for (int i = 0; i < M * G; ++i) X[i] = 3.0 * frand(&seed);
    for (int i = 0; i < N * G; ++i) Y[i] = 3.0 * frand(&seed);
    if (nabo_pairwise(X, M, Y, N, G, NABO_METRIC_EUCLIDEAN, 0.0, D, 0) != NABO_OK) { fprintf(stderr, "%s\n", nabo_last_error()); return 3; }
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            double td = 0.0;
            for (int k = 0; k < G; ++k) { double t = X[i * G + k] - Y[j * G + k]; td += t * t; }
            if (sqrt(td) != D[i * N + j]) { fprintf(stderr, "pairwise differs at %d,%d\n", i, j); return 4; }
        }
    if (nabo_knn(X, M, Y, N, G, K, NABO_METRIC_EUCLIDEAN, 0.0, NULL, 0, gi, gd, 0) != NABO_OK) { fprintf(stderr, "%s\n", nabo_last_error()); return 5; }
    for (int i = 0; i < M; ++i) {
        int64_t prev_j = -1;
        double prev_d = -1.0;
        for (int p = 0; p < K; ++p) {          /* p-th smallest (distance, index) after the previous pick */
            int64_t bj = -1;
            double bd = 0.0;
            for (int j = 0; j < N; ++j) {
                const double d = D[i * N + j];
                if (d < prev_d || (d == prev_d && j <= prev_j)) continue;
                if (bj < 0 || d < bd) { bd = d; bj = j; }
            }
            if (gi[i * K + p] != bj || gd[i * K + p] != bd) { fprintf(stderr, "knn differs at row %d pos %d\n", i, p); return 6; }
            prev_j = bj;
            prev_d = bd;
        }
    }
    /* error convention: bad argument -> NABO_E_INVALID and a message */
    if (nabo_knn(X, M, Y, N, G, N + 1, NABO_METRIC_EUCLIDEAN, 0.0, NULL, 0, gi, gd, 0) != NABO_E_INVALID || !*nabo_last_error()) return 7;
    /* the sharded entry point with n_devices = 1: nabo_comm_create_all (RCCL, ncclCommInitAll) -> resident index ->
     * nabo_sharded_query under both protocols -> the same [M,K] result as nabo_knn above */
    {
        nabo_comm *comm = NULL;
        nabo_index *ix = NULL;
        const int32_t devs[1] = {0};
        void *dX = NULL, *dI = NULL, *dD = NULL;
        static int64_t si[M * K];
        static double sd[M * K];
        if (nabo_comm_create_all(&comm, devs, 1) != NABO_OK) { fprintf(stderr, "%s\n", nabo_last_error()); return 8; }
        if (nabo_comm_world(comm) != 1 || nabo_comm_rank(comm) != 0 || nabo_comm_barrier(comm) != NABO_OK) return 9;
        if (nabo_index_create(&ix, 0, N, G, NABO_METRIC_EUCLIDEAN, 0.0, 0) != NABO_OK || nabo_index_set_ref(ix, Y, 0, NULL) != NABO_OK) { fprintf(stderr, "%s\n", nabo_last_error()); return 10; }
        if (nabo_dev_malloc(0, &dX, sizeof(X)) || nabo_dev_malloc(0, &dI, sizeof(si)) || nabo_dev_malloc(0, &dD, sizeof(sd)) ||
            nabo_memcpy_h2d(0, dX, X, sizeof(X))) return 11;
        for (int protocol = 1; protocol <= 2; ++protocol) {
            if (nabo_sharded_query(comm, ix, (const double *)dX, M, K, 0, (int64_t *)dI, (double *)dD, protocol) != NABO_OK) { fprintf(stderr, "%s\n", nabo_last_error()); return 12; }
            if (nabo_memcpy_d2h(0, si, dI, sizeof(si)) || nabo_memcpy_d2h(0, sd, dD, sizeof(sd))) return 13;
            if (memcmp(si, gi, sizeof(si)) != 0 || memcmp(sd, gd, sizeof(sd)) != 0) { fprintf(stderr, "sharded query (protocol %d) differs from nabo_knn\n", protocol); return 14; }
        }
        nabo_dev_free(0, dX); nabo_dev_free(0, dI); nabo_dev_free(0, dD);
        nabo_index_destroy(ix);
        nabo_comm_destroy(comm);
    }
    printf("C ABI ok: pairwise %dx%dx%d, k-NN (k=%d) and the sharded query (1 device, both protocols) bit-equal to the C restatement\n", M, N, G, K);
    return 0;
}
