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
                          (void *)nabo_pairwise, (void *)nabo_index_create, (void *)nabo_index_destroy,
                          (void *)nabo_index_set_ref, (void *)nabo_index_set_mask, (void *)nabo_index_query,
                          (void *)nabo_index_query_candidates, (void *)nabo_index_last_stats, (void *)nabo_merge_topk,
                          (void *)nabo_snn_counts, (void *)nabo_score_null, (void *)nabo_score_null_edges, (void *)nabo_dev_malloc, (void *)nabo_dev_free,
                          (void *)nabo_memcpy_h2d, (void *)nabo_memcpy_d2h, (void *)nabo_dev_synchronize};
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
    printf("C ABI ok: pairwise %dx%dx%d and k-NN (k=%d) bit-equal to the C restatement\n", M, N, G, K);
    return 0;
}
