/*
 * nabo_oracle.c -- CPU restatement of Nabo's k-NN mapping hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / reported CPU baseline.  The product (nabo_amd) never links,
 * imports or falls back to it.
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks every function here against
 * golden vectors produced by the reference's own nabo/_mapping.py (oracle/gen_golden.py,
 * run in the build container; fixtures in tests/golden/).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (see Makefile).
 * -ffp-contract=off matters: the reference (numba without fastmath) evaluates
 * `td += temp * temp` as a rounded multiply followed by a rounded add, never an FMA.
 *
 * Citations are to /root/reference/nabo/_mapping.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* _euclidean_dist, _mapping.py:16-26: d[i,j] = sqrt(sum_k (x[i,k]-y[j,k])^2), k ascending,
 * float64, separate multiply and add, np.sqrt (correctly rounded). */
static inline double euclid_pair(const double *x, const double *y, int g)
{
    double td = 0.0;
    for (int k = 0; k < g; ++k) {
        double temp = x[k] - y[k];
        td += temp * temp;
    }
    return sqrt(td);
}

/* _mod_canberra_dist, _mapping.py:29-45: per dimension, num=|x-y|; if num < f*|x| then
 * dist += num/(|x|+|y|+0.01) else dist += 1.  Asymmetric in x (target) vs y (reference). */
static inline double canberra_pair(const double *x, const double *y, int g, double f)
{
    double dist = 0.0;
    for (int k = 0; k < g; ++k) {
        double absx = fabs(x[k]);
        double num = fabs(x[k] - y[k]);
        if (num < f * absx) {
            double absy = fabs(y[k]);
            double den = (absx + absy + 0.01);
            dist += num / den;
        } else {
            dist += 1;
        }
    }
    return dist;
}

/* EXTENSION -- NOT in the reference (BASELINE.json's config 5 asks for a cosine
 * metric the reference does not have), so this definition is the build's own and its parity is
 * UNPINNED by any reference output: cosine distance 1 - <x,y> / (sqrt<x,x> * sqrt<y,y>), the three sums
 * accumulated in ascending k with separate multiply and add like a1; a zero vector is at distance 1
 * from everything. */
static inline double cosine_pair(const double *x, const double *y, int g)
{
    double dot = 0.0, nx = 0.0, ny = 0.0;
    for (int k = 0; k < g; ++k) {
        dot += x[k] * y[k];
        nx += x[k] * x[k];
        ny += y[k] * y[k];
    }
    if (nx == 0.0 || ny == 0.0) return 1.0;
    return 1.0 - dot / (sqrt(nx) * sqrt(ny));
}

/* metric: 0 euclidean (a1), 1 modified canberra (a2), 2 cosine (extension, see above) */
static inline double pair_dist(const double *x, const double *y, int g, int metric, double f)
{
    return metric == 0 ? euclid_pair(x, y, g) : metric == 1 ? canberra_pair(x, y, g, f) : cosine_pair(x, y, g);
}

/* Literal a1/a2 kernel seam (_mapping.py:120-124): caller-allocated D[m,n]. */
int oracle_pairwise(const double *X, int64_t m, const double *Y, int64_t n, int32_t g,
                    int32_t metric, double dist_factor, double *D, int32_t nthreads)
{
    if (metric < 0 || metric > 2) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < n; ++j)
            D[i * n + j] = pair_dist(X + i * g, Y + j * g, g, metric, dist_factor);
    return 0;
}

/* Total order used wherever the reference's unstable argsort leaves ties undefined:
 * (masked last, dist ascending, ref index ascending).  The reference masks ignored refs
 * to NaN so that they sort to the END of the order row (_mapping.py:135-144,
 * numpy.ma argsort endwith=True). */
typedef struct { double d; int64_t j; int masked; } cand_t;

static inline int cand_less(const cand_t *a, const cand_t *b)
{
    if (a->masked != b->masked) return a->masked < b->masked;
    if (!a->masked) {
        if (a->d < b->d) return 1;
        if (a->d > b->d) return 0;
    }
    return a->j < b->j;
}

/* _calc_dist tile loop + mask + sort (_mapping.py:98-146), restated array-in/array-out:
 * for every target row the first (k) entries of the order row after the optional
 * positional `[1:]` drop (intra_ref, :142), with their distances.
 * out_idx/out_dist are [m,k].  Requires k + drop_first <= n. */
int oracle_knn(const double *X, int64_t m, const double *Y, int64_t n, int32_t g, int32_t k,
               int32_t metric, double dist_factor, const uint8_t *ref_mask, int32_t drop_first,
               int64_t *out_idx, double *out_dist, int32_t nthreads)
{
    if (metric < 0 || metric > 2) return -1;
    int kk = k + (drop_first ? 1 : 0);
    if (k < 1 || kk > n) return -2;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    int err = 0;
#pragma omp parallel
    {
        cand_t *best = (cand_t *)malloc(sizeof(cand_t) * (size_t)kk);
        if (!best) {
#pragma omp atomic write
            err = 1;
        }
#pragma omp for schedule(static)
        for (int64_t i = 0; i < m; ++i) {
            if (!best) continue;
            int cnt = 0;
            const double *x = X + i * g;
            for (int64_t j = 0; j < n; ++j) {
                cand_t c;
                c.d = pair_dist(x, Y + j * g, g, metric, dist_factor);
                c.j = j;
                c.masked = ref_mask ? (ref_mask[j] != 0) : 0;
                if (cnt == kk && !cand_less(&c, &best[kk - 1])) continue;
                int p = cnt < kk ? cnt++ : kk - 1;
                while (p > 0 && cand_less(&c, &best[p - 1])) { best[p] = best[p - 1]; --p; }
                best[p] = c;
            }
            for (int t = 0; t < k; ++t) {
                const cand_t *c = &best[t + (drop_first ? 1 : 0)];
                out_idx[i * k + t] = c->j;
                out_dist[i * k + t] = c->d;
            }
        }
        free(best);
    }
    return err ? -3 : 0;
}

/* _calc_snn (_mapping.py:186-198), integer part: for target row t and each j in
 * a = set(order_t[:k]): snn = |a  intersect  set(order_ref_j[:k])|.  Emits one (t, j, snn)
 * triple per (t, slot) with snn > 0 (:195); the weight round(snn/(2(k-1)-snn), 2) (:194) is
 * applied by the caller (Python round()).  out_* are [m*k]; returns the number of edges. */
int64_t oracle_snn_counts(const int64_t *t_idx, int64_t m, const int64_t *r_idx, int64_t n, int32_t k,
                          int64_t *out_t, int64_t *out_j, int32_t *out_snn)
{
    int64_t ne = 0;
    for (int64_t t = 0; t < m; ++t) {
        const int64_t *a = t_idx + t * k;
        for (int s = 0; s < k; ++s) {
            int64_t j = a[s];
            if (j < 0 || j >= n) continue;
            const int64_t *b = r_idx + j * k;
            int snn = 0;
            for (int p = 0; p < k; ++p)
                for (int q = 0; q < k; ++q)
                    if (a[p] == b[q]) { ++snn; break; }
            if (snn > 0) {
                out_t[ne] = t; out_j[ne] = j; out_snn[ne] = snn; ++ne;
            }
        }
    }
    return ne;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
