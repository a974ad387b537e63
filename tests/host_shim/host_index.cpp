// host_index.cpp -- TEST INFRASTRUCTURE: the part of libnabo_knn.so that sharded.hip calls into, restated for the host
// build (hip_shim.h): a nabo_index is a record whose candidate query / certified query are CALLBACKS the test installs,
// the merge and gather kernels are plain loops in the canonical (distance, index) order of refine.hip's merge_kernel.
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <vector>

#include "../../include/nabo_knn.h"
#include "hip_shim.h"

thread_local dim3 blockIdx, threadIdx, blockDim;
static thread_local char g_err[512] = "";
static std::atomic<int> g_fail_malloc{0};

hipError_t hipMalloc(void **p, size_t bytes)
{
    if (g_fail_malloc.load() > 0 && g_fail_malloc.fetch_sub(1) == 1) { *p = nullptr; return hipErrorOutOfMemory; }
    *p = malloc(bytes ? bytes : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}

struct nabo_index { int device, g, metric, id; int64_t n; bool shard_mode; int cand_slack; double ms_topk; };

typedef int (*cand_cb_t)(int id, const double *X, int64_t m, int32_t n_cand, int64_t *out_idx, double *out_dist, double *out_bound);
typedef int (*query_cb_t)(int id, const double *X, int64_t m, int32_t k, int64_t *out_idx, double *out_dist);
static cand_cb_t g_cand = nullptr;
static query_cb_t g_query = nullptr;

namespace nabo {
int api_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int index_device(const nabo_index *ix) { return ix->device; }
int index_g(const nabo_index *ix) { return ix->g; }
int64_t index_n(const nabo_index *ix) { return ix->n; }
int index_metric(const nabo_index *ix) { return ix->metric; }
bool index_can_emit_candidates(const nabo_index *ix) { return ix->metric != NABO_METRIC_MOD_CANBERRA; }
void index_set_shard_mode(nabo_index *ix, bool on) { ix->shard_mode = on; }
void index_set_cand_slack(nabo_index *ix, int s) { ix->cand_slack = s; }

// parts [n_parts][m][kp] (idx < 0: absent) -> the first k entries by (distance, index) after dropping `drop` leading ones
hipError_t merge_parts_launch(const double *pd, const int64_t *pi, int n_parts, int64_t m, int kp, int k, int drop,
                              int64_t *oi, double *od, hipStream_t)
{
    std::vector<std::pair<double, int64_t>> v;
    for (int64_t r = 0; r < m; ++r) {
        v.clear();
        for (int p = 0; p < n_parts; ++p)
            for (int c = 0; c < kp; ++c) {
                const int64_t j = pi[((int64_t)p * m + r) * kp + c];
                if (j >= 0) v.emplace_back(pd[((int64_t)p * m + r) * kp + c], j);
            }
        std::sort(v.begin(), v.end());
        for (int c = 0; c < k; ++c) {
            const size_t e = (size_t)(c + drop);
            oi[r * k + c] = e < v.size() ? v[e].second : -1;
            od[r * k + c] = e < v.size() ? v[e].first : NAN;
        }
    }
    return hipSuccess;
}
hipError_t gather_rows_launch(const double *X, const uint32_t *rows, int64_t nrows, int g, double *out, hipStream_t)
{
    for (int64_t i = 0; i < nrows; ++i) memcpy(out + i * g, X + (int64_t)rows[i] * g, (size_t)g * sizeof(double));
    return hipSuccess;
}
}  // namespace nabo

extern "C" {
const char *nabo_last_error(void) { return g_err; }
int nabo_index_query(nabo_index *ix, const double *X, int32_t, int64_t m, int32_t k, int32_t drop_first, int64_t *oi, double *od, int32_t)
{
    if (!g_query || drop_first) return nabo::api_fail(NABO_E_INVALID, "host index: no query callback");
    if ((int64_t)k > ix->n) return nabo::api_fail(NABO_E_INVALID, "k = %d exceeds the %lld references", k, (long long)ix->n);
    const int rc = g_query(ix->id, X, m, k, oi, od);
    return rc ? nabo::api_fail(rc, "host index %d: injected query failure", ix->id) : NABO_OK;
}
int nabo_index_query_candidates(nabo_index *ix, const double *X, int32_t, int64_t m, int32_t n_cand, int64_t *oi, double *od, double *ob)
{
    if (!g_cand) return nabo::api_fail(NABO_E_INVALID, "host index: no candidate callback");
    const int rc = g_cand(ix->id, X, m, n_cand, oi, od, ob);
    return rc ? nabo::api_fail(rc, "host index %d: injected candidate-query failure", ix->id) : NABO_OK;
}
int nabo_index_last_stats(const nabo_index *ix, double ms[5], int64_t counters[4])
{
    if (ms) { for (int i = 0; i < 5; ++i) ms[i] = 0.0; ms[1] = ix->ms_topk; }
    if (counters) for (int i = 0; i < 4; ++i) counters[i] = 0;
    return NABO_OK;
}
// ---- test hooks --------------------------------------------------------------------------------------------------------
nabo_index *nabo_host_index_create(int id, int device, int64_t n, int g, int metric)
{
    return new nabo_index{device, g, metric, id, n, false, 0, 1.0};
}
void nabo_host_index_destroy(nabo_index *ix) { delete ix; }
int nabo_host_index_shard_mode(const nabo_index *ix) { return ix->shard_mode ? 1 : 0; }
void nabo_host_set_callbacks(cand_cb_t c, query_cb_t q) { g_cand = c; g_query = q; }
void nabo_host_fail_nth_malloc(int n) { g_fail_malloc.store(n); }
void *nabo_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void nabo_host_free(void *p) { free(p); }
}
