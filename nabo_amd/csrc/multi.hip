// multi.hip -- nabo_knn_devices: the array-in / array-out k-NN of nabo_knn with the reference rows sharded over several GPUs of
// one node, for a C caller that has no threads, communicators or device buffers of its own.
//
// Reference call site: Mapping.calc_dist (nabo/_mapping.py:408-444) -> _calc_dist (:48-148), one device there; SURVEY
// section 8(b) sketched the boundary as nabo_knn(..., devices, n_devices, ...).  Everything below is composition of entry points
// this library already exports -- nabo_comm_create_all / _loopback, nabo_index_*, nabo_sharded_query -- one host thread per
// device (what nabo_amd/_sharded.py: ShardedGroup does from Python): rank r holds reference rows [n r / N, n (r + 1) / N) and
// reports global indices (ref_index_base), every rank sees all m target rows, the sharded query leaves the merged, certified
// result on every rank and rank 0's copy goes back to the caller.  One device: nabo_knn itself.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../include/nabo_knn.h"

namespace nabo {
int api_fail(int code, const char *fmt, ...);
}

namespace {

struct RankJob {
    int rc = NABO_OK;
    char msg[256] = "";
};

void run_rank(int r, int N, nabo_comm *comm, int device, const double *X, int64_t m, const double *Y, int64_t n, int g, int k,
              int metric, double f, const uint8_t *ref_mask, int drop_first, int64_t *out_idx, double *out_dist, RankJob *job)
{
    nabo_index *ix = nullptr;
    void *dX = nullptr, *dI = nullptr, *dD = nullptr;
    auto fail_here = [&](int rc) {
        job->rc = rc;
        snprintf(job->msg, sizeof(job->msg), "rank %d (device %d): %s", r, device, nabo_last_error());
    };
    const int64_t lo = n * r / N, hi = n * (r + 1) / N;
    int rc = nabo_index_create(&ix, device, hi - lo, g, metric, f, lo);
    if (!rc) rc = nabo_index_set_ref(ix, Y + lo * g, 0, ref_mask ? ref_mask + lo : nullptr);
    if (!rc) rc = nabo_dev_malloc(device, &dX, (size_t)m * g * sizeof(double));
    if (!rc) rc = nabo_dev_malloc(device, &dI, (size_t)m * k * sizeof(int64_t));
    if (!rc) rc = nabo_dev_malloc(device, &dD, (size_t)m * k * sizeof(double));
    if (!rc) rc = nabo_memcpy_h2d(device, dX, X, (size_t)m * g * sizeof(double));
    // (a rank that failed alone still enters the collective: nabo_sharded_query agrees on the status before anything is
    // exchanged -- with a NULL index / buffers it reports its error and every rank returns)
    if (rc) fail_here(rc);
    const int qrc = nabo_sharded_query(comm, rc ? nullptr : ix, static_cast<const double *>(dX), m, k, drop_first,
                                       static_cast<int64_t *>(dI), static_cast<double *>(dD), 0);
    if (!rc && qrc) fail_here(qrc);
    if (!job->rc && r == 0) {
        rc = nabo_memcpy_d2h(device, out_idx, dI, (size_t)m * k * sizeof(int64_t));
        if (!rc) rc = nabo_memcpy_d2h(device, out_dist, dD, (size_t)m * k * sizeof(double));
        if (rc) fail_here(rc);
    }
    if (dX) nabo_dev_free(device, dX);
    if (dI) nabo_dev_free(device, dI);
    if (dD) nabo_dev_free(device, dD);
    if (ix) nabo_index_destroy(ix);
}

}  // namespace

extern "C" int nabo_knn_devices(const double *X, int64_t m, const double *Y, int64_t n, int32_t g, int32_t k, int32_t metric,
                                double dist_factor, const uint8_t *ref_mask, int32_t drop_first, const int32_t *devices,
                                int32_t n_devices, int32_t transport, int64_t *out_idx, double *out_dist)
{
    if (!X || !Y || !out_idx || !out_dist || !devices) return nabo::api_fail(NABO_E_INVALID, "NULL argument");
    if (n_devices < 1 || n_devices > 64) return nabo::api_fail(NABO_E_INVALID, "n_devices=%d: 1 .. 64 devices of one node", n_devices);
    if (transport != 0 && transport != 1) return nabo::api_fail(NABO_E_INVALID, "transport: 0 = RCCL, 1 = loopback (one GPU, rehearsal)");
    if (n_devices == 1)
        return nabo_knn(X, m, Y, n, g, k, metric, dist_factor, ref_mask, drop_first, out_idx, out_dist, devices[0]);
    if (n < n_devices) return nabo::api_fail(NABO_E_INVALID, "fewer reference rows (%lld) than devices (%d)", (long long)n, n_devices);
    if (m < 1 || k < 1 || g < 1) return nabo::api_fail(NABO_E_INVALID, "bad shape");
    std::vector<nabo_comm *> comms((size_t)n_devices, nullptr);
    int rc = transport == 1 ? nabo_comm_create_loopback(comms.data(), devices, n_devices)
                            : nabo_comm_create_all(comms.data(), devices, n_devices);
    if (rc) return rc;
    std::vector<RankJob> jobs((size_t)n_devices);
    std::vector<std::thread> th;
    int started = 0;
    try {
        for (int r = 0; r < n_devices; ++r) {
            th.emplace_back(run_rank, r, (int)n_devices, comms[(size_t)r], (int)devices[r], X, m, Y, n, (int)g, (int)k, (int)metric,
                            dist_factor, ref_mask, (int)drop_first, out_idx, out_dist, &jobs[(size_t)r]);
            ++started;
        }
    } catch (...) {
        // a rank that never started would leave the others waiting in their first collective: give the communicators up
        for (int r = 0; r < n_devices; ++r) nabo_comm_abort(comms[(size_t)r]);
    }
    for (std::thread &t : th) t.join();
    rc = started == n_devices ? NABO_OK : NABO_E_NOMEM;
    const char *msg = started == n_devices ? "" : "could not start a host thread per device";
    for (int r = 0; r < n_devices && !rc; ++r)           // the first rank with an error of its OWN, else any error
        if (jobs[(size_t)r].rc && jobs[(size_t)r].rc != NABO_E_COMM) { rc = jobs[(size_t)r].rc; msg = jobs[(size_t)r].msg; }
    for (int r = 0; r < n_devices && !rc; ++r)
        if (jobs[(size_t)r].rc) { rc = jobs[(size_t)r].rc; msg = jobs[(size_t)r].msg; }
    for (nabo_comm *c : comms) nabo_comm_destroy(c);
    return rc ? nabo::api_fail(rc, "%s", msg) : NABO_OK;
}
