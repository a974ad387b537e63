// sharded.hip -- reference rows sharded over the GPUs of one node: communicators and the
// sharded query of the C ABI (include/nabo_knn.h: nabo_comm_*, nabo_sharded_query).  No torch, no MPI.
//
// The reference has no multi-device path (its _calc_dist, nabo/_mapping.py:48-148, is one Python loop); the call
// site this entry point serves is Mapping.calc_dist (nabo/_mapping.py:441-444).
//
// Transports
//   * RCCL over xGMI (the product path): librccl.so is dlopen'ed on first use -- a host that never shards does not
//     need it -- one communicator per GPU, created either per rank (one process per GPU: ncclCommInitRank with a
//     unique id the caller passes around) or for all devices of one process (ncclCommInitAll; one host thread per
//     device then drives its rank).  Every exchange is ONE grouped operation: the candidate lists, their distances and
//     the bounds go out as ncclSend/ncclRecv pairs inside a single ncclGroupStart/End, the result slices as two
//     ncclAllGather in one group.  xGMI is point to point: each peer pair moves only the m/N rows the receiver owns.
//   * loopback: N ranks as host threads of ONE process, rendezvous through a host barrier and device-to-device
//     copies.  Same call sequence, same buffers, same kernels -- it exists so that the whole protocol can be run (and
//     is tested) with N shards on a single GPU; it is also a correct transport for several peer-accessible devices.
//
// Protocol (rank r of N holds reference rows [base_r, base_r + n_r); every rank sees all m target rows; rank r OWNS
// target rows [r*mr, (r+1)*mr), mr = ceil(m/N)):
//   global certification (Euclidean / cosine, k' = k + drop_first):
//     1. nabo_index_query_candidates: the first Ls entries of the shard's order rows + a lower bound on the squared
//        distance of everything the shard did not emit (Ls = nabo_candidates_per_shard(k', N, m));
//     2. one exchange: owner(row) receives N lists + N bounds;
//     3. merge by (distance, index) to k' entries; the owner accepts a row when d_k'^2 (1+1e-12) < min bound: no
//        unreported reference anywhere can then enter or tie;
//     4. refused rows (a shard held >= Ls of the global top-k'): MAX-all-reduce of the count; if any, all-gather of the
//        row ids, exact local top-k' of just those rows on every shard (nabo_index_query), all-gather, merge;
//     5. positional drop (nabo/_mapping.py:142 is positional) AFTER the merge, all-gather of the [mr,k] slices.
//   local certification (modified Canberra, or k'/N beyond the candidate lists): every shard's certified top-k',
//     the same exchange and merge.
// The merge is deterministic, so N shards == 1 shard bit for bit.
//
// Failure semantics (the reference is one process, nabo/_mapping.py:48-148: there is nothing to match -- the rule here
// is "no rank ever waits for a peer that has already given up"):
//   * every phase a rank can fail in ALONE (argument checks, buffer reservation, its local queries) ends in a status
//     agreement -- one small MAX all-reduce that also checks that all ranks were handed the same m / k / drop_first /
//     protocol -- so either every rank goes on or every rank returns an error; the communicator stays usable;
//   * an error INSIDE a collective (RCCL failure, a kernel launch between two collectives, a peer that never arrives)
//     aborts the communicator: ncclCommAbort for RCCL, the hub's abort flag for the loopback transport; an opened
//     RCCL group is always closed first.  Peers blocked in the same collective then return NABO_E_COMM instead of
//     hanging: host waits on an RCCL stream poll hipStreamQuery + ncclCommGetAsyncError with a deadline
//     (nabo_comm_set_timeout, NABO_COMM_TIMEOUT_S, default 600 s), the loopback barrier is a timed condition wait;
//   * a shard with fewer than k' references takes part with absent entries (-1) instead of failing its local query.
#include <dlfcn.h>
#ifdef NABO_SHARDED_HOST
// tests/host_shim: this file compiled with g++ against host memory, so that a box WITHOUT a GPU runs the compiled control
// flow of the protocol below (tests/test_sharded_host.py); the product build never defines it
#include "hip_shim.h"
#else
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#endif
#include <pthread.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <utility>
#include <vector>

#include "../../include/nabo_knn.h"

namespace nabo {
hipError_t merge_parts_launch(const double *parts_d, const int64_t *parts_i, int n_parts, int64_t m, int kp, int k,
                              int drop, int64_t *out_idx, double *out_dist, hipStream_t st);
hipError_t gather_rows_launch(const double *X, const uint32_t *rows, int64_t nrows, int g, double *out, hipStream_t st);
// api.hip
int api_fail(int code, const char *fmt, ...);
int index_device(const nabo_index *ix);
int index_g(const nabo_index *ix);
int64_t index_n(const nabo_index *ix);
int index_metric(const nabo_index *ix);
bool index_can_emit_candidates(const nabo_index *ix);
void index_set_shard_mode(nabo_index *ix, bool on);
void index_set_cand_slack(nabo_index *ix, int s);
}  // namespace nabo

namespace {

using nabo::api_fail;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            (void)hipGetLastError(); /* (the thread's sticky copy: a later launch check must not report THIS failure) */ \
            return api_fail(e__ == hipErrorOutOfMemory ? NABO_E_NOMEM : NABO_E_HIP, "%s failed: %s", \
                            #expr, hipGetErrorString(e__));                                        \
        }                                                                                          \
    } while (0)

// ---- librccl.so, resolved at run time -------------------------------------------------------------------
struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl g_rccl;
pthread_mutex_t g_rccl_lock = PTHREAD_MUTEX_INITIALIZER;

int load_rccl()
{
    pthread_mutex_lock(&g_rccl_lock);
    if (!g_rccl.h) {
        const char *env = getenv("NABO_RCCL_LIB");
        const char *names[] = {env, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
        void *h = nullptr;
        for (const char *nm : names)
            if (nm && *nm && (h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) {
            pthread_mutex_unlock(&g_rccl_lock);
            return api_fail(NABO_E_UNSUPPORTED, "librccl.so could not be loaded (%s): multi-GPU sharding needs RCCL", dlerror());
        }
        bool ok = true;
#define NABO_SYM(field, name)                                                    \
    do {                                                                         \
        g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
        ok = ok && g_rccl.field;                                                 \
    } while (0)
        NABO_SYM(GetUniqueId, "ncclGetUniqueId");
        NABO_SYM(CommInitRank, "ncclCommInitRank");
        NABO_SYM(CommInitAll, "ncclCommInitAll");
        NABO_SYM(CommDestroy, "ncclCommDestroy");
        NABO_SYM(CommAbort, "ncclCommAbort");
        NABO_SYM(CommCount, "ncclCommCount");
        NABO_SYM(CommGetAsyncError, "ncclCommGetAsyncError");
        NABO_SYM(AllReduce, "ncclAllReduce");
        NABO_SYM(AllGather, "ncclAllGather");
        NABO_SYM(Send, "ncclSend");
        NABO_SYM(Recv, "ncclRecv");
        NABO_SYM(GroupStart, "ncclGroupStart");
        NABO_SYM(GroupEnd, "ncclGroupEnd");
        NABO_SYM(GetErrorString, "ncclGetErrorString");
#undef NABO_SYM
        if (!ok) {
            dlclose(h);
            pthread_mutex_unlock(&g_rccl_lock);
            return api_fail(NABO_E_UNSUPPORTED, "librccl.so lacks a required entry point");
        }
        g_rccl.h = h;
    }
    pthread_mutex_unlock(&g_rccl_lock);
    return NABO_OK;
}

// ---- loopback rendezvous ---------------------------------------------------------------------------------
// An abortable, timed barrier: a rank that fails, or nabo_comm_abort from any thread, releases everyone who waits (and
// everyone who will), and a rank whose peers never arrive gives up after the deadline and aborts the hub itself.
constexpr int NABO_AGREE_MAX = 8;

struct LoopHub {
    int n = 0;
    int refs = 0;
    pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
    pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
    int arrived = 0;
    unsigned long gen = 0;
    bool aborted = false;
    std::vector<const void *> ptr;
    std::vector<int64_t> vals;          // [n][NABO_AGREE_MAX]
};

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return NABO_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return api_fail(NABO_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e)); }
        cap = want;
        return NABO_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double default_timeout_s()
{
    const char *s = getenv("NABO_COMM_TIMEOUT_S");
    const double v = (s && *s) ? atof(s) : 0.0;
    return v > 0.0 ? v : 600.0;
}

}  // namespace

struct nabo_comm {
    int kind = 0;                    // 0 RCCL, 1 loopback
    int device = 0, rank = 0, world = 1;
    ncclComm_t nccl = nullptr;
    LoopHub *hub = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    DevBuf ci, cd, cb, ri, rd, rb, mi, md, oi, od, fulli, fulld, cnt, bad, ids, allids, sel, xb, bi, bd, gi, gd, fi, fd, scratch, agree, pi, pd;
    double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t counters[4] = {0, 0, 0, 0};
    // 2-D layout (nabo_comm_set_ref_shards): the references are cut into ref_shards pieces, rank r holds piece
    // r % ref_shards and answers for target slice r / ref_shards; 0 = world (every rank its own piece: the 1-D form)
    int ref_shards = 0;
    double timeout_s = 600.0;        // deadline of every wait on a peer
    std::atomic<bool> aborted{false};   // set once (exchange: one thread wins); every later call fails with NABO_E_COMM
    // nabo_comm_abort may come from ANY thread while the rank's own thread polls the handle: ncclCommAbort frees it, so the
    // handle is only touched under this lock and never again once nccl_dead is set (the pointer itself stays until destroy)
    pthread_mutex_t nccl_lock = PTHREAD_MUTEX_INITIALIZER;
    bool nccl_dead = false;
    bool group_open = false;         // the rank's thread has an RCCL group open (closed BEFORE an abort: see rccl_failed)
    bool agreed = false;             // the error being returned was agreed on by all ranks (no abort needed)
};

namespace {

// The communicator is finished: release whoever waits on it.  RCCL: ncclCommAbort (peers' pending operations end with an
// error, our own stream is released); loopback: the hub's flag + a broadcast.  Idempotent; callable from any thread.
void comm_abort(nabo_comm *c)
{
    if (!c || c->aborted.exchange(true)) return;             // one caller aborts, every other one returns
    if (c->kind == 0) {
        pthread_mutex_lock(&c->nccl_lock);
        if (c->nccl && !c->nccl_dead && g_rccl.CommAbort) { (void)g_rccl.CommAbort(c->nccl); c->nccl_dead = true; }
        pthread_mutex_unlock(&c->nccl_lock);
    } else if (c->hub) {
        pthread_mutex_lock(&c->hub->lock);
        c->hub->aborted = true;
        pthread_cond_broadcast(&c->hub->cv);
        pthread_mutex_unlock(&c->hub->lock);
    }
}

int comm_dead(nabo_comm *c)
{
    return api_fail(NABO_E_COMM, "rank %d: the communicator was aborted (an earlier collective failed or timed out)", c->rank);
}

// An RCCL call of the rank's own thread failed: an open group is closed FIRST (ncclGroupEnd on operations of a freed
// communicator is undefined), then the communicator is aborted.
void rccl_failed(nabo_comm *c)
{
    if (c->group_open) {
        c->group_open = false;
        (void)g_rccl.GroupEnd();
    }
    comm_abort(c);
}

#define RCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        if (c->aborted.load()) return comm_dead(c);      /* the handle may be gone: never enqueue on it */ \
        ncclResult_t r__ = (expr);                                                                      \
        if (r__ != ncclSuccess) {                                                                       \
            const int rc__ = api_fail(NABO_E_COMM, "%s failed: %s", #expr, g_rccl.GetErrorString(r__)); \
            rccl_failed(c);                                                                             \
            return rc__;                                                                                \
        }                                                                                               \
    } while (0)

// One RCCL group, closed on every path out of the scope that opened it.
struct Group {
    nabo_comm *c;
    bool open = false;
    explicit Group(nabo_comm *cc) : c(cc) {}
    int begin()
    {
        if (c->kind == 0) {
            RCCL_TRY(g_rccl.GroupStart());
            open = c->group_open = true;
        }
        return NABO_OK;
    }
    int end()
    {
        if (open) {
            open = false;
            if (!c->group_open) return comm_dead(c);          // rccl_failed closed it on the way out of a failed call
            c->group_open = false;
            ncclResult_t r = g_rccl.GroupEnd();
            if (r != ncclSuccess) {
                const int rc = api_fail(NABO_E_COMM, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(r));
                comm_abort(c);
                return rc;
            }
        }
        return NABO_OK;
    }
    ~Group()
    {
        // (an error return between begin and end: the thread's group state must not leak into its next RCCL call)
        if (open && c->group_open) {
            c->group_open = false;
            (void)g_rccl.GroupEnd();
        }
    }
};

// Host wait for the communicator's stream.  Work that depends on peers (RCCL kernels) is waited for by polling, with
// the asynchronous error state of the communicator and a deadline in the loop: a peer that died or never entered the
// collective turns into an error here, not a hang.
int stream_wait(nabo_comm *c)
{
    if (c->kind != 0 || c->world == 1 || !c->nccl) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        return NABO_OK;
    }
    if (c->aborted.load()) return comm_dead(c);
    const double t0 = now_s();
    for (unsigned spins = 1;; ++spins) {
        const hipError_t e = hipStreamQuery(c->stream);
        if (e == hipSuccess) return NABO_OK;
        if (e != hipErrorNotReady) {
            const int rc = api_fail(NABO_E_HIP, "hipStreamQuery failed: %s", hipGetErrorString(e));
            comm_abort(c);
            return rc;
        }
        if ((spins & 255) == 0) {
            ncclResult_t ar = ncclSuccess;
            bool dead;
            pthread_mutex_lock(&c->nccl_lock);                // (an abort from another thread frees the handle)
            dead = c->nccl_dead;
            const bool have = !dead && g_rccl.CommGetAsyncError(c->nccl, &ar) == ncclSuccess;
            pthread_mutex_unlock(&c->nccl_lock);
            if (dead) return comm_dead(c);
            if (have && ar != ncclSuccess && ar != ncclInProgress) {
                const int rc = api_fail(NABO_E_COMM, "rank %d: RCCL reported an asynchronous error: %s", c->rank, g_rccl.GetErrorString(ar));
                comm_abort(c);
                return rc;
            }
            if (now_s() - t0 > c->timeout_s) {
                const int rc = api_fail(NABO_E_COMM, "rank %d: a collective did not complete within %.0f s (a peer is missing or has failed); "
                                        "communicator aborted", c->rank, c->timeout_s);
                comm_abort(c);
                return rc;
            }
            if (spins > 65536) usleep(50);
        }
    }
}

// loopback barrier: NABO_OK when all n ranks arrived; NABO_E_COMM when the hub was aborted or the deadline passed
int hub_wait(nabo_comm *c)
{
    LoopHub *h = c->hub;
    int rc = NABO_OK;
    pthread_mutex_lock(&h->lock);
    if (h->aborted) {
        rc = NABO_E_COMM;
    } else {
        const unsigned long gen0 = h->gen;
        if (++h->arrived == h->n) {
            h->arrived = 0;
            ++h->gen;
            pthread_cond_broadcast(&h->cv);
        } else {
            timespec dl;
            clock_gettime(CLOCK_REALTIME, &dl);
            const double t = (double)dl.tv_sec + 1e-9 * (double)dl.tv_nsec + c->timeout_s;
            dl.tv_sec = (time_t)t;
            dl.tv_nsec = (long)((t - (double)dl.tv_sec) * 1e9);
            while (h->gen == gen0 && !h->aborted)
                if (pthread_cond_timedwait(&h->cv, &h->lock, &dl) != 0 && h->gen == gen0) {      // ETIMEDOUT: give up for everyone
                    h->aborted = true;
                    pthread_cond_broadcast(&h->cv);
                }
            if (h->gen == gen0) rc = NABO_E_COMM;
        }
    }
    pthread_mutex_unlock(&h->lock);
    if (rc) {
        c->aborted.store(true);
        return api_fail(NABO_E_COMM, "rank %d: the loopback group was aborted (a peer failed, or did not arrive within %.0f s)", c->rank, c->timeout_s);
    }
    return NABO_OK;
}

// ---- collectives (device pointers, on c->stream) ----------------------------------------------------------
// Among the ranks [first, first + count) (the caller's rank is one of them; every rank of the world makes the call,
// with its own group): block b of `send` (bytes each) goes to peer first + b; block b of `recv` comes from peer first + b.
// RCCL: the caller holds an open Group (several exchanges travel as one grouped operation).
int all_to_all(nabo_comm *c, const void *send, void *recv, size_t bytes, int first = 0, int count = -1)
{
    const int N = count < 0 ? c->world : count;
    const int me = c->rank - first;
    if (bytes == 0) return NABO_OK;
    if (c->kind == 0) {
        for (int b = 0; b < N; ++b) {
            RCCL_TRY(g_rccl.Send(static_cast<const char *>(send) + (size_t)b * bytes, bytes, ncclUint8, first + b, c->nccl, c->stream));
            RCCL_TRY(g_rccl.Recv(static_cast<char *>(recv) + (size_t)b * bytes, bytes, ncclUint8, first + b, c->nccl, c->stream));
        }
        return NABO_OK;
    }
    int rc;
    HIP_TRY(hipStreamSynchronize(c->stream));                 // my send buffer is final
    c->hub->ptr[c->rank] = send;
    if ((rc = hub_wait(c))) return rc;
    for (int b = 0; b < N; ++b)
        HIP_TRY(hipMemcpyAsync(static_cast<char *>(recv) + (size_t)b * bytes,
                               static_cast<const char *>(c->hub->ptr[first + b]) + (size_t)me * bytes, bytes, hipMemcpyDefault,
                               c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return hub_wait(c);                                       // nobody reuses a send buffer before all have copied
}

int all_gather(nabo_comm *c, const void *send, void *recv, size_t bytes)
{
    const int N = c->world;
    if (bytes == 0) return NABO_OK;
    if (c->kind == 0) {
        RCCL_TRY(g_rccl.AllGather(send, recv, bytes, ncclUint8, c->nccl, c->stream));
        return NABO_OK;
    }
    int rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->hub->ptr[c->rank] = send;
    if ((rc = hub_wait(c))) return rc;
    for (int p = 0; p < N; ++p)
        HIP_TRY(hipMemcpyAsync(static_cast<char *>(recv) + (size_t)p * bytes, c->hub->ptr[p], bytes, hipMemcpyDefault, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return hub_wait(c);
}

// MAX over ranks of `nv` (<= NABO_AGREE_MAX) int64 values living on the device (in place) -- returned on the host too
int all_reduce_max(nabo_comm *c, int64_t *dev_val, int64_t *host_out, int nv = 1)
{
    int rc;
    if (c->kind == 0) {
        RCCL_TRY(g_rccl.AllReduce(dev_val, dev_val, (size_t)nv, ncclInt64, ncclMax, c->nccl, c->stream));
        HIP_TRY(hipMemcpyAsync(host_out, dev_val, sizeof(int64_t) * nv, hipMemcpyDeviceToHost, c->stream));
        return stream_wait(c);
    }
    int64_t mine[NABO_AGREE_MAX];
    HIP_TRY(hipMemcpyAsync(mine, dev_val, sizeof(int64_t) * nv, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < nv; ++i) c->hub->vals[(size_t)c->rank * NABO_AGREE_MAX + i] = mine[i];
    if ((rc = hub_wait(c))) return rc;
    for (int i = 0; i < nv; ++i) {
        int64_t mx = c->hub->vals[i];
        for (int p = 1; p < c->world; ++p) mx = std::max(mx, c->hub->vals[(size_t)p * NABO_AGREE_MAX + i]);
        host_out[i] = mx;
    }
    return hub_wait(c);
}

// Status agreement at the end of a phase a rank can fail in ALONE: every rank enters with its own status; either all
// return NABO_OK or all return an error (a rank with a local error keeps its own code and message).  `args` (optional,
// n_args <= 3 values): what every rank must have been handed identically -- a mismatch is an error on every rank.
int agree(nabo_comm *c, int rc_local, const char *phase, const int64_t *args = nullptr, int n_args = 0)
{
    c->agreed = false;
    if (c->world == 1) { c->agreed = rc_local != NABO_OK; return rc_local; }
    char keep[512] = "";
    if (rc_local) snprintf(keep, sizeof(keep), "%s", nabo_last_error());
    int64_t v[NABO_AGREE_MAX] = {0, 0, 0, 0, 0, 0, 0, 0}, out[NABO_AGREE_MAX];
    v[0] = rc_local ? -(int64_t)rc_local : 0;                  // status codes are negative
    for (int i = 0; i < n_args && i < 3; ++i) { v[1 + 2 * i] = args[i]; v[2 + 2 * i] = -args[i]; }
    const int nv = 1 + 2 * (n_args < 3 ? n_args : 3);
    int rc = c->agree.reserve(sizeof(v));
    if (!rc) {
        hipError_t e = hipMemcpyAsync(c->agree.p, v, sizeof(int64_t) * nv, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);           // v is a stack array
        if (e != hipSuccess) rc = api_fail(NABO_E_HIP, "status agreement: %s", hipGetErrorString(e));
    }
    if (rc) {               // this rank cannot even take part: release the others
        comm_abort(c);
        return rc;
    }
    if ((rc = all_reduce_max(c, c->agree.as<int64_t>(), out, nv))) return rc;      // (the communicator is aborted already)
    c->agreed = true;
    if (rc_local) return api_fail(rc_local, "%s", keep);
    if (out[0] != 0)
        return api_fail(NABO_E_COMM, "rank %d: a peer failed in the %s phase (status %lld); no rank went on", c->rank, phase, -(long long)out[0]);
    for (int i = 0; i < n_args && i < 3; ++i)
        if (out[1 + 2 * i] != -out[2 + 2 * i])
            return api_fail(NABO_E_INVALID, "rank %d: the ranks were handed different arguments (%s: argument %d ranges over [%lld, %lld])",
                            c->rank, phase, i, -(long long)out[2 + 2 * i], (long long)out[1 + 2 * i]);
    c->agreed = false;
    return NABO_OK;
}

// ---- small kernels of the protocol ------------------------------------------------------------------------
// owner's certificate: row r of my slice is final when its k'-th merged distance lies below every shard's bound
__global__ void certify_kernel(const int64_t *__restrict__ mi, const double *__restrict__ md, int kk,
                               const double *__restrict__ bounds /*[N][mr]*/, int N, int64_t mr, int64_t row0, int64_t m,
                               int64_t *__restrict__ bad_rows, unsigned long long *__restrict__ bad_count)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= mr || row0 + r >= m) return;                     // padding rows are nobody's
    double b = bounds[r];
    for (int s = 1; s < N; ++s) b = fmin(b, bounds[(int64_t)s * mr + r]);
    const double dk = md[r * kk + kk - 1];
    const bool ok = mi[r * kk + kk - 1] >= 0 && dk * dk * (1.0 + 1e-12) < b;
    if (!ok) bad_rows[atomicAdd(bad_count, 1ull)] = row0 + r;
}

// columns [d0, d0+k) of the merged [mr, kk] rows (the positional drop comes AFTER the merge)
__global__ void slice_kernel(const int64_t *__restrict__ mi, const double *__restrict__ md, int64_t mr, int kk, int d0,
                             int k, int64_t *__restrict__ oi, double *__restrict__ od)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= mr * k) return;
    const int64_t r = e / k;
    const int c = (int)(e - r * k);
    oi[e] = mi[r * kk + d0 + c];
    od[e] = md[r * kk + d0 + c];
}

// second round: rows of `sel` that I own replace my merged rows
__global__ void adopt_kernel(const uint32_t *__restrict__ sel, int64_t nb, const int64_t *__restrict__ fi,
                             const double *__restrict__ fd, int kk, int64_t row0, int64_t mr, int64_t *__restrict__ mi,
                             double *__restrict__ md)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nb * kk) return;
    const int64_t b = e / kk;
    const int64_t row = (int64_t)sel[b] - row0;
    if (row < 0 || row >= mr) return;
    mi[row * kk + (e - b * kk)] = fi[e];
    md[row * kk + (e - b * kk)] = fd[e];
}

__global__ void fill_absent_kernel(int64_t *__restrict__ idx, double *__restrict__ dist, int64_t n_idx, double *__restrict__ bnd, int64_t n_bnd)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_idx) { idx[e] = -1; dist[e] = __builtin_inf(); }
    if (bnd && e < n_bnd) bnd[e] = __builtin_inf();
}

// rows of kq entries -> rows of kk >= kq entries, the tail absent (a shard with fewer than k' references)
__global__ void widen_kernel(const int64_t *__restrict__ si, const double *__restrict__ sd, int64_t m, int kq, int kk,
                             int64_t *__restrict__ oi, double *__restrict__ od)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * kk) return;
    const int64_t r = e / kk;
    const int c = (int)(e - r * kk);
    oi[e] = c < kq ? si[r * kq + c] : -1;
    od[e] = c < kq ? sd[r * kq + c] : __builtin_inf();
}

int use_dev(int device)
{
    HIP_TRY(hipSetDevice(device));
    return NABO_OK;
}

int comm_alloc(nabo_comm **out, int kind, int device, int rank, int world)
{
    nabo_comm *c = new (std::nothrow) nabo_comm();
    if (!c) return api_fail(NABO_E_NOMEM, "host allocation failed");
    c->kind = kind; c->device = device; c->rank = rank; c->world = world;
    c->timeout_s = default_timeout_s();
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; i < 8 && e == hipSuccess; ++i) e = hipEventCreate(&c->ev[i]);
    if (e != hipSuccess) {
        (void)hipGetLastError();          // (the thread's sticky copy: a later launch check must not report THIS failure)
        delete c;
        return api_fail(e == hipErrorInvalidDevice ? NABO_E_NODEVICE : NABO_E_HIP, "communicator on device %d: %s", device,
                        hipGetErrorString(e));
    }
    *out = c;
    return NABO_OK;
}

float ev_ms(nabo_comm *c, int a, int b)
{
    float t = 0;
    (void)hipEventElapsedTime(&t, c->ev[a], c->ev[b]);
    return t;
}

}  // namespace

extern "C" {

int32_t nabo_candidates_per_shard(int32_t kk, int32_t world, int64_t m)
{
    // smallest list length that leaves an expected < 0.1 rows of the batch for the second round:
    // world * m * P[Bin(kk, 1/world) >= Ls] < 0.1 (exchangeable shards); never more than kk+1, at least ceil(kk/world)
    if (kk < 1 || world < 1) return 0;
    const int cap = std::min(kk + 1, 32);
    const double p = 1.0 / world;
    double tail = 0.0;
    int ls = cap;
    for (int j = kk; j >= 1; --j) {
        double cb = 1.0;                       // C(kk, j)
        for (int i = 1; i <= j; ++i) cb = cb * (double)(kk - j + i) / (double)i;
        tail += cb * std::pow(p, j) * std::pow(1.0 - p, kk - j);
        if (tail * world * (double)(m > 0 ? m : 1) >= 0.1) { ls = j + 1; break; }
        ls = j;
    }
    ls = std::min(ls, cap);
    ls = std::max(ls, (kk + world - 1) / world);
    return std::max(ls, 1);
}

int nabo_comm_unique_id(void *id128)
{
    if (!id128) return api_fail(NABO_E_INVALID, "NULL argument");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return api_fail(NABO_E_COMM, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString(r));
    static_assert(sizeof(id) == NABO_COMM_ID_BYTES, "unique id size");
    memcpy(id128, &id, sizeof(id));
    return NABO_OK;
}

int nabo_comm_create(nabo_comm **out, int32_t device, int32_t rank, int32_t world, const void *id128)
{
    if (!out || !id128) return api_fail(NABO_E_INVALID, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return api_fail(NABO_E_INVALID, "rank %d of %d", rank, world);
    int rc = load_rccl();
    if (rc) return rc;
    nabo_comm *c = nullptr;
    if ((rc = comm_alloc(&c, 0, device, rank, world))) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
        c->nccl = nullptr;
        nabo_comm_destroy(c);
        return api_fail(NABO_E_COMM, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, g_rccl.GetErrorString(r));
    }
    *out = c;
    return NABO_OK;
}

int nabo_comm_create_all(nabo_comm **out, const int32_t *devices, int32_t n)
{
    if (!out || !devices || n < 1) return api_fail(NABO_E_INVALID, "bad argument");
    int rc = load_rccl();
    if (rc) return rc;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return api_fail(NABO_E_NODEVICE, "no HIP device is available");
    for (int i = 0; i < n; ++i) {
        if (devices[i] < 0 || devices[i] >= cnt)
            return api_fail(NABO_E_NODEVICE, "device %d out of range (have %d): RCCL needs one GPU per rank", devices[i], cnt);
        for (int j = 0; j < i; ++j)
            if (devices[j] == devices[i])
                return api_fail(NABO_E_INVALID, "device %d listed twice: RCCL needs one GPU per rank (the loopback transport allows repeats)", devices[i]);
    }
    std::vector<ncclComm_t> comms((size_t)n);
    std::vector<int> devs(devices, devices + n);
    const ncclResult_t r = g_rccl.CommInitAll(comms.data(), n, devs.data());
    if (r != ncclSuccess) return api_fail(NABO_E_COMM, "ncclCommInitAll over %d devices failed: %s", n, g_rccl.GetErrorString(r));
    for (int i = 0; i < n; ++i) out[i] = nullptr;
    for (int i = 0; i < n; ++i) {
        nabo_comm *c = nullptr;
        if ((rc = comm_alloc(&c, 0, devices[i], i, n))) {
            for (int j = 0; j < n; ++j) {
                if (out[j]) { out[j]->nccl = nullptr; nabo_comm_destroy(out[j]); out[j] = nullptr; }
                (void)g_rccl.CommDestroy(comms[(size_t)j]);
            }
            return rc;
        }
        c->nccl = comms[(size_t)i];
        out[i] = c;
    }
    return NABO_OK;
}

int nabo_comm_create_loopback(nabo_comm **out, const int32_t *devices, int32_t n)
{
    if (!out || !devices || n < 1) return api_fail(NABO_E_INVALID, "bad argument");
    LoopHub *hub = new (std::nothrow) LoopHub();
    if (!hub) return api_fail(NABO_E_NOMEM, "host allocation failed");
    hub->n = n;
    hub->refs = n;
    hub->ptr.assign((size_t)n, nullptr);
    hub->vals.assign((size_t)n * NABO_AGREE_MAX, 0);
    for (int i = 0; i < n; ++i) out[i] = nullptr;
    for (int i = 0; i < n; ++i) {
        nabo_comm *c = nullptr;
        int rc = comm_alloc(&c, 1, devices[i], i, n);
        if (rc) {
            for (int j = 0; j < i; ++j) { out[j]->hub = nullptr; nabo_comm_destroy(out[j]); out[j] = nullptr; }
            delete hub;
            return rc;
        }
        c->hub = hub;
        out[i] = c;
    }
    return NABO_OK;
}

int nabo_comm_destroy(nabo_comm *c)
{
    if (!c) return NABO_OK;
    (void)hipSetDevice(c->device);
    if (c->nccl && !c->nccl_dead && g_rccl.CommDestroy) {
        // a stream that still waits for a peer must not block the teardown
        if (c->stream && hipStreamQuery(c->stream) == hipErrorNotReady && g_rccl.CommAbort) (void)g_rccl.CommAbort(c->nccl);
        else (void)g_rccl.CommDestroy(c->nccl);
        c->nccl = nullptr;
    }
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->hub) {
        pthread_mutex_lock(&c->hub->lock);
        const int left = --c->hub->refs;
        if (left > 0 && c->hub->arrived > 0) {          // peers are waiting for a rank that is going away
            c->hub->aborted = true;
            pthread_cond_broadcast(&c->hub->cv);
        }
        pthread_mutex_unlock(&c->hub->lock);
        if (left == 0) delete c->hub;
    }
    for (int i = 0; i < 8; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return NABO_OK;
}

int nabo_comm_rank(const nabo_comm *c) { return c ? c->rank : -1; }
int nabo_comm_world(const nabo_comm *c) { return c ? c->world : -1; }

int nabo_comm_transport_ranks(nabo_comm *c)
{
    if (!c) return api_fail(NABO_E_INVALID, "NULL communicator");
    if (c->aborted.load()) return comm_dead(c);
    if (c->kind != 0) return c->hub ? c->hub->n : 1;
    int count = -1;
    pthread_mutex_lock(&c->nccl_lock);
    const ncclResult_t r = (c->nccl && !c->nccl_dead) ? g_rccl.CommCount(c->nccl, &count) : ncclSuccess;
    pthread_mutex_unlock(&c->nccl_lock);
    if (r != ncclSuccess || count < 0) return api_fail(NABO_E_COMM, "ncclCommCount failed");
    return count;
}

int nabo_comm_abort(nabo_comm *c)
{
    if (!c) return api_fail(NABO_E_INVALID, "NULL communicator");
    comm_abort(c);
    return NABO_OK;
}

int nabo_comm_set_timeout(nabo_comm *c, double seconds)
{
    if (!c || !(seconds > 0.0)) return api_fail(NABO_E_INVALID, "bad argument");
    c->timeout_s = seconds;
    return NABO_OK;
}

int nabo_comm_set_ref_shards(nabo_comm *c, int32_t ref_shards)
{
    if (!c) return api_fail(NABO_E_INVALID, "NULL communicator");
    if (ref_shards < 0 || (ref_shards > 0 && c->world % ref_shards != 0))
        return api_fail(NABO_E_INVALID, "ref_shards = %d does not divide the world size %d", ref_shards, c->world);
    c->ref_shards = ref_shards == c->world ? 0 : ref_shards;
    return NABO_OK;
}

int nabo_comm_allreduce_max_f64(nabo_comm *c, double *value)
{
    if (!c || !value) return api_fail(NABO_E_INVALID, "NULL argument");
    if (c->aborted) return comm_dead(c);
    int rc = use_dev(c->device);
    if (rc) return rc;
    if (c->world == 1) return NABO_OK;
    if ((rc = c->scratch.reserve(64))) { comm_abort(c); return rc; }
    if (c->kind == 0) {
        HIP_TRY(hipMemcpyAsync(c->scratch.p, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
        RCCL_TRY(g_rccl.AllReduce(c->scratch.p, c->scratch.p, 1, ncclFloat64, ncclMax, c->nccl, c->stream));
        HIP_TRY(hipMemcpyAsync(value, c->scratch.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        return stream_wait(c);
    }
    int64_t bits;
    memcpy(&bits, value, sizeof(bits));        // callers pass non-negative times: IEEE order == integer order
    c->hub->vals[(size_t)c->rank * NABO_AGREE_MAX] = bits;
    if ((rc = hub_wait(c))) return rc;
    int64_t mx = c->hub->vals[0];
    for (int p = 1; p < c->world; ++p) mx = std::max(mx, c->hub->vals[(size_t)p * NABO_AGREE_MAX]);
    if ((rc = hub_wait(c))) return rc;
    memcpy(value, &mx, sizeof(mx));
    return NABO_OK;
}

int nabo_comm_barrier(nabo_comm *c)
{
    double z = 0.0;
    return nabo_comm_allreduce_max_f64(c, &z);
}

static int sharded_query_impl(nabo_comm *c, nabo_index *ix, const double *X, int64_t m, int32_t k, int32_t drop_first,
                              int64_t *out_idx, double *out_dist, int32_t protocol);

int nabo_sharded_query(nabo_comm *c, nabo_index *ix, const double *X, int64_t m, int32_t k, int32_t drop_first,
                       int64_t *out_idx, double *out_dist, int32_t protocol)
{
    if (!c) return api_fail(NABO_E_INVALID, "NULL communicator");
    if (c->aborted) return comm_dead(c);
    // With more than one shard, a shard's local queries must not continue a short row with its MASKED references
    // (api.hip: tail_len): they would enter the merge as neighbours.  Rows with fewer than k' unmasked references in
    // the WHOLE reference set then end in absent entries (-1 / NaN) instead of the ignored references by index.
    const bool shards = ix && c->world > 1;
    if (shards) nabo::index_set_shard_mode(ix, true);
    c->agreed = false;
    const int rc = sharded_query_impl(c, ix, X, m, k, drop_first, out_idx, out_dist, protocol);
    if (shards) nabo::index_set_shard_mode(ix, false);
    // An error every rank agreed on leaves the communicator usable; anything else happened between two collectives
    // (or inside one): peers may be waiting for this rank -- release them.
    if (rc && !c->agreed && c->world > 1) {
        char keep[512];
        snprintf(keep, sizeof(keep), "%s", nabo_last_error());
        comm_abort(c);
        (void)api_fail(rc, "%s", keep);
    }
    return rc;
}

// This rank's first kk order-row entries of rows X [m,g] into out_i / out_d [m,kk]; a shard with fewer than kk
// references answers with what it has and absent entries behind (nabo_index_query itself refuses kk > n_ref).
static int local_topk(nabo_comm *c, nabo_index *ix, const double *X, int64_t m, int kk, int64_t *out_i, double *out_d)
{
    const int64_t n = nabo::index_n(ix);
    if ((int64_t)kk <= n) return nabo_index_query(ix, X, 1, m, kk, 0, out_i, out_d, 1);
    const int kq = (int)n;
    int rc;
    if ((rc = c->pi.reserve((size_t)m * kq * 8)) || (rc = c->pd.reserve((size_t)m * kq * 8))) return rc;
    if ((rc = nabo_index_query(ix, X, 1, m, kq, 0, c->pi.as<int64_t>(), c->pd.as<double>(), 1))) return rc;
    (void)hipSetDevice(c->device);
    const unsigned blk = 256;
    hipLaunchKernelGGL(widen_kernel, dim3((unsigned)((m * kk + blk - 1) / blk)), dim3(blk), 0, c->stream, c->pi.as<int64_t>(),
                       c->pd.as<double>(), m, kq, kk, out_i, out_d);
    HIP_TRY(hipGetLastError());
    return NABO_OK;
}

static int sharded_query_impl(nabo_comm *c, nabo_index *ix, const double *X, int64_t m, int32_t k, int32_t drop_first,
                              int64_t *out_idx, double *out_dist, int32_t protocol)
{
    // ---- phase 0 (local): arguments, protocol, buffers; agreed on before anything is exchanged ----------------
    int rc = NABO_OK;
    const int N = c->world;
    const int d0 = drop_first ? 1 : 0, kk = k + d0;
    const int R = c->ref_shards > 0 ? c->ref_shards : N;
    bool global = false;
    int g = 0, Ls = 0;
    if (!ix || !X || !out_idx || !out_dist) rc = api_fail(NABO_E_INVALID, "NULL argument");
    else if (m < 1 || k < 1) rc = api_fail(NABO_E_INVALID, "bad shape m=%lld k=%d", (long long)m, k);
    else if (nabo::index_device(ix) != c->device) rc = api_fail(NABO_E_INVALID, "index and communicator live on different devices");
    else rc = use_dev(c->device);
    const int64_t mr = m > 0 ? (m + N - 1) / N : 0, m_pad = mr * N, row0 = (int64_t)c->rank * mr;
    // 2-D layout: R reference pieces x N / R target slices; my group = the R ranks [gfirst, gfirst + R) that hold the
    // pieces for my slice, rows [s0, s0 + R mr) (ms of them exist).  R = N: one group, the whole batch (the 1-D form).
    const int gfirst = (c->rank / R) * R;
    const int64_t s0 = (int64_t)gfirst * mr, ms_pad = (int64_t)R * mr;
    const int64_t ms = m - s0 < 0 ? 0 : (m - s0 < ms_pad ? m - s0 : ms_pad);
    hipStream_t st = c->stream;
    if (!rc) {
        g = nabo::index_g(ix);
        // protocol: 0 auto, 1 global certification, 2 local certification
        const bool can_cand = nabo::index_can_emit_candidates(ix) && (kk + R - 1) / R <= 32;
        if (protocol < 0 || protocol > 2) rc = api_fail(NABO_E_INVALID, "protocol %d (0 auto, 1 global, 2 local certification)", protocol);
        else if (R != N && R != 1 && (protocol == 2 || !can_cand))
            rc = api_fail(NABO_E_UNSUPPORTED, "the 2-D shard layout (ref_shards = %d of %d ranks) needs the global-certification protocol", R, N);
        else if (R == 1 && N > 1 && protocol == 2)
            rc = api_fail(NABO_E_UNSUPPORTED, "pure target slicing (ref_shards = 1) runs through the global protocol's merge and gather (protocol 0 or 1)");
        else if (protocol == 1 && !can_cand && R != 1)
            rc = api_fail(NABO_E_UNSUPPORTED, "global certification needs the Euclidean / cosine filter and k'/N <= 32");
        else {
            // (R = 1: every metric -- a rank's certified local query of its own slice needs no candidate lists)
            global = protocol == 1 || (protocol == 0 && (can_cand || R == 1) && N > 1);
            // (R = 1, pure target slicing: a rank holds ALL the references, its own certified first k' entries ARE the answer --
            // they travel through the same merge / certificate / gather with a bound of +inf)
            Ls = global ? (R == 1 ? kk : nabo_candidates_per_shard(kk, R, m)) : 0;
            // the owner's merge sorts one wave-wide batch of at most 1024 (distance, index) pairs per row
            if (global && (int64_t)R * Ls > 1024)
                rc = api_fail(NABO_E_UNSUPPORTED, "ref_shards * candidates per shard = %d x %d exceeds the merge width 1024", R, Ls);
            else if ((int64_t)(global ? R : N) * kk > 1024)
                rc = api_fail(NABO_E_UNSUPPORTED, "shards * (k + drop_first) = %d x %d exceeds the merge width 1024", global ? R : N, kk);
        }
    }
    for (double &v : c->ms) v = 0.0;
    c->counters[0] = c->counters[1] = c->counters[2] = 0;
    c->counters[3] = global ? 1 : 2;
    c->counters[1] = Ls;
    // every buffer of the call is reserved HERE, before the agreement: an allocation that fails later would fail
    // between two collectives
    auto reserve_all = [&rc](std::initializer_list<std::pair<DevBuf *, size_t>> bufs) {
        for (const auto &b : bufs)
            if (!rc) rc = b.first->reserve(b.second);
    };
    if (!rc) {
        const size_t li = global ? (size_t)ms_pad * Ls * 8 : (size_t)m_pad * kk * 8, lb = global ? (size_t)ms_pad * 8 : 0;
        const size_t full = m_pad != m ? (size_t)m_pad * k * 8 : 0;
        reserve_all({{&c->mi, (size_t)mr * kk * 8}, {&c->md, (size_t)mr * kk * 8}, {&c->ci, li}, {&c->cd, li}, {&c->ri, li},
                     {&c->rd, li}, {&c->cb, lb}, {&c->rb, lb}, {&c->cnt, 64}, {&c->bad, (size_t)mr * 8},
                     {&c->oi, (size_t)mr * k * 8}, {&c->od, (size_t)mr * k * 8}, {&c->fulli, full}, {&c->fulld, full}});
    }
    {
        const int64_t args[3] = {m, (int64_t)k * 2 + d0, protocol};
        if ((rc = agree(c, rc, "argument", args, 3))) return rc;
    }
    HIP_TRY(hipEventRecord(c->ev[0], st));
    const unsigned blk = 256;

    // ---- phase 1 (local): this shard's lists --------------------------------------------------------------------
    if (global) {
        if (ms_pad != ms) {        // ragged tail of my slice: absent entries, +inf bounds
            const int64_t nx = (ms_pad - ms) * Ls;
            hipLaunchKernelGGL(fill_absent_kernel, dim3((unsigned)((std::max(nx, ms_pad - ms) + blk - 1) / blk)), dim3(blk), 0, st,
                               c->ci.as<int64_t>() + ms * Ls, c->cd.as<double>() + ms * Ls, nx, c->cb.as<double>() + ms, ms_pad - ms);
            if (hipGetLastError() != hipSuccess) rc = api_fail(NABO_E_HIP, "fill_absent_kernel launch failed");
        }
        // One-product first pass (api.hip): with few pieces a shard's Ls-th candidate is close to the global k'-th, and the
        // certificate needs the exact distance of the first candidate left out, not the one-product threshold (three
        // kept entries more than emitted); with many pieces it lies far beyond it and the shorter lists win (one rank of
        // eight: 25 instead of 32 ms, one refused row at 1M x 1M).
        nabo::index_set_cand_slack(ix, Ls >= kk ? 3 : 0);
        if (!rc && ms > 0 && R == 1) {
            // one piece: the certified local query (with its whole chain of passes behind the first filter), nothing is left out
            hipLaunchKernelGGL(fill_absent_kernel, dim3((unsigned)((ms + blk - 1) / blk)), dim3(blk), 0, st, (int64_t *)nullptr,
                               (double *)nullptr, (int64_t)0, c->cb.as<double>(), ms);
            if (hipGetLastError() != hipSuccess) rc = api_fail(NABO_E_HIP, "fill_absent_kernel launch failed");
            if (!rc) rc = local_topk(c, ix, X + s0 * g, ms, kk, c->ci.as<int64_t>(), c->cd.as<double>());
        } else if (!rc && ms > 0)
            rc = nabo_index_query_candidates(ix, X + s0 * g, 1, ms, Ls, c->ci.as<int64_t>(), c->cd.as<double>(), c->cb.as<double>());
    } else {
        // local certification: every shard's own first k' order-row entries
        if (m_pad != m) {
            const int64_t nx = (m_pad - m) * kk;
            hipLaunchKernelGGL(fill_absent_kernel, dim3((unsigned)((nx + blk - 1) / blk)), dim3(blk), 0, st,
                               c->ci.as<int64_t>() + m * kk, c->cd.as<double>() + m * kk, nx, (double *)nullptr, (int64_t)0);
            if (hipGetLastError() != hipSuccess) rc = api_fail(NABO_E_HIP, "fill_absent_kernel launch failed");
        }
        if (!rc) rc = local_topk(c, ix, X, m, kk, c->ci.as<int64_t>(), c->cd.as<double>());
    }
    (void)hipSetDevice(c->device);
    if (!rc) {          // the dominant kernel of this rank's share (a second round would overwrite the index's own record)
        double ims[5] = {0, 0, 0, 0, 0};
        (void)nabo_index_last_stats(ix, ims, nullptr);
        c->ms[7] = ims[1];
    }
    if ((rc = agree(c, rc, "local query"))) return rc;
    HIP_TRY(hipEventRecord(c->ev[1], st));

    // ---- phase 2 (collective): exchange, merge, certificate ------------------------------------------------------
    if (global) {
        {
            Group grp(c);
            if ((rc = grp.begin())) return rc;
            if ((rc = all_to_all(c, c->ci.p, c->ri.p, (size_t)mr * Ls * 8, gfirst, R))) return rc;
            if ((rc = all_to_all(c, c->cd.p, c->rd.p, (size_t)mr * Ls * 8, gfirst, R))) return rc;
            if ((rc = all_to_all(c, c->cb.p, c->rb.p, (size_t)mr * 8, gfirst, R))) return rc;
            if ((rc = grp.end())) return rc;
        }
        HIP_TRY(hipEventRecord(c->ev[2], st));
        HIP_TRY(nabo::merge_parts_launch(c->rd.as<double>(), c->ri.as<int64_t>(), R, mr, Ls, kk, 0, c->mi.as<int64_t>(),
                                         c->md.as<double>(), st));
        HIP_TRY(hipMemsetAsync(c->cnt.p, 0, 16, st));
        hipLaunchKernelGGL(certify_kernel, dim3((unsigned)((mr + blk - 1) / blk)), dim3(blk), 0, st, c->mi.as<int64_t>(),
                           c->md.as<double>(), kk, c->rb.as<double>(), R, mr, row0, m, c->bad.as<int64_t>(),
                           c->cnt.as<unsigned long long>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev[3], st));
        // my count stays in cnt[0]; the MAX over ranks goes through cnt[1]
        HIP_TRY(hipMemcpyAsync(c->cnt.as<int64_t>() + 1, c->cnt.p, 8, hipMemcpyDeviceToDevice, st));
        int64_t nb_max = 0;
        if ((rc = all_reduce_max(c, c->cnt.as<int64_t>() + 1, &nb_max))) return rc;
        if (nb_max > 0) {
            // ---- second round: rows some owner refused, re-solved exactly on every piece ------------------------
            int64_t mine = 0;
            std::vector<int64_t> ids((size_t)nb_max, -1);
            int64_t nb = 0;
            std::vector<uint32_t> sel;                                 // rank-major, identical on every rank
            HIP_TRY(hipMemcpyAsync(&mine, c->cnt.p, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (mine > 0) {
                HIP_TRY(hipMemcpyAsync(ids.data(), c->bad.p, (size_t)mine * 8, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                std::sort(ids.begin(), ids.begin() + mine);           // the atomics' order is not reproducible
            }
            rc = NABO_OK;
            reserve_all({{&c->ids, (size_t)nb_max * 8}, {&c->allids, (size_t)N * nb_max * 8}});
            if ((rc = agree(c, rc, "second-round buffers"))) return rc;
            HIP_TRY(hipMemcpyAsync(c->ids.p, ids.data(), (size_t)nb_max * 8, hipMemcpyHostToDevice, st));
            if ((rc = all_gather(c, c->ids.p, c->allids.p, (size_t)nb_max * 8))) return rc;
            std::vector<int64_t> all((size_t)N * nb_max);
            HIP_TRY(hipMemcpyAsync(all.data(), c->allids.p, all.size() * 8, hipMemcpyDeviceToHost, st));
            if ((rc = stream_wait(c))) return rc;
            for (int64_t v : all)
                if (v >= 0) sel.push_back((uint32_t)v);
            nb = (int64_t)sel.size();
            c->counters[0] = nb;
            // (local again: buffers for, and the exact query of, the refused rows on this rank's piece)
            rc = NABO_OK;
            reserve_all({{&c->sel, (size_t)nb * 4}, {&c->xb, (size_t)nb * g * 8}, {&c->bi, (size_t)nb * kk * 8},
                         {&c->bd, (size_t)nb * kk * 8}, {&c->gi, (size_t)N * nb * kk * 8}, {&c->gd, (size_t)N * nb * kk * 8},
                         {&c->fi, (size_t)nb * kk * 8}, {&c->fd, (size_t)nb * kk * 8}});
            if (!rc) {
                hipError_t e = hipMemcpyAsync(c->sel.p, sel.data(), (size_t)nb * 4, hipMemcpyHostToDevice, st);
                if (e == hipSuccess) e = nabo::gather_rows_launch(X, c->sel.as<uint32_t>(), nb, g, c->xb.as<double>(), st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
                if (e != hipSuccess) rc = api_fail(NABO_E_HIP, "second round: %s", hipGetErrorString(e));
            }
            if (!rc) rc = local_topk(c, ix, c->xb.as<double>(), nb, kk, c->bi.as<int64_t>(), c->bd.as<double>());
            (void)hipSetDevice(c->device);
            if ((rc = agree(c, rc, "second-round query"))) return rc;
            {
                Group grp(c);
                if ((rc = grp.begin())) return rc;
                if ((rc = all_gather(c, c->bi.p, c->gi.p, (size_t)nb * kk * 8))) return rc;
                if ((rc = all_gather(c, c->bd.p, c->gd.p, (size_t)nb * kk * 8))) return rc;
                if ((rc = grp.end())) return rc;
            }
            // (every rank re-solved every refused row on its reference piece; the R parts of MY group cover all pieces)
            HIP_TRY(nabo::merge_parts_launch(c->gd.as<double>() + (size_t)gfirst * nb * kk, c->gi.as<int64_t>() + (size_t)gfirst * nb * kk,
                                             R, nb, kk, kk, 0, c->fi.as<int64_t>(), c->fd.as<double>(), st));
            hipLaunchKernelGGL(adopt_kernel, dim3((unsigned)((nb * kk + blk - 1) / blk)), dim3(blk), 0, st, c->sel.as<uint32_t>(), nb,
                               c->fi.as<int64_t>(), c->fd.as<double>(), kk, row0, mr, c->mi.as<int64_t>(), c->md.as<double>());
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(c->ev[4], st));
    } else {
        if (N > 1) {
            Group grp(c);
            if ((rc = grp.begin())) return rc;
            if ((rc = all_to_all(c, c->ci.p, c->ri.p, (size_t)mr * kk * 8))) return rc;
            if ((rc = all_to_all(c, c->cd.p, c->rd.p, (size_t)mr * kk * 8))) return rc;
            if ((rc = grp.end())) return rc;
        }
        HIP_TRY(hipEventRecord(c->ev[2], st));
        HIP_TRY(nabo::merge_parts_launch(N > 1 ? c->rd.as<double>() : c->cd.as<double>(), N > 1 ? c->ri.as<int64_t>() : c->ci.as<int64_t>(),
                                         N, mr, kk, kk, 0, c->mi.as<int64_t>(), c->md.as<double>(), st));
        HIP_TRY(hipEventRecord(c->ev[3], st));
        HIP_TRY(hipEventRecord(c->ev[4], st));
    }
    // positional drop after the merge, then every rank gets every owner's slice
    hipLaunchKernelGGL(slice_kernel, dim3((unsigned)((mr * k + blk - 1) / blk)), dim3(blk), 0, st, c->mi.as<int64_t>(),
                       c->md.as<double>(), mr, kk, d0, k, c->oi.as<int64_t>(), c->od.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev[5], st));
    int64_t *fi = out_idx;
    double *fd = out_dist;
    if (m_pad != m) {
        fi = c->fulli.as<int64_t>();
        fd = c->fulld.as<double>();
    }
    if (N > 1) {
        Group grp(c);
        if ((rc = grp.begin())) return rc;
        if ((rc = all_gather(c, c->oi.p, fi, (size_t)mr * k * 8))) return rc;
        if ((rc = all_gather(c, c->od.p, fd, (size_t)mr * k * 8))) return rc;
        if ((rc = grp.end())) return rc;
    } else {
        HIP_TRY(hipMemcpyAsync(fi, c->oi.p, (size_t)mr * k * 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(fd, c->od.p, (size_t)mr * k * 8, hipMemcpyDeviceToDevice, st));
    }
    if (m_pad != m) {
        HIP_TRY(hipMemcpyAsync(out_idx, fi, (size_t)m * k * 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(out_dist, fd, (size_t)m * k * 8, hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipEventRecord(c->ev[6], st));
    if ((rc = stream_wait(c))) return rc;
    // ms: [0] local query, [1] exchange, [2] merge + certificate, [3] second round, [4] slice, [5] gather, [6] total,
    // [7] the distance + top-k kernel inside [0]
    for (int i = 0; i < 6; ++i) c->ms[i] = ev_ms(c, i, i + 1);
    c->ms[6] = ev_ms(c, 0, 6);
    return NABO_OK;
}

int nabo_sharded_last_stats(const nabo_comm *c, double ms[8], int64_t counters[4])
{
    if (!c) return api_fail(NABO_E_INVALID, "NULL communicator");
    if (ms) memcpy(ms, c->ms, sizeof(c->ms));
    if (counters) memcpy(counters, c->counters, sizeof(c->counters));
    return NABO_OK;
}

}  // extern "C"
