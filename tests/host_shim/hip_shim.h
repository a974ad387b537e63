// hip_shim.h -- TEST INFRASTRUCTURE: just enough of the HIP runtime, on HOST memory, to compile nabo_amd/csrc/sharded.hip
// with g++ (-DNABO_SHARDED_HOST) and run its control flow on a box without a GPU: communicators, the loopback
// rendezvous, status agreements, the exchange / merge / certificate / second round / gather sequence of
// nabo_sharded_query, its failure semantics.  "Device" memory is malloc'ed, streams are synchronous, a kernel launch runs
// its grid as nested host loops (blockIdx / threadIdx are thread-local variables), RCCL is absent (the loader fails, as on
// a host without librccl).  What a rank's nabo_index would compute on the GPU -- its candidate lists, its certified
// local top-k -- is INJECTED by the test through callbacks (tests/test_sharded_host.py fills them from the oracle).
#pragma once
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorInvalidValue = 1, hipErrorInvalidDevice = 101 };
typedef struct shim_stream *hipStream_t;
struct shim_event { std::chrono::steady_clock::time_point t; };
typedef shim_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1 };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
#define __global__
#define __restrict__
extern thread_local dim3 blockIdx, threadIdx, blockDim;

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : e == hipErrorOutOfMemory ? "out of memory" : "error"; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 64; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipMalloc(void **p, size_t bytes);                 // (fails on request: nabo_host_fail_next_malloc)
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new shim_event(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }

// a launch = the grid as host loops (x only: the protocol's kernels are one-dimensional)
template <typename F> inline void shim_launch(dim3 g, dim3 b, F &&body)
{
    blockDim = b;
    for (unsigned bx = 0; bx < g.x; ++bx)
        for (unsigned tx = 0; tx < b.x; ++tx) {
            blockIdx = dim3(bx);
            threadIdx = dim3(tx);
            body();
        }
}
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) shim_launch((grid), (block), [&]() { kern(__VA_ARGS__); })

// ---- RCCL: types only; load_rccl() fails in the host build (no multi-GPU transport on a CPU box) ---------------------------
typedef struct shim_nccl_comm *ncclComm_t;
typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclInProgress = 7 };
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
enum { ncclUint8 = 1, ncclInt64 = 4, ncclFloat64 = 8, ncclMax = 2 };
ncclResult_t ncclGetUniqueId(ncclUniqueId *);
ncclResult_t ncclCommInitRank(ncclComm_t *, int, ncclUniqueId, int);
ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclCommAbort(ncclComm_t);
ncclResult_t ncclCommCount(const ncclComm_t, int *);
ncclResult_t ncclCommGetAsyncError(ncclComm_t, ncclResult_t *);
ncclResult_t ncclAllReduce(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
ncclResult_t ncclAllGather(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
ncclResult_t ncclSend(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclRecv(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
const char *ncclGetErrorString(ncclResult_t);
