"""ctypes binding of libnabo_knn.so (include/nabo_knn.h).  No torch, no CPU fallback:
if the HIP library is missing or no GPU is visible, every compute entry point raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# NABO_KNN_SO: load another build of the library (kernel A/B experiments, tools/ab) WITHOUT overwriting the product's
SO_PATH = os.environ.get("NABO_KNN_SO") or os.path.join(HERE, "libnabo_knn.so")

EUCLIDEAN = 0
MOD_CANBERRA = 1
COSINE = 2          # extension: not in the reference (include/nabo_knn.h)
MAX_COMPS = 128
MAX_K = 56

E_INVALID, E_NODEVICE, E_HIP, E_NOMEM, E_UNSUPPORTED, E_COMM = -1, -2, -3, -4, -5, -6

_lib = None

# every symbol include/nabo_knn.h declares (tests check the .so exports exactly these)
SYMBOLS = [
    "nabo_version", "nabo_last_error", "nabo_device_count", "nabo_knn", "nabo_pairwise",
    "nabo_index_create", "nabo_index_destroy", "nabo_index_set_option", "nabo_query_plan", "nabo_index_set_ref", "nabo_index_set_mask", "nabo_index_query", "nabo_index_query_async", "nabo_index_query_wait",
    "nabo_index_query_candidates",
    "nabo_index_last_stats", "nabo_index_last_kernel", "nabo_index_last_passes", "nabo_index_last_row_pass", "nabo_merge_topk", "nabo_snn_counts", "nabo_pyset_order", "nabo_component_labels", "nabo_group_edges", "nabo_score_null", "nabo_score_null_edges", "nabo_dev_malloc", "nabo_dev_free",
    "nabo_memcpy_h2d", "nabo_memcpy_d2h", "nabo_dev_synchronize", "nabo_dev_mem_info",
    "nabo_comm_unique_id", "nabo_comm_create", "nabo_comm_create_all", "nabo_comm_create_loopback", "nabo_comm_destroy",
    "nabo_comm_rank", "nabo_comm_world", "nabo_comm_transport_ranks", "nabo_comm_abort", "nabo_comm_set_timeout", "nabo_comm_set_ref_shards", "nabo_comm_barrier", "nabo_comm_allreduce_max_f64", "nabo_candidates_per_shard",
    "nabo_sharded_query", "nabo_sharded_last_stats", "nabo_knn_devices",
]


class NaboError(RuntimeError):
    pass


def lib():
    """Load libnabo_knn.so; fail loudly if it has not been built (python -m nabo_amd._build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise NaboError("libnabo_knn.so not found at %s -- build it with `python -m nabo_amd._build` "
                        "(there is no CPU fallback)" % SO_PATH)
    L = C.CDLL(SO_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.nabo_version.restype = C.c_char_p
    L.nabo_last_error.restype = C.c_char_p
    L.nabo_device_count.restype = C.c_int
    L.nabo_knn.argtypes = [vp, i64, vp, i64, i32, i32, i32, dbl, vp, i32, vp, vp, i32]
    L.nabo_pairwise.argtypes = [vp, i64, vp, i64, i32, i32, dbl, vp, i32]
    L.nabo_index_create.argtypes = [C.POINTER(vp), i32, i64, i32, i32, dbl, i64]
    L.nabo_index_destroy.argtypes = [vp]
    L.nabo_index_set_option.argtypes = [vp, C.c_char_p, i64]
    L.nabo_query_plan.argtypes = [i64, i32, i32, i64, i32, i32, i32, i32, C.c_char_p, C.c_char_p, C.POINTER(i64), C.c_char_p, C.c_size_t]
    L.nabo_index_set_ref.argtypes = [vp, vp, i32, vp]
    L.nabo_index_set_mask.argtypes = [vp, vp]
    L.nabo_index_query.argtypes = [vp, vp, i32, i64, i32, i32, vp, vp, i32]
    L.nabo_index_query_async.argtypes = [vp, vp, i32, i64, i32, i32, vp, vp, i32]
    L.nabo_index_query_wait.argtypes = [vp]
    L.nabo_index_query_candidates.argtypes = [vp, vp, i32, i64, i32, vp, vp, vp]
    L.nabo_index_last_stats.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64)]
    L.nabo_index_last_kernel.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.nabo_index_last_passes.argtypes = [vp, C.POINTER(i64)]
    L.nabo_index_last_row_pass.argtypes = [vp, vp, i64]
    L.nabo_merge_topk.argtypes = [i32, vp, vp, i32, i64, i32, i32, i32, vp, vp]
    L.nabo_snn_counts.argtypes = [i32, vp, i64, vp, i64, i32, vp]
    L.nabo_dev_malloc.argtypes = [i32, C.POINTER(vp), C.c_size_t]
    L.nabo_dev_free.argtypes = [i32, vp]
    L.nabo_memcpy_h2d.argtypes = [i32, vp, vp, C.c_size_t]
    L.nabo_memcpy_d2h.argtypes = [i32, vp, vp, C.c_size_t]
    L.nabo_dev_synchronize.argtypes = [i32]
    L.nabo_dev_mem_info.argtypes = [i32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.nabo_comm_unique_id.argtypes = [vp]
    L.nabo_comm_create.argtypes = [C.POINTER(vp), i32, i32, i32, vp]
    L.nabo_comm_create_all.argtypes = [C.POINTER(vp), C.POINTER(i32), i32]
    L.nabo_comm_create_loopback.argtypes = [C.POINTER(vp), C.POINTER(i32), i32]
    L.nabo_comm_destroy.argtypes = [vp]
    L.nabo_comm_rank.argtypes = [vp]
    L.nabo_comm_world.argtypes = [vp]
    L.nabo_comm_transport_ranks.argtypes = [vp]
    L.nabo_comm_set_ref_shards.argtypes = [vp, i32]
    L.nabo_comm_abort.argtypes = [vp]
    L.nabo_comm_set_timeout.argtypes = [vp, dbl]
    L.nabo_comm_barrier.argtypes = [vp]
    L.nabo_comm_allreduce_max_f64.argtypes = [vp, C.POINTER(dbl)]
    L.nabo_candidates_per_shard.argtypes = [i32, i32, i64]
    L.nabo_sharded_query.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, i32]
    L.nabo_sharded_last_stats.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64)]
    L.nabo_knn_devices.argtypes = [vp, i64, vp, i64, i32, i32, i32, dbl, vp, i32, C.POINTER(i32), i32, i32, vp, vp]
    for name in SYMBOLS:
        if name not in ("nabo_version", "nabo_last_error"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def so_digest():
    """sha256 (first 16 hex digits) of the library file in use: bench.py prints it with every line"""
    import hashlib
    with open(SO_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


# sources a kernel's counter record depends on (profiles/pmc.json): the kernel, what it includes, the operand packing,
# the launch logic and the compiler flags
KERNEL_SOURCES = {
    "euclid": ["l2c_topk.hip", "l2q_topk.hip", "topk_lists.h", "knn_common.h", "pack.hip", "api.hip", "_build.py"],
    "canberra": ["canberra_f32.hip", "canberra_bits.hip", "knn_common.h", "api.hip", "_build.py"],
}


def src_digest(names=None):
    """sha256 (first 16 hex digits) over sources the library is built from -- by default all of nabo_amd/csrc/*.hip,
    *.h, include/nabo_knn.h and nabo_amd/_build.py (the compiler flags); `names` restricts it (KERNEL_SOURCES).  Unlike
    the bytes of the .so it is the same after a rebuild on another box: profiles/pmc.json keys its records by it."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(HERE, "csrc")
    if names is None:
        files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
        files += [os.path.join(HERE, "..", "include", "nabo_knn.h"), os.path.join(HERE, "_build.py")]
    else:
        files = [os.path.join(HERE if f.endswith(".py") else csrc, f) for f in sorted(names)]
    for fn in files:
        h.update(os.path.basename(fn).encode())
        with open(fn, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def check(rc):
    """Map a C status to the reference's exception convention: bad parameters -> ValueError
    (nabo/_mapping.py:301,429,520,...), everything else -> NaboError."""
    if rc == 0:
        return
    msg = lib().nabo_last_error().decode("utf-8", "replace")
    if rc in (E_INVALID, E_UNSUPPORTED):
        raise ValueError("ERROR: " + msg)
    raise NaboError("nabo_knn error %d: %s" % (rc, msg))


def device_count():
    return int(lib().nabo_device_count())


def mem_info(device=0):
    """(free, total) bytes of the device's memory"""
    f, t = C.c_size_t(), C.c_size_t()
    check(lib().nabo_dev_mem_info(int(device), C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)
