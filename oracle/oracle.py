"""ctypes wrapper over oracle/libnabo_oracle.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libnabo_oracle.so")
EUCLIDEAN, MOD_CANBERRA = 0, 1
COSINE = 2        # extension, not in the reference (parity unpinned) -- see nabo_oracle.c
_lib = None


def build(force=False):
    src = os.path.join(HERE, "nabo_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "-B", "libnabo_oracle.so"])
    return SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(SO)
        dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        ip = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
        L.oracle_pairwise.argtypes = [dp, C.c_int64, dp, C.c_int64, C.c_int32, C.c_int32, C.c_double, dp, C.c_int32]
        L.oracle_pairwise.restype = C.c_int
        L.oracle_knn.argtypes = [dp, C.c_int64, dp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                 C.c_void_p, C.c_int32, ip, dp, C.c_int32]
        L.oracle_knn.restype = C.c_int
        L.oracle_snn_counts.argtypes = [ip, C.c_int64, ip, C.c_int64, C.c_int32, ip, ip,
                                        np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")]
        L.oracle_snn_counts.restype = C.c_int64
        L.oracle_max_threads.restype = C.c_int
        _lib = L
    return _lib


def max_threads():
    return int(lib().oracle_max_threads())


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def pairwise(X, Y, metric=EUCLIDEAN, dist_factor=0.25, nthreads=1):
    """Literal a1/a2 (nabo/_mapping.py:16-45): dense D[m,n] float64."""
    X, Y = _f64(X), _f64(Y)
    assert X.ndim == 2 and Y.ndim == 2 and X.shape[1] == Y.shape[1]
    D = np.empty((X.shape[0], Y.shape[0]), dtype=np.float64)
    rc = lib().oracle_pairwise(X, X.shape[0], Y, Y.shape[0], X.shape[1], metric, float(dist_factor), D, nthreads)
    if rc:
        raise ValueError("oracle_pairwise rc=%d" % rc)
    return D


def knn(X, Y, k, metric=EUCLIDEAN, dist_factor=0.25, ref_mask=None, drop_first=False, nthreads=1):
    """First k entries of the order rows of nabo/_mapping.py:135-146 (+ their distances),
    ties/masked ordered by the canonical (masked last, dist, idx) rule."""
    X, Y = _f64(X), _f64(Y)
    assert X.ndim == 2 and Y.ndim == 2 and X.shape[1] == Y.shape[1]
    m, n = X.shape[0], Y.shape[0]
    idx = np.empty((m, k), dtype=np.int64)
    dist = np.empty((m, k), dtype=np.float64)
    mp = None
    if ref_mask is not None:
        ref_mask = np.ascontiguousarray(ref_mask, dtype=np.uint8)
        assert ref_mask.shape == (n,)
        mp = ref_mask.ctypes.data
    rc = lib().oracle_knn(X, m, Y, n, X.shape[1], k, metric, float(dist_factor), mp, int(bool(drop_first)),
                          idx, dist, nthreads)
    if rc:
        raise ValueError("oracle_knn rc=%d" % rc)
    return idx, dist


def snn_weight(snn, k):
    """nabo/_mapping.py:185,194: round(snn / (2*(k-1) - snn), 2) with Python's round()."""
    return round(snn / (2 * (k - 1) - snn), 2)


def snn_edges(t_idx, r_idx, k):
    """nabo/_mapping.py:186-198: (t, j, weight) for every j in NN_t[:k] with a shared neighbour."""
    t_idx = np.ascontiguousarray(t_idx[:, :k], dtype=np.int64)
    r_idx = np.ascontiguousarray(r_idx[:, :k], dtype=np.int64)
    m, n = t_idx.shape[0], r_idx.shape[0]
    ot = np.empty(m * k, dtype=np.int64)
    oj = np.empty(m * k, dtype=np.int64)
    os_ = np.empty(m * k, dtype=np.int32)
    ne = lib().oracle_snn_counts(t_idx, m, r_idx, n, k, ot, oj, os_)
    w = np.array([snn_weight(int(s), k) for s in os_[:ne]], dtype=np.float64)
    return ot[:ne].copy(), oj[:ne].copy(), w


# ---- permutation null of the mapping score (EXTENSION; definition in include/nabo_knn.h / score_null.hip) --------
def _null_keys(seed, p, n_t, key_bits):
    t = np.arange(n_t, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (np.full(n_t, seed & (2 ** 64 - 1), dtype=np.uint64) +
             np.full(n_t, ((p + 1) * 0x9E3779B97F4A7C15) & (2 ** 64 - 1), dtype=np.uint64) +
             t * np.uint64(0xD1B54A32D192ED03))
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return z >> np.uint64(64 - key_bits)


def score_null(edge_t, edge_ref_idx, edge_weight, group, n_ref, n_perm, seed=0, score_multiplier=1000, key_bits=64):
    """CPU statement of nabo_score_null: plain loops, float64 sums in CSR (stable by reference node) order."""
    edge_t = np.asarray(edge_t, dtype=np.int64)
    edge_r = np.asarray(edge_ref_idx, dtype=np.int64)
    w = np.asarray(edge_weight, dtype=np.float64)
    group = np.asarray(group) != 0
    n_t = group.shape[0]
    n_a = int(group.sum())
    lab = np.empty((n_perm + 1, n_t), dtype=bool)
    sizes = np.empty(n_perm, dtype=np.int64)
    for p in range(n_perm):
        k = _null_keys(seed, p, n_t, key_bits)
        thr = np.partition(k, n_a - 1)[n_a - 1]
        lab[p] = k <= thr
        sizes[p] = int(lab[p].sum())
    lab[n_perm] = group
    order = np.argsort(edge_r, kind="stable")
    acc = np.zeros((n_ref, n_perm + 1))
    for e in order:
        acc[edge_r[e]] = acc[edge_r[e]] + np.where(lab[:, edge_t[e]], w[e], 0.0)
    obs = (score_multiplier * acc[:, n_perm]) / float(n_a)
    sp = (score_multiplier * acc[:, :n_perm]) / sizes[None, :].astype(np.float64)
    n_ge = (sp >= obs[:, None]).sum(axis=1).astype(np.int64)
    return {"obs": obs, "n_ge": n_ge, "null_mean": sp.mean(axis=1), "null_sd": sp.std(axis=1), "sizes": sizes,
            "scores": sp}
