"""numpy-facing wrappers over the C ABI: the array-in / array-out form of Nabo's
`_calc_dist` tile loop + mask + sort (nabo/_mapping.py:98-146).

Everything here runs on the GPU through libnabo_knn.so; there is no CPU path.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import EUCLIDEAN, MOD_CANBERRA  # noqa: F401


def _f64(a, name):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 2:
        raise ValueError("ERROR: %s must be a 2-D array" % name)
    return a


def _mask(ref_mask, n):
    if ref_mask is None:
        return None, None
    mk = np.ascontiguousarray(ref_mask, dtype=np.uint8)
    if mk.shape != (n,):
        raise ValueError("ERROR: ref_mask must have one entry per reference cell")
    return mk, mk.ctypes.data


def knn(X, Y, k, metric=EUCLIDEAN, dist_factor=0.25, ref_mask=None, drop_first=False, device=0, options=None):
    """First k entries of every order row + their float64 distances (host arrays in/out).

    X [m,g] targets, Y [n,g] references.  Returns (idx int64 [m,k], dist float64 [m,k]).
    options ({name: value} of nabo_index_set_option): goes through a KnnIndex instead of the one-shot nabo_knn."""
    X, Y = _f64(X, "X"), _f64(Y, "Y")
    if X.shape[1] != Y.shape[1]:
        raise ValueError("ERROR: X and Y must have the same number of components")
    m, n, g = X.shape[0], Y.shape[0], X.shape[1]
    if options:
        ix = KnnIndex(n, g, metric=metric, dist_factor=dist_factor, device=device, options=options)
        try:
            return ix.set_ref(Y, ref_mask=ref_mask).query(X, k, drop_first=drop_first)
        finally:
            ix.close()
    mk, mp = _mask(ref_mask, n)
    idx = np.empty((m, k), dtype=np.int64)
    dist = np.empty((m, k), dtype=np.float64)
    rc = _lib.lib().nabo_knn(X.ctypes.data, m, Y.ctypes.data, n, g, int(k), int(metric), float(dist_factor),
                             mp, int(bool(drop_first)), idx.ctypes.data, dist.ctypes.data, int(device))
    _lib.check(rc)
    return idx, dist


def knn_devices(X, Y, k, devices, metric=EUCLIDEAN, dist_factor=0.25, ref_mask=None, drop_first=False, transport="rccl"):
    """nabo_knn_devices: the same call with the reference rows sharded over `devices` (one host thread per device inside the
    library; transport "loopback": devices may repeat -- the whole protocol on one GPU).  Same results as knn()."""
    X, Y = _f64(X, "X"), _f64(Y, "Y")
    if X.shape[1] != Y.shape[1]:
        raise ValueError("ERROR: X and Y must have the same number of components")
    m, n, g = X.shape[0], Y.shape[0], X.shape[1]
    mk, mp = _mask(ref_mask, n)
    dev = (C.c_int32 * len(devices))(*[int(d) for d in devices])
    idx = np.empty((m, k), dtype=np.int64)
    dist = np.empty((m, k), dtype=np.float64)
    if transport not in ("rccl", "loopback"):
        raise ValueError("ERROR: transport must be 'rccl' or 'loopback'")
    _lib.check(_lib.lib().nabo_knn_devices(X.ctypes.data, m, Y.ctypes.data, n, g, int(k), int(metric), float(dist_factor), mp,
                                           int(bool(drop_first)), dev, len(devices), 1 if transport == "loopback" else 0,
                                           idx.ctypes.data, dist.ctypes.data))
    return idx, dist


PLAN_FIELDS = ("first_pass", "geometry", "rows_per_wg", "workgroups_main", "workgroups_tail", "splits", "splits_tail", "lkeep",
               "list_len", "tiles_per_split", "tournament_tiles", "tournament_group", "resident_workgroups", "workgroups",
               "rows_padded", "operand_steps", "pieces", "piece_tiles")


def query_plan(n_ref, g, m, k, metric=EUCLIDEAN, drop_first=False, n_cand=0, n_cu=256, l2_mode=None, options=None):
    """What a Euclidean / cosine query of this shape would launch first (nabo_query_plan: no index, no device)."""
    out = (C.c_int64 * len(PLAN_FIELDS))()
    kern = C.create_string_buffer(192)
    opts = ",".join("%s=%d" % (a, int(b)) for a, b in (options or {}).items())
    _lib.check(_lib.lib().nabo_query_plan(int(n_ref), int(g), int(metric), int(m), int(k), int(bool(drop_first)), int(n_cand),
                                          int(n_cu), l2_mode.encode() if l2_mode else None, opts.encode() if opts else None,
                                          out, kern, 192))
    d = {name: int(out[i]) for i, name in enumerate(PLAN_FIELDS)}
    d["kernel"] = kern.value.decode("ascii", "replace")
    return d


def pairwise(X, Y, metric=EUCLIDEAN, dist_factor=0.25, device=0):
    """Literal a1 / a2 (nabo/_mapping.py:16-45): dense float64 D[m,n] computed on the GPU."""
    X, Y = _f64(X, "X"), _f64(Y, "Y")
    if X.shape[1] != Y.shape[1]:
        raise ValueError("ERROR: X and Y must have the same number of components")
    D = np.empty((X.shape[0], Y.shape[0]), dtype=np.float64)
    rc = _lib.lib().nabo_pairwise(X.ctypes.data, X.shape[0], Y.ctypes.data, Y.shape[0], X.shape[1], int(metric),
                                  float(dist_factor), D.ctypes.data, int(device))
    _lib.check(rc)
    return D


class KnnIndex:
    """References resident in HBM; query many target batches (nabo_index_* in nabo_knn.h).

    `Y` may be a host array or, with `y_device_ptr`, a raw device pointer (borrowed)."""

    def __init__(self, n_ref, g, metric=EUCLIDEAN, dist_factor=0.25, ref_index_base=0, device=0, options=None):
        """options: {name: value} of nabo_index_set_option (tuning / test hooks; every setting returns the same bits)"""
        self._h = C.c_void_p()
        self.n_ref, self.g, self.metric, self.device = int(n_ref), int(g), int(metric), int(device)
        _lib.check(_lib.lib().nabo_index_create(C.byref(self._h), self.device, self.n_ref, self.g, self.metric,
                                                float(dist_factor), int(ref_index_base)))
        self._keep = None
        for name, value in (options or {}).items():
            self.set_option(name, value)

    def set_option(self, name, value):
        _lib.check(_lib.lib().nabo_index_set_option(self._h, str(name).encode(), int(value)))
        return self

    def set_ref(self, Y=None, ref_mask=None, y_device_ptr=None):
        mk, mp = _mask(ref_mask, self.n_ref)
        if y_device_ptr is not None:
            _lib.check(_lib.lib().nabo_index_set_ref(self._h, int(y_device_ptr), 1, mp))
        else:
            Y = _f64(Y, "Y")
            if Y.shape != (self.n_ref, self.g):
                raise ValueError("ERROR: Y must be [n_ref, g]")
            _lib.check(_lib.lib().nabo_index_set_ref(self._h, Y.ctypes.data, 0, mp))
        return self

    def set_mask(self, ref_mask=None):
        """New ignore mask for the references already resident (no re-upload)."""
        mk, mp = _mask(ref_mask, self.n_ref)
        _lib.check(_lib.lib().nabo_index_set_mask(self._h, mp))
        return self

    def query(self, X, k, drop_first=False):
        X = _f64(X, "X")
        if X.shape[1] != self.g:
            raise ValueError("ERROR: X must have g components")
        m = X.shape[0]
        idx = np.empty((m, k), dtype=np.int64)
        dist = np.empty((m, k), dtype=np.float64)
        _lib.check(_lib.lib().nabo_index_query(self._h, X.ctypes.data, 0, m, int(k), int(bool(drop_first)),
                                               idx.ctypes.data, dist.ctypes.data, 0))
        return idx, dist

    def query_device(self, x_ptr, m, k, drop_first, out_idx_ptr, out_dist_ptr):
        """Device pointers in and out (X [m,g] f64; out_idx [m,k] i64; out_dist [m,k] f64)."""
        _lib.check(_lib.lib().nabo_index_query(self._h, int(x_ptr), 1, int(m), int(k), int(bool(drop_first)),
                                               int(out_idx_ptr), int(out_dist_ptr), 1))

    def query_device_async(self, x_ptr, m, k, drop_first, out_idx_ptr, out_dist_ptr):
        """nabo_index_query_async: returns at once; wait() returns when the results are in the output buffers.  One query in
        flight per index; every other call on it is refused until wait()."""
        _lib.check(_lib.lib().nabo_index_query_async(self._h, int(x_ptr), 1, int(m), int(k), int(bool(drop_first)),
                                                     int(out_idx_ptr), int(out_dist_ptr), 1))
        return self

    def wait(self):
        _lib.check(_lib.lib().nabo_index_query_wait(self._h))
        return self

    def query_candidates_device(self, x_ptr, m, n_cand, out_idx_ptr, out_dist_ptr, out_bound_ptr):
        """Shard mode: first n_cand order-row entries + the bound on everything else (device pointers)."""
        _lib.check(_lib.lib().nabo_index_query_candidates(self._h, int(x_ptr), 1, int(m), int(n_cand),
                                                          int(out_idx_ptr), int(out_dist_ptr), int(out_bound_ptr)))

    def last_stats(self):
        ms = (C.c_double * 5)()
        cn = (C.c_int64 * 4)()
        _lib.check(_lib.lib().nabo_index_last_stats(self._h, ms, cn))
        ps = (C.c_int64 * 3)()
        _lib.check(_lib.lib().nabo_index_last_passes(self._h, ps))
        return {"ms_pack": ms[0], "ms_topk": ms[1], "ms_refine": ms[2], "ms_fallback": ms[3], "ms_total": ms[4],
                "fallback_rows": int(cn[0]), "splits": int(cn[1]), "list_len": int(cn[2]), "workgroups": int(cn[3]),
                "seeded_pass_rows": int(ps[0]), "second_pass_rows": int(ps[1]), "wide_list_rows": int(ps[2])}

    PASS_NAMES = ("one_product", "seeded", "second_filter", "wide_lists", "exact", "canberra_filter")

    def last_row_pass(self, m):
        """uint8 [m]: which pass answered each row of the last query of m rows (include/nabo_knn.h: NABO_PASS_*)"""
        out = np.empty(int(m), dtype=np.uint8)
        _lib.check(_lib.lib().nabo_index_last_row_pass(self._h, out.ctypes.data, int(m)))
        return out

    def last_kernel(self):
        """name of the dominant kernel the last query ran (which filter the launch logic picked)"""
        buf = C.create_string_buffer(192)
        _lib.check(_lib.lib().nabo_index_last_kernel(self._h, buf, 192))
        return buf.value.decode("ascii", "replace")

    def close(self):
        if self._h is not None and self._h.value:
            _lib.lib().nabo_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    """hipMalloc'ed bytes through the C ABI (for hosts that have no other GPU binding)."""

    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = int(device), int(nbytes)
        p = C.c_void_p()
        _lib.check(_lib.lib().nabo_dev_malloc(self.device, C.byref(p), self.nbytes))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _lib.check(_lib.lib().nabo_memcpy_h2d(self.device, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _lib.check(_lib.lib().nabo_memcpy_d2h(self.device, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            _lib.lib().nabo_dev_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def merge_topk_device(parts_idx_ptr, parts_dist_ptr, n_parts, m, kp, k, drop_first, out_idx_ptr, out_dist_ptr,
                      device=0):
    _lib.check(_lib.lib().nabo_merge_topk(int(device), int(parts_idx_ptr), int(parts_dist_ptr), int(n_parts), int(m),
                                          int(kp), int(k), int(bool(drop_first)), int(out_idx_ptr),
                                          int(out_dist_ptr)))


def snn_counts(t_idx, r_idx, k, device=0):
    """Shared-neighbour counts of nabo/_mapping.py:190-193 computed on the GPU.
    t_idx [m,>=k], r_idx [n,>=k] host int arrays -> int32 [m,k]."""
    t = np.ascontiguousarray(np.asarray(t_idx)[:, :k], dtype=np.int64)
    r = np.ascontiguousarray(np.asarray(r_idx)[:, :k], dtype=np.int64)
    m, n = t.shape[0], r.shape[0]
    dt = DeviceBuffer(t.nbytes, device).upload(t)
    dr = DeviceBuffer(r.nbytes, device).upload(r)
    do = DeviceBuffer(m * k * 4, device)
    try:
        _lib.check(_lib.lib().nabo_snn_counts(int(device), dt.ptr, m, dr.ptr, n, int(k), do.ptr))
        return do.download((m, k), np.int32)
    finally:
        dt.free(); dr.free(); do.free()
