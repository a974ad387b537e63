"""The PRODUCT's sharded-query control flow, compiled, on a box without a GPU.

tests/test_dist_gloo.py runs a torch.distributed *restatement* of the protocol (tests/_dist_spec.py): it pins the
protocol, not the implementation.  Here nabo_amd/csrc/sharded.hip ITSELF is compiled with g++ against tests/host_shim
(-DNABO_SHARDED_HOST: "device" memory is malloc'ed, streams are synchronous, a kernel launch is a host loop) and driven
through its C entry points with the loopback transport, N ranks as host threads: communicator creation, the 2-D layout,
status agreements, the grouped exchange, merge_parts + certify_kernel, the second round (adopt_kernel), the ragged-slice
fill, the final gather, and the failure semantics (a rank failing alone, mismatched arguments, an allocation failure, a
peer that never arrives, nabo_comm_abort from another thread).  What a rank's nabo_index would compute on the GPU -- its
candidate lists with their bound, its certified local top-k -- is injected from the oracle.  First contact of this code
with RCCL is the driver's 8-GPU run; everything around the ncclSend/Recv calls has run here first.
"""
import ctypes as C
import os
import subprocess
import threading
import time

import numpy as np
import pytest

import oracle
from nabo_amd._sharded import shard_bounds, candidates_per_shard
from nabo_amd._synth import pca_like

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(REPO, "tests", "host_shim")
SO = os.path.join(SHIM, "build", "libnabo_sharded_host.so")
E_INVALID, E_HIP, E_NOMEM, E_COMM = -1, -3, -4, -6

CAND_CB = C.CFUNCTYPE(C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p)
QUERY_CB = C.CFUNCTYPE(C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p)


@pytest.fixture(scope="module")
def host():
    srcs = [os.path.join(REPO, "nabo_amd", "csrc", "sharded.hip"), os.path.join(SHIM, "host_index.cpp")]
    deps = srcs + [os.path.join(SHIM, "hip_shim.h"), os.path.join(REPO, "include", "nabo_knn.h")]
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-x", "c++", "-DNABO_SHARDED_HOST", "-I" + SHIM]
                              + srcs + ["-o", SO, "-lpthread", "-ldl"])
    L = C.CDLL(SO)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.nabo_last_error.restype = C.c_char_p
    L.nabo_comm_create_loopback.argtypes = [C.POINTER(vp), C.POINTER(i32), i32]
    L.nabo_comm_destroy.argtypes = [vp]
    L.nabo_comm_abort.argtypes = [vp]
    L.nabo_comm_set_timeout.argtypes = [vp, C.c_double]
    L.nabo_comm_set_ref_shards.argtypes = [vp, i32]
    L.nabo_comm_transport_ranks.argtypes = [vp]
    L.nabo_comm_barrier.argtypes = [vp]
    L.nabo_sharded_query.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, i32]
    L.nabo_sharded_last_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64)]
    L.nabo_host_index_create.restype = vp
    L.nabo_host_index_create.argtypes = [C.c_int, C.c_int, i64, C.c_int, C.c_int]
    L.nabo_host_index_destroy.argtypes = [vp]
    L.nabo_host_index_shard_mode.argtypes = [vp]
    L.nabo_host_set_callbacks.argtypes = [CAND_CB, QUERY_CB]
    L.nabo_host_fail_nth_malloc.argtypes = [C.c_int]
    return L


class Group:
    """N loopback ranks over one reference set Y (piece r % R of R on rank r), candidate lists injected from the oracle."""

    def __init__(self, L, N, R, Y, metric=0, timeout=20.0):
        self.L, self.N, self.R, self.Y, self.metric = L, N, R, np.ascontiguousarray(Y), metric
        self.g = Y.shape[1]
        hs = (C.c_void_p * N)()
        dv = (C.c_int32 * N)(*([0] * N))
        assert L.nabo_comm_create_loopback(hs, dv, N) == 0
        self.comms = [C.c_void_p(hs[i]) for i in range(N)]
        self.bounds = [shard_bounds(Y.shape[0], R, r % R) for r in range(N)]
        self.indices = [C.c_void_p(L.nabo_host_index_create(r, 0, self.bounds[r][1] - self.bounds[r][0], self.g, metric)) for r in range(N)]
        for c in self.comms:
            assert L.nabo_comm_set_ref_shards(c, R) == 0
            assert L.nabo_comm_set_timeout(c, timeout) == 0
        self.refuse_rows = set()          # target rows (global) whose bound is spoiled: the owner must refuse them
        self.fail_cand = self.fail_query = None
        self.calls = {"cand": 0, "query": 0}
        self.row_base = {}                # rank -> first global row of the slice its candidate query sees (set per call)
        self._cand = CAND_CB(self._cand_cb)
        self._query = QUERY_CB(self._query_cb)
        L.nabo_host_set_callbacks(self._cand, self._query)

    def _rows(self, ptr, m):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(m, self.g))

    def _cand_cb(self, rid, xp, m, n_cand, oi, od, ob):
        try:
            self.calls["cand"] += 1
            if self.fail_cand == rid:
                return E_HIP
            lo, hi = self.bounds[rid]
            X = np.ascontiguousarray(self._rows(xp, m))
            kq = min(n_cand + 1, hi - lo)
            qi, qd = oracle.knn(X, self.Y[lo:hi], kq, self.metric, nthreads=2)
            I = np.full((m, n_cand), -1, dtype=np.int64)
            D = np.full((m, n_cand), np.inf)
            kc = min(n_cand, kq)
            I[:, :kc], D[:, :kc] = qi[:, :kc] + lo, qd[:, :kc]
            B = np.full(m, np.inf) if kq <= n_cand else qd[:, n_cand] ** 2 * (1 - 1e-12)
            base = self.row_base.get(rid, 0)
            for r in self.refuse_rows:
                if base <= r < base + m:
                    B[r - base] = 0.0
            C.memmove(oi, I.ctypes.data, I.nbytes)
            C.memmove(od, D.ctypes.data, D.nbytes)
            C.memmove(ob, B.ctypes.data, B.nbytes)
            return 0
        except Exception:      # noqa: BLE001  (never let an exception cross the C boundary)
            return E_HIP

    def _query_cb(self, rid, xp, m, k, oi, od):
        try:
            self.calls["query"] += 1
            if self.fail_query == rid:
                return E_HIP
            lo, hi = self.bounds[rid]
            X = np.ascontiguousarray(self._rows(xp, m))
            qi, qd = oracle.knn(X, self.Y[lo:hi], k, self.metric, nthreads=2)
            qi = qi + lo
            C.memmove(oi, qi.ctypes.data, qi.nbytes)
            C.memmove(od, qd.ctypes.data, qd.nbytes)
            return 0
        except Exception:      # noqa: BLE001
            return E_HIP

    def query(self, X, k, drop=False, protocol=0, ks=None, skip=(), join_timeout=60.0):
        """every rank (except `skip`) calls nabo_sharded_query on its own thread; returns per-rank (rc, message, idx, dist)"""
        X = np.ascontiguousarray(X, dtype=np.float64)
        m = X.shape[0]
        mr = (m + self.N - 1) // self.N
        for r in range(self.N):
            self.row_base[r] = (r // self.R) * self.R * mr
        out = [None] * self.N

        def work(r):
            kr = ks[r] if ks else k
            gi = np.full((m, kr), -7, dtype=np.int64)
            gd = np.full((m, kr), -7.0)
            rc = self.L.nabo_sharded_query(self.comms[r], self.indices[r], X.ctypes.data, m, kr, int(drop), gi.ctypes.data,
                                           gd.ctypes.data, protocol)
            out[r] = (rc, self.L.nabo_last_error().decode() if rc else "", gi, gd)

        ths = [threading.Thread(target=work, args=(r,)) for r in range(self.N) if r not in skip]
        for t in ths:
            t.start()
        for t in ths:
            t.join(join_timeout)
        assert not any(t.is_alive() for t in ths), "a rank thread hangs"
        return out

    def stats(self, r=0):
        ms = (C.c_double * 8)()
        cn = (C.c_int64 * 4)()
        assert self.L.nabo_sharded_last_stats(self.comms[r], ms, cn) == 0
        return {"uncertified": int(cn[0]), "candidates": int(cn[1]), "protocol": int(cn[3])}

    def close(self):
        for ix in self.indices:
            self.L.nabo_host_index_destroy(ix)
        for c in self.comms:
            self.L.nabo_comm_destroy(c)


def _data(n=700, m=131, g=12):
    return pca_like(n, g, seed=71), pca_like(m, g, seed=72)


@pytest.mark.parametrize("N,R,k,drop,protocol", [(2, 2, 5, 0, 0), (3, 3, 5, 1, 0), (3, 3, 7, 0, 2), (4, 2, 6, 0, 1), (3, 1, 5, 0, 0), (1, 1, 4, 1, 0)])
def test_compiled_protocol_equals_the_oracle(host, N, R, k, drop, protocol):
    """world 1 .. 4, the 1-D form, the 2 x 2 layout, pure target slicing, both protocols, a ragged last slice (131 rows):
    every rank's copy of every row equals the oracle's order rows (nabo/_mapping.py:139-145 + the positional drop)."""
    Y, X = _data()
    if drop:
        X = Y[:131]
    grp = Group(host, N, R, Y)
    try:
        assert all(host.nabo_comm_transport_ranks(c) == N for c in grp.comms)
        res = grp.query(X, k, drop=bool(drop), protocol=protocol)
        oi, od = oracle.knn(X, Y, k, 0, drop_first=bool(drop), nthreads=4)
        for rc, msg, gi, gd in res:
            assert rc == 0, msg
            assert np.array_equal(gi, oi) and np.array_equal(gd, od)
        st = grp.stats()
        assert st["protocol"] == (2 if protocol == 2 or N == 1 else 1)
        if st["protocol"] == 1 and R > 1:
            assert st["candidates"] == candidates_per_shard(k + drop, R, X.shape[0]) and grp.calls["cand"] == N
        assert all(host.nabo_host_index_shard_mode(ix) == 0 for ix in grp.indices)          # switched back after the call
    finally:
        grp.close()


def test_refused_rows_take_the_second_round(host):
    """Owners must refuse rows whose k'-th merged distance does not lie below every shard's bound (certify_kernel), agree on
    the count, re-solve exactly those rows on every piece and adopt the merged answer (adopt_kernel): rows spread over all
    three owners, the SAME result on every rank, the count reported identically."""
    Y, X = _data()
    grp = Group(host, 3, 3, Y)
    try:
        grp.refuse_rows = {0, 1, 43, 44, 45, 90, 130}
        res = grp.query(X, 5)
        oi, od = oracle.knn(X, Y, 5, 0, nthreads=4)
        for rc, msg, gi, gd in res:
            assert rc == 0, msg
            assert np.array_equal(gi, oi) and np.array_equal(gd, od)
        assert [grp.stats(r)["uncertified"] for r in range(3)] == [7, 7, 7]
        assert grp.calls["query"] == 3                        # one exact re-solve per rank
    finally:
        grp.close()


def test_a_rank_that_fails_alone_fails_everywhere_and_the_communicators_stay_usable(host):
    """Status agreement: one rank's candidate query fails / one rank's second-round query fails / one rank cannot allocate /
    one rank is handed another k -> EVERY rank returns an error within seconds (the failing rank its own status, the others
    NABO_E_COMM or the argument mismatch), nobody hangs, and the same communicators answer the next call correctly."""
    Y, X = _data()
    grp = Group(host, 3, 3, Y)
    oi, od = oracle.knn(X, Y, 5, 0, nthreads=4)

    def good():
        for rc, msg, gi, gd in grp.query(X, 5):
            assert rc == 0, msg
            assert np.array_equal(gi, oi) and np.array_equal(gd, od)

    try:
        good()
        grp.fail_cand = 1
        res = grp.query(X, 5)
        assert [r[0] for r in res] == [E_COMM, E_HIP, E_COMM] and "peer failed" in res[0][1]
        grp.fail_cand = None
        good()
        grp.refuse_rows, grp.fail_query = {5, 60}, 2
        res = grp.query(X, 5)
        assert [r[0] for r in res] == [E_COMM, E_COMM, E_HIP]
        grp.refuse_rows, grp.fail_query = set(), None
        good()
        res = grp.query(X, 5, ks=[5, 6, 5])
        assert all(r[0] == E_INVALID and "different arguments" in r[1] for r in res)
        good()
        host.nabo_host_fail_nth_malloc(1)                     # the next "device" allocation of whichever rank fails
        grp2 = Group(host, 3, 3, Y)                           # (fresh communicators: their buffers are still to be reserved)
        res = grp2.query(X, 5)
        assert sorted(r[0] for r in res) == [E_COMM, E_COMM, E_NOMEM]
        host.nabo_host_fail_nth_malloc(0)
        for rc, msg, gi, gd in grp2.query(X, 5):
            assert rc == 0, msg
            assert np.array_equal(gi, oi) and np.array_equal(gd, od)
        grp2.close()
        host.nabo_host_set_callbacks(grp._cand, grp._query)
        good()
    finally:
        host.nabo_host_fail_nth_malloc(0)
        grp.close()


def test_a_missing_peer_times_out_and_abort_releases_the_waiting_ranks(host):
    """A rank that never enters the collective: the others give up after the deadline with NABO_E_COMM (no hang) and every
    later call on those communicators fails at once; nabo_comm_abort from another thread releases waiting ranks early."""
    Y, X = _data()
    grp = Group(host, 3, 3, Y, timeout=1.0)
    try:
        t0 = time.time()
        res = grp.query(X, 5, skip=(2,))
        assert time.time() - t0 < 15
        assert [r[0] for r in res[:2]] == [E_COMM, E_COMM] and res[2] is None
        t0 = time.time()
        res = grp.query(X, 5)
        assert all(r[0] == E_COMM for r in res) and time.time() - t0 < 5
    finally:
        grp.close()
    grp = Group(host, 3, 3, Y, timeout=120.0)
    try:
        threading.Timer(0.5, lambda: host.nabo_comm_abort(grp.comms[0])).start()
        t0 = time.time()
        res = grp.query(X, 5, skip=(1,))
        assert time.time() - t0 < 20 and res[0][0] == E_COMM and res[2][0] == E_COMM
        assert host.nabo_comm_abort(grp.comms[0]) == 0 and host.nabo_comm_abort(grp.comms[2]) == 0      # idempotent, any thread
        assert host.nabo_comm_transport_ranks(grp.comms[0]) == E_COMM
    finally:
        grp.close()
