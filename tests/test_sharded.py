"""Reference rows sharded over GPUs behind the C ABI (nabo_comm_*, nabo_sharded_query; no torch).

CPU side: the list-length rule (C vs Python), the unique-id hand-off between ranks, and that the product never imports
torch.  GPU side (one MI355X): the whole protocol with N = 2, 3, 8 shard-ranks on one GPU through the loopback
transport (same call sequence, buffers and kernels as the RCCL transport), the RCCL transport itself with one rank
(dlopen, unique id, communicator, grouped send/recv, all-gather, all-reduce), bench.py's N>1 code path, and the
launcher hand-off under torch.distributed.run.  RCCL with N > 1 ranks needs N GPUs: the driver's scaling run."""
import itertools
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

import oracle
from nabo_amd import _lib, _sharded
from nabo_amd._synth import pca_like

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_list_length_rule_c_equals_python():
    L = _lib.lib()
    for kk, w, m in itertools.product([1, 2, 5, 11, 15, 16, 24, 31, 51, 56], [1, 2, 3, 4, 8, 16], [1, 1000, 1000000, 5000000]):
        assert _sharded.candidates_per_shard(kk, w, m) == L.nabo_candidates_per_shard(kk, w, m), (kk, w, m)
    # the values DESIGN.md section 5 quotes (k' = 15, 1M rows)
    assert [_sharded.candidates_per_shard(15, n, 1000000) for n in (8, 4, 2)] == [12, 15, 16]
    assert _sharded.candidates_per_shard(50, 8, 5000000) == 24


def test_unique_id_reaches_every_rank_through_a_file(tmp_path):
    world = 5
    path = str(tmp_path / "id")
    want = bytes(range(128))
    got = [None] * world

    def run(r):
        got[r] = _sharded.exchange_unique_id(r, world, lambda: want, path=path, timeout=20)[0]

    th = [threading.Thread(target=run, args=(r,)) for r in range(world - 1, -1, -1)]      # readers first
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got == [want] * world
    with pytest.raises(_lib.NaboError):
        _sharded.exchange_unique_id(1, 2, None, path=str(tmp_path / "never"), timeout=0.05)


def test_unique_id_left_by_an_older_job_is_not_accepted(tmp_path):
    """A crashed job leaves its id file behind; a restart under the same launcher pid and port must not hand that id to
    the non-zero ranks while rank 0 is still creating the new one: readers only take a file written
    after the launcher started, rank 0 removes the leftover before it publishes, files are private to the user."""
    path = str(tmp_path / "id")
    stale, fresh = bytes([7]) * 128, bytes(range(128))
    with open(path, "wb") as f:
        f.write(stale)
    old = time.time() - 3600.0
    os.utime(path, (old, old))
    with pytest.raises(_lib.NaboError):                      # a reader alone never accepts the leftover
        _sharded.exchange_unique_id(1, 2, None, path=path, timeout=0.2, not_before=time.time() - 60.0)
    got = {}

    def reader():
        got["blob"] = _sharded.exchange_unique_id(1, 2, None, path=path, timeout=20, not_before=time.time() - 60.0)[0]

    t = threading.Thread(target=reader)
    t.start()
    time.sleep(0.1)
    assert _sharded.exchange_unique_id(0, 2, lambda: fresh, path=path)[0] == fresh
    t.join()
    assert got["blob"] == fresh
    assert (os.stat(path).st_mode & 0o077) == 0
    # the default name carries the launcher's pid, the port, torchrun's run id and restart count, in a 0700 directory
    os.environ["NABO_ID_DIR"] = str(tmp_path / "ids")
    try:
        a = _sharded.id_file_path()
        os.environ["TORCHELASTIC_RESTART_COUNT"] = "1"
        b = _sharded.id_file_path()
    finally:
        os.environ.pop("NABO_ID_DIR", None)
        os.environ.pop("TORCHELASTIC_RESTART_COUNT", None)
    assert a != b and str(os.getppid()) in a
    assert (os.stat(str(tmp_path / "ids")).st_mode & 0o077) == 0
    assert 0 < _sharded._launcher_start_time() <= time.time()


def test_product_and_bench_never_import_torch():
    """torch is the LAUNCHER (python -m torch.distributed.run) and the gloo test harness (tests/_dist_spec.py), not the
    product: nothing under nabo_amd/, nor bench.py, nor the driver entry points may import it."""
    files = [os.path.join(REPO, "bench.py"), os.path.join(REPO, "__graft_entry__.py")]
    for root, _, fs in os.walk(os.path.join(REPO, "nabo_amd")):
        files += [os.path.join(root, f) for f in fs if f.endswith(".py")]
    for f in files:
        for ln in open(f).read().splitlines():
            code = ln.split("#")[0]
            assert "import torch" not in code and "from torch" not in code, (f, ln)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


CASES = [
    # N, m, n, g, k, drop, metric, protocol, sorted refs
    (2, 1001, 6000, 20, 11, False, 0, "auto", False),
    (3, 500, 4001, 30, 15, True, 0, "auto", False),           # ragged rows and shards, positional drop after the merge
    (8, 4000, 40000, 50, 15, False, 0, "auto", False),
    (8, 3000, 40000, 50, 15, False, 0, "global", True),       # neighbours in ONE shard: the second round must repair
    (4, 700, 5000, 25, 23, True, 2, "auto", False),           # cosine (extension)
    (2, 300, 3000, 16, 9, False, 1, "auto", False),           # modified Canberra: local certification only
    (3, 257, 2000, 12, 40, False, 0, "local", False),         # k' beyond the candidate lists of a shard
    (2, 64, 900, 8, 5, True, 0, "global", False),
]


@pytest.mark.gpu
@pytest.mark.parametrize("N,m,n,g,k,drop,metric,protocol,sorted_refs", CASES)
def test_sharded_query_on_one_gpu_equals_unsharded_and_oracle(gpu_lib, N, m, n, g, k, drop, metric, protocol, sorted_refs):
    Y = pca_like(n, g, seed=400 + n)
    if sorted_refs:
        Y = np.ascontiguousarray(Y[np.argsort(Y[:, 0], kind="stable")])
    X = Y[:m].copy() if drop else pca_like(m, g, seed=500 + m)
    grp = _sharded.LoopbackGroup(N, 0, n, g, metric, Y, protocol=protocol).set_ref()
    gi, gd = grp.query(X, k, drop_first=drop)
    st = [grp.last_stats(r) for r in range(N)]
    grp.close()
    oi, od = oracle.knn(X, Y, k, metric, 0.25, drop_first=drop, nthreads=8)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    want = "local" if (metric == 1 or protocol == "local") else "global"
    assert all(s["protocol"] == want for s in st)
    if sorted_refs:
        assert st[0]["uncertified"] > 0                          # the second round really ran
    assert len({s["uncertified"] for s in st}) == 1              # every rank saw the same second round


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_shards_with_fewer_unmasked_references_than_k_do_not_leak_ignored_ones(gpu_lib, metric):
    """Found by tools/stress_sweep2.py: a 40-reference shard with 60 % of its references ignored has fewer than k'
    unmasked ones; its local query used to continue the row with the IGNORED references (the one-device rule for short
    rows, nabo/_mapping.py:135-146) and those entered the global merge as neighbours."""
    rng = np.random.default_rng(74)
    n, m, g, k, N = 200, 301, 19, 14, 5
    Y = pca_like(n, g, seed=741)
    Y[rng.integers(0, n, n // 3)] = Y[int(rng.integers(0, n))]          # a block of identical references: ties
    X = pca_like(m, g, seed=742)
    mask = (rng.random(n) < 0.6).astype(np.uint8)
    mask[:40] = 1
    mask[3:11] = 0                                                     # shard 0 keeps 8 of its 40 references
    assert int((mask == 0).sum()) >= k and min(int((mask[lo:hi] == 0).sum()) for lo, hi in
                                               (_sharded.shard_bounds(n, N, r) for r in range(N))) < k
    grp = _sharded.LoopbackGroup(N, 0, n, g, metric, Y, ref_mask=mask).set_ref()
    gi, gd = grp.query(X, k)
    grp.close()
    oi, od = oracle.knn(X, Y, k, metric, 0.25, ref_mask=mask, nthreads=8)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    assert not mask[gi].any()


@pytest.mark.gpu
@pytest.mark.parametrize("metric,protocol", [(0, "auto"), (0, "local"), (1, "auto"), (2, "global")])
def test_a_shard_with_fewer_references_than_k_takes_part_with_absent_entries(gpu_lib, metric, protocol):
    """58 references over 5 shards = 11, 12, 11, 12, 12 rows, k' = 12: nabo_index_query refuses k' > n_ref, so shards 0
    and 2 used to fail alone while their peers waited in the exchange.  They now answer with what
    they have; the forced second round (protocol "global" on references sorted along a component) takes the same path."""
    n, m, g, k, N = 58, 300, 9, 11, 5
    Y = pca_like(n, g, seed=58)
    Y = np.ascontiguousarray(Y[np.argsort(Y[:, 0], kind="stable")])
    X = Y[:m % n + 40].copy()
    X = np.concatenate([X, pca_like(m - X.shape[0], g, seed=59)])
    sizes = [hi - lo for lo, hi in (_sharded.shard_bounds(n, N, r) for r in range(N))]
    assert min(sizes) < k + 1 <= max(sizes)
    grp = _sharded.LoopbackGroup(N, 0, n, g, metric, Y, protocol=protocol).set_ref()
    gi, gd = grp.query(X, k, drop_first=True)
    gi2, gd2 = grp.query(X[:7], k + 20)                       # k' beyond EVERY shard (but not beyond the reference set)
    grp.close()
    oi, od = oracle.knn(X, Y, k, metric, 0.25, drop_first=True, nthreads=8)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    oi, od = oracle.knn(X[:7], Y, k + 20, metric, 0.25, nthreads=8)
    assert np.array_equal(gi2, oi) and np.array_equal(gd2, od)


@pytest.mark.gpu
@pytest.mark.parametrize("fault", ["k", "m", "null", "protocol"])
def test_a_rank_handed_a_bad_argument_fails_every_rank_and_the_group_stays_usable(gpu_lib, fault):
    """Failure semantics (include/nabo_knn.h): what a rank gets wrong ALONE is agreed on before anything is
    exchanged -- every rank returns an error within seconds, nobody waits in a collective for the rank that gave up,
    and the communicators remain usable for the next (correct) call."""
    from nabo_amd import _knn
    N, m, n, g, k = 4, 900, 6000, 16, 9
    Y, X = pca_like(n, g, seed=81), pca_like(m, g, seed=82)
    grp = _sharded.LoopbackGroup(N, 0, n, g, 0, Y, timeout=30.0).set_ref()
    dx, di, dd = _knn.DeviceBuffer(X.nbytes).upload(X), _knn.DeviceBuffer(m * k * 8), _knn.DeviceBuffer(m * k * 8)
    errs = [None] * N

    def call(r):
        kw = {"k": k, "m": m, "x": dx.ptr, "protocol": grp.shards[r].protocol}
        if r == 2:
            kw.update({"k": {"k": 0}, "m": {"m": m - 1}, "null": {"x": 0}, "protocol": {"protocol": 7}}[fault])
        rc = _lib.lib().nabo_sharded_query(grp.comms[r]._h, grp.indices[r]._h, kw["x"], kw["m"], kw["k"], 0, di.ptr, dd.ptr,
                                           kw["protocol"])
        errs[r] = (rc, _lib.lib().nabo_last_error().decode())

    t0 = time.time()
    grp._each(call)
    assert time.time() - t0 < 20.0
    assert all(rc != 0 for rc, _ in errs), errs
    if fault == "m":                                          # nobody's argument is wrong by itself: the mismatch is the error
        assert all(rc == _lib.E_INVALID and "different arguments" in msg for rc, msg in errs), errs
    else:
        assert errs[2][0] == _lib.E_INVALID and all(rc == _lib.E_COMM and "peer failed" in msg for rc, msg in errs[:2] + errs[3:]), errs
    gi, gd = grp.query(X, k)                                  # the same communicators, next call
    grp.close()
    oi, od = oracle.knn(X, Y, k, 0, nthreads=8)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)


@pytest.mark.gpu
def test_a_missing_rank_times_out_into_an_error_on_every_rank(gpu_lib):
    """One of four ranks never enters the collective: the other three give up after the communicator's timeout and
    return NABO_E_COMM, the group is dead afterwards (every later call fails at once) -- no hang."""
    from nabo_amd import _knn
    N, m, n, g, k = 4, 500, 4000, 12, 7
    Y, X = pca_like(n, g, seed=83), pca_like(m, g, seed=84)
    grp = _sharded.LoopbackGroup(N, 0, n, g, 0, Y, timeout=2.0).set_ref()
    dx, di, dd = _knn.DeviceBuffer(X.nbytes).upload(X), _knn.DeviceBuffer(m * k * 8), _knn.DeviceBuffer(m * k * 8)
    errs = [None] * N

    def call(r):
        if r == 1:
            return                                            # this rank "died"
        rc = _lib.lib().nabo_sharded_query(grp.comms[r]._h, grp.indices[r]._h, dx.ptr, m, k, 0, di.ptr, dd.ptr, 0)
        errs[r] = (rc, _lib.lib().nabo_last_error().decode())

    t0 = time.time()
    grp._each(call)
    assert time.time() - t0 < 15.0
    assert all(errs[r][0] == _lib.E_COMM for r in (0, 2, 3)), errs
    with pytest.raises(_lib.NaboError):
        grp.query(X, k)
    grp.close()


@pytest.mark.gpu
def test_abort_from_another_thread_releases_the_waiting_ranks(gpu_lib):
    """nabo_comm_abort (what ShardedGroup._each calls when a rank thread does not come back): ranks blocked in a
    collective return NABO_E_COMM at once."""
    from nabo_amd import _knn
    N, m, n, g, k = 3, 400, 3000, 10, 5
    Y, X = pca_like(n, g, seed=85), pca_like(m, g, seed=86)
    grp = _sharded.LoopbackGroup(N, 0, n, g, 0, Y, timeout=120.0).set_ref()
    dx, di, dd = _knn.DeviceBuffer(X.nbytes).upload(X), _knn.DeviceBuffer(m * k * 8), _knn.DeviceBuffer(m * k * 8)
    errs = [None] * N

    def call(r):
        if r == 0:
            time.sleep(1.0)
            grp.comms[0].abort()
            return
        rc = _lib.lib().nabo_sharded_query(grp.comms[r]._h, grp.indices[r]._h, dx.ptr, m, k, 0, di.ptr, dd.ptr, 0)
        errs[r] = (rc, _lib.lib().nabo_last_error().decode())

    t0 = time.time()
    grp._each(call)
    assert time.time() - t0 < 10.0
    assert all(errs[r][0] == _lib.E_COMM for r in (1, 2)), errs
    grp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,R,m,n,g,k,drop,metric,sorted_refs", [
    (8, 2, 3001, 9000, 24, 15, False, 0, False), (4, 2, 700, 5000, 50, 11, True, 0, False),
    (8, 4, 1234, 12000, 30, 10, False, 2, False), (6, 3, 999, 6000, 16, 15, True, 0, True),
    (8, 1, 515, 4000, 20, 7, False, 0, False), (4, 2, 5, 3000, 12, 6, False, 0, False),
    (4, 1, 777, 5000, 30, 11, True, 1, False), (3, 1, 100, 2500, 64, 30, False, 2, False)])      # target slices: any metric
def test_two_dimensional_layout_equals_oracle(gpu_lib, N, R, m, n, g, k, drop, metric, sorted_refs):
    """nabo_comm_set_ref_shards: R reference pieces x N / R target slices (exchange, merge and certificate inside each
    group of R ranks, the gather over all N) -- same answer as one device, on every rank, ragged slices and the second
    round included."""
    from nabo_amd import _knn
    Y = pca_like(n, g, seed=900 + n)
    if sorted_refs:
        Y = np.ascontiguousarray(Y[np.argsort(Y[:, 0], kind="stable")])
    X = Y[:m].copy() if drop else pca_like(m, g, seed=950 + m)
    grp = _sharded.LoopbackGroup(N, 0, n, g, metric, Y, ref_shards=R).set_ref()
    dx = _knn.DeviceBuffer(X.nbytes).upload(X)
    outs = [(_knn.DeviceBuffer(m * k * 8), _knn.DeviceBuffer(m * k * 8)) for _ in range(N)]
    grp.query_device(dx.ptr, m, k, drop, [a.ptr for a, _ in outs], [b.ptr for _, b in outs])
    st = [grp.last_stats(r) for r in range(N)]
    res = [(a.download((m, k), np.int64), b.download((m, k), np.float64)) for a, b in outs]
    grp.close()
    oi, od = oracle.knn(X, Y, k, metric, 0.25, drop_first=drop, nthreads=8)
    for gi, gd in res:
        assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    assert all(s["protocol"] == "global" for s in st)
    if sorted_refs:
        assert st[0]["uncertified"] > 0


@pytest.mark.gpu
def test_two_dimensional_layout_rejects_what_it_cannot_do(gpu_lib):
    Y = pca_like(2000, 10, seed=5)
    with pytest.raises(ValueError):
        _sharded.LoopbackGroup(4, 0, 2000, 10, 0, Y, ref_shards=3)
    grp = _sharded.LoopbackGroup(4, 0, 2000, 10, 1, Y, ref_shards=2).set_ref()       # modified Canberra: local protocol only
    with pytest.raises(Exception):
        grp.query(pca_like(10, 10, seed=6), 3)
    grp.close()


@pytest.mark.gpu
def test_every_rank_ends_with_the_same_full_result(gpu_lib):
    from nabo_amd import _knn
    N, m, n, g, k = 4, 1234, 9000, 24, 10
    Y, X = pca_like(n, g, seed=61), pca_like(m, g, seed=62)
    grp = _sharded.LoopbackGroup(N, 0, n, g, 0, Y).set_ref()
    dx = _knn.DeviceBuffer(X.nbytes).upload(X)
    outs = [(_knn.DeviceBuffer(m * k * 8), _knn.DeviceBuffer(m * k * 8)) for _ in range(N)]
    grp.query_device(dx.ptr, m, k, False, [a.ptr for a, _ in outs], [b.ptr for _, b in outs])
    res = [(a.download((m, k), np.int64), b.download((m, k), np.float64)) for a, b in outs]
    grp.close()
    oi, od = oracle.knn(X, Y, k, 0, nthreads=8)
    for ri, rd in res:
        assert np.array_equal(ri, oi) and np.array_equal(rd, od)


@pytest.mark.gpu
@pytest.mark.parametrize("protocol", ["global", "local"])
def test_rccl_transport_with_one_rank(gpu_lib, protocol):
    """librccl.so through the C ABI: unique id, ncclCommInitRank, the grouped send/recv exchange (to self), all-reduce,
    all-gather -- every RCCL call of the protocol, with the only world size one GPU allows."""
    import ctypes as C
    from nabo_amd import _knn
    L = _lib.lib()
    buf = C.create_string_buffer(128)
    _lib.check(L.nabo_comm_unique_id(buf))
    h = C.c_void_p()
    _lib.check(L.nabo_comm_create(C.byref(h), 0, 0, 1, buf))
    comm = _sharded.Comm(h, 0, 0, 1)
    assert comm.allreduce_max(3.25) == 3.25
    comm.barrier()
    m, n, g, k = 999, 8000, 20, 12
    Y, X = pca_like(n, g, seed=71), pca_like(m, g, seed=72)
    ix = gpu_lib.KnnIndex(n, g, metric=0).set_ref(Y)
    sk = _sharded.ShardedIndex(comm, ix, protocol)
    dx, di, dd = _knn.DeviceBuffer(X.nbytes).upload(X), _knn.DeviceBuffer(m * k * 8), _knn.DeviceBuffer(m * k * 8)
    sk.query_device(dx.ptr, m, k, True, di.ptr, dd.ptr)
    st = sk.last_stats()
    gi, gd = di.download((m, k), np.int64), dd.download((m, k), np.float64)
    ix.close()
    comm.close()
    oi, od = oracle.knn(X, Y, k, 0, drop_first=True, nthreads=8)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    assert st["protocol"] == protocol


@pytest.mark.gpu
def test_bench_sharded_code_path_with_loopback_ranks():
    """bench.py's N>1 step (per-shard index with a global index base, nabo_sharded_query, stats in the line) with 4
    shard-ranks on ONE GPU; the result is compared with an unsharded index inside bench.py (NABO_BENCH_CHECK)."""
    env = dict(os.environ, NABO_BENCH_LOOPBACK="4", NABO_BENCH_CHECK="1")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "1", "--targets", "20001",
           "--refs", "50000", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "sharded == unsharded: True" in r.stdout and '"candidates_per_shard"' in r.stdout, r.stdout[-3000:]


@pytest.mark.gpu
def test_bench_asked_for_more_gpus_than_visible_exits_loudly():
    """`python bench.py --gpus N` without a launcher drives N devices itself (nabo_comm_create_all, one thread per rank);
    on a box with fewer GPUs it must refuse -- a line saying n_gpus: 1 for a --gpus 8 request is worse than no line."""
    import nabo_amd
    have = nabo_amd.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NABO_BENCH_LOOPBACK")}
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(have + 1), "--steps", "1", "--warmup", "0", "--targets", "2000",
           "--refs", "5000", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600, cwd=REPO)
    assert r.returncode == 2, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "n_gpus" not in r.stdout and "refusing" in r.stderr


@pytest.mark.gpu
def test_bench_prints_the_baseline_layout_as_headline_and_the_2d_layout_beside_it():
    """N>1: the headline is BASELINE configs[3]'s layout (references sharded N ways); the 2 x N/2 layout runs in the same
    invocation as `alt_layout` and must give the same bits.  Eight loopback ranks on the one GPU."""
    import json
    env = dict(os.environ, NABO_BENCH_LOOPBACK="8", NABO_BENCH_CHECK="1")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "1", "--targets", "30001",
           "--refs", "60000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stdout[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "refs sharded 8-way" in line["config"]["workload"] and "slices" not in line["config"]["workload"]
    assert line["sharded"]["layout"] == {"ref_shards": 8, "target_slices": 1} and len(line["sharded"]["per_rank_ms"]) == 8
    assert line["alt_layout"]["layout"] == {"ref_shards": 2, "target_slices": 4}
    assert line["alt_layout"]["same_bits_as_headline_layout"] is True and line["sampled_rows_equal_oracle"] is True
    # pure target slicing (every rank holds all the references, certifies its own slice, no exchange) beside both
    ts = line["alt_layout_target_slices"]
    assert ts["layout"] == {"ref_shards": 1, "target_slices": 8} and ts["same_bits_as_headline_layout"] is True
    assert ts["second_round_rows"] == 0
    assert all("ms_exchange" in e["sharded"] and "ms_topk" in e["index"] for e in line["sharded"]["per_rank_ms"])


@pytest.mark.gpu
def test_bench_plain_launch_drives_its_ranks_from_one_process():
    """`python bench.py --gpus N` with no launcher environment: nabo_comm_create_all (ncclCommInitAll), one persistent host
    thread per rank, the shards resident in HBM -- rehearsed with the one rank a single GPU allows."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NABO_BENCH_LOOPBACK")}
    env.update(NABO_BENCH_FORCE_THREADS="1", NABO_BENCH_CHECK="1")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--targets", "20001",
           "--refs", "50000", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stdout[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "one RCCL communicator per GPU" in line["config"]["launch"] and line["sharded"]["rccl_world"] == 1
    assert "sharded == unsharded: True" in r.stdout and line["sampled_rows_equal_oracle"] is True


@pytest.mark.gpu
def test_bench_under_the_launcher_creates_its_communicator_without_torch():
    """The driver's launch line (python -m torch.distributed.run ... bench.py --gpus N) with the one rank a single
    GPU allows: RANK / WORLD_SIZE / LOCAL_RANK from the launcher, the unique id through the file, ncclCommInitRank,
    nabo_sharded_query under global certification, barrier and max-over-ranks through nabo_comm_*."""
    env = dict(os.environ, NABO_BENCH_FORCE_COMM="1", NABO_BENCH_CHECK="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--targets", "20001", "--refs", "50000", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "sharded == unsharded: True" in r.stdout and '"sharded"' in r.stdout, r.stdout[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("N,m,n,g,k,drop,metric", [(3, 700, 9000, 30, 11, 1, 0), (4, 1200, 20000, 50, 15, 0, 0), (2, 300, 5000, 20, 11, 0, 1),
                                                   (8, 2000, 40000, 100, 50, 0, 2), (1, 500, 3000, 30, 11, 1, 0)])
def test_knn_devices_one_call_from_host_arrays_equals_one_device(gpu_lib, N, m, n, g, k, drop, metric):
    """nabo_knn_devices (multi.hip): host arrays in and out, the references sharded over `devices` inside the library -- one host
    thread, one index and one communicator per device, met in nabo_sharded_query.  Loopback transport (the devices repeat):
    N shards = one device bit for bit, and the oracle; a masked reference set; the error of a rank that fails alone."""
    Y = pca_like(n, g, seed=91)
    X = pca_like(m, g, seed=92) if not drop else Y[:m]
    mask = np.zeros(n, dtype=np.uint8)
    mask[::7] = 1
    for rm in (None, mask):
        gi, gd = gpu_lib.knn_devices(X, Y, k, [0] * N, metric=metric, ref_mask=rm, drop_first=bool(drop), transport="loopback")
        ri, rd = gpu_lib.knn(X, Y, k, metric=metric, ref_mask=rm, drop_first=bool(drop))
        assert np.array_equal(gi, ri) and np.array_equal(gd, rd)
    oi, od = oracle.knn(X, Y, k, metric, ref_mask=mask, drop_first=bool(drop), nthreads=8)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    if N > 1:
        with pytest.raises(ValueError):                       # k beyond a shard's ... beyond the whole set: every rank refuses
            gpu_lib.knn_devices(X, Y, 0, [0] * N, metric=metric, transport="loopback")
        with pytest.raises(Exception):                        # a device that does not exist: that rank fails alone, the call fails
            gpu_lib.knn_devices(X, Y, k, [0] * (N - 1) + [99], metric=metric, transport="loopback")
        gi2, gd2 = gpu_lib.knn_devices(X, Y, k, [0] * N, metric=metric, transport="loopback")      # ... and the library is still usable
        ri2, rd2 = gpu_lib.knn(X, Y, k, metric=metric)
        assert np.array_equal(gi2, ri2) and np.array_equal(gd2, rd2)
