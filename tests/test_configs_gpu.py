"""BASELINE.json configs[3] and configs[4] at their full shapes on ONE MI355X.

configs[3]: 1M ref x 1M target, d=50, k=15, references sharded 8 ways.  The driver measures the real 8-GPU run;
here the product's own sharded entry point (nabo_sharded_query: candidate query, grouped exchange, merge, the owner's
certificate, second round, gather) runs with 8 loopback ranks on the one GPU, in the 1-D form and in the 2 x 4 layout,
and every rank's result must equal the unsharded index on ALL rows bit for bit and the oracle on a row sample.

configs[4]: 5M ref x 5M target, d=100, k=50, cosine + 1000-permutation null (both EXTENSIONS: the reference has
neither, parity is pinned by this build's own oracle only).  One rank's share at full size (625k references x
5M targets, 24 candidates per shard), the whole 8-shard protocol on a row sample against the unsharded index over
all 5M references and the oracle, and the permutation null at 15M edges x 1000 permutations against the oracle on
a sampled sub-graph.
"""
import numpy as np
import pytest

import oracle
from nabo_amd import _knn
from nabo_amd._sharded import shard_bounds, candidates_per_shard
from nabo_amd._synth import pca_like, pca_like_big

pytestmark = pytest.mark.gpu


def _protocol_one_gpu(gpu_lib, X, Y, k, N, metric, Ls, rows_chunk=None):
    """Every shard's candidate query, the merge and the owner's certificate, as tests/_dist_spec.py and nabo_sharded_query sequence them
    (exchange = host stack).  Returns merged idx/dist [m,k], certified flags [m] and per-shard stats."""
    m, n = X.shape[0], Y.shape[0]
    dx = _knn.DeviceBuffer(X.nbytes).upload(X)
    pi = np.empty((N, m, Ls), dtype=np.int64)
    pd = np.empty((N, m, Ls), dtype=np.float64)
    pb = np.empty((N, m), dtype=np.float64)
    stats = []
    for r in range(N):
        lo, hi = shard_bounds(n, N, r)
        sx = gpu_lib.KnnIndex(hi - lo, Y.shape[1], metric=metric, ref_index_base=lo).set_ref(Y[lo:hi])
        di, dd, db = _knn.DeviceBuffer(m * Ls * 8), _knn.DeviceBuffer(m * Ls * 8), _knn.DeviceBuffer(m * 8)
        sx.query_candidates_device(dx.ptr, m, Ls, di.ptr, dd.ptr, db.ptr)
        stats.append(sx.last_stats())
        pi[r], pd[r], pb[r] = di.download((m, Ls), np.int64), dd.download((m, Ls), np.float64), db.download((m,), np.float64)
        sx.close()
        for b in (di, dd, db):
            b.free()
    dx.free()
    dpi, dpd = _knn.DeviceBuffer(pi.nbytes).upload(pi), _knn.DeviceBuffer(pd.nbytes).upload(pd)
    doi, dod = _knn.DeviceBuffer(m * k * 8), _knn.DeviceBuffer(m * k * 8)
    _knn.merge_topk_device(dpi.ptr, dpd.ptr, N, m, Ls, k, False, doi.ptr, dod.ptr)
    mi, md = doi.download((m, k), np.int64), dod.download((m, k), np.float64)
    for b in (dpi, dpd, doi, dod):
        b.free()
    dk = md[:, k - 1]
    ok = (mi[:, k - 1] >= 0) & (dk * dk * (1 + 1e-12) < pb.min(axis=0))          # the owner certificate (sharded.hip certify_kernel)
    return mi, md, ok, stats, (pi, pd, pb)


def _second_round(gpu_lib, X, Y, k, N, metric, bad, mi, md):
    """rows the owner could not certify: exact local top-k of just those rows on every shard, merged"""
    if bad.size == 0:
        return
    Xb = np.ascontiguousarray(X[bad])
    n = Y.shape[0]
    pi = np.empty((N, bad.size, k), dtype=np.int64)
    pd = np.empty((N, bad.size, k), dtype=np.float64)
    for r in range(N):
        lo, hi = shard_bounds(n, N, r)
        sx = gpu_lib.KnnIndex(hi - lo, Y.shape[1], metric=metric, ref_index_base=lo).set_ref(Y[lo:hi])
        pi[r], pd[r] = sx.query(Xb, k)
        sx.close()
    dpi, dpd = _knn.DeviceBuffer(pi.nbytes).upload(pi), _knn.DeviceBuffer(pd.nbytes).upload(pd)
    doi, dod = _knn.DeviceBuffer(bad.size * k * 8), _knn.DeviceBuffer(bad.size * k * 8)
    _knn.merge_topk_device(dpi.ptr, dpd.ptr, N, bad.size, k, k, False, doi.ptr, dod.ptr)
    mi[bad], md[bad] = doi.download((bad.size, k), np.int64), dod.download((bad.size, k), np.float64)
    for b in (dpi, dpd, doi, dod):
        b.free()


def _loopback_all_ranks(N, R, X, Y, k, metric=0, drop=False):
    """nabo_sharded_query itself (certify_kernel, adopt_kernel, the grouped exchange, the ragged-slice fill) with N
    shard-ranks on this one GPU: every rank writes its own copy of the [m,k] result."""
    from nabo_amd import _sharded
    m = X.shape[0]
    grp = _sharded.LoopbackGroup(N, 0, Y.shape[0], Y.shape[1], metric, Y, ref_shards=R).set_ref()
    dx = _knn.DeviceBuffer(X.nbytes).upload(X)
    outs = [(_knn.DeviceBuffer(m * k * 8), _knn.DeviceBuffer(m * k * 8)) for _ in range(N)]
    try:
        grp.query_device(dx.ptr, m, k, drop, [a.ptr for a, _ in outs], [b.ptr for _, b in outs])
        st = [grp.last_stats(r) for r in range(N)]
        kern = grp.indices[0].last_stats()
        res = [(a.download((m, k), np.int64), b.download((m, k), np.float64)) for a, b in outs]
    finally:
        grp.close()
        dx.free()
        for a, b in outs:
            a.free(); b.free()
    return res, st, kern


@pytest.mark.parametrize("N,R", [(8, 8), (8, 2), (2, 2)])
def test_baseline_configs3_1M_sharded_query_equals_unsharded(gpu_lib, N, R):
    """BASELINE configs[3] at full size THROUGH THE PRODUCT'S ENTRY POINT: 1M x 1M, d=50, k=15, nabo_sharded_query with
    N loopback ranks -- the 1-D form (one reference piece per rank: the layout BASELINE names) and the 2 x 4 layout.
    Every rank's copy of every row equals the unsharded index, a row sample equals the oracle."""
    n = m = 1000000
    d, k = 50, 15
    Y = pca_like(n, d, seed=1003)
    X = pca_like(m, d, seed=2003)
    ix = gpu_lib.KnnIndex(n, d, metric=0).set_ref(Y)
    ri, rd = ix.query(X, k)
    ix.close()
    res, st, kern = _loopback_all_ranks(N, R, X, Y, k)
    for gi, gd in res:
        assert np.array_equal(gi, ri) and np.array_equal(gd, rd)              # EVERY row, indices and distances
    rows = np.random.default_rng(8).choice(m, 32, replace=False)
    oi, od = oracle.knn(X[rows], Y, k, 0, nthreads=8)
    assert np.array_equal(res[0][0][rows], oi) and np.array_equal(res[0][1][rows], od)
    assert all(s["protocol"] == "global" for s in st)
    assert st[0]["candidates"] == candidates_per_shard(k, R, m)
    assert st[0]["uncertified"] < 100, "the list-length rule expects < 0.1 uncertified rows per batch"
    assert len({s["uncertified"] for s in st}) == 1


def test_baseline_configs3_sorted_references_take_the_second_round(gpu_lib):
    """References ordered along the first component: a target's neighbours sit in ONE shard, which then holds more
    than Ls of the global top-k; the owners must refuse those rows (certify_kernel) and the second round must repair
    them (adopt_kernel) -- 200k target rows through nabo_sharded_query, 8 ranks, every rank's copy compared."""
    n, m, d, k, N = 1000000, 200000, 50, 15, 8
    Y = pca_like(n, d, seed=1003)
    Y = np.ascontiguousarray(Y[np.argsort(Y[:, 0], kind="stable")])
    X = pca_like(m, d, seed=2003)
    ix = gpu_lib.KnnIndex(n, d, metric=0).set_ref(Y)
    ri, rd = ix.query(X, k)
    ix.close()
    res, st, _ = _loopback_all_ranks(N, N, X, Y, k)
    assert st[0]["uncertified"] > 100, "sorted references should defeat the exchangeable-shard list length (%d rows)" % st[0]["uncertified"]
    for gi, gd in res:
        assert np.array_equal(gi, ri) and np.array_equal(gd, rd)


def test_baseline_configs4_one_ranks_share_and_protocol_sample(gpu_lib):
    """BASELINE configs[4] (EXTENSION metric, parity vs this build's oracle): rank 0's share at full size --
    625k references x 5M targets, d=100, k=50, cosine, 24 candidates per shard -- then the 8-shard protocol on a
    row sample against the unsharded index over all 5M references and against the oracle."""
    n = m = 5000000
    d, k, N = 100, 50, 8
    MET = gpu_lib.COSINE
    Y = pca_like_big(n, d, seed=1004)
    X = pca_like_big(m, d, seed=2004)
    Ls = candidates_per_shard(k, N, m)
    assert Ls == 24
    lo, hi = shard_bounds(n, N, 0)
    sx = gpu_lib.KnnIndex(hi - lo, d, metric=MET, ref_index_base=lo).set_ref(Y[lo:hi])
    batch = 1000000
    di, dd, db = _knn.DeviceBuffer(batch * Ls * 8), _knn.DeviceBuffer(batch * Ls * 8), _knn.DeviceBuffer(batch * 8)
    dx = _knn.DeviceBuffer(batch * d * 8)
    rng = np.random.default_rng(9)
    ms = 0.0
    for b0 in range(0, m, batch):
        xb = np.ascontiguousarray(X[b0:b0 + batch])
        dx.upload(xb)
        sx.query_candidates_device(dx.ptr, batch, Ls, di.ptr, dd.ptr, db.ptr)
        ms += sx.last_stats()["ms_total"]
        ci, cd, cb = di.download((batch, Ls), np.int64), dd.download((batch, Ls), np.float64), db.download((batch,), np.float64)
        # all rows: global indices of THIS shard, no repeats, ascending distances, a usable bound
        assert ci.min() >= lo and ci.max() < hi
        assert (np.diff(np.sort(ci, axis=1), axis=1) > 0).all()
        assert (np.diff(cd, axis=1) >= 0).all() and (cd >= -1e-15).all() and (cd <= 2.0 + 1e-12).all()
        assert np.isfinite(cb).all() and (cb > 0).all()
        # sampled rows: the list is the oracle's order row of the shard, and the bound really bounds the rest
        rows = rng.choice(batch, 6, replace=False)
        oi, od = oracle.knn(xb[rows], Y[lo:hi], Ls + 1, oracle.COSINE, nthreads=8)
        assert np.array_equal(ci[rows], oi[:, :Ls] + lo) and np.array_equal(cd[rows], od[:, :Ls])
        assert (cb[rows] <= od[:, Ls] ** 2 * (1 + 1e-12)).all()
    sx.close()
    for b in (di, dd, db, dx):
        b.free()
    assert ms < 20000, "rank share took %.0f ms" % ms          # 4.9 s in round 1; a gross regression guard only
    # the whole protocol for a sample of targets
    sample = 20000
    rows = np.sort(rng.choice(m, sample, replace=False))
    Xs = np.ascontiguousarray(X[rows])
    del X
    ix = gpu_lib.KnnIndex(n, d, metric=MET).set_ref(Y)
    ri, rd = ix.query(Xs, k)
    ix.close()
    mi, md, ok, _, _ = _protocol_one_gpu(gpu_lib, Xs, Y, k, N, MET, Ls)
    bad = np.nonzero(~ok)[0]
    assert bad.size < 20
    _second_round(gpu_lib, Xs, Y, k, N, MET, bad, mi, md)
    assert np.array_equal(mi, ri) and np.array_equal(md, rd)
    oi, od = oracle.knn(Xs[:16], Y, k, oracle.COSINE, nthreads=8)
    assert np.array_equal(mi[:16], oi) and np.array_equal(md[:16], od)


def test_baseline_configs4_permutation_null_15M_edges_1000_permutations(gpu_lib):
    """BASELINE configs[4], second half (EXTENSION): nabo_score_null_edges at 1M reference nodes, 1M pooled
    target cells, 15M edges, 1000 permutations; the oracle restates it for a sampled sub-graph (the edges of 150
    reference nodes, all target labels)."""
    n_ref = n_t = 1000000
    k, P = 15, 1000
    rng = np.random.default_rng(21)
    e_t = np.repeat(np.arange(n_t, dtype=np.int64), k)
    e_r = rng.integers(0, n_ref, n_t * k)
    hot = rng.choice(n_ref, 150, replace=False)               # make sure sampled nodes have some edges
    e_r[rng.choice(n_t * k, 6000, replace=False)] = rng.choice(hot, 6000)
    w = rng.choice(np.round(np.arange(1, k + 1) / (2.0 * (k - 1) - np.arange(1, k + 1)), 2), n_t * k)
    group = (rng.random(n_t) < 0.4).astype(np.uint8)
    res = gpu_lib.mapping_score_null(e_t, e_r, w, group, n_ref, n_perm=P, seed=3)
    # observed score == the reference's mapping score (Graph.get_mapping_score) of the group, all nodes
    keep = group[e_t] != 0
    sc = gpu_lib.mapping_score_from_edges(n_ref, e_r[keep], w[keep], int(group.sum()))
    assert np.allclose(res["obs"], sc, rtol=1e-12, atol=0)
    assert (res["sizes"] == int(group.sum())).all()
    assert ((res["n_ge"] >= 0) & (res["n_ge"] <= P)).all()
    sel = np.isin(e_r, hot)
    hs = np.sort(hot)
    o = oracle.score_null(e_t[sel], np.searchsorted(hs, e_r[sel]), w[sel], group, hs.size, P, seed=3)   # nodes renumbered
    for key in ("obs", "n_ge"):                                # same float64 operations in the same order; integers
        assert np.array_equal(res[key][hs], o[key]), key
    assert np.allclose(res["null_mean"][hs], o["null_mean"], rtol=1e-12, atol=1e-12)     # reduction order over P differs
    assert np.allclose(res["null_sd"][hs], o["null_sd"], rtol=1e-9, atol=1e-9)
    assert np.array_equal(res["sizes"], o["sizes"])
