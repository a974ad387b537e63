"""BASELINE.json configs[3] and configs[4] at their full shapes on ONE MI355X.

configs[3]: 1M ref x 1M target, d=50, k=15, references sharded 8 ways.  The driver measures the real 8-GPU run;
here the product's own sharded entry point (nabo_sharded_query: candidate query, grouped exchange, merge, the owner's
certificate, second round, gather) runs with 8 loopback ranks on the one GPU, in the 1-D form and in the 2 x 4 layout,
and every rank's result must equal the unsharded index on ALL rows bit for bit and the oracle on a row sample.

configs[4]: 5M ref x 5M target, d=100, k=50, cosine + 1000-permutation null (both EXTENSIONS: the reference has
neither, parity is pinned by this build's own oracle only).  Run WHOLE on the one GPU (tools/config4_one_gpu.py): all
5M x 5M pairs on one unsharded index, all of them again with the references sharded 8 ways (loopback ranks, every
rank's copy of every row compared), the reference graph, the SNN weights and the permutation null on the resulting
~250M-edge graph; the oracle re-solves stratified row samples and a sub-graph.  The 15M-edge null test keeps the
1000-permutation comparison with the oracle.
"""
import numpy as np
import pytest

import oracle
from nabo_amd import _knn
from nabo_amd._sharded import candidates_per_shard
from nabo_amd._synth import pca_like

pytestmark = pytest.mark.gpu


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


def test_baseline_configs4_whole_on_one_gpu(gpu_lib):
    """BASELINE configs[4] run WHOLE on one MI355X (EXTENSION metric + EXTENSION null: parity vs this build's oracle only):
    5M x 5M, d=100, k=50, cosine -- ALL 5M target rows on one unsharded index, ALL of them again through
    nabo_sharded_query with the reference rows sharded 8 ways (loopback ranks: every rank's copy of every row must equal
    the unsharded result), the reference <-> reference graph, the SNN weights, and nabo_score_null_edges on the resulting
    ~250M-edge graph with 1000 permutations (tools/config4_one_gpu.py).  Oracle: >= 256 target rows stratified by the
    pass that answered them + reference rows (order rows as nabo/_mapping.py:139-145 defines them, cosine expression of
    oracle/nabo_oracle.c), the SNN counts of a row sample, the mapping score (nabo/_graph.py:644-653) and the null on
    the sub-graph of 150 reference nodes."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "config4_one_gpu", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "config4_one_gpu.py"))
    c4 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(c4)
    n = m = 5000000
    d, k, N, P = 100, 50, 8, 1000
    rep, A = c4.run(n, m, d, k, ranks=N, perms=P, keep=True)
    X, Y, ti, td, ri, rd = A["X"], A["Y"], A["ti"], A["td"], A["ri"], A["rd"]
    # ---- all rows: size-independent properties ----------------------------------------------------------------------
    for gi, gd in ((ti, td), (ri, rd)):
        assert gi.min() >= 0 and gi.max() < n
        assert (np.diff(gd, axis=1) >= 0).all() and (gd >= -1e-15).all() and (gd <= 2.0 + 1e-12).all()
        srt = np.sort(gi, axis=1)
        assert (np.diff(srt, axis=1) > 0).all()                               # no repeated neighbour in any row
    assert not (ri == np.arange(n)[:, None]).any()                            # positional self-drop (no exact twins in this data)
    assert rep["sharded_loopback"]["all_rows_of_all_ranks_equal_unsharded"]
    assert rep["sharded_loopback"]["candidates_per_shard"] == candidates_per_shard(k, N, 1000000) == 23
    assert rep["sharded_loopback"]["second_round_rows_max"] < 100
    assert "l2c_topk_kernel<4," in rep["target_knn"]["kernel"], rep["target_knn"]["kernel"]
    assert rep["target_knn"]["rows_by_pass"]["exact"] < 1000
    # ---- >= 256 target rows + 64 reference rows against the oracle, stratified by the pass that answered them ------------
    rng = np.random.default_rng(9)
    rp = A["rp_t"]
    rare = np.nonzero(rp >= 1)[0]
    rows = np.unique(np.concatenate([rng.choice(rare, min(rare.size, 128), replace=False),
                                     np.nonzero(rp >= 2)[0][:64], rng.choice(m, 160, replace=False)]))
    assert rows.size >= 256
    oi, od = oracle.knn(X[rows], Y, k, oracle.COSINE, nthreads=16)
    assert np.array_equal(ti[rows], oi) and np.array_equal(td[rows], od)
    rrows = np.unique(np.concatenate([np.nonzero(A["rp_r"] >= 1)[0][:32], rng.choice(n, 32, replace=False)]))
    oi, od = oracle.knn(Y[rrows], Y, k, oracle.COSINE, drop_first=True, nthreads=16)
    assert np.array_equal(ri[rrows], oi) and np.array_equal(rd[rrows], od)
    # ---- SNN counts of a row sample (nabo/_mapping.py:186-198) -------------------------------------------------------------
    srows = rng.choice(m, 2000, replace=False)
    for t in srows[:200]:
        st = set(ti[t].tolist())
        want = [len(st & set(ri[j].tolist())) for j in ti[t]]
        assert A["snn"][t].tolist() == want
    assert rep["edges"] == int((A["snn"] > 0).sum()) and rep["edges"] > 100000000
    # ---- the null: observed score == the reference's mapping score; oracle on a sub-graph, first 48 permutations ----------
    e_t, e_r, w, group, null = A["e_t"], A["e_r"], A["w"], A["group"], A["null"]
    keep = group[e_t] != 0
    sc = gpu_lib.mapping_score_from_edges(n, e_r[keep], w[keep], int(group.sum()))
    assert np.allclose(null["obs"], sc, rtol=1e-12, atol=0)
    assert (null["sizes"] == int(group.sum())).all() and ((null["n_ge"] >= 0) & (null["n_ge"] <= P)).all()
    deg = np.bincount(e_r, minlength=n)
    hot = np.sort(rng.choice(np.nonzero(deg > 0)[0], 150, replace=False))
    sel = np.isin(e_r, hot)
    P2 = 48                                                                    # (the oracle labels all 5M pooled cells per permutation)
    small = gpu_lib.mapping_score_null(e_t[sel], np.searchsorted(hot, e_r[sel]), w[sel], group, hot.size, n_perm=P2, seed=3)
    o = oracle.score_null(e_t[sel], np.searchsorted(hot, e_r[sel]), w[sel], group, hot.size, P2, seed=3)
    for key in ("obs", "n_ge", "sizes"):
        assert np.array_equal(small[key], o[key]), key
    assert np.allclose(small["null_mean"], o["null_mean"], rtol=1e-12, atol=1e-12)
    assert np.allclose(small["null_sd"], o["null_sd"], rtol=1e-9, atol=1e-9)
    # the whole-graph run and the sub-graph run agree where they overlap (same permutations: a prefix of the 1000)
    assert np.array_equal(null["obs"][hot], small["obs"])
    assert (null["n_ge"][hot] >= small["n_ge"]).all()


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
