"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference-generated
golden vectors.  Indices must be bit-exact, distances are compared bit-for-bit as well
(the contract allows 1e-4 relative; the implementation recomputes them in the reference's own
float64 arithmetic, so equality is asserted and the tolerance is only the documented bound)."""
import os

import numpy as np
import pytest

import oracle
from nabo_amd._synth import pca_like

pytestmark = pytest.mark.gpu
RTOL = 1e-4        # north_star tolerance for distances; we assert exact equality below


def _check(gi, gd, oi, od):
    assert np.array_equal(gi, oi), "indices differ in %d rows" % int((gi != oi).any(axis=1).sum())
    assert np.array_equal(gd, od), "max rel dist err %g" % float(np.max(np.abs(gd - od) / np.maximum(od, 1e-300)))


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("d", [7, 30, 50])
def test_pairwise_bit_exact_vs_golden(gpu_lib, golden, metric, d):
    g = golden("kernels")
    x, y = g["x_%d" % d], g["y_%d" % d]
    if metric == 0:
        assert np.array_equal(gpu_lib.pairwise(x, y, 0), g["euclid_%d" % d])
    else:
        for f in (0.1, 0.25, 1.0):
            assert np.array_equal(gpu_lib.pairwise(x, y, 1, f), g["canberra_%d_%s" % (d, str(f).replace(".", "p"))])


def test_device_sqrt_and_divide_are_correctly_rounded(gpu_lib):
    """The refine kernel relies on IEEE sqrt/div on the device; check them on 1e5 random pairs."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((300, 1)) * np.exp(rng.uniform(-20, 20, (300, 1)))
    y = rng.standard_normal((400, 1)) * np.exp(rng.uniform(-20, 20, (400, 1)))
    assert np.array_equal(gpu_lib.pairwise(x, y, 0), np.abs(x - y.T))        # sqrt(t*t) == |t|
    assert np.array_equal(gpu_lib.pairwise(x, y, 1, 1e9), oracle.pairwise(x, y, 1, 1e9))   # always divides
    x2 = rng.standard_normal((300, 2)) * np.exp(rng.uniform(-5, 5, (300, 1)))
    y2 = rng.standard_normal((400, 2)) * np.exp(rng.uniform(-5, 5, (400, 1)))
    assert np.array_equal(gpu_lib.pairwise(x2, y2, 0), oracle.pairwise(x2, y2, 0))


SHAPES = [
    # m, n, g, k, drop
    (1, 40, 5, 3, False),
    (33, 64, 16, 8, False),
    (100, 1000, 30, 11, True),
    (257, 4097, 50, 15, False),
    (1000, 1000, 15, 11, True),
    (64, 20000, 50, 15, False),
    (3000, 3000, 100, 23, True),
    (130, 5000, 64, 30, False),      # L=64 lists
    (200, 3000, 128, 50, True),      # max components, k=50 (BASELINE config 5 neighbour count)
]


@pytest.mark.parametrize("m,n,g,k,drop", SHAPES)
def test_euclidean_knn_vs_oracle(gpu_lib, m, n, g, k, drop):
    Y = pca_like(n, g, seed=1000 + n + g)
    X = Y[:m].copy() if drop else pca_like(m, g, seed=2000 + m + g)
    gi, gd = gpu_lib.knn(X, Y, k, metric=0, drop_first=drop)
    oi, od = oracle.knn(X, Y, k, 0, drop_first=drop, nthreads=8)
    _check(gi, gd, oi, od)


@pytest.mark.parametrize("m,n,g,k,drop", [(50, 300, 10, 5, False), (500, 3000, 30, 11, False),
                                          (300, 2000, 50, 30, True), (40, 70, 7, 50, False)])
def test_canberra_knn_vs_oracle(gpu_lib, m, n, g, k, drop):
    Y = pca_like(n, g, seed=1000 + n + g)
    X = pca_like(m, g, seed=2000 + m + g)
    for f in (0.25, 1.0):
        gi, gd = gpu_lib.knn(X, Y, k, metric=1, dist_factor=f, drop_first=drop)
        oi, od = oracle.knn(X, Y, k, 1, f, drop_first=drop, nthreads=8)
        _check(gi, gd, oi, od)


@pytest.mark.parametrize("m,n,g,k,drop,f,masked", [
    (50, 300, 10, 5, False, 0.25, False),            # one block, mostly padding bits
    (500, 3000, 30, 11, False, 0.25, True),          # two blocks, ignored references
    (333, 5000, 50, 15, True, 0.25, False),          # positional drop; 56 padded dimensions = 7 groups (odd: buffer parity flips)
    (2100, 9000, 50, 20, False, 1.0, False),         # f = 1: wide windows, many survivors; 66 waves of 32 rows
    (70, 40000, 63, 24, False, 0.1, True),           # the largest g / k' the bit-sliced pass is instantiated for, reference splits
    (40, 2600, 7, 3, False, 0.25, False),
    (1500, 70000, 16, 9, False, 0.25, False),        # g = 16: two groups (even)
])
def test_canberra_bit_sliced_count_gives_the_same_bits(gpu_lib, m, n, g, k, drop, f, masked):
    """canberra_bits.hip (the default counting pass from 25k references on; NABO_CANBERRA_MODE=bits forces it at any
    size): cumulative per-bucket bitmaps, carry-save count, bit-sliced comparator.  Same candidates' certificate, same
    float64 refine: the oracle's bits -- and the SWAR pass (NABO_CANBERRA_MODE=swar) on the same inputs too."""
    Y = pca_like(n, g, seed=4100 + n + g)
    Y[11] = Y[5]                                             # an exact tie
    X = Y[:m].copy() if drop else pca_like(m, g, seed=4200 + m + g)
    X[0] = 1e-3 * X[0]                                       # tiny windows: everything far away sits on the plateau (distance g)
    mask = (np.random.default_rng(n).random(n) < 0.25).astype(np.uint8) if masked else None
    oi, od = oracle.knn(X, Y, k, 1, f, ref_mask=mask, drop_first=drop, nthreads=8)
    for mode in ("bits", "swar"):
        os.environ["NABO_CANBERRA_MODE"] = mode
        try:
            ix = gpu_lib.KnnIndex(n, g, metric=1, dist_factor=f).set_ref(Y, ref_mask=mask)
        finally:
            os.environ.pop("NABO_CANBERRA_MODE", None)
        gi, gd = ix.query(X, k, drop_first=drop)
        kern = ix.last_kernel()
        if mask is not None and mode == "bits":              # a new ignore list on the resident index: the valid bits follow
            mask2 = np.roll(mask, 17)
            ix.set_mask(mask2)
            g2i, g2d = ix.query(X[:64], k, drop_first=drop)
            o2i, o2d = oracle.knn(X[:64], Y, k, 1, f, ref_mask=mask2, drop_first=drop, nthreads=8)
            _check(g2i, g2d, o2i, o2d)
        ix.close()
        assert ("cbb_filter" if mode == "bits" else "cbf_filter") in kern, kern
        _check(gi, gd, oi, od)


def test_canberra_exact_ties_follow_canonical_order(gpu_lib):
    """Far-apart pairs all score exactly g: the top-k is tie-filled and must come out in
    ascending index order (the canonical (dist, idx) rule)."""
    rng = np.random.default_rng(9)
    Y = rng.standard_normal((900, 12)) * 100.0
    X = rng.standard_normal((70, 12)) * 0.01
    gi, gd = gpu_lib.knn(X, Y, 20, metric=1, dist_factor=0.25)
    oi, od = oracle.knn(X, Y, 20, 1, 0.25)
    assert (od == 12.0).mean() > 0.5            # the case really is tie-dominated
    _check(gi, gd, oi, od)


@pytest.mark.parametrize("metric", [0, 1])
def test_ignore_mask_and_masked_tail(gpu_lib, metric):
    Y = pca_like(500, 20, seed=77)
    X = pca_like(90, 20, seed=78)
    mask = np.zeros(500, np.uint8)
    mask[np.random.default_rng(1).choice(500, 60, replace=False)] = 1
    gi, gd = gpu_lib.knn(X, Y, 15, metric=metric, ref_mask=mask)
    oi, od = oracle.knn(X, Y, 15, metric, ref_mask=mask)
    _check(gi, gd, oi, od)
    assert not mask[gi].any()
    # fewer un-ignored refs than k: ignored refs follow in ascending index order with true distances
    Ys = Y[:30]
    mk = np.zeros(30, np.uint8)
    mk[[1, 5, 6, 20, 29]] = 1
    gi, gd = gpu_lib.knn(X, Ys, 28, metric=metric, ref_mask=mk)
    oi, od = oracle.knn(X, Ys, 28, metric, ref_mask=mk)
    _check(gi, gd, oi, od)
    assert np.array_equal(gi[:, 25:], np.tile([1, 5, 6], (90, 1)))


def test_duplicates_force_the_exact_fallback(gpu_lib):
    """40 identical reference cells: the fp32 filter cannot separate them, the guard must flag the
    rows and the float64 fallback must still return the canonical answer."""
    Y = pca_like(3000, 30, seed=5)
    Y[100:140] = Y[100]
    X = np.vstack([Y[100:103], pca_like(50, 30, seed=6)])
    ix = gpu_lib.KnnIndex(3000, 30, metric=0).set_ref(Y)
    gi, gd = ix.query(X, 15)
    st = ix.last_stats()
    oi, od = oracle.knn(X, Y, 15, 0)
    _check(gi, gd, oi, od)
    assert st["fallback_rows"] >= 3
    assert np.array_equal(gi[0], np.arange(100, 115))
    ix.close()


def test_golden_mapping_small_through_the_abi(gpu_lib, golden):
    g = golden("mapping_small")
    uc, k, _ = g["params"]
    names = list(g["ref_names"])
    ref = g["ref"][[names.index(c) for c in g["ref_cells"]]][:, :uc]
    gi, gd = gpu_lib.knn(ref, ref, 32, metric=0, drop_first=True)
    assert np.array_equal(gi, g["ref_idx"]) and np.array_equal(gd, g["ref_dist"])
    for t in ("ME", "IG"):
        tn = list(g["t_%s_names" % t])
        X = g["t_%s_data" % t][[tn.index(c) for c in g["t_%s_cells" % t]]][:, :uc]
        mask = np.isin(g["ref_cells"], g["t_%s_ignore" % t]).astype(np.uint8)
        gi, gd = gpu_lib.knn(X, ref, 32, metric=1, dist_factor=float(g["dist_factor"]), ref_mask=mask)
        assert np.array_equal(gi, g["t_%s_idx" % t]) and np.array_equal(gd, g["t_%s_dist" % t])


def test_golden_c1_full_size(gpu_lib, golden):
    """BASELINE.json configs[0] against the reference's own output."""
    g = golden("c1_3k")
    ref, tgt = pca_like(3000, 30, 1001), pca_like(3000, 30, 2001)
    gi, gd = gpu_lib.knn(ref, ref, 16, metric=0, drop_first=True)
    assert np.array_equal(gi, g["ref_idx"]) and np.array_equal(gd[:, :12], g["ref_dist"])
    gi, gd = gpu_lib.knn(tgt, ref, 16, metric=1, dist_factor=0.25)
    assert np.array_equal(gi, g["t_ME_idx"]) and np.array_equal(gd[:, :12], g["t_ME_dist"])


def test_reference_splits_do_not_change_results(gpu_lib):
    Y = pca_like(20000, 50, seed=31)
    X = pca_like(300, 50, seed=32)
    base = None
    for s in ("1", "2", "5", "8"):
        r = gpu_lib.knn(X, Y, 15, metric=0, options={"splits": int(s)})
        if base is None:
            base = r
        assert np.array_equal(r[0], base[0]) and np.array_equal(r[1], base[1])
    oi, od = oracle.knn(X, Y, 15, 0, nthreads=8)
    _check(base[0], base[1], oi, od)


def test_resident_index_shard_base_and_merge(gpu_lib):
    """Reference rows sharded 3 ways on ONE device, merged with nabo_merge_topk == unsharded."""
    from nabo_amd import _knn
    from nabo_amd._sharded import shard_bounds
    Y = pca_like(10001, 30, seed=41)
    X = pca_like(777, 30, seed=42)
    k, kk = 11, 12
    pi, pd = [], []
    for r in range(3):
        lo, hi = shard_bounds(10001, 3, r)
        ix = gpu_lib.KnnIndex(hi - lo, 30, metric=0, ref_index_base=lo).set_ref(Y[lo:hi])
        i, d = ix.query(X, kk)
        ix.close()
        assert i.min() >= lo and i.max() < hi
        pi.append(i)
        pd.append(d)
    pi, pd = np.stack(pi), np.stack(pd)
    dpi = _knn.DeviceBuffer(pi.nbytes).upload(pi)
    dpd = _knn.DeviceBuffer(pd.nbytes).upload(pd)
    doi, dod = _knn.DeviceBuffer(777 * k * 8), _knn.DeviceBuffer(777 * k * 8)
    _knn.merge_topk_device(dpi.ptr, dpd.ptr, 3, 777, kk, k, True, doi.ptr, dod.ptr)
    gi, gd = doi.download((777, k), np.int64), dod.download((777, k), np.float64)
    oi, od = oracle.knn(X, Y, k, 0, drop_first=True, nthreads=8)
    _check(gi, gd, oi, od)


@pytest.mark.parametrize("metric", [0, 2])
def test_shard_candidates_and_their_bound(gpu_lib, metric):
    """nabo_index_query_candidates: the emitted list is the exact head of the shard's order row and the
    bound is a true lower bound on the squared distance of everything not emitted (global certification)."""
    from nabo_amd import _knn
    n, g, m = 6001, 30, 513
    Y = pca_like(n, g, seed=43)
    X = pca_like(m, g, seed=44)
    mask = np.zeros(n, dtype=np.uint8)
    mask[::7] = 1
    full_i, full_d = oracle.knn(X, Y, 40, metric, ref_mask=mask, nthreads=8)
    ix = gpu_lib.KnnIndex(n, g, metric=metric, ref_index_base=1000).set_ref(Y, ref_mask=mask)
    dx = _knn.DeviceBuffer(X.nbytes).upload(X)
    for nc in (1, 9, 16, 32):
        di, dd, db = _knn.DeviceBuffer(m * nc * 8), _knn.DeviceBuffer(m * nc * 8), _knn.DeviceBuffer(m * 8)
        ix.query_candidates_device(dx.ptr, m, nc, di.ptr, dd.ptr, db.ptr)
        ci, cd, cb = di.download((m, nc), np.int64), dd.download((m, nc), np.float64), db.download((m,), np.float64)
        have = ci >= 0
        # whatever is emitted is the exact prefix, in canonical order, with exact float64 distances
        assert np.array_equal(np.where(have, ci - 1000, -1), np.where(have, full_i[:, :nc], -1))
        assert np.array_equal(np.where(have, cd, 0.0), np.where(have, full_d[:, :nc], 0.0))
        assert np.all(np.isinf(cd[~have]))
        assert have[:, 0].all() or nc == 1
        # bound: every reference not emitted is at squared distance >= bound (or the bound says "unknown")
        cnt = have.sum(1)
        nxt = full_d[np.arange(m), np.minimum(cnt, 39)] ** 2
        known = np.isfinite(cb)
        assert known.mean() > 0.95
        assert np.all(cb[known] <= nxt[known] * (1 + 1e-12))
        assert np.all(cb[known] > 0)
        assert not np.any(cb == np.inf)          # n - masked > n_cand here: something else always exists
    ix.close()


def test_snn_counts_kernel_vs_oracle(gpu_lib):
    Y = pca_like(2000, 20, seed=51)
    idx, _ = oracle.knn(Y, Y, 11, 0, drop_first=True, nthreads=8)
    cnt = gpu_lib.snn_counts(idx, idx, 11)
    t, j, w = oracle.snn_edges(idx, idx, 11)
    tt, ss = np.nonzero(cnt > 0)
    assert np.array_equal(tt, t) and np.array_equal(idx[tt, ss], j)
    assert np.array_equal(np.array([oracle.snn_weight(int(c), 11) for c in cnt[tt, ss]]), w)


def test_config2_100k_sampled_rows(gpu_lib):
    """BASELINE config[1] (100k x 100k, d=50, k=15): full GPU run; the oracle re-solves a sample of
    rows exactly, and size-independent properties are checked on all rows."""
    n = 100000
    Y = pca_like(n, 50, seed=1002)
    X = pca_like(n, 50, seed=2002)
    ix = gpu_lib.KnnIndex(n, 50, metric=0).set_ref(Y)
    gi, gd = ix.query(X, 15)
    st = ix.last_stats()
    ix.close()
    assert (np.diff(gd, axis=1) >= 0).all()                       # sorted
    assert gi.min() >= 0 and gi.max() < n
    assert all(len(set(r)) == 15 for r in gi[::997])              # no repeated neighbour
    rows = np.random.default_rng(3).choice(n, 256, replace=False)
    oi, od = oracle.knn(X[rows], Y, 15, 0, nthreads=8)
    _check(gi[rows], gd[rows], oi, od)
    # distances returned are the reference formula evaluated at the returned indices
    chk = oracle.pairwise(X[rows[:8]], Y[gi[rows[0]]], 0)[0]
    assert np.array_equal(chk, gd[rows[0]])
    assert st["fallback_rows"] < n // 100


def _stratified_rows(rp, n_total, n_rare, seed):
    """>= n_total rows for the oracle: EVERY row a pass behind the seeded one answered, up to n_rare (at least) of the rows
    the seeded pass answered, the rest uniform over the rows of the first pass (nabo_index_last_row_pass)."""
    rng = np.random.default_rng(seed)
    beyond = np.nonzero(rp >= 2)[0]
    seeded = np.nonzero(rp == 1)[0]
    take = [beyond[:2048], rng.choice(seeded, min(seeded.size, n_rare), replace=False)]
    first = np.nonzero(rp == 0)[0]
    left = max(n_total - sum(t.size for t in take), 256)
    take.append(rng.choice(first, min(first.size, left), replace=False))
    return np.unique(np.concatenate(take))


def test_config3_full_size_1M_properties(gpu_lib):
    """BASELINE configs[2] at FULL size (1M x 1M, d=50, k=15, the bench workload) through the resident-index
    entry points: size-independent properties on all rows; >= 1024 rows re-solved by the oracle, STRATIFIED by the pass
    that answered them (nabo_index_last_row_pass): at least 256 of the ~0.7 % of rows the seeded pass answered, every
    row a pass behind it answered, the rest uniform (the order rows must equal nabo/_mapping.py:139-145 whichever
    pass produced them); and the reference-vs-itself form (positional self-drop) on a slice."""
    n = 1000000
    Y = pca_like(n, 50, seed=1003)
    X = pca_like(n, 50, seed=2003)
    ix = gpu_lib.KnnIndex(n, 50, metric=0).set_ref(Y)
    gi, gd = ix.query(X, 15)
    st = ix.last_stats()
    rp = ix.last_row_pass(n)
    assert st["fallback_rows"] < 1000
    # the per-row record agrees with the per-pass counters
    assert int((rp >= 1).sum()) == st["seeded_pass_rows"]
    assert int((rp >= 2).sum()) == st["second_pass_rows"] and int((rp >= 3).sum()) >= st["wide_list_rows"]
    assert int((rp == 4).sum()) == st["fallback_rows"]
    assert "l2c_topk_kernel" in ix.last_kernel() and st["seeded_pass_rows"] >= 256, "the default chain: one-product pass, then seeded"
    assert (np.diff(gd, axis=1) >= 0).all()                       # sorted rows
    assert gi.min() >= 0 and gi.max() < n
    s = np.sort(gi, axis=1)
    assert (np.diff(s, axis=1) > 0).all()                         # no repeated neighbour in any row
    rows = _stratified_rows(rp, 1024, 384, seed=5)
    assert rows.size >= 1024 and int((rp[rows] == 1).sum()) >= 256 and set(np.nonzero(rp >= 2)[0][:2048]) <= set(rows)
    oi, od = oracle.knn(X[rows], Y, 15, 0, nthreads=16)
    _check(gi[rows], gd[rows], oi, od)
    # every returned distance is the reference formula at the returned index (checksum over a sample of rows)
    rs = rows[:16]
    d_chk = np.stack([oracle.pairwise(X[r:r + 1], Y[gi[r]], 0)[0] for r in rs])
    assert np.array_equal(d_chk, gd[rs])
    # ref <-> ref with the positional [1:] drop: a cell never lists itself unless it has an exact twin
    gi2, gd2 = ix.query(Y[:20000], 15, drop_first=True)
    ix.close()
    assert not (gi2 == np.arange(20000)[:, None]).any()
    assert (gd2[:, 0] > 0).all()
    o2i, o2d = oracle.knn(Y[rows[:8] % 20000], Y, 15, 0, drop_first=True, nthreads=8)
    _check(gi2[rows[:8] % 20000], gd2[rows[:8] % 20000], o2i, o2d)


def test_config3_full_size_1M_every_filter_gives_the_same_bits_on_all_rows(gpu_lib, monkeypatch):
    """ALL 1M rows of BASELINE configs[2] through three different first filters -- the default chain (one-product pass,
    seeded pass, ...), the f16x3 split first (NABO_L2_MODE=f16x3) and the fp32-MFMA filter (NABO_L2_MODE=f32: no f16
    arithmetic anywhere) -- must agree bit for bit, indices and distances: the all-rows cross-check of bench.py's alt
    blocks as a test.  (They share refine_kernel's float64 evaluation; the filters, their error budgets and the rows
    that leave the first pass differ: the 0.7 % of rows the seeded pass answers in the default chain are certified by
    the first pass in the other two.)"""
    n = 1000000
    Y = pca_like(n, 50, seed=1003)
    X = pca_like(n, 50, seed=2003)
    res = {}
    for mode in ("", "f16x3", "f32"):
        if mode:
            monkeypatch.setenv("NABO_L2_MODE", mode)
        else:
            monkeypatch.delenv("NABO_L2_MODE", raising=False)
        ix = gpu_lib.KnnIndex(n, 50, metric=0).set_ref(Y)
        gi, gd = ix.query(X, 15)
        res[mode] = (gi, gd, ix.last_kernel(), ix.last_row_pass(n), ix.last_stats())
        ix.close()
    monkeypatch.delenv("NABO_L2_MODE", raising=False)
    assert "l2c_topk_kernel" in res[""][2] and "l2q_topk_kernel" in res["f16x3"][2] and "f32_32x32x2" in res["f32"][2]
    for mode in ("f16x3", "f32"):
        assert np.array_equal(res[mode][0], res[""][0]), mode
        assert np.array_equal(res[mode][1], res[""][1]), mode
        assert (res[mode][3] >= 2).all()                          # these chains start at the "second" filter
    # the rows the default chain could not certify in its first pass were certified by the first pass of the others
    seeded = res[""][3] == 1
    assert seeded.sum() >= 256 and (res["f32"][3][seeded] == 2).mean() > 0.99


def test_canberra_tail_round_rows_are_exact(gpu_lib):
    """Many target rows: the last, partially filled round of workgroups is launched with its own reference
    split (api.hip); rows of both launches must equal the oracle."""
    n, g, k = 20000, 30, 11
    m = 2048 * 16 + 700                       # one full round of one-wave workgroups + a tail
    Y = pca_like(n, g, seed=95)
    X = pca_like(m, g, seed=96)
    gi, gd = gpu_lib.knn(X, Y, k, metric=1, dist_factor=0.25)
    rows = np.concatenate([np.arange(0, 40), np.arange(2048 * 16 - 20, 2048 * 16 + 20), np.arange(m - 40, m)])
    oi, od = oracle.knn(X[rows], Y, k, 1, 0.25, nthreads=8)
    _check(gi[rows], gd[rows], oi, od)
    hi, hd = gpu_lib.knn(X, Y, k, metric=1, dist_factor=0.25, options={"tail_split": 0})
    assert np.array_equal(gi, hi) and np.array_equal(gd, hd)


def test_canberra_full_size_1M_sampled(gpu_lib):
    """The reference's target<->reference metric at 1M x 1M (d=50, k=15): sorted rows, valid indices, and a
    row sample equal to the oracle bit for bit."""
    n = 1000000
    Y = pca_like(n, 50, seed=1003)
    X = pca_like(n, 50, seed=2003)
    ix = gpu_lib.KnnIndex(n, 50, metric=1, dist_factor=0.25).set_ref(Y)
    gi, gd = ix.query(X, 15)
    st = ix.last_stats()
    ix.close()
    assert (np.diff(gd, axis=1) >= 0).all()
    assert gi.min() >= 0 and gi.max() < n
    rows = np.random.default_rng(6).choice(n, 48, replace=False)
    oi, od = oracle.knn(X[rows], Y, 15, 1, 0.25, nthreads=8)
    _check(gi[rows], gd[rows], oi, od)
    assert st["fallback_rows"] < 1000


def test_far_from_origin_and_badly_scaled_inputs(gpu_lib):
    """Centring + the certification must keep results exact when the cloud sits far from the origin
    and when components differ by many orders of magnitude."""
    Y = pca_like(5000, 30, seed=61)
    X = pca_like(300, 30, seed=62)
    for shift, scale in ((1e4, 1.0), (0.0, 1e-6), (-3e5, 1e3)):
        Ys, Xs = Y * scale + shift, X * scale + shift
        gi, gd = gpu_lib.knn(Xs, Ys, 15, metric=0)
        oi, od = oracle.knn(Xs, Ys, 15, 0, nthreads=8)
        _check(gi, gd, oi, od)


def test_any_unit_of_the_input_is_fine_and_fp32_overflow_is_noticed(gpu_lib):
    """The filter works on data scaled by a power of two taken from the references, so magnitudes whose squares
    would overflow or underflow fp32 (1e+25, 1e-30 per component; found as a wrong answer at 1e+19 by
    tools/stress_sweep.py before the scaling existed) go through the fast path; TARGETS that dwarf the
    references by more than the fp32 range still overflow, the guard notices and the float64 kernels answer."""
    Yb = pca_like(600, 8, seed=63)
    Xb = pca_like(20, 8, seed=64)
    for sc in (1e25, 1e19, 1e-30, 1e-150):
        ix = gpu_lib.KnnIndex(600, 8, metric=0).set_ref(Yb * sc)
        gi, gd = ix.query(Xb * sc, 5)
        st = ix.last_stats()
        ix.close()
        oi, od = oracle.knn(Xb * sc, Yb * sc, 5, 0)
        _check(gi, gd, oi, od)
        assert st["fallback_rows"] == 0, sc
    Xh = Xb.copy()
    Xh[:7] *= 1e45                                            # scaled targets exceed fp32: inf / NaN scores
    ix = gpu_lib.KnnIndex(600, 8, metric=0).set_ref(Yb)
    gi, gd = ix.query(Xh, 5)
    st = ix.last_stats()
    ix.close()
    oi, od = oracle.knn(Xh, Yb, 5, 0)
    _check(gi, gd, oi, od)
    assert st["fallback_rows"] >= 7


def test_zero_target_rows_is_a_no_op(gpu_lib):
    Y = pca_like(100, 5, seed=65)
    gi, gd = gpu_lib.knn(np.empty((0, 5)), Y, 3)
    assert gi.shape == (0, 3) and gd.shape == (0, 3)


@pytest.mark.parametrize("m,n,g,k,drop", [(257, 4097, 50, 15, False), (1000, 1000, 15, 11, True),
                                          (2000, 30000, 30, 20, True), (130, 5000, 64, 30, False),
                                          (64, 3000, 100, 11, False)])
def test_both_filter_kernels_give_the_same_bits(gpu_lib, m, n, g, k, drop):
    """The Euclidean filter runs on the f16 matrix pipe -- the one-product first pass (l2c_topk.hip, the default where it
    is instantiated: g <= 125 and k + drop + 4 <= 64; rows it cannot certify go on to the seeded pass and the filter behind
    it), the f16x3 split as the first pass (NABO_L2_MODE=f16x3: l2q_topk.hip, g < 64 and k' <= 28) -- or on the fp32 MFMA
    (everything else, or NABO_L2_MODE=f32): the float64 refine + certification make all of them return exactly what the
    oracle does.  (The 32x32x16 kernels of rounds 1-2, l2h_topk.hip / l2s_topk.hip, live in the experiments build.)"""
    Y = pca_like(n, g, seed=1000 + n + g)
    X = Y[:m].copy() if drop else pca_like(m, g, seed=2000 + m + g)
    oi, od = oracle.knn(X, Y, k, 0, drop_first=drop, nthreads=8)
    kernels = {}
    for mode in ("f16x1", "f16x3", "f32", None):
        if mode:
            os.environ["NABO_L2_MODE"] = mode
        try:
            ix = gpu_lib.KnnIndex(n, g, metric=0).set_ref(Y)
            gi, gd = ix.query(X, k, drop_first=drop)
            st = ix.last_stats()
            kernels[mode] = ix.last_kernel()
            ix.close()
        finally:
            os.environ.pop("NABO_L2_MODE", None)
        _check(gi, gd, oi, od)
        assert st["fallback_rows"] == 0
    assert "l2_topk_kernel" in kernels["f32"]
    assert kernels[None] == kernels["f16x1"]                       # the one-product first pass is the default
    kk = k + drop
    h_ok = g < 64 and kk + 4 <= 32
    assert kernels["f16x3"].startswith("l2q_topk" if h_ok else "l2_topk"), kernels          # 16x16x32 MFMA shape
    # the one-product pass: g <= 125 and k' + 4 <= 64 (lists of 23 / 32 / 64 kept entries: geometries B / A / C of l2c_topk.hip)
    c_ok = g <= 125 and kk + 4 <= 64
    assert kernels["f16x1"].startswith("l2c_topk" if c_ok else "l2_topk") and ("one-product" in kernels["f16x1"]) == c_ok, kernels
    # the planner says the same without a device (nabo_query_plan)
    from nabo_amd import _knn
    assert _knn.query_plan(n, g, m, k, drop_first=drop)["kernel"] == kernels[None]
    assert _knn.query_plan(n, g, m, k, drop_first=drop, l2_mode="f16x3")["kernel"] == kernels["f16x3"]
    assert _knn.query_plan(n, g, m, k, drop_first=drop, l2_mode="f32")["kernel"] == kernels["f32"]


@pytest.mark.parametrize("mode,full_round", [("f32", 547), ("f16x3", 274)])
def test_tail_round_split_rows_are_exact(gpu_lib, mode, full_round):
    """More target workgroups than resident slots: the last, partially filled round is launched with its
    own reference split (api.hip "tail round").  Rows of BOTH launches must match the oracle."""
    m, n = 140000, 20000     # fp32 kernel: 547 workgroups of 256 rows on 512 slots; f16x3: 274 of 512 rows on 256 slots
    Y = pca_like(n, 50, seed=71)
    X = pca_like(m, 50, seed=72)
    os.environ["NABO_L2_MODE"] = mode
    try:
        ix = gpu_lib.KnnIndex(n, 50, metric=0).set_ref(Y)
        gi, gd = ix.query(X, 15)
        st = ix.last_stats()
        ix.close()
    finally:
        del os.environ["NABO_L2_MODE"]
    assert st["workgroups"] > full_round     # the tail really was split
    rows = np.concatenate([np.arange(0, 300), np.arange(131072 - 150, 131072 + 150), np.arange(m - 300, m)])
    oi, od = oracle.knn(X[rows], Y, 15, 0, nthreads=8)
    _check(gi[rows], gd[rows], oi, od)
    assert (np.diff(gd, axis=1) >= 0).all() and gi.min() >= 0 and gi.max() < n


@pytest.mark.parametrize("g,k", [(1, 1), (2, 3), (3, 1)])
def test_tiny_dimensionality(gpu_lib, g, k):
    Y = pca_like(700, g, seed=81)
    X = pca_like(90, g, seed=82)
    for metric in (0, 1):
        gi, gd = gpu_lib.knn(X, Y, k, metric=metric)
        oi, od = oracle.knn(X, Y, k, metric)
        _check(gi, gd, oi, od)


def test_canberra_filter_path_stress(gpu_lib):
    """The fp32 lower-bound filter + float64 refine + exact re-solve of uncertified rows must equal the
    exact kernel (NABO_CANBERRA_MODE=exact) and the oracle on adversarial inputs: dist_factor > 1,
    values near the window boundary, tiny and huge magnitudes, exact duplicates, plateau ties."""
    rng = np.random.default_rng(17)
    cases = []
    Y = pca_like(6000, 24, seed=91); X = pca_like(400, 24, seed=92)
    cases.append(("plain", X, Y, 0.25))
    cases.append(("f>1", X, Y, 2.5))
    cases.append(("f tiny", X, Y, 1e-3))
    Yb = Y.copy(); Xb = X.copy()
    Yb[:, :8] = Xb[rng.integers(0, 400, 6000), :8] * (1 + 0.25 * rng.choice([-1, 1], (6000, 8)) * (1 + rng.uniform(-1e-7, 1e-7, (6000, 8))))
    cases.append(("boundary", Xb, Yb, 0.25))             # |x-y| ~ f|x| to 1e-7: borderline window tests
    cases.append(("scaled 1e-30", X * 1e-30, Y * 1e-30, 0.25))
    cases.append(("scaled 1e+20", X * 1e20, Y * 1e20, 0.25))
    Yd = Y.copy(); Yd[100:160] = Yd[100]
    cases.append(("duplicates", np.vstack([Yd[100:102], X[:50]]), Yd, 0.25))
    cases.append(("far apart", rng.standard_normal((80, 24)) * 0.01, rng.standard_normal((3000, 24)) * 100, 0.25))
    Xz = X.copy(); Xz[:, ::3] = 0.0
    cases.append(("zeros in x", Xz, Y, 0.25))
    # the packed-f16 counting pass: per-dimension power-of-two scales, half-precision window edges, f16 range
    col = 10.0 ** (np.arange(24) - 12.0)
    cases.append(("column scales 1e-12..1e11", X * col, Y * col, 0.25))
    Yo = Y.copy(); Yo[np.arange(24) * 7, np.arange(24)] *= 1e6          # one huge outlier per column: the rest of the
    cases.append(("column outliers", X, Yo, 0.25))                      # column lands in the f16 denormal range
    cases.append(("targets far outside the reference range", X * 3e4, Y, 0.25))     # x' overflows f16: never counted
    Yh = Y.copy()
    Yh[:, :12] = X[rng.integers(0, 400, 6000), :12] * (1 + 0.25 * rng.choice([-1, 1], (6000, 12)) * (1 + rng.uniform(-2e-3, 2e-3, (6000, 12))))
    cases.append(("edges at f16 resolution", X, Yh, 0.25))
    cases.append(("negated references", X, -Y, 0.25))
    cases.append(("f = 1", X, Y, 1.0))
    for name, Xc, Yc, f in cases:
        gi, gd = gpu_lib.knn(Xc, Yc, 15, metric=1, dist_factor=f)
        oi, od = oracle.knn(Xc, Yc, 15, 1, f, nthreads=8)
        assert np.array_equal(gi, oi), name
        assert np.array_equal(gd, od), name
    # magnitudes beyond fp32: the pack kernels flag it and the exact kernel answers
    gi, gd = gpu_lib.knn(X * 1e60, Y * 1e60, 7, metric=1)
    oi, od = oracle.knn(X * 1e60, Y * 1e60, 7, 1, 0.25, nthreads=8)
    _check(gi, gd, oi, od)


def test_canberra_filter_equals_exact_kernel(gpu_lib):
    Y = pca_like(20000, 50, seed=93); X = pca_like(3000, 50, seed=94)
    mask = np.zeros(20000, np.uint8); mask[::7] = 1
    a = gpu_lib.knn(X, Y, 20, metric=1, dist_factor=0.25, ref_mask=mask, drop_first=True)
    os.environ["NABO_CANBERRA_MODE"] = "exact"
    try:
        b = gpu_lib.knn(X, Y, 20, metric=1, dist_factor=0.25, ref_mask=mask, drop_first=True)
    finally:
        del os.environ["NABO_CANBERRA_MODE"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ---- cosine metric: an EXTENSION (BASELINE.json configs[4]); no reference counterpart, so parity is
# against this build's own oracle definition only ("parity unpinned") --------------------
def test_cosine_pairwise_bit_exact_vs_oracle(gpu_lib):
    rng = np.random.default_rng(9)
    for g in (1, 7, 50, 100):
        x = rng.standard_normal((70, g)) * np.exp(rng.uniform(-8, 8, (70, 1)))
        y = rng.standard_normal((90, g)) * np.exp(rng.uniform(-8, 8, (90, 1)))
        y[5] = 0.0
        x[3] = 0.0
        y[7] = x[11] * 3.0                  # parallel rows: distance ~0 (may round to +-1 ulp of 0)
        assert np.array_equal(gpu_lib.pairwise(x, y, 2), oracle.pairwise(x, y, 2))


@pytest.mark.parametrize("m,n,g,k,drop", [(1, 40, 5, 3, False), (100, 1000, 30, 11, True), (257, 4097, 50, 15, False),
                                          (64, 20000, 50, 15, False), (300, 6000, 100, 50, True),
                                          (200, 3000, 128, 30, False)])
def test_cosine_knn_vs_oracle(gpu_lib, m, n, g, k, drop):
    Y = pca_like(n, g, seed=3000 + n + g)
    X = Y[:m].copy() if drop else pca_like(m, g, seed=4000 + m + g)
    gi, gd = gpu_lib.knn(X, Y, k, metric=2, drop_first=drop)
    oi, od = oracle.knn(X, Y, k, 2, drop_first=drop, nthreads=8)
    _check(gi, gd, oi, od)


def test_cosine_is_scale_free_and_handles_zero_rows_masks_and_duplicates(gpu_lib):
    rng = np.random.default_rng(21)
    n, m, g, k = 5000, 300, 40, 15
    Y = pca_like(n, g, seed=77) * np.exp(rng.uniform(-6, 6, (n, 1)))      # wildly different row lengths
    X = pca_like(m, g, seed=78) * np.exp(rng.uniform(-6, 6, (m, 1)))
    Y[10] = 0.0
    Y[11] = 0.0
    X[5] = 0.0                               # zero target: every distance is 1 -> index order, exact path
    Y[200:216] = Y[100] * np.arange(1, 17)[:, None]       # 16 parallel copies: exact ties in angle
    X[7] = Y[100] * 0.5
    mask = np.zeros(n, dtype=np.uint8)
    mask[::5] = 1
    ix = gpu_lib.KnnIndex(n, g, metric=2).set_ref(Y, ref_mask=mask)
    gi, gd = ix.query(X, k)
    st = ix.last_stats()
    ix.close()
    oi, od = oracle.knn(X, Y, k, 2, ref_mask=mask, nthreads=8)
    _check(gi, gd, oi, od)
    assert st["fallback_rows"] >= 1          # the zero row cannot be certified by the filter
    assert st["fallback_rows"] < m // 4      # ... but ordinary rows are


def test_cosine_config5_shape_sampled(gpu_lib):
    """BASELINE configs[4] shape (d=100, k=50, cosine) at 200k references; a row sample against the oracle."""
    n, g, k = 200000, 100, 50
    Y = pca_like(n, g, seed=5005)
    ix = gpu_lib.KnnIndex(n, g, metric=2).set_ref(Y)
    gi, gd = ix.query(Y[:20000], k, drop_first=True)
    st = ix.last_stats()
    ix.close()
    rows = np.random.default_rng(1).choice(20000, 64, replace=False)
    oi, od = oracle.knn(Y[rows], Y, k, 2, drop_first=True, nthreads=8)
    _check(gi[rows], gd[rows], oi, od)
    assert st["fallback_rows"] < 200


# ---- permutation null of the mapping score: EXTENSION (no reference counterpart; own oracle) --------------------
def _null_case(n_ref, n_t, k, seed):
    rng = np.random.default_rng(seed)
    edge_t = np.repeat(np.arange(n_t), k)
    edge_r = rng.integers(0, n_ref, n_t * k)
    edge_r[:5 * k] = 3                                    # a heavily hit reference node
    w = rng.choice(np.round(np.arange(1, 11) / (20.0 - np.arange(1, 11)), 2), n_t * k)
    group = (rng.random(n_t) < 0.3).astype(np.uint8)
    group[:5] = 1
    return edge_t, edge_r, w, group


@pytest.mark.parametrize("n_ref,n_t,k,P,bits", [(40, 200, 5, 64, 64), (300, 1500, 7, 257, 64), (50, 400, 4, 1000, 8),
                                                (17, 90, 3, 31, 16)])
def test_mapping_score_permutation_null_vs_oracle(gpu_lib, n_ref, n_t, k, P, bits):
    import nabo_amd
    from oracle import oracle as orc
    edge_t, edge_r, w, group = _null_case(n_ref, n_t, k, 7 + n_ref)
    res = nabo_amd.mapping_score_null(edge_t, edge_r, w, group, n_ref, n_perm=P, seed=12345, key_bits=bits)
    ref = orc.score_null(edge_t, edge_r, w, group, n_ref, P, seed=12345, key_bits=bits)
    assert np.array_equal(res["sizes"], ref["sizes"])                     # thresholds / ties: integer-exact
    if bits == 8:
        assert (ref["sizes"] > int(group.sum())).any()                    # 8-bit keys tie at the threshold
    assert np.array_equal(res["obs"], ref["obs"])                         # same float64 operations, same order
    assert np.array_equal(res["n_ge"], ref["n_ge"])
    assert np.allclose(res["null_mean"], ref["null_mean"], rtol=1e-12, atol=1e-12)    # reduction order differs
    assert np.allclose(res["null_sd"], ref["null_sd"], rtol=1e-9, atol=1e-9)
    # the observed score is the reference's mapping score of the sample of interest (nabo/_graph.py:644-653)
    keep = group[edge_t] != 0
    sc = nabo_amd.mapping_score_from_edges(n_ref, edge_r[keep], w[keep], int(group.sum()))
    assert np.allclose(res["obs"], sc, rtol=1e-13, atol=0)
    assert res["pvalue"].min() >= 1.0 / (P + 1) and res["pvalue"].max() <= 1.0


def test_mapping_score_null_rejects_bad_input(gpu_lib):
    import nabo_amd
    edge_t, edge_r, w, group = _null_case(10, 30, 2, 1)
    with pytest.raises(ValueError):
        nabo_amd.mapping_score_null(edge_t, edge_r, w, np.zeros(30), 10, n_perm=8)          # empty group
    with pytest.raises(ValueError):
        nabo_amd.mapping_score_null(edge_t, edge_r + 100, w, group, 10, n_perm=8)           # ref index out of range
    with pytest.raises(ValueError):
        nabo_amd.mapping_score_null(edge_t, edge_r, w, group, 10, n_perm=5000)               # > 4096 permutations


def test_mapping_score_null_edge_list_in_any_order(gpu_lib):
    """The device-side CSR build (csr_build.hip) is a STABLE sort by reference node: a shuffled edge list, long
    enough for several radix-sort blocks, gives the oracle's float64 sums (which follow the listed order inside a
    row), and the CSR entry point fed with a host-side stable sort gives the same bits."""
    import ctypes as C
    import nabo_amd
    from nabo_amd import _lib
    from oracle import oracle as orc
    n_ref, n_t, k, P = 3000, 40000, 6, 96
    edge_t, edge_r, w, group = _null_case(n_ref, n_t, k, 99)
    rng = np.random.default_rng(4)
    sh = rng.permutation(edge_t.shape[0])
    edge_t, edge_r, w = edge_t[sh], edge_r[sh], w[sh]
    res = nabo_amd.mapping_score_null(edge_t, edge_r, w, group, n_ref, n_perm=P, seed=77)
    ref = orc.score_null(edge_t, edge_r, w, group, n_ref, P, seed=77)
    assert np.array_equal(res["obs"], ref["obs"]) and np.array_equal(res["n_ge"], ref["n_ge"])
    assert np.array_equal(res["sizes"], ref["sizes"])
    # CSR entry point (nabo_score_null) on the host-sorted edges
    order = np.argsort(edge_r, kind="stable")
    rp = np.zeros(n_ref + 1, dtype=np.int64)
    np.cumsum(np.bincount(edge_r, minlength=n_ref), out=rp[1:])
    et, ew = np.ascontiguousarray(edge_t[order], dtype=np.int64), np.ascontiguousarray(w[order])
    obs, mean, sd = np.empty(n_ref), np.empty(n_ref), np.empty(n_ref)
    nge, sizes = np.empty(n_ref, dtype=np.int64), np.empty(P, dtype=np.int64)
    L = _lib.lib()
    L.nabo_score_null.argtypes = [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_int32, C.c_uint64, C.c_int32, C.c_double] + [C.c_void_p] * 5
    _lib.check(L.nabo_score_null(0, n_ref, rp.ctypes.data, et.ctypes.data, ew.ctypes.data, n_t, group.ctypes.data, P, 77,
                                 64, 1000.0, obs.ctypes.data, nge.ctypes.data, mean.ctypes.data, sd.ctypes.data,
                                 sizes.ctypes.data))
    assert np.array_equal(obs, res["obs"]) and np.array_equal(nge, res["n_ge"])
    assert np.array_equal(mean, res["null_mean"]) and np.array_equal(sd, res["null_sd"])
    # no edges at all: every score is 0, every permutation ties with it
    z = nabo_amd.mapping_score_null(edge_t[:0], edge_r[:0], w[:0], group, n_ref, n_perm=8)
    assert not z["obs"].any() and (z["n_ge"] == 8).all()
    with pytest.raises(ValueError):
        nabo_amd.mapping_score_null(edge_t + n_t, edge_r, w, group, n_ref, n_perm=8)          # target out of range


# ---- randomized sweep: many small shapes, every metric, masks, duplicates, drop_first ---------------------------
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_random_small_shapes_sweep(gpu_lib, metric):
    rng = np.random.default_rng(1000 + metric)
    for case in range(60):
        n = int(rng.choice([1, 2, 3, 5, 17, 31, 32, 33, 63, 64, 65, 100, 257, 700, 1500]))
        m = int(rng.choice([1, 2, 7, 31, 32, 33, 64, 129, 300]))
        g = int(rng.choice([1, 2, 3, 8, 15, 16, 17, 30, 49, 50, 64, 65, 100, 128]))
        drop = bool(rng.integers(0, 2)) and n >= 2
        kmax = min(n - (1 if drop else 0), 55)
        if kmax < 1:
            continue
        k = int(rng.integers(1, kmax + 1))
        Y = pca_like(n, g, seed=int(rng.integers(1, 1 << 30)))
        if drop and m <= n:
            X = Y[:m].copy()
        else:
            X = pca_like(m, g, seed=int(rng.integers(1, 1 << 30)))
            drop = False
        mask = None
        flavour = int(rng.integers(0, 5))
        if flavour == 1 and n > 4:                       # ignored references (some rows run out of valid ones)
            mask = (rng.random(n) < rng.choice([0.1, 0.5, 0.9])).astype(np.uint8)
            if mask.all():
                mask[0] = 0
        elif flavour == 2 and n > 8:                     # exact duplicates among the references
            Y[rng.integers(0, n, n // 2)] = Y[rng.integers(0, n)]
        elif flavour == 3:                               # coarse grid: many exact ties
            Y = np.round(Y)
            X = np.round(X)
        gi, gd = gpu_lib.knn(X, Y, k, metric=metric, dist_factor=0.25, ref_mask=mask, drop_first=drop)
        oi, od = oracle.knn(X, Y, k, metric, 0.25, ref_mask=mask, drop_first=drop, nthreads=8)
        ok = np.array_equal(gi, oi) and np.array_equal(gd, od, equal_nan=True)
        assert ok, "case %d: m=%d n=%d g=%d k=%d drop=%s flavour=%d metric=%d" % (case, m, n, g, k, drop, flavour, metric)


def test_tie_heavy_rows_take_the_wide_second_chance(gpu_lib):
    """References on the 4-D integer lattice: around a site the distance shells hold 8, 24, 32 ... points, so the
    15th AND the 24th neighbour sit in the same shell (no certificate from 32-entry lists) while the 64th lies
    in the next one.  Such rows are re-filtered with 64-entry lists before anything is brute-forced; the answer
    is the oracle's either way.  (One reference split, so that the lists are not S times longer.)"""
    rng = np.random.default_rng(8)
    side = 9
    Y = np.stack(np.meshgrid(*[np.arange(side)] * 4, indexing="ij"), -1).reshape(-1, 4).astype(np.float64)   # 6561 points
    interior = np.nonzero(((Y >= 2) & (Y <= side - 3)).all(axis=1))[0]
    X = Y[rng.choice(interior, 600, replace=False)] + 0.0                                                     # on lattice sites
    k = 15
    oi, od = oracle.knn(X, Y, k, 0, nthreads=8)
    ix = gpu_lib.KnnIndex(len(Y), 4, metric=0, options={"splits": 1}).set_ref(Y)
    gi, gd = ix.query(X, k)
    with_retry = ix.last_stats()["fallback_rows"]
    ix.set_option("wide_retry", 0)
    hi, hd = ix.query(X, k)
    without = ix.last_stats()["fallback_rows"]
    ix.close()
    _check(gi, gd, oi, od)
    _check(hi, hd, oi, od)
    assert without >= 16                 # the situation really arises on this input
    assert with_retry < without          # ... and the second chance certifies part of it without brute force


def test_canberra_plateau_with_window_edge_candidates(gpu_lib):
    """Regression (found by tools/stress_sweep.py): references whose every dimension lies EXACTLY on the window
    edge are at distance g like the all-out ones, but the filter cannot prove them out, so they enter the
    candidate list with a key below the plateau whatever their index.  The plateau certificate must not accept
    a row whose k'-th entry could be preceded by a dropped (all-out, lower-index) reference."""
    n, k = 200, 15
    Y = np.stack([100.0 + np.arange(n), -50.0 - np.arange(n)], 1).astype(np.float64)      # all-out for the target
    Y[150:175] = [3.0, 3.0]                                                                # |x-y| == 0.25*|x| in both dims
    X = np.array([[4.0, 4.0], [4.0, 4.0], [8.0, -8.0]])
    gi, gd = gpu_lib.knn(X, Y, k, metric=1, dist_factor=0.25)
    oi, od = oracle.knn(X, Y, k, 1, 0.25)
    assert (od[0] == 2.0).all() and list(oi[0]) == list(range(15))
    _check(gi, gd, oi, od)


@pytest.mark.parametrize("metric,m,n,g,k,drop", [(0, 150, 3000, 30, 100, True), (0, 120, 2500, 200, 15, False),
                                                  (1, 100, 2000, 20, 100, False), (2, 90, 2000, 200, 60, True),
                                                  (1, 60, 1500, 200, 11, False), (0, 70, 900, 129, 57, True)])
def test_no_cliff_beyond_the_filter_kernels_limits(gpu_lib, metric, m, n, g, k, drop):
    """The reference accepts any k and use_comps (nabo/_mapping.py:495-524).  Past the instantiated filter kernels
    (k + drop_first > 56, g > 128) every row is answered by the exact float64 kernels: same results as the oracle."""
    Y = pca_like(n, g, seed=3000 + n + g)
    X = Y[:m].copy() if drop else pca_like(m, g, seed=4000 + m + g)
    ix = gpu_lib.KnnIndex(n, g, metric=metric, dist_factor=0.25).set_ref(Y)
    gi, gd = ix.query(X, k, drop_first=drop)
    st = ix.last_stats()
    ix.close()
    oi, od = oracle.knn(X, Y, k, metric, 0.25, drop_first=drop, nthreads=8)
    _check(gi, gd, oi, od)
    if k + drop > 56 or (g > 128 and metric != 1):
        assert st["fallback_rows"] == m


def test_no_cliff_masked_tail_longer_than_any_list(gpu_lib):
    """k larger than the un-ignored references: the order row continues with ALL ignored references by ascending
    index (numpy.ma's NaN-fill, nabo/_mapping.py:135-144) -- also past the 56 entries the filter path can emit."""
    Y = pca_like(150, 12, seed=81)
    X = pca_like(40, 12, seed=82)
    mask = np.zeros(150, np.uint8)
    mask[np.random.default_rng(2).choice(150, 70, replace=False)] = 1
    for metric in (0, 1):
        gi, gd = gpu_lib.knn(X, Y, 140, metric=metric, ref_mask=mask)
        oi, od = oracle.knn(X, Y, 140, metric, ref_mask=mask)
        _check(gi, gd, oi, od)
        assert np.array_equal(gi[:, 80:], np.tile(np.nonzero(mask)[0][:60], (40, 1)))


@pytest.mark.parametrize("splits", ["1", "3"])
def test_partly_overflowing_targets_never_reach_the_filter_as_nan(gpu_lib, splits):
    """The fp32 score kernel is compiled with -fno-honor-nans, so NaN must not occur in it: targets whose scaled
    components leave the safe range (here only SOME components of SOME rows) are zeroed at pack time and carry a NaN
    norm, which refine.hip can never certify -- the exact kernels answer those rows, the others stay on the fast
    path; with reference splits (several lists per row) and in candidate (shard) mode the same holds."""
    n, g, k = 4000, 24, 9
    Y = pca_like(n, g, seed=91)
    X = pca_like(300, g, seed=92)
    big = np.array([3, 64, 65, 127, 128, 255, 299])
    X[big[:3], 5] *= 1e42                    # one component
    X[big[3:5], ::2] *= -1e45                # every other component, negative
    X[big[5:], :] *= 1e60                    # the whole row
    ix = gpu_lib.KnnIndex(n, g, metric=0, options={"splits": int(splits)} if splits else None).set_ref(Y)
    gi, gd = ix.query(X, k)
    st = ix.last_stats()
    # shard mode: the bound of such a row must be -inf (unknown), never a number
    dx = _knn_buf(gpu_lib, X)
    di, dd, db = (_knn_mod().DeviceBuffer(300 * 12 * 8), _knn_mod().DeviceBuffer(300 * 12 * 8),
                  _knn_mod().DeviceBuffer(300 * 8))
    ix.query_candidates_device(dx.ptr, 300, 12, di.ptr, dd.ptr, db.ptr)
    cb = db.download((300,), np.float64)
    ix.close()
    oi, od = oracle.knn(X, Y, k, 0, nthreads=8)
    _check(gi, gd, oi, od)
    assert st["fallback_rows"] == big.size
    assert np.isneginf(cb[big]).all() and np.isfinite(np.delete(cb, big)).all()


def _knn_mod():
    from nabo_amd import _knn
    return _knn


def _knn_buf(gpu_lib, arr):
    return _knn_mod().DeviceBuffer(arr.nbytes).upload(arr)


def _offset_clusters(n, g, seed, offset):
    """Tight clusters far from the origin of the centred data: ||x|| ||y|| is large next to the neighbour distances, which is
    where the one-product pass's bound (2^-9 ||x|| ||y|| below the real score) is weakest."""
    centres = np.random.default_rng(5).standard_normal((6, g)) * offset
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, 6, size=n)
    return centres[lab] + rng.standard_normal((n, g)) * 0.5


@pytest.mark.parametrize("geo", ["a", "b"])
@pytest.mark.parametrize("offset", [0.5, 10.0, 40.0, 400.0])
def test_one_product_pass_chain_equals_the_oracle(gpu_lib, geo, offset):
    """The default Euclidean filter starts with the one-product pass (l2c_topk.hip, geometry A = one wave per SIMD / 32-entry
    lists, B = two waves per SIMD / 23-entry lists); rows its lower bound cannot certify go through the SEEDED one-product pass,
    then the f16x3 pass, the 64-entry lists and the exact kernels.  The further the clusters sit from the origin the more rows
    travel down that chain -- the answer is the oracle's bits wherever a row ends up."""
    n, m, g, k = 30000, 3000, 40, 15
    Y = _offset_clusters(n, g, 11, offset)
    X = _offset_clusters(m, g, 12, offset)
    oi, od = oracle.knn(X, Y, k, 0, nthreads=8)
    ix = gpu_lib.KnnIndex(n, g, metric=0, options={"l2c_geo": {"a": 0, "b": 1}[geo]}).set_ref(Y)
    gi, gd = ix.query(X, k)
    st, kern = ix.last_stats(), ix.last_kernel()
    ix.set_option("seeded_pass", 0)              # the chain without its seeded link ...
    ix.set_option("coarse_adapt", 0)             # ... and with the first pass even where the first query found it weak
    hi, hd = ix.query(X, k)
    st2 = ix.last_stats()
    ix.close()
    _check(gi, gd, oi, od)
    _check(hi, hd, oi, od)
    assert kern.startswith("l2c_topk_kernel<2,1,%s>" % ("33,8,64,4,1" if geo == "a" else "23,6,32,4,2")), kern
    assert st2["seeded_pass_rows"] == 0 and st2["second_pass_rows"] == st["seeded_pass_rows"]
    if offset <= 0.5:
        assert st["seeded_pass_rows"] < m // 10          # well-centred data: the first pass answers (nearly) everything
    if offset >= 10.0:
        assert st["seeded_pass_rows"] > 0                # the weak bound really sends rows on ...
    if offset <= 40.0:
        assert st["fallback_rows"] <= m // 20            # ... and the filter passes, not the exact kernels, answer them
    # (offset 400: clusters 800 noise widths from the centre of the data -- even the fp32 filter's error bound exceeds the
    # neighbour gaps, and the exact kernels answer; a global centre cannot serve such data, the result is right anyway)


def test_seeded_pass_serves_most_of_what_the_first_pass_leaves(gpu_lib):
    """Lists without slack (option lkeep = k': the threshold IS the k'-th kept one-product score, which lies below the k'-th exact
    distance) make the first pass fail nearly every row; the seeded pass -- every row starts from the threshold its failed
    certificate implies -- answers them without the f16x3 pass."""
    n, m, g, k = 60000, 5000, 50, 15
    Y = pca_like(n, g, seed=71)
    X = pca_like(m, g, seed=72)
    oi, od = oracle.knn(X, Y, k, 0, nthreads=8)
    # (one reference split: S splits would hand refine S lists per row)
    ix = gpu_lib.KnnIndex(n, g, metric=0, options={"lkeep": k, "splits": 1}).set_ref(Y)
    gi, gd = ix.query(X, k)
    st = ix.last_stats()
    ix.close()
    _check(gi, gd, oi, od)
    assert st["seeded_pass_rows"] > m // 10, st
    assert st["second_pass_rows"] <= st["seeded_pass_rows"] // 20, st
    assert st["fallback_rows"] == 0, st


def test_small_query_pads_cost_no_list_work(gpu_lib):
    """One row is still a whole workgroup of the filter: the padding rows start from threshold -inf (l2c_topk.hip), so the
    kernel time of a 1-row query stays a fraction of a 512-row one over the same references."""
    n, g, k = 200000, 50, 15
    Y = pca_like(n, g, seed=81)
    ix = gpu_lib.KnnIndex(n, g, metric=0).set_ref(Y)
    X = pca_like(512, g, seed=82)
    oi, od = oracle.knn(X[:1], Y, k, 0, nthreads=8)
    ix.query(X, k)
    t512 = ix.last_stats()["ms_topk"]
    best1 = 1e9
    for _ in range(3):
        gi, gd = ix.query(X[:1], k)
        best1 = min(best1, ix.last_stats()["ms_topk"])
    ix.close()
    _check(gi, gd, oi, od)
    assert best1 < 0.6 * t512, (best1, t512)


@pytest.mark.parametrize("m,mode", [(300, ""), (70000, ""), (300, "f16x3"), (300, "f32")])
def test_large_reference_sets_take_more_splits(gpu_lib, monkeypatch, m, mode):
    """A list entry holds the reference as a 25-bit offset into its split (topk_lists.h: the entry is a double -- key,
    slot | offset -- so that a rescan is one v_max_f64 per entry); api.hip raises the split count of larger sets.  With the
    bound lowered to 5000 references per split, 60 000 references need 13 splits where the planner would take fewer (one
    for 70 000 rows, main and tail launch): same neighbours, same distances, and entries of every split carry the split's
    first reference back (indices beyond 2^25 / 5000 come out right)."""
    n, g, k = 60000, 50, 15
    Y = pca_like(n, g, seed=91)
    X = pca_like(m, g, seed=92)
    rows = np.arange(m) if m <= 2000 else np.random.default_rng(3).choice(m, 600, replace=False)
    oi, od = oracle.knn(X[rows], Y, k, 0, nthreads=8)
    if mode:
        monkeypatch.setenv("NABO_L2_MODE", mode)
    ix = gpu_lib.KnnIndex(n, g, metric=0, options={"split_refs_max": 5000}).set_ref(Y)
    gi, gd = ix.query(X, k)
    st = ix.last_stats()
    ix.close()
    assert st["splits"] >= 13, st
    _check(gi[rows], gd[rows], oi, od)
    assert st["fallback_rows"] == 0, st


@pytest.mark.parametrize("m,n,g,k,drop,metric", [
    (3000, 40000, 100, 50, False, 2),        # BASELINE configs[4]'s metric / shape: cosine, d = 100, k = 50 -> four steps, 64-entry lists
    (2000, 30000, 100, 50, True, 0),
    (1500, 20000, 64, 30, False, 0),         # first g without an f16x3 kernel, lists of 38
    (2500, 50000, 30, 40, False, 0),         # g < 64 with k' > 28: 64-entry lists on two steps (the hand-scheduled statements)
    (700, 9000, 125, 15, False, 0),          # the last g the one-product operands are instantiated for (128 slots)
    (700, 9000, 126, 15, False, 0),          # ... and the first one beyond: fp32 filter
    (900, 12000, 90, 56, False, 0),          # k' = 56: the longest lists (64 kept entries)
    (900, 12000, 90, 56, True, 0),           # k' = 57: beyond them, the exact kernels
])
def test_one_product_pass_wide_lists_and_many_components(gpu_lib, m, n, g, k, drop, metric):
    """The one-product pass serves every g <= 125 (g + 3 operand slots in steps of 32) and every k' <= 56 (geometry C of
    l2c_topk.hip: 64-entry lists); behind it sit the seeded pass and, where no f16x3 kernel exists (g >= 64 or k' > 28), the
    fp32-MFMA filter.  Same bits as the oracle, and as the fp32 filter run alone."""
    Y = pca_like(n, g, seed=4100 + n + g)
    X = Y[:m].copy() if drop else pca_like(m, g, seed=4200 + m + g)
    oi, od = oracle.knn(X, Y, k, metric, 0.25, drop_first=drop, nthreads=8)
    ix = gpu_lib.KnnIndex(n, g, metric=metric).set_ref(Y)
    gi, gd = ix.query(X, k, drop_first=drop)
    st, kern = ix.last_stats(), ix.last_kernel()
    ix.close()
    _check(gi, gd, oi, od)
    kk = k + (1 if drop else 0)
    if kk > 56:                                          # NABO_MAX_K: the exact float64 kernels answer every row
        assert kern.startswith("exact_dist_rows_kernel"), kern
        return
    if g <= 125:
        ks = (g + 3 + 31) // 32
        geo = "1,23,6,32,4,2" if (kk + 8 <= 23 and ks <= 2) else ("1,33,8,64,4,1" if kk <= 24 else "2,65,4,64,4,1")
        assert kern.startswith("l2c_topk_kernel<%d,%s>" % (ks, geo)), kern
    else:
        assert kern.startswith("l2_topk_kernel"), kern
    assert st["fallback_rows"] <= m // 50, st


def test_weak_one_product_bound_is_remembered_until_the_references_change(gpu_lib):
    """Tight clusters far from the centre of the data: the one-product passes leave (nearly) every row to the f16x3 pass.  The
    index notices (more than a quarter of >= 1024 rows) and starts the NEXT queries with the f16x3 filter; set_ref starts over.
    Same bits every time."""
    n, m, g, k = 30000, 3000, 40, 15
    Y, X = _offset_clusters(n, g, 21, 40.0), _offset_clusters(m, g, 22, 40.0)
    oi, od = oracle.knn(X, Y, k, 0, nthreads=8)
    ix = gpu_lib.KnnIndex(n, g, metric=0).set_ref(Y)
    kernels, second = [], []
    for it in range(3):
        if it == 2:
            ix.set_ref(Y)
        gi, gd = ix.query(X, k)
        _check(gi, gd, oi, od)
        kernels.append(ix.last_kernel())
        second.append(ix.last_stats()["second_pass_rows"])
    ix.close()
    assert kernels[0].startswith("l2c_topk") and second[0] > m // 4, (kernels, second)
    assert kernels[1].startswith("l2q_topk") and second[1] == 0, (kernels, second)
    assert kernels[2].startswith("l2c_topk"), kernels


def test_create_and_destroy_many_indices_returns_the_device_memory(gpu_lib):
    """nabo_index_destroy frees EVERY buffer of an index (round-3 advisory: the release list missed the one-product tiles,
    128 MB at 1M references -- every one-shot nabo_knn call, Mapping call and bench layout leaked them): free device memory
    after 24 create / set_ref / query / destroy cycles over all three metrics (and every pass of the Euclidean chain: the
    tight far-away cluster sends rows through the seeded pass and the pass behind it) is back at the baseline."""
    from nabo_amd import _lib
    rng = np.random.default_rng(77)
    Y = pca_like(60000, 50, seed=31)
    Y[:20000] = 400.0 + 1e-3 * rng.standard_normal((20000, 50))       # weak one-product bound: later passes allocate too
    X = np.concatenate([pca_like(3000, 50, seed=32), Y[:3000] + 1e-4])
    ix = gpu_lib.KnnIndex(60000, 50, metric=0).set_ref(Y)              # warm-up: code objects, allocator pools
    ix.query(X, 15)
    ix.close()
    _lib.check(_lib.lib().nabo_dev_synchronize(0))
    free0, total = _lib.mem_info(0)
    for it in range(24):
        metric = it % 3
        ix = gpu_lib.KnnIndex(60000, 50, metric=metric, dist_factor=0.25).set_ref(Y)
        ix.query(X, 15)
        ix.close()
        gpu_lib.knn(X[:64], Y, 5, metric=metric)                       # the one-shot entry point creates and destroys its own
    _lib.check(_lib.lib().nabo_dev_synchronize(0))
    free1, _ = _lib.mem_info(0)
    assert free0 - free1 < 32 << 20, "device memory not returned: %.1f MB gone after 24 cycles" % ((free0 - free1) / 2 ** 20)


def test_row_pass_record_names_the_pass_that_answered_each_row(gpu_lib, monkeypatch):
    """nabo_index_last_row_pass: the chain test's data (a tight cluster far from the centre: the one-product bound fails
    there) -- the record's counts equal nabo_index_last_passes, every pass of the chain occurs, all rows equal the oracle;
    pinning the f16x3 filter as the first pass renames every row; the Canberra record separates filter and exact rows."""
    rng = np.random.default_rng(5)
    n, g, k = 40000, 50, 15
    Y = pca_like(n, g, seed=41)
    Y[:8000] = 300.0 + 2e-3 * rng.standard_normal((8000, g))
    X = np.concatenate([pca_like(1500, g, seed=42), Y[:1500] + 1e-5 * rng.standard_normal((1500, g))])
    oi, od = oracle.knn(X, Y, k, 0, nthreads=8)
    # (merge_lists = 0: the seeded pass re-evaluates ALL its S x 32 list entries -- on this weak-bound data a seed has several
    # hundred references below it -- and answers some rows itself; by default it keeps the 128 best-scored ones, what does not
    # fit goes on to the f16x3 pass: the same results, fewer rows named 1)
    seen = {}
    for opts in ({"merge_lists": 0}, {}):
        ix = gpu_lib.KnnIndex(n, g, metric=0, options=opts).set_ref(Y)
        gi, gd = ix.query(X, k)
        st, rp = ix.last_stats(), ix.last_row_pass(X.shape[0])
        with pytest.raises(ValueError):
            ix.last_row_pass(X.shape[0] + 1)
        ix.close()
        _check(gi, gd, oi, od)
        assert int((rp >= 1).sum()) == st["seeded_pass_rows"] > 0 and int((rp >= 2).sum()) == st["second_pass_rows"]
        assert int((rp == 4).sum()) == st["fallback_rows"]
        seen[len(opts)] = set(np.unique(rp).tolist())
    assert {1, 2} <= seen[1], "the far-away tight cluster sends rows past the first pass and the seeded one"
    assert 2 in seen[0] and 1 not in seen[0] or {1, 2} <= seen[0]      # (on this data nothing is certified by the first pass)
    monkeypatch.setenv("NABO_L2_MODE", "f16x3")
    ix = gpu_lib.KnnIndex(n, g, metric=0).set_ref(Y)
    monkeypatch.delenv("NABO_L2_MODE")
    hi, hd = ix.query(X, k)
    rp2 = ix.last_row_pass(X.shape[0])
    ix.close()
    assert np.array_equal(hi, gi) and np.array_equal(hd, gd) and (rp2 >= 2).all()
    # modified Canberra: duplicated references make exact ties the filter cannot certify
    Yc = pca_like(3000, 20, seed=43)
    Yc[1500:1540] = Yc[100]
    Xc = np.concatenate([pca_like(200, 20, seed=44), Yc[100:101]])
    ix = gpu_lib.KnnIndex(3000, 20, metric=1, dist_factor=0.25).set_ref(Yc)
    ci, cd = ix.query(Xc, 11)
    st, rpc = ix.last_stats(), ix.last_row_pass(Xc.shape[0])
    ix.close()
    oi, od = oracle.knn(Xc, Yc, 11, 1, 0.25, nthreads=8)
    _check(ci, cd, oi, od)
    assert set(np.unique(rpc)) <= {4, 5} and int((rpc == 4).sum()) == st["fallback_rows"]


@pytest.mark.parametrize("m,n,g,k,drop,metric,splits", [(3000, 60000, 50, 15, 0, 0, "0"), (700, 200000, 50, 15, 0, 0, "0"),
                                                       (2500, 90000, 30, 40, 1, 0, "0"), (1500, 70000, 100, 50, 0, 2, "0"),
                                                       (900, 50000, 50, 15, 0, 0, "3"), (1200, 40000, 80, 11, 1, 2, "2")])
def test_tournament_seeds_change_no_result(gpu_lib, m, n, g, k, drop, metric, splits):
    """l2c_pre_kernel (the one-product pass starts every list from an upper bound of its lkeep-th smallest score among the
    split's first references instead of +inf): results with it, with four times the planned tournament and without it are
    the same bits as the oracle's -- all three geometries, one to four operand steps, reference splits, the tail launch,
    cosine.  With seeds the first pass must not certify fewer rows than the margin its shorter warm-up can explain."""
    Y = pca_like(n, g, seed=51)
    X = pca_like(m, g, seed=52) if not drop else Y[:m]
    oi, od = oracle.knn(X, Y, k, metric, drop_first=bool(drop), nthreads=16)
    res = {}
    for pre in ("100", "400", "0"):
        ix = gpu_lib.KnnIndex(n, g, metric=metric, options={"splits": int(splits), "prepass": int(pre)}).set_ref(Y)
        gi, gd = ix.query(X, k, drop_first=bool(drop))
        res[pre] = ix.last_stats()
        assert "l2c_topk_kernel" in ix.last_kernel()
        ix.close()
        _check(gi, gd, oi, od)
    assert res["100"]["seeded_pass_rows"] <= res["0"]["seeded_pass_rows"] + max(8, m // 50)


@pytest.mark.parametrize("m,n,g,k,drop,metric", [(5000, 20000, 50, 15, 0, 0), (30000, 30000, 50, 15, 1, 0), (20000, 100000, 50, 15, 0, 0),
                                                 (9000, 50000, 100, 50, 0, 2), (2000, 150000, 30, 40, 1, 0), (700, 3000, 50, 15, 0, 0),
                                                 (45000, 9000, 20, 11, 0, 0), (101000, 40000, 50, 15, 0, 0),
                                                 (110000, 300000, 50, 15, 0, 0)])      # (a long stream: one split more + a tail launch)
def test_cut_launches_and_merged_lists_change_no_result(gpu_lib, m, n, g, k, drop, metric):
    """Fewer column-workgroups than slots.  (Experiments builds: the one-product launch cut into equal pieces of the (column,
    reference tile) space -- api.hip: cut_pieces; option "pieces", a no-op in the product library.)  The several lists of a row are merged by their
    filter keys before the float64 re-evaluation (refine.hip: merge_lists_kernel); by default such a query is cut into ONE round
    of workgroups -- uniform splits, the columns that do not fit as a tail launch (api.hip: plan_l2, one_round).  The cut
    launch, the one-round plan, the cost model's plan, merged and unmerged lists and caller-chosen splits return the same bits; rows sampled against the oracle (all three geometries,
    cosine, the positional self-drop)."""
    Y = pca_like(n, g, seed=61)
    X = pca_like(m, g, seed=62) if not drop else (Y[:m] if m <= n else np.concatenate([Y, pca_like(m - n, g, seed=63)]))
    ref = None
    for opts in ({"pieces": 1}, {}, {"one_round": 0}, {"merge_lists": 0}, {"pieces": 1, "merge_lists": 0}, {"splits": 3}, {"splits": 3, "merge_lists": 0}):
        ix = gpu_lib.KnnIndex(n, g, metric=metric, options=opts).set_ref(Y)
        gi, gd = ix.query(X, k, drop_first=bool(drop))
        st = ix.last_stats()
        assert "l2c_topk_kernel" in ix.last_kernel()
        ix.close()
        if ref is None:
            ref = (gi, gd, st)
        assert np.array_equal(gi, ref[0]) and np.array_equal(gd, ref[1]), opts
    plan = _knn_mod().query_plan(n, g, m, k, metric=metric, drop_first=bool(drop), options={"pieces": 1})
    if plan["pieces"]:
        assert ref[2]["workgroups"] == plan["workgroups"] and ref[2]["splits"] == plan["splits"]
    rows = np.random.default_rng(5).choice(m, min(m, 400), replace=False)
    oi, od = oracle.knn(X[rows], Y, k, metric, drop_first=bool(drop), nthreads=16)
    _check(ref[0][rows], ref[1][rows], oi, od)


def test_asynchronous_queries_overlap_and_return_the_same_bits(gpu_lib):
    """nabo_index_query_async / _wait: the call returns before the query is done (the caller's thread is free), two indices on
    one device have their queries in flight together, results equal the synchronous call's; while a query is in flight every
    other call on that index is refused; an error of the query surfaces in wait()."""
    import time
    K = _knn_mod()
    n, g, k = 300000, 50, 15
    Y1, Y2 = pca_like(n, g, seed=101), pca_like(n, g, seed=102)
    X = pca_like(60000, g, seed=103)
    dx = _knn_buf(gpu_lib, X)
    m = X.shape[0]
    ixs = [K.KnnIndex(n, g, metric=0).set_ref(Y) for Y in (Y1, Y2)]
    outs = [(K.DeviceBuffer(m * k * 8), K.DeviceBuffer(m * k * 8)) for _ in ixs]
    sync = []
    for ix, (di, dd) in zip(ixs, outs):
        ix.query_device(dx.ptr, m, k, False, di.ptr, dd.ptr)
        sync.append((di.download((m, k), np.int64), dd.download((m, k), np.float64)))
    t_sync = []
    for ix, (di, dd) in zip(ixs, outs):
        t0 = time.perf_counter()
        ix.query_device(dx.ptr, m, k, False, di.ptr, dd.ptr)
        t_sync.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for ix, (di, dd) in zip(ixs, outs):
        ix.query_device_async(dx.ptr, m, k, False, di.ptr, dd.ptr)
    t_issue = time.perf_counter() - t0
    with pytest.raises(ValueError):
        ixs[0].query_device(dx.ptr, m, k, False, outs[0][0].ptr, outs[0][1].ptr)          # in flight: refused
    with pytest.raises(ValueError):
        ixs[0].set_mask(None)
    for ix in ixs:
        ix.wait()
    assert t_issue < 0.8 * min(t_sync), (t_issue, t_sync)                                # returned before ONE query could have finished (blocking calls: the sum of two)
    for (di, dd), (si, sd) in zip(outs, sync):
        assert np.array_equal(di.download((m, k), np.int64), si) and np.array_equal(dd.download((m, k), np.float64), sd)
    ixs[0].wait()                                                                         # nothing in flight: a no-op
    ixs[0].query_device_async(dx.ptr, m, n + 5, False, outs[0][0].ptr, outs[0][1].ptr)    # k beyond the references
    with pytest.raises(ValueError):
        ixs[0].wait()
    oi, od = oracle.knn(X[:200], Y1, k, 0, nthreads=8)
    _check(sync[0][0][:200], sync[0][1][:200], oi, od)
    for ix in ixs:
        ix.close()


@pytest.mark.parametrize("m,n,expect", [(4000, 30000, "cbb_filter"), (3000, 120000, "cbb_filter"), (4000, 20000, "cbf_filter")])
def test_canberra_default_counting_pass_by_size(gpu_lib, m, n, expect):
    """Which counting pass a modified-Canberra index picks when nobody pins it (api.hip: the bitmaps from 12 blocks = 24 576
    references on, round 4; the SWAR count below), a masked reference set included -- equal to the oracle either way."""
    g, k = 50, 15
    Y = pca_like(n, g, seed=111)
    X = pca_like(m, g, seed=112)
    mask = np.zeros(n, dtype=np.uint8)
    mask[5::11] = 1
    ix = gpu_lib.KnnIndex(n, g, metric=1, dist_factor=0.25).set_ref(Y, ref_mask=mask)
    gi, gd = ix.query(X, k)
    kern = ix.last_kernel()
    ix.close()
    assert expect in kern, kern
    rows = np.random.default_rng(3).choice(m, 600, replace=False)
    oi, od = oracle.knn(X[rows], Y, k, 1, 0.25, ref_mask=mask, nthreads=16)
    _check(gi[rows], gd[rows], oi, od)
