"""The CPU oracle (oracle/nabo_oracle.c) against golden vectors produced by the reference's
own nabo/_mapping.py (oracle/gen_golden.py).  This is what PINS the oracle."""
import numpy as np
import pytest

import oracle
from nabo_amd._synth import pca_like, digest


def _fkey(f):
    return str(f).replace(".", "p")


@pytest.mark.parametrize("d", [7, 30, 50])
def test_pairwise_kernels_bit_exact(golden, d):
    g = golden("kernels")
    x, y = g["x_%d" % d], g["y_%d" % d]
    assert np.array_equal(oracle.pairwise(x, y, oracle.EUCLIDEAN), g["euclid_%d" % d])
    for f in (0.1, 0.25, 1.0):
        assert np.array_equal(oracle.pairwise(x, y, oracle.MOD_CANBERRA, f), g["canberra_%d_%s" % (d, _fkey(f))])


def test_synth_inputs_are_reproducible(golden):
    g = golden("c1_3k")
    s_ref, s_tgt = g["seeds"]
    assert digest(pca_like(3000, 30, int(s_ref))) == str(g["ref_digest"])
    assert digest(pca_like(3000, 30, int(s_tgt))) == str(g["t_ME_digest"])


def _ordered(g, prefix_names, prefix_cells, data_key):
    names = list(g[prefix_names])
    order = [names.index(c) for c in g[prefix_cells]]
    return g[data_key][order]


def test_mapping_small_order_rows(golden):
    """Lexicographic HDF5 cell order, chunk_size that does not divide N, [1:] drop, ignore mask."""
    g = golden("mapping_small")
    uc, k, _ = g["params"]
    f = float(g["dist_factor"])
    assert list(g["ref_cells"][:4]) == ["R0", "R1", "R10", "R100"]          # name order, not insertion order
    ref = _ordered(g, "ref_names", "ref_cells", "ref")[:, :uc]
    idx, dist = oracle.knn(ref, ref, 32, oracle.EUCLIDEAN, drop_first=True)
    assert not g["ref_ties"].any()
    assert np.array_equal(idx, g["ref_idx"]) and np.array_equal(dist, g["ref_dist"])
    assert int(g["ref_order_len"]) == 399
    for t in ("ME", "IG"):
        X = _ordered(g, "t_%s_names" % t, "t_%s_cells" % t, "t_%s_data" % t)[:, :uc]
        mask = np.isin(g["ref_cells"], g["t_%s_ignore" % t]).astype(np.uint8)
        idx, dist = oracle.knn(X, ref, 32, oracle.MOD_CANBERRA, f, ref_mask=mask)
        assert not g["t_%s_ties" % t].any()
        assert np.array_equal(idx, g["t_%s_idx" % t]) and np.array_equal(dist, g["t_%s_dist" % t])
    # ignored refs stay in the order row, at its very end
    assert list(g["t_IG_ignore_pos"]) == [395, 396, 397, 398, 399] and int(g["t_IG_order_len"]) == 400


def test_mapping_small_masked_tail(golden):
    """k larger than the number of un-ignored refs: ignored refs follow, ascending index."""
    g = golden("mapping_small")
    uc = g["params"][0]
    ref = _ordered(g, "ref_names", "ref_cells", "ref")[:12, :uc]
    X = _ordered(g, "t_IG_names", "t_IG_cells", "t_IG_data")[:5, :uc]
    mask = np.zeros(12, np.uint8)
    mask[[2, 7, 9]] = 1
    idx, dist = oracle.knn(X, ref, 12, oracle.MOD_CANBERRA, 0.25, ref_mask=mask)
    assert np.array_equal(idx[:, 9:], np.tile([2, 7, 9], (5, 1)))
    assert set(idx[0, :9]) == set(range(12)) - {2, 7, 9}


def test_dup_case_positional_drop(golden):
    """Duplicate reference cells: `[1:]` is positional, so a cell can keep ITSELF as a neighbour
    (the lower-index twin sorts first and is the one dropped)."""
    g = golden("dup")
    uc, k, _ = g["params"]
    ref = _ordered(g, "ref_names", "ref_cells", "ref")[:, :uc]
    idx, dist = oracle.knn(ref, ref, 59, oracle.EUCLIDEAN, drop_first=True)
    assert np.array_equal(dist, g["ref_dist"])                 # distances (sorted) always agree
    # wherever a distance is unique within its row the index must agree (twins 3/7 and 40/41 tie
    # in EVERY row, which is exactly where the reference's unstable sort is free)
    uniq = np.ones_like(dist, dtype=bool)
    uniq[:, 1:] &= dist[:, 1:] != dist[:, :-1]
    uniq[:, :-1] &= dist[:, :-1] != dist[:, 1:]
    assert np.array_equal(idx[uniq], g["ref_idx"][uniq]) and uniq.mean() > 0.9
    # rows 3/7 and 40/41 are exact duplicates: the tie at distance 0 is where numpy's unstable
    # sort is free; the canonical rule (dist, idx) must drop the lower twin from both rows
    assert idx[7, 0] == 7 and idx[3, 0] == 7 and idx[41, 0] == 41 and idx[40, 0] == 41
    assert g["ref_idx"][7, 0] == 7 and g["ref_idx"][3, 0] == 7     # ...and that is what the reference did
    X = _ordered(g, "t_TG_names", "t_TG_cells", "t_TG_data")[:, :uc]
    mask = np.isin(g["ref_cells"], g["t_TG_ignore"]).astype(np.uint8)
    idx, dist = oracle.knn(X, ref, 59, oracle.MOD_CANBERRA, float(g["dist_factor"]), ref_mask=mask)
    # compare distances only over the un-ignored part of the order row (the reference stores the
    # true distance of ignored refs but sorts them last)
    nv = 60 - int(mask.sum())
    assert np.array_equal(dist[:, :nv - 1], g["t_TG_dist"][:, :nv - 1])
    tt = g["t_TG_ties"]
    assert np.array_equal(idx[~tt][:, :k], g["t_TG_idx"][~tt][:, :k])


def test_c1_full_size(golden):
    """BASELINE.json configs[0]: 3k x 3k, d=30, k=11 -- idx and dist bit-equal to the reference."""
    g = golden("c1_3k")
    ref, tgt = pca_like(3000, 30, 1001), pca_like(3000, 30, 2001)
    idx, dist = oracle.knn(ref, ref, 16, oracle.EUCLIDEAN, drop_first=True, nthreads=8)
    assert not g["ref_ties"].any()
    assert np.array_equal(idx, g["ref_idx"]) and np.array_equal(dist[:, :12], g["ref_dist"])
    idx, dist = oracle.knn(tgt, ref, 16, oracle.MOD_CANBERRA, 0.25, nthreads=8)
    assert not g["t_ME_ties"].any()
    assert np.array_equal(idx, g["t_ME_idx"]) and np.array_equal(dist[:, :12], g["t_ME_dist"])


def _graph_sets(g, prefix):
    return {(s, d, w) for s, d, w in zip(g[prefix + "_src"], g[prefix + "_dst"], g[prefix + "_w"])}


def test_snn_edges_match_reference_graph(golden):
    """a5 (nabo/_mapping.py:186-198): SNN edges + weights from the first k of the order rows."""
    g = golden("mapping_small")
    k = int(g["params"][1])
    cells = list(g["ref_cells"])
    ridx = g["ref_idx"].astype(np.int64)
    t, j, w = oracle.snn_edges(g["t_ME_idx"].astype(np.int64), ridx, k)
    mine = {(str(g["t_ME_cells"][a]) + "_ME", cells[b] + "_WT", float(c)) for a, b, c in zip(t, j, w)}
    assert mine == _graph_sets(g, "t_ME_graph")
    # reference graph: undirected, every edge listed under both end points; repair edges carry
    # weight 0.5/(2(k-1)-0.5) and are the only ones with a non 2-decimal weight
    t, j, w = oracle.snn_edges(ridx, ridx, k)
    mine = set()
    for a, b, c in zip(t, j, w):
        mine.add((cells[a] + "_WT", cells[b] + "_WT", float(c)))
        mine.add((cells[b] + "_WT", cells[a] + "_WT", float(c)))
    fixw = 0.5 / ((2 * (k - 1)) - 0.5)
    ref_edges = _graph_sets(g, "ref_graph")
    assert mine == {e for e in ref_edges if e[2] != fixw}


@pytest.mark.parametrize("tag", ["mini_%d" % i for i in range(6)])
def test_mini_cases_order_rows_and_target_graph(golden, tag):
    """Six small full runs of the reference over a spread of parameters (use_comps < stored components, k 3..15,
    dist_factor 0.1..2, chunk sizes that do / do not divide the cell counts, ignore lists, three naming styles):
    the oracle reproduces the order rows, their distances and the target SNN graph."""
    g = golden(tag)
    uc, k, _ = [int(v) for v in g["params"]]
    f = float(g["dist_factor"])
    nk = g["ref_idx"].shape[1]
    ref = _ordered(g, "ref_names", "ref_cells", "ref")[:, :uc]
    idx, dist = oracle.knn(ref, ref, nk, oracle.EUCLIDEAN, drop_first=True)
    ok = ~g["ref_ties"]
    assert np.array_equal(idx[ok][:, :k], g["ref_idx"][ok][:, :k]) and np.array_equal(dist[:, :k], g["ref_dist"][:, :k])
    X = _ordered(g, "t_TG_names", "t_TG_cells", "t_TG_data")[:, :uc]
    mask = np.isin(g["ref_cells"], g["t_TG_ignore"]).astype(np.uint8)
    idx, dist = oracle.knn(X, ref, nk, oracle.MOD_CANBERRA, f, ref_mask=mask)
    tt = g["t_TG_ties"]
    assert np.array_equal(idx[~tt][:, :k], g["t_TG_idx"][~tt][:, :k]) and np.array_equal(dist[:, :k], g["t_TG_dist"][:, :k])
    if not tt.any() and not g["ref_ties"].any():
        cells = list(g["ref_cells"])
        t, j, w = oracle.snn_edges(g["t_TG_idx"].astype(np.int64), g["ref_idx"].astype(np.int64), k)
        mine = {(str(g["t_TG_cells"][a]) + "_TG", cells[b] + "_WT", float(c)) for a, b, c in zip(t, j, w)}
        assert mine == _graph_sets(g, "t_TG_graph")
