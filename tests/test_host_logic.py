"""Host-side logic that needs neither a GPU nor HDF5."""
import numpy as np
import pytest

from nabo_amd._mapping import pyset_iteration_order, snn_edges_from_counts, snn_weight_table


@pytest.mark.parametrize("k", [1, 3, 4, 5, 7, 11, 15, 18, 19, 20, 31, 50, 56, 100])
def test_pyset_iteration_order_is_cpythons(k):
    """the reference visits a cell's neighbours with `for j in set(order[:k])` (nabo/_mapping.py:190-191);
    the product restates CPython's set layout: compare with the interpreter's own set"""
    rng = np.random.default_rng(k)
    for hi in (k + 1, 40, 400, 100000, 5000000):
        if hi < k:
            continue
        rows = np.stack([rng.choice(hi, k, replace=False) for _ in range(200)])
        perm = pyset_iteration_order(rows)
        for r in range(rows.shape[0]):
            assert list(set(rows[r])) == list(rows[r][perm[r]]), (k, hi, r)


def test_pyset_iteration_order_edges():
    assert pyset_iteration_order(np.zeros((0, 5), dtype=np.int64)).shape == (0, 5)
    assert pyset_iteration_order(np.zeros((3, 0), dtype=np.int64)).shape == (3, 0)
    with pytest.raises(ValueError):
        pyset_iteration_order(np.array([[1, -1]]))


def test_snn_edges_from_counts_weights_and_order():
    k = 5
    t_idx = np.array([[9, 3, 40, 8, 17], [1, 2, 3, 4, 5]])
    cnt = np.array([[2, 0, 5, 1, 3], [0, 0, 0, 0, 0]], dtype=np.int32)
    et, ej, ew = snn_edges_from_counts(t_idx, cnt, k)
    tab = snn_weight_table(k)
    want = [(j, tab[c]) for j, c in ((j, cnt[0][list(t_idx[0]).index(j)]) for j in set(t_idx[0])) if c > 0]
    assert list(et) == [0] * len(want) and [(int(j), float(w)) for j, w in zip(ej, ew)] == want
    for s in range(1, k + 1):
        assert tab[s] == round(s / (2 * (k - 1) - s), 2)
    with pytest.raises(ZeroDivisionError):              # k = 2, snn = 2: the reference divides by zero (:194)
        snn_edges_from_counts(np.array([[0, 1]]), np.array([[2, 1]], dtype=np.int32), 2)


def test_native_host_helpers_equal_their_numpy_restatements():
    """csrc/host_graph.hip against tests/_host_numpy.py (the forms the product used before): the set order on many rows
    (threads), component labels of random forests + cycles, edge grouping with repeated pairs (first position, last weight)."""
    from nabo_amd._mapping import _component_labels, group_edges
    from _host_numpy import component_labels_numpy, group_edges_numpy, pyset_iteration_order_numpy
    rng = np.random.default_rng(5)
    for n, k, hi in ((20000, 15, 1000000), (5000, 50, 300), (9000, 23, 1 << 40)):
        rows = np.argsort(rng.random((n, min(hi, 4 * k))), axis=1)[:, :k].astype(np.int64)
        if hi > 4 * k:
            rows = rows * (hi // (4 * k)) + rng.integers(0, hi // (4 * k), (n, 1))
        assert np.array_equal(pyset_iteration_order(rows), pyset_iteration_order_numpy(rows))
    for n, e in ((1, 0), (50, 0), (1000, 300), (20000, 15000), (20000, 90000)):
        a, b = rng.integers(0, n, e), rng.integers(0, n, e)
        lab = _component_labels(n, a, b)
        assert np.array_equal(lab, component_labels_numpy(n, a, b))
        assert (lab <= np.arange(n)).all() and np.array_equal(lab[lab], lab)
    for n_nodes, rows_n, span in ((1, 0, 1), (7, 40, 5), (3000, 60000, 40), (50000, 400000, 100000)):
        node = rng.integers(0, n_nodes, rows_n)
        nb = rng.integers(0, span, rows_n)
        w = np.round(rng.random(rows_n), 2)
        s1, n1, w1 = group_edges(n_nodes, node, nb, w)
        s2, n2, w2 = group_edges_numpy(n_nodes, node, nb, w)
        assert np.array_equal(s1, s2) and np.array_equal(n1, n2) and np.array_equal(w1, w2)
    with pytest.raises(ValueError):
        group_edges(3, np.array([0, 3]), np.array([1, 1]), np.array([0.5, 0.5]))
    with pytest.raises(ValueError):
        _component_labels(3, np.array([0]), np.array([5]))


def test_list_entries_order_as_doubles_like_their_fp32_keys():
    """topk_lists.h keeps a list entry as ONE double -- high word the fp32 key, low word slot | offset -- and finds the
    largest entry with v_max_f64.  The property it rests on, checked here on the host: for non-NaN keys (finite, +-inf,
    zeros, denormals) the order of such doubles refines the order of the keys; none of them is a NaN or an infinity."""
    rng = np.random.default_rng(11)
    special = np.array([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45, 1.17549435e-38, -1.17549435e-38, 3.4028235e38, -3.4028235e38,
                        1.0, -1.0], dtype=np.float32)
    keys = np.concatenate([special, rng.standard_normal(4000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 4000).astype(np.float32),
                           rng.integers(0, 1 << 32, 4000, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    keys = keys[~np.isnan(keys)]
    lo = rng.integers(0, 1 << 32, keys.shape[0], dtype=np.uint64)
    ent = ((keys.view(np.uint32).astype(np.uint64) << np.uint64(32)) | lo).view(np.float64)
    assert np.isfinite(ent).all()
    a, b = rng.integers(0, keys.shape[0], 200000), rng.integers(0, keys.shape[0], 200000)
    lt = keys[a] < keys[b]
    assert (ent[a][lt] < ent[b][lt]).all()                      # strictly smaller key => strictly smaller entry
    m = np.maximum(ent[a], ent[b])                               # the maximum's key is the larger key
    assert np.array_equal((m.view(np.uint64) >> np.uint64(32)).astype(np.uint32).view(np.float32) >= np.maximum(keys[a], keys[b]),
                          np.ones(a.shape[0], dtype=bool))


# ---- the launch planner (nabo_query_plan: a pure function of shapes and options -- no index, no device) -----------------------
def test_query_plan_fills_the_chip_and_follows_the_list_rules():
    """api.hip: plan_l2 through nabo_query_plan.  BASELINE's shapes: the default first pass is the one-product kernel in the
    geometry its lists want; a query with enough work is cut into at least as many workgroups as the part has CUs
    (configs[1], 100k x 100k, was 131 workgroups on 256 CUs in round 3); list lengths, split bounds and the tournament
    follow their documented rules; the fp32 / f16x3 modes and the exact route are planned, not discovered on the device."""
    from nabo_amd import _knn
    P = _knn.query_plan
    c2 = P(1000000, 50, 1000000, 15)                                      # BASELINE configs[2]
    assert c2["kernel"].startswith("l2c_topk_kernel<2,1,23,6,32,4,2>") and c2["first_pass"] == 0 and c2["geometry"] == 1
    assert c2["lkeep"] == 15 + 8 and c2["list_len"] == 32 and c2["rows_per_wg"] == 384 and c2["resident_workgroups"] == 512
    assert c2["workgroups_main"] % 512 == 0 and c2["workgroups_tail"] * c2["splits_tail"] <= 512
    assert c2["workgroups_main"] * 384 + c2["workgroups_tail"] * 384 == c2["rows_padded"] >= 1000000
    assert 0 < c2["tournament_tiles"] <= c2["tiles_per_split"] // 4 and c2["tournament_tiles"] % c2["tournament_group"] == 0
    c1 = P(100000, 50, 100000, 15)                                        # BASELINE configs[1]
    assert c1["workgroups"] >= 256, c1
    # fewer column-workgroups than slots: ONE round of workgroups -- uniform splits where they fill >= 80 % of the slots, on long
    # reference streams one split more plus a tail launch; otherwise the cost model (the "pieces" cut of the same space is
    # an experiments-build option: a no-op here)
    assert c1["pieces"] == 0 and P(100000, 50, 100000, 15, options={"pieces": 1})["pieces"] == 0
    p = P(100000, 50, 49152, 15)
    assert p["workgroups_main"] == 128 and p["splits"] == 4 and p["workgroups"] == 512 == p["resident_workgroups"]
    p = P(30000, 50, 30000, 15)
    assert p["splits"] == 6 and 0.8 * 512 <= p["workgroups"] <= 512
    p = P(1000000, 50, 120000, 15)
    assert (p["workgroups_main"], p["splits"], p["workgroups_tail"]) == (256, 2, 57) and p["splits_tail"] >= 4
    p = P(100000, 50, 100000, 15, options={"one_round": 0})
    assert p["workgroups_tail"] == 0 and p["splits"] == 1
    c4 = P(5000000, 100, 1000000, 50, metric=2)                           # BASELINE configs[4]: cosine, d = 100, k = 50
    assert c4["kernel"].startswith("l2c_topk_kernel<4,2,65,4,64,4,1>") and c4["lkeep"] == 64 and c4["list_len"] == 64
    shard = P(125000, 50, 1000000, 15, n_cand=12)                         # one rank of eight: candidate mode, 12 emitted + 3 kept
    assert shard["lkeep"] == 15 and P(125000, 50, 1000000, 15, n_cand=12, options={"cand_slack": 0})["lkeep"] == 12
    assert P(1000000, 50, 1000, 15, l2_mode="f32")["kernel"].startswith("l2_topk_kernel") and P(1000000, 50, 1000, 15, l2_mode="f32")["first_pass"] == 2
    assert P(1000000, 50, 1000, 15, l2_mode="f16x3")["kernel"].startswith("l2q_topk_kernel<10,")
    assert P(1000000, 126, 1000, 15)["kernel"].startswith("l2_topk_kernel")               # beyond the one-product operands
    assert P(1000000, 200, 1000, 15)["first_pass"] == 4 and P(1000000, 50, 1000, 60)["first_pass"] == 4      # no-cliff limits
    # options: the caller's split count, the geometry pin, the tournament switch; unknown names are refused
    assert P(100000, 50, 5000, 15, options={"splits": 7})["splits"] == 7
    assert P(100000, 50, 5000, 15, options={"l2c_geo": 0})["kernel"].startswith("l2c_topk_kernel<2,1,33,8,64,4,1>")
    assert P(1000000, 50, 1000000, 15, options={"prepass": 0})["tournament_tiles"] == 0
    assert P(200000000, 50, 1000, 15)["splits"] >= 6                      # 25 bits of offset per split: > 2^25 references need more
    with pytest.raises(ValueError):
        P(1000, 50, 1000, 15, options={"no_such_option": 1})
    with pytest.raises(ValueError):
        P(1000, 50, 1000, 15, metric=1)
