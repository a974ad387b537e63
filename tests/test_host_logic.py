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
