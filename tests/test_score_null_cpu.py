"""CPU-side checks of the permutation-null definition (EXTENSION, no reference counterpart): the oracle's
observed score is the reference's mapping score, group sizes are preserved, ties follow the documented rule."""
import numpy as np

from oracle import oracle as orc
from nabo_amd._score import mapping_score_from_edges


def test_oracle_null_definition_properties():
    rng = np.random.default_rng(3)
    n_ref, n_t, k, P = 25, 120, 4, 40
    edge_t = np.repeat(np.arange(n_t), k)
    edge_r = rng.integers(0, n_ref, n_t * k)
    w = rng.choice([0.05, 0.11, 0.25, 1.0], n_t * k)
    group = rng.random(n_t) < 0.35
    r = orc.score_null(edge_t, edge_r, w, group, n_ref, P, seed=9)
    keep = group[edge_t]
    sc = mapping_score_from_edges(n_ref, edge_r[keep], w[keep], int(group.sum()))
    assert np.allclose(r["obs"], sc, rtol=1e-13, atol=0)               # nabo/_graph.py:644-653
    assert (r["sizes"] == int(group.sum())).all()                      # 64-bit keys: no ties, exact permutations
    assert r["n_ge"].min() >= 0 and r["n_ge"].max() <= P
    # a permuted score is a score: total mass is conserved in expectation, never negative
    assert (r["scores"] >= 0).all()
    r8 = orc.score_null(edge_t, edge_r, w, group, n_ref, P, seed=9, key_bits=8)
    assert (r8["sizes"] >= int(group.sum())).all() and (r8["sizes"] > int(group.sum())).any()
    # different seeds give different permutations, the same seed the same ones
    assert np.array_equal(orc.score_null(edge_t, edge_r, w, group, n_ref, P, seed=9)["n_ge"], r["n_ge"])
    assert not np.array_equal(orc.score_null(edge_t, edge_r, w, group, n_ref, P, seed=10)["scores"], r["scores"])
