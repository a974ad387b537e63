#!/usr/bin/env python
"""BASELINE.json configs[4] run WHOLE on one MI355X: 5M ref x 5M target, d=100, k=50, cosine + the 1000-permutation null
of the mapping scores (both EXTENSIONS: the reference has neither a cosine metric nor a permutation test -- parity is
pinned by this build's own oracle only).

    python tools/config4_one_gpu.py [--cells 5000000] [--dims 100] [--neighbors 50] [--perms 1000] [--ranks 8]

Steps (product entry points only; tests/test_configs_gpu.py adds the oracle checks, bench.py reports the times as its
`config4_one_gpu` block):
  1. target <-> reference k-NN of ALL rows on ONE unsharded index (the references resident: 4 GB of float64 rows + 4 GB of
     unit rows + 1.3 GB of f16 tiles of the 288 GB), the targets in batches of 1M rows;
  2. (ranks > 1) the same through nabo_sharded_query with `ranks` loopback shard-ranks on the one GPU -- BASELINE's
     "8 x MI355X" layout, the reference rows sharded 8 ways: every batch of every rank must equal step 1 on all rows;
  3. reference <-> reference k-NN with the positional self-drop (nabo/_mapping.py:142), the SNN counts
     (nabo/_mapping.py:186-198) of every target against it, weights round(snn / (2 (k - 1) - snn), 2), edges with snn > 0;
  4. nabo_score_null_edges on that graph (up to 250M edges), `perms` permutations of the pooled target cells.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def snn_weights(snn, k):
    """nabo/_mapping.py:185,194: round(snn / (2 (k - 1) - snn), 2) with Python's round(), as a table over the k + 1 counts"""
    tab = np.array([round(s / (2 * (k - 1) - s), 2) if 2 * (k - 1) - s != 0 else np.inf for s in range(k + 1)])
    return tab[snn]


def knn_all_rows(ix, X, k, drop_first, batch, dev=0, on_batch=None):
    """every row of X (host) through ix in batches; returns idx, dist (host), sum of on-stream ms, per-row pass ids"""
    from nabo_amd import _knn
    m, d = X.shape
    gi = np.empty((m, k), dtype=np.int64)
    gd = np.empty((m, k), dtype=np.float64)
    rp = np.empty(m, dtype=np.uint8)
    dx = _knn.DeviceBuffer(batch * d * 8, dev)
    di, dd = _knn.DeviceBuffer(batch * k * 8, dev), _knn.DeviceBuffer(batch * k * 8, dev)
    ms, stats = 0.0, []
    try:
        for b0 in range(0, m, batch):
            nb = min(batch, m - b0)
            dx.upload(np.ascontiguousarray(X[b0:b0 + nb]))
            ix.query_device(dx.ptr, nb, k, drop_first, di.ptr, dd.ptr)
            st = ix.last_stats()
            ms += st["ms_total"]
            stats.append(st)
            gi[b0:b0 + nb] = di.download((nb, k), np.int64)
            gd[b0:b0 + nb] = dd.download((nb, k), np.float64)
            rp[b0:b0 + nb] = ix.last_row_pass(nb)
            if on_batch:
                on_batch(b0, nb, dx)
    finally:
        for b in (dx, di, dd):
            b.free()
    return gi, gd, ms, rp, stats


def run(n=5000000, m=5000000, d=100, k=50, ranks=8, perms=1000, batch=1000000, dev=0, seed_ref=1004, seed_tgt=2004,
        group_frac=0.4, keep=False, log=None):
    """Returns (report dict, arrays dict or None).  `keep`: also return X, Y, the k-NN results and the edge list (tests)."""
    import nabo_amd
    from nabo_amd import _knn, _sharded
    from nabo_amd._synth import pca_like_big
    say = log or (lambda *a: None)
    rep = {"workload": "%dM ref x %dM target, d=%d, k=%d, cosine (EXTENSION: parity vs this build's oracle only), one GPU"
                       % (n // 1000000, m // 1000000, d, k)}
    t0 = time.perf_counter()
    Y = pca_like_big(n, d, seed=seed_ref)
    X = pca_like_big(m, d, seed=seed_tgt)
    rep["host_generate_s"] = time.perf_counter() - t0
    say("generated", rep["host_generate_s"])
    MET = nabo_amd.COSINE
    dY = _knn.DeviceBuffer(Y.nbytes, dev).upload(Y)
    ix = nabo_amd.KnnIndex(n, d, metric=MET, device=dev)
    t0 = time.perf_counter()
    ix.set_ref(y_device_ptr=dY.ptr)
    rep["set_ref_s"] = time.perf_counter() - t0
    # 1. all target rows on the unsharded index
    t0 = time.perf_counter()
    ti, td, ms_t, rp_t, st_t = knn_all_rows(ix, X, k, False, batch, dev)
    rep["target_knn"] = {"gpu_ms": ms_t, "wall_s": time.perf_counter() - t0, "kernel": ix.last_kernel(),
                         "kernel_ms": sum(s["ms_topk"] for s in st_t), "pairs_per_s": m * n / (ms_t * 1e-3),
                         "rows_by_pass": {nm: int((rp_t == c).sum()) for c, nm in enumerate(nabo_amd.KnnIndex.PASS_NAMES)},
                         "frac_of_f16_mfma_peak": 2.0 * m * n * d / (sum(s["ms_topk"] for s in st_t) * 1e-3) / 2516.6e12}
    say("target knn", rep["target_knn"])
    # 3a. reference <-> reference (positional self-drop)
    t0 = time.perf_counter()
    ri, rd, ms_r, rp_r, st_r = knn_all_rows(ix, Y, k, True, batch, dev)
    rep["ref_knn"] = {"gpu_ms": ms_r, "wall_s": time.perf_counter() - t0, "kernel_ms": sum(s["ms_topk"] for s in st_r),
                      "rows_by_pass": {nm: int((rp_r == c).sum()) for c, nm in enumerate(nabo_amd.KnnIndex.PASS_NAMES)}}
    say("ref knn", rep["ref_knn"])
    ix.close()
    # 2. the reference rows sharded `ranks` ways (loopback ranks on this GPU): all rows of all batches equal step 1
    if ranks > 1:
        grp = _sharded.LoopbackGroup(ranks, dev, n, d, MET, Y).set_ref()
        dx = _knn.DeviceBuffer(batch * d * 8, dev)
        outs = [(_knn.DeviceBuffer(batch * k * 8, dev), _knn.DeviceBuffer(batch * k * 8, dev)) for _ in range(ranks)]
        gpu_ms, second, same = 0.0, 0, True
        per_rank_ms = np.zeros(ranks)
        t0 = time.perf_counter()
        try:
            for b0 in range(0, m, batch):
                nb = min(batch, m - b0)
                dx.upload(np.ascontiguousarray(X[b0:b0 + nb]))
                grp.query_device(dx.ptr, nb, k, False, [a.ptr for a, _ in outs], [b.ptr for _, b in outs])
                st = [grp.last_stats(r) for r in range(ranks)]
                per_rank_ms += np.array([s["ms_total"] for s in st])
                second = max(second, st[0]["uncertified"])
                for a, b in outs:                       # EVERY rank's copy of EVERY row
                    same = same and np.array_equal(a.download((nb, k), np.int64), ti[b0:b0 + nb]) \
                        and np.array_equal(b.download((nb, k), np.float64), td[b0:b0 + nb])
            cand = st[0]["candidates"]
        finally:
            grp.close()
            dx.free()
            for a, b in outs:
                a.free(); b.free()
        gpu_ms = float(per_rank_ms.sum())
        rep["sharded_loopback"] = {"ranks": ranks, "gpu_ms_all_ranks": gpu_ms, "max_rank_ms": float(per_rank_ms.max()),
                                   "wall_s": time.perf_counter() - t0, "candidates_per_shard": int(cand),
                                   "second_round_rows_max": int(second), "all_rows_of_all_ranks_equal_unsharded": bool(same)}
        say("sharded", rep["sharded_loopback"])
    dY.free()
    # 3b. SNN counts -> weighted bipartite edges (nabo/_mapping.py:186-198)
    t0 = time.perf_counter()
    snn = _knn.snn_counts(ti, ri, k, device=dev)
    rep["snn_s"] = time.perf_counter() - t0
    keep_e = snn > 0
    e_t = np.repeat(np.arange(m, dtype=np.int64), k)[keep_e.ravel()]
    e_r = ti.ravel()[keep_e.ravel()]
    w = snn_weights(snn.ravel()[keep_e.ravel()], k)
    rep["edges"] = int(e_t.size)
    say("edges", rep["edges"], rep["snn_s"])
    # 4. the permutation null
    group = (np.random.default_rng(21).random(m) < group_frac).astype(np.uint8)
    t0 = time.perf_counter()
    null = nabo_amd.mapping_score_null(e_t, e_r, w, group, n, n_perm=perms, seed=3, device=dev)
    rep["null"] = {"perms": perms, "wall_s": time.perf_counter() - t0, "edge_permutations_per_s": e_t.size * perms / (time.perf_counter() - t0),
                   "nodes_with_edges": int((np.bincount(e_r, minlength=n) > 0).sum()),
                   "nodes_p_below_0.01": int((null["pvalue"] < 0.01).sum())}
    say("null", rep["null"])
    arrays = None
    if keep:
        arrays = {"X": X, "Y": Y, "ti": ti, "td": td, "rp_t": rp_t, "ri": ri, "rd": rd, "rp_r": rp_r, "snn": snn,
                  "e_t": e_t, "e_r": e_r, "w": w, "group": group, "null": null}
    return rep, arrays


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=5000000)
    ap.add_argument("--dims", type=int, default=100)
    ap.add_argument("--neighbors", type=int, default=50)
    ap.add_argument("--perms", type=int, default=1000)
    ap.add_argument("--ranks", type=int, default=8)
    a = ap.parse_args()
    rep, _ = run(a.cells, a.cells, a.dims, a.neighbors, a.ranks, a.perms, log=lambda *x: print(*x, file=sys.stderr, flush=True))
    print(json.dumps(rep))
