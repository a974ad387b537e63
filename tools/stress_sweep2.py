"""Randomised parity sweep, part 2: every Euclidean filter kernel (NABO_L2_MODE drawn at random), the sharded query
behind the C ABI (nabo_sharded_query over the loopback transport, 2-8 ranks on one GPU, masks, second round), mask swaps
on a resident index, SNN counts and the permutation null.
    python tools/stress_sweep2.py [n_cases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import nabo_amd  # noqa: E402
import oracle  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from nabo_amd import _knn  # noqa: E402
from nabo_amd import _sharded  # noqa: E402
from nabo_amd._synth import pca_like  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()


def fail(msg):
    print("MISMATCH " + msg)
    sys.exit(1)


def data(n, m, g, flavour):
    Y = pca_like(n, g, seed=int(rng.integers(1, 1 << 30)))
    X = pca_like(m, g, seed=int(rng.integers(1, 1 << 30)))
    if flavour == 1:
        Y[rng.integers(0, n, max(1, n // 3))] = Y[rng.integers(0, n)]
    elif flavour == 2:
        Y, X = np.round(Y), np.round(X)
    elif flavour == 3:
        sc = 10.0 ** rng.integers(-15, 15)
        Y, X = Y * sc, X * sc
    elif flavour == 4:
        Y = Y[np.argsort(Y[:, 0])]                  # spatially sorted: neighbours concentrate in one shard
    return X, Y


counts = {"f16x3": 0, "sharded": 0, "set_mask": 0, "snn": 0, "null": 0, "sequence": 0, "lattice": 0, "pairwise": 0}
for case in range(n_cases):
    kind = ["f16x3", "sharded", "set_mask", "snn", "null", "sequence", "lattice", "pairwise"][int(rng.integers(0, 8))]
    if kind == "f16x3":
        n = int(rng.choice([40, 300, 3000, 20000])); m = int(rng.choice([1, 33, 400, 1500])); g = int(rng.integers(1, 64))
        k = int(rng.integers(1, min(n, 24) + 1)); fl = int(rng.integers(0, 4))
        X, Y = data(n, m, g, fl)
        mask = (rng.random(n) < 0.3).astype(np.uint8) if rng.random() < 0.3 else None
        drop = bool(rng.integers(0, 2)) and m <= n and k < n
        if drop:
            X = Y[:m].copy()
        os.environ["NABO_L2_MODE"] = str(rng.choice(["f16x3", "f16x3q", "f16x3h", "f16x3s", "f32"]))
        try:
            gi, gd = nabo_amd.knn(X, Y, k, metric=0, ref_mask=mask, drop_first=drop)
        finally:
            del os.environ["NABO_L2_MODE"]
        oi, od = oracle.knn(X, Y, k, 0, ref_mask=mask, drop_first=drop, nthreads=16)
        if not (np.array_equal(gi, oi) and np.array_equal(gd, od, equal_nan=True)):
            fail("f16x3 case %d n=%d m=%d g=%d k=%d flavour=%d" % (case, n, m, g, k, fl))
    elif kind == "sharded":
        N = int(rng.integers(2, 9)); n = int(rng.choice([200, 3000, 20000])); m = int(rng.choice([5, 64, 301])); g = int(rng.integers(2, 60))
        metric = int(rng.choice([0, 1, 2])); drop = bool(rng.integers(0, 2)) and m <= n
        k = int(rng.integers(1, min(n // N, 30) + 1)); fl = int(rng.integers(0, 5))
        X, Y = data(n, m, g, fl)
        if drop:
            X = Y[:m].copy()
        kk = k + (1 if drop else 0)
        mask = None
        if rng.random() < 0.4:
            mask = (rng.random(n) < rng.choice([0.1, 0.6])).astype(np.uint8)
            if int((mask == 0).sum()) < kk:
                mask = None
        if min(_sharded.shard_bounds(n, N, r)[1] - _sharded.shard_bounds(n, N, r)[0] for r in range(N)) < kk:
            continue                                      # a shard must hold k' references (include/nabo_knn.h)
        # the 2-D layout (R reference pieces x N / R target slices) for the metrics that certify globally
        R = int(rng.choice([r for r in range(1, N + 1) if N % r == 0])) if metric != 1 else N
        if min(_sharded.shard_bounds(n, R, r)[1] - _sharded.shard_bounds(n, R, r)[0] for r in range(R)) < kk:
            R = N
        grp = _sharded.LoopbackGroup(N, 0, n, g, metric, Y, ref_mask=mask, ref_shards=R).set_ref()
        gi, gd = grp.query(X, k, drop_first=drop)
        grp.close()
        oi, od = oracle.knn(X, Y, k, metric, 0.25, ref_mask=mask, drop_first=drop, nthreads=16)
        if not (np.array_equal(gi, oi) and np.array_equal(gd, od, equal_nan=True)):
            bad = np.where((gi != oi).any(1) | ~((gd == od) | (np.isnan(gd) & np.isnan(od))).all(1))[0]
            r = int(bad[0])
            print("rows differing: %d of %d; first %d\n got idx  %s\n want idx %s\n got d  %s\n want d %s" % (len(bad), m, r, gi[r], oi[r], gd[r], od[r]))
            fail("sharded case %d N=%d R=%d n=%d m=%d g=%d k=%d drop=%s metric=%d flavour=%d" % (case, N, R, n, m, g, k, drop, metric, fl))
    elif kind == "set_mask":
        n = int(rng.choice([100, 2000, 9000])); m = int(rng.choice([3, 70, 400])); g = int(rng.integers(1, 80))
        metric = int(rng.integers(0, 3)); k = int(rng.integers(1, 12))
        X, Y = data(n, m, g, int(rng.integers(0, 3)))
        ix = nabo_amd.KnnIndex(n, g, metric=metric).set_ref(Y)
        for rep in range(3):
            mask = (rng.random(n) < rng.choice([0.0, 0.3, 0.97])).astype(np.uint8)
            if mask.all():
                mask[0] = 0
            ix.set_mask(mask if mask.any() else None)
            gi, gd = ix.query(X, k)
            oi, od = oracle.knn(X, Y, k, metric, 0.25, ref_mask=mask, nthreads=16)
            if not (np.array_equal(gi, oi) and np.array_equal(gd, od, equal_nan=True)):
                fail("set_mask case %d rep %d n=%d m=%d g=%d k=%d metric=%d" % (case, rep, n, m, g, k, metric))
        ix.close()
    elif kind == "sequence":
        # one resident index, several queries of different shapes / list widths / modes: workspace reuse
        n = int(rng.choice([300, 4000, 25000])); g = int(rng.integers(1, 100)); metric = int(rng.integers(0, 3))
        _, Y = data(n, 1, g, int(rng.integers(0, 3)))
        ix = nabo_amd.KnnIndex(n, g, metric=metric).set_ref(Y)
        for rep in range(4):
            m = int(rng.choice([1, 40, 300, 1200])); k = int(rng.choice([1, 5, 15, 24, 25, 40, 55])); k = min(k, n - 1)
            drop = bool(rng.integers(0, 2)) and m <= n
            X = Y[:m].copy() if drop else pca_like(m, g, seed=int(rng.integers(1, 1 << 30)))
            gi, gd = ix.query(X, k, drop_first=drop)
            oi, od = oracle.knn(X, Y, k, metric, 0.25, drop_first=drop, nthreads=16)
            if not (np.array_equal(gi, oi) and np.array_equal(gd, od, equal_nan=True)):
                fail("sequence case %d rep %d n=%d m=%d g=%d k=%d metric=%d drop=%s" % (case, rep, n, m, g, k, metric, drop))
        ix.close()
    elif kind == "lattice":
        # integer lattices: large shells of exactly equal distances (second-chance pass, exact kernels, tie order)
        gdim = int(rng.integers(2, 6)); side = int(rng.integers(4, 11)); metric = int(rng.choice([0, 0, 2, 1]))
        Y = np.stack(np.meshgrid(*[np.arange(side)] * gdim, indexing="ij"), -1).reshape(-1, gdim).astype(np.float64)
        if len(Y) > 12000:
            Y = Y[rng.choice(len(Y), 12000, replace=False)]
        if metric != 0:
            Y = Y + 1.0                                   # keep away from the zero vector / zero components
        m = int(rng.choice([10, 200, 900])); k = int(rng.choice([3, 15, 24, 30])); k = min(k, len(Y) - 1)
        X = Y[rng.choice(len(Y), m, replace=True)].copy()
        if rng.random() < 0.5:
            os.environ["NABO_SPLITS"] = "1"
        try:
            gi, gd = nabo_amd.knn(X, Y, k, metric=metric, dist_factor=0.25)
        finally:
            os.environ.pop("NABO_SPLITS", None)
        oi, od = oracle.knn(X, Y, k, metric, 0.25, nthreads=16)
        if not (np.array_equal(gi, oi) and np.array_equal(gd, od)):
            fail("lattice case %d dim=%d side=%d n=%d m=%d k=%d metric=%d" % (case, gdim, side, len(Y), m, k, metric))
    elif kind == "pairwise":
        m = int(rng.integers(1, 200)); n = int(rng.integers(1, 400)); g = int(rng.integers(1, 129)); metric = int(rng.integers(0, 3))
        X, Y = data(n, m, g, int(rng.integers(0, 4)))
        f = float(rng.choice([0.1, 0.25, 1.0, 3.0]))
        if not np.array_equal(nabo_amd.pairwise(X, Y, metric, f), oracle.pairwise(X, Y, metric, f)):
            fail("pairwise case %d m=%d n=%d g=%d metric=%d f=%g" % (case, m, n, g, metric, f))
    elif kind == "snn":
        n = int(rng.choice([50, 700, 5000])); k = int(rng.integers(3, 20)); m = int(rng.choice([10, 300, 2000]))   # k = 2: the reference divides by zero
        r_idx = np.stack([rng.choice(n, k, replace=False) for _ in range(n)]) if n > k else None
        if r_idx is None:
            continue
        t_idx = np.stack([rng.choice(n, k, replace=False) for _ in range(m)])
        cnt = nabo_amd.snn_counts(t_idx, r_idx, k)
        t, j, w = oracle.snn_edges(t_idx, r_idx, k)
        tt, ss = np.nonzero(cnt > 0)
        if not (np.array_equal(tt, t) and np.array_equal(t_idx[tt, ss], j)):
            fail("snn case %d n=%d m=%d k=%d" % (case, n, m, k))
    else:
        n_ref = int(rng.choice([5, 60, 700])); n_t = int(rng.choice([20, 300, 2500])); kq = int(rng.integers(1, 9))
        P = int(rng.choice([1, 31, 32, 33, 255, 256, 257, 1000])); bits = int(rng.choice([8, 16, 64]))
        et = np.repeat(np.arange(n_t), kq); er = rng.integers(0, n_ref, n_t * kq)
        w = rng.choice(np.round(np.arange(1, 11) / (20.0 - np.arange(1, 11)), 2), n_t * kq)
        grp = (rng.random(n_t) < rng.choice([0.1, 0.5, 0.9])).astype(np.uint8); grp[0] = 1
        sd = int(rng.integers(0, 1 << 62))
        res = nabo_amd.mapping_score_null(et, er, w, grp, n_ref, n_perm=P, seed=sd, key_bits=bits)
        ref = orc.score_null(et, er, w, grp, n_ref, P, seed=sd, key_bits=bits)
        if not (np.array_equal(res["sizes"], ref["sizes"]) and np.array_equal(res["obs"], ref["obs"]) and
                np.array_equal(res["n_ge"], ref["n_ge"]) and np.allclose(res["null_mean"], ref["null_mean"], rtol=1e-11, atol=1e-11)):
            fail("null case %d n_ref=%d n_t=%d kq=%d P=%d bits=%d" % (case, n_ref, n_t, kq, P, bits))
    counts[kind] += 1
    if case % 50 == 49:
        print("%d cases ok (%.0f s)" % (case + 1, time.time() - t0), flush=True)
print("all cases equal to the oracle: %s (%.0f s)" % (counts, time.time() - t0))
