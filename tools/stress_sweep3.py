"""Randomised parity sweep aimed at the PASS CHAIN of the Euclidean / cosine filter (one-product first pass in its three
geometries, seeded pass, f16x3 / fp32 pass, 64-entry lists, exact kernels): random shapes up to g = 128 and k = 56, data
whose one-product bound is weak (clusters far from the centre, huge / tiny scales, quantised values, duplicates), random
switches that move rows between the passes (list lengths, geometry pins, splits, links of the chain switched off), several
queries per index (the weak-bound memory), masks.  Every result must equal the oracle's bits whichever pass answered.
    python tools/stress_sweep3.py [n_cases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import nabo_amd  # noqa: E402
import oracle  # noqa: E402
from nabo_amd._synth import pca_like  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
SWITCHES = ["NABO_L2C_GEO", "NABO_LKEEP", "NABO_COARSE_SLACK", "NABO_SEEDED_PASS", "NABO_COARSE_ADAPT", "NABO_SPLITS",
            "NABO_WIDE_RETRY", "NABO_L2_MODE", "NABO_COSINE_CENTRE", "NABO_TAIL_SPLIT"]
passes = {"first": 0, "seeded": 0, "second": 0, "wide": 0, "exact": 0}
for case in range(n_cases):
    metric = int(rng.choice([0, 0, 0, 2]))
    n = int(rng.choice([40, 300, 2000, 9000, 30000, 70000]))
    m = int(rng.choice([1, 33, 300, 1500, 4000]))
    g = int(rng.integers(1, 129)) if rng.random() < 0.6 else int(rng.choice([29, 30, 50, 61, 62, 64, 93, 94, 100, 125, 126]))
    drop = bool(rng.integers(0, 2)) and m <= n
    kmax = min(n - (1 if drop else 0), 56)
    k = int(rng.integers(1, kmax + 1)) if rng.random() < 0.5 else int(min(kmax, rng.choice([11, 15, 20, 24, 25, 28, 29, 50, 55])))
    centres = np.random.default_rng(int(rng.integers(1, 1 << 30))).standard_normal((6, g)) * float(rng.choice([0.0, 0.5, 10.0, 40.0, 400.0]))
    lab = rng.integers(0, 6, size=n)
    spread = float(rng.choice([0.5, 1.0]))
    Y = centres[lab] + pca_like(n, g, seed=int(rng.integers(1, 1 << 30))) * spread * 0.2
    X = Y[:m].copy() if drop else centres[rng.integers(0, 6, size=m)] + pca_like(m, g, seed=int(rng.integers(1, 1 << 30))) * spread * 0.2
    flavour = int(rng.integers(0, 6))
    mask = None
    if flavour == 1 and n > 4:
        mask = (rng.random(n) < rng.choice([0.05, 0.5, 0.9])).astype(np.uint8)
        if mask.all():
            mask[int(rng.integers(0, n))] = 0
    elif flavour == 2 and n > 8:
        Y[rng.integers(0, n, n // 3)] = Y[rng.integers(0, n)]
    elif flavour == 3:
        q = float(rng.choice([0.25, 1.0]))
        Y, X = np.round(Y / q) * q, np.round(X / q) * q
    elif flavour == 4:
        sc = 10.0 ** rng.integers(-15, 15)
        Y, X = Y * sc, X * sc
    env = {}
    if rng.random() < 0.7:
        env["NABO_L2C_GEO"] = str(rng.choice(["a", "b", "c"]))
    if rng.random() < 0.4:
        env["NABO_LKEEP"] = str(int(k + (1 if drop else 0) + rng.integers(0, 4)))
    if rng.random() < 0.3:
        env["NABO_COARSE_SLACK"] = str(int(rng.integers(-6, 7)))
    if rng.random() < 0.25:
        env["NABO_SEEDED_PASS"] = "0"
    if rng.random() < 0.25:
        env["NABO_COARSE_ADAPT"] = "0"
    if rng.random() < 0.4:
        env["NABO_SPLITS"] = str(int(rng.choice([1, 1, 2, 5])))
    if rng.random() < 0.15:
        env["NABO_WIDE_RETRY"] = "0"
    if rng.random() < 0.15:
        env["NABO_L2_MODE"] = str(rng.choice(["f16x3", "f32", "f16x1h"]))
    if rng.random() < 0.2:
        env["NABO_COSINE_CENTRE"] = "0"
    for s in SWITCHES:
        os.environ.pop(s, None)
    os.environ.update(env)
    oi, od = oracle.knn(X, Y, k, metric, 0.25, ref_mask=mask, drop_first=drop, nthreads=16)
    ix = nabo_amd.KnnIndex(n, g, metric=metric).set_ref(Y, ref_mask=mask)
    for rep in range(int(rng.choice([1, 1, 2, 3]))):
        gi, gd = ix.query(X, k, drop_first=drop)
        st = ix.last_stats()
        if not (np.array_equal(gi, oi) and np.array_equal(gd, od, equal_nan=True)):
            bad = np.nonzero((gi != oi).any(1) | ~((gd == od) | (np.isnan(gd) & np.isnan(od))).all(1))[0]
            print("MISMATCH case %d rep %d: metric=%d m=%d n=%d g=%d k=%d drop=%s flavour=%d env=%s kernel=%s stats=%s rows=%s" %
                  (case, rep, metric, m, n, g, k, drop, flavour, env, ix.last_kernel(), st, bad[:5]))
            print(" gpu", gi[bad[0]], gd[bad[0]])
            print(" ora", oi[bad[0]], od[bad[0]])
            sys.exit(1)
        passes["first"] += m - st["seeded_pass_rows"] - (st["second_pass_rows"] if not st["seeded_pass_rows"] else 0)
        passes["seeded"] += st["seeded_pass_rows"]
        passes["second"] += st["second_pass_rows"]
        passes["wide"] += st["wide_list_rows"]
        passes["exact"] += st["fallback_rows"]
    ix.close()
    if case % 25 == 24:
        print("%d cases ok (%.0f s) rows by pass %s" % (case + 1, time.time() - t0, passes), flush=True)
for s in SWITCHES:
    os.environ.pop(s, None)
print("all %d cases equal to the oracle; rows sent on by pass: %s (%.0f s)" % (n_cases, passes, time.time() - t0))
