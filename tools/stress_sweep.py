"""Randomised parity sweep (longer than the pytest one): random shapes / metrics / masks / duplicates / ties /
scales against the oracle.  Prints the first mismatch and exits 1, or a summary.
    python tools/stress_sweep.py [n_cases] [seed]"""
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
stats = {0: 0, 1: 0, 2: 0}
for case in range(n_cases):
    metric = int(rng.integers(0, 3))
    n = int(rng.choice([1, 2, 5, 31, 33, 64, 100, 257, 1000, 3000, 9000, 30000]))
    m = int(rng.choice([1, 3, 32, 33, 100, 257, 700, 2500]))
    g = int(rng.integers(1, 129)) if rng.random() < 0.7 else int(rng.choice([1, 2, 16, 50, 64, 65, 100, 128]))
    drop = bool(rng.integers(0, 2)) and n >= 2 and m <= n
    kmax = min(n - (1 if drop else 0), 55)
    if kmax < 1:
        continue
    k = int(rng.integers(1, kmax + 1))
    Y = pca_like(n, g, seed=int(rng.integers(1, 1 << 30)))
    X = Y[:m].copy() if drop else pca_like(m, g, seed=int(rng.integers(1, 1 << 30)))
    flavour = int(rng.integers(0, 8))
    mask = None
    f = float(rng.choice([0.25, 0.25, 0.1, 1.0, 2.5]))
    if flavour == 1 and n > 4:
        mask = (rng.random(n) < rng.choice([0.05, 0.5, 0.95])).astype(np.uint8)
        if mask.all():
            mask[int(rng.integers(0, n))] = 0
    elif flavour == 2 and n > 8:
        Y[rng.integers(0, n, n // 3)] = Y[rng.integers(0, n)]
    elif flavour == 3:
        q = float(rng.choice([1.0, 0.5, 4.0]))
        Y = np.round(Y / q) * q
        X = np.round(X / q) * q
    elif flavour == 4:
        sc = 10.0 ** rng.integers(-20, 20)
        Y = Y * sc
        X = X * sc
    elif flavour == 5:
        col = 10.0 ** rng.uniform(-6, 6, g)
        Y = Y * col
        X = X * col
    elif flavour == 6:
        X = X + rng.choice([0.0, 50.0, 1e4])            # far from the references' centre
    elif flavour == 7 and g > 2:
        Y[:, ::2] = 0.0
        X[:, 1::3] = 0.0
    gi, gd = nabo_amd.knn(X, Y, k, metric=metric, dist_factor=f, ref_mask=mask, drop_first=drop)
    oi, od = oracle.knn(X, Y, k, metric, f, ref_mask=mask, drop_first=drop, nthreads=16)
    if not (np.array_equal(gi, oi) and np.array_equal(gd, od, equal_nan=True)):
        bad = np.nonzero((gi != oi).any(1) | ~((gd == od) | (np.isnan(gd) & np.isnan(od))).all(1))[0]
        print("MISMATCH case %d: metric=%d m=%d n=%d g=%d k=%d drop=%s flavour=%d f=%g rows=%s" %
              (case, metric, m, n, g, k, drop, flavour, f, bad[:5]))
        r = bad[0]
        print(" gpu", gi[r], gd[r])
        print(" ora", oi[r], od[r])
        sys.exit(1)
    stats[metric] += 1
    if case % 50 == 49:
        print("%d cases ok (%.0f s)" % (case + 1, time.time() - t0), flush=True)
print("all %d cases equal to the oracle: euclidean %d, canberra %d, cosine %d (%.0f s)" %
      (sum(stats.values()), stats[0], stats[1], stats[2], time.time() - t0))
