"""Randomised parity checks at sizes where the launch geometry changes (tail rounds, splits, many workgroup
rounds): full GPU runs, rows sampled against the oracle (boundaries of the launches included).
    python tools/stress_large.py [n_cases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import nabo_amd  # noqa: E402
import oracle  # noqa: E402
from nabo_amd._synth import pca_like  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
for case in range(n_cases):
    metric = int(rng.integers(0, 3))
    m = int(rng.choice([32768 + 5, 70001, 131072, 131072 + 257, 200003, 262144 + 31]))
    n = int(rng.choice([2000, 9001, 40000]))
    g = int(rng.choice([3, 20, 50, 64, 100]))
    k = int(rng.choice([1, 11, 15, 24, 25, 40]))
    drop = bool(rng.integers(0, 2)) and m <= n
    Y = pca_like(n, g, seed=int(rng.integers(1, 1 << 30)))
    X = pca_like(m, g, seed=int(rng.integers(1, 1 << 30)))
    mask = None
    if rng.random() < 0.3:
        mask = (rng.random(n) < 0.2).astype(np.uint8)
    gi, gd = nabo_amd.knn(X, Y, k, metric=metric, dist_factor=0.25, ref_mask=mask, drop_first=drop)
    edges = [0, 255, 256, 32767, 32768, 131071, 131072, m - 1]
    rows = np.unique(np.concatenate([rng.choice(m, 160, replace=False), [e for e in edges if e < m],
                                     np.arange(max(0, m - 40), m)]))
    oi, od = oracle.knn(X[rows], Y, k, metric, 0.25, ref_mask=mask, drop_first=drop, nthreads=16)
    if not (np.array_equal(gi[rows], oi) and np.array_equal(gd[rows], od, equal_nan=True)):
        bad = rows[np.nonzero((gi[rows] != oi).any(1))[0]]
        print("MISMATCH case %d metric=%d m=%d n=%d g=%d k=%d rows=%s" % (case, metric, m, n, g, k, bad[:6]))
        sys.exit(1)
    if not (np.diff(gd, axis=1) >= 0).all():
        print("MISMATCH case %d: unsorted rows (metric=%d m=%d n=%d g=%d k=%d)" % (case, metric, m, n, g, k))
        sys.exit(1)
    print("case %d ok: metric=%d m=%d n=%d g=%d k=%d (%.0f s)" % (case, metric, m, n, g, k, time.time() - t0), flush=True)
print("all %d large cases equal to the oracle on the sampled rows" % n_cases)
