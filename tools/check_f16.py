"""Parity + timing of the experimental f16x3 mode against the oracle (run with NABO_L2_MODE=f16x3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nabo_amd, oracle
from nabo_amd._synth import pca_like
print("mode", os.environ.get("NABO_L2_MODE"))
for (m, n, g, k, drop) in [(33, 64, 16, 8, False), (257, 4097, 50, 15, False), (1000, 1000, 15, 11, True),
                           (3000, 3000, 100, 23, True), (130, 5000, 64, 30, False), (200, 3000, 128, 50, True),
                           (5000, 50000, 50, 15, False)]:
    Y = pca_like(n, g, seed=1000 + n + g)
    X = Y[:m].copy() if drop else pca_like(m, g, seed=2000 + m + g)
    ix = nabo_amd.KnnIndex(n, g, metric=0).set_ref(Y)
    gi, gd = ix.query(X, k, drop_first=drop)
    st = ix.last_stats(); ix.close()
    oi, od = oracle.knn(X, Y, k, 0, drop_first=drop, nthreads=8)
    print((m, n, g, k, drop), "idx", np.array_equal(gi, oi), "dist", np.array_equal(gd, od), "fallback", st["fallback_rows"], "ms_topk %.2f" % st["ms_topk"])
for shift, scale in ((1e4, 1.0), (0.0, 1e-6), (-3e5, 1e3)):
    Y = pca_like(5000, 30, seed=61) * scale + shift; X = pca_like(300, 30, seed=62) * scale + shift
    ix = nabo_amd.KnnIndex(5000, 30, metric=0).set_ref(Y); gi, gd = ix.query(X, 15); st = ix.last_stats(); ix.close()
    oi, od = oracle.knn(X, Y, 15, 0, nthreads=8)
    print("shift/scale", shift, scale, np.array_equal(gi, oi), np.array_equal(gd, od), "fallback", st["fallback_rows"])
