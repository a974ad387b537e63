"""First contact of a new filter kernel with the GPU: a few small parity cases against the oracle, each printed as it
finishes (run under `timeout`: a hang must not take the box's whole limit)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import nabo_amd  # noqa: E402
import oracle  # noqa: E402
from nabo_amd._synth import pca_like  # noqa: E402

ok = True
for (m, n, g, k, drop) in [(100, 1000, 30, 11, True), (2000, 30000, 30, 20, True), (257, 4097, 50, 15, False),
                           (1000, 1000, 15, 11, True), (5000, 200000, 50, 15, False), (33, 64, 16, 8, False), (64, 3000, 7, 5, False)]:
    Y = pca_like(n, g, seed=1000 + n + g)
    X = Y[:m].copy() if drop else pca_like(m, g, seed=2000 + m + g)
    t0 = time.time()
    ix = nabo_amd.KnnIndex(n, g, metric=0).set_ref(Y)
    gi, gd = ix.query(X, k, drop_first=drop)
    st, kern = ix.last_stats(), ix.last_kernel()
    ix.close()
    oi, od = oracle.knn(X, Y, k, 0, drop_first=drop, nthreads=8)
    same = bool(np.array_equal(gi, oi) and np.array_equal(gd, od))
    ok = ok and same and st["fallback_rows"] == 0
    print("m=%d n=%d g=%d k=%d drop=%s: same=%s fallback=%d splits=%d  %.2fs  %s" % (m, n, g, k, drop, same, st["fallback_rows"],
          st["splits"], time.time() - t0, kern), flush=True)
sys.exit(0 if ok else 1)
