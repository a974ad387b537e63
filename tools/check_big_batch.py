"""One 5M-row query in ONE call equals the same rows asked in 1M-row batches (64-bit indexing of the candidate buffers).
    python tools/check_big_batch.py [m n d k metric]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import nabo_amd  # noqa: E402
from nabo_amd._synth import pca_like  # noqa: E402

a = [int(v) for v in sys.argv[1:]]
m, n, d, k, met = a + [5000000, 625000, 100, 50, 2][len(a):]
Y = pca_like(n, d, seed=1004)
X = pca_like(m, d, seed=2004)
ix = nabo_amd.KnnIndex(n, d, metric=met).set_ref(Y)
t0 = time.perf_counter()
bi, bd = ix.query(X, k)
print("one call: %.2f s, fallback rows %d" % (time.perf_counter() - t0, ix.last_stats()["fallback_rows"]), flush=True)
ok = True
for b0 in range(0, m, 1000000):
    b1 = min(m, b0 + 1000000)
    i, dd = ix.query(np.ascontiguousarray(X[b0:b1]), k)
    same = bool(np.array_equal(i, bi[b0:b1]) and np.array_equal(dd, bd[b0:b1]))
    print("  rows %d..%d equal: %s" % (b0, b1, same), flush=True)
    ok = ok and same
sys.exit(0 if ok else 1)
