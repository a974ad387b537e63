"""Times the exact mod-Canberra k-NN kernel (diagnostic)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nabo_amd
from nabo_amd._synth import pca_like
m, n, d, k = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (100000, 100000, 50, 15))]
Y = pca_like(n, d, 1003); X = pca_like(m, d, 2003)
ix = nabo_amd.KnnIndex(n, d, metric=nabo_amd.MOD_CANBERRA, dist_factor=0.25).set_ref(Y)
ix.query(X[:1000], k)
t0 = time.perf_counter(); gi, gd = ix.query(X, k); dt = time.perf_counter() - t0
st = ix.last_stats()
print("canberra %dx%dx%d k=%d: %.3f s wall, topk kernel %.1f ms, %.3g pairs/s, splits=%d" % (m, n, d, k, dt, st["ms_topk"], m * n / (st["ms_topk"] * 1e-3), st["splits"]))
