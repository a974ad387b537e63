"""Cost of a small nabo_index_query (the sharded protocol's second round solves a handful of refused rows on every shard):
m rows against n references, device pointers.    python tools/bench_small_query.py [n] [g] [k]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from nabo_amd import _knn  # noqa: E402
from nabo_amd._synth import pca_like  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
g = int(sys.argv[2]) if len(sys.argv) > 2 else 50
k = int(sys.argv[3]) if len(sys.argv) > 3 else 15
Y = pca_like(n, g, seed=1003)
ix = _knn.KnnIndex(n, g, metric=0).set_ref(Y)
for m in (1, 16, 64, 256, 1000, 4000, 16000):
    X = pca_like(m, g, seed=2003 + m)
    dx = _knn.DeviceBuffer(X.nbytes).upload(X)
    di, dd = _knn.DeviceBuffer(m * k * 8), _knn.DeviceBuffer(m * k * 8)
    ts = []
    for it in range(5):
        t0 = time.perf_counter()
        ix.query_device(dx.ptr, m, k, False, di.ptr, dd.ptr)
        ts.append((time.perf_counter() - t0) * 1e3)
    st = ix.last_stats()
    print(json.dumps({"n": n, "m": m, "ms_best": round(min(ts[1:]), 3), "ms_first": round(ts[0], 3),
                      "stats": {a: (round(b, 3) if isinstance(b, float) else b) for a, b in st.items()}, "kernel": ix.last_kernel()[:24]}))
ix.close()
