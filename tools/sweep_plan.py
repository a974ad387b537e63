"""How a query should be cut: the same m x n query under pinned index options (reference splits, kernel geometry, tournament
length), on-stream phase times of each and a check that every setting returns the same bits.
    python tools/sweep_plan.py m n g k [metric] ["splits=2,l2c_geo=0" ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from nabo_amd import _knn  # noqa: E402
from nabo_amd._synth import pca_like  # noqa: E402

m, n, g, k = (int(a) for a in sys.argv[1:5])
metric = int(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5].isdigit() else 0
sets = [a for a in sys.argv[5:] if "=" in a or a == "default"] or ["default"]
Y = pca_like(n, g, seed=1003)
X = pca_like(m, g, seed=2003)
dx = _knn.DeviceBuffer(X.nbytes).upload(X)
di, dd = _knn.DeviceBuffer(m * k * 8), _knn.DeviceBuffer(m * k * 8)
ref = None
for s in sets:
    opts = {} if s == "default" else {a.split("=")[0]: int(a.split("=")[1]) for a in s.split(",")}
    ix = _knn.KnnIndex(n, g, metric=metric, options=opts).set_ref(Y)
    ts = []
    for it in range(6):
        t0 = time.perf_counter()
        ix.query_device(dx.ptr, m, k, False, di.ptr, dd.ptr)
        ts.append((time.perf_counter() - t0) * 1e3)
    st = ix.last_stats()
    gi, gd = di.download((m, k), np.int64), dd.download((m, k), np.float64)
    if ref is None:
        ref = (gi, gd)
    same = bool(np.array_equal(gi, ref[0]) and np.array_equal(gd, ref[1]))
    print(json.dumps({"set": s, "ms_best": round(min(ts[1:]), 3), "same_bits": same,
                      "stats": {a: (round(b, 3) if isinstance(b, float) else b) for a, b in st.items()},
                      "kernel": ix.last_kernel()[:40]}), flush=True)
    ix.close()
