"""Secondary benchmark: the exact mod-Canberra k-NN kernel on one MI355X,
with the oracle's CPU timing beside it.  Prints one JSON line (same field meanings as bench.py).

    python tools/bench_canberra.py [targets refs dims k]      default 100000 100000 50 15

Roofline: VALU bound (compare / add / divide per dimension; nothing for MFMA).  The dominant kernel is the
lower-bound filter (canberra_f32.hip): a packed-f16 counting pass of 1.5 VALU instructions per pair and dimension
proves "out of window" for most dimensions and drops all but ~1e-3 of the pairs; the fp32 bound (~15 slots per
dimension) runs on the survivors and the float64 expression only on the <= 32 candidates per target.  Algorithmic work per (pair, dimension)
= the reference's 9 operations (nabo/_mapping.py:37-44: abs, sub, abs, mul, cmp, abs, add, add, div/add); peak =
78.6e12 fp32 VALU instructions/s (MI355X: 157.3 TFLOP/s fp32 vector, an FMA counting 2).  With
NABO_CANBERRA_MODE=exact the float64 kernel of canberra.hip runs instead (peak 39.3e12 float64 instructions/s).
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import nabo_amd  # noqa: E402
import oracle  # noqa: E402
from nabo_amd._synth import pca_like  # noqa: E402

m, n, d, k = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (100000, 100000, 50, 15))]
Y = pca_like(n, d, 1003)
X = pca_like(m, d, 2003)
ix = nabo_amd.KnnIndex(n, d, metric=nabo_amd.MOD_CANBERRA, dist_factor=0.25).set_ref(Y)
ix.query(X[:1000], k)
steps = 3
t0 = time.perf_counter()
for _ in range(steps):
    gi, gd = ix.query(X, k)
dt = (time.perf_counter() - t0) / steps
st = ix.last_stats()
ix.close()
rows = np.random.default_rng(0).choice(m, 128, replace=False)
cores = max(1, min(os.cpu_count() or 1, 16))
t0 = time.perf_counter()
oi, od = oracle.knn(X[rows], Y, k, oracle.MOD_CANBERRA, 0.25, nthreads=cores)
tc = time.perf_counter() - t0
assert np.array_equal(gi[rows], oi) and np.array_equal(gd[rows], od), "GPU result differs from the oracle"
t_k = st["ms_topk"] * 1e-3
alg_ops = 9.0 * m * n * d
exact = os.environ.get("NABO_CANBERRA_MODE") == "exact"
peak = 39.3 if exact else 78.6
print(json.dumps({
    "metric": "cell-pair distances/s (mod-Canberra k-NN, dist_factor 0.25)", "value": m * n / dt,
    "unit": "cell-pair distances/s", "n_gpus": 1, "ms_per_step": dt * 1e3, "higher_is_better": True,
    "dtype": "f64" if exact else "f32", "data": "synthetic",
    "config": {"workload": "%dk ref x %dk target, d=%d, k=%d, modified Canberra" % (n // 1000, m // 1000, d, k)},
    "roofline": {"bound": "valu-f64" if exact else "valu-f32", "achieved": alg_ops / t_k / 1e12, "peak": peak,
                 "unit": "T ops/s", "frac": alg_ops / t_k / (peak * 1e12),
                 "kernel": "canberra_topk_kernel (float64)" if exact else "cbf_filter_kernel (f16 count + fp32 lower bound)",
                 "kernel_ms": st["ms_topk"]},
    "phases_ms": {kk_: st[kk_] for kk_ in ("ms_pack", "ms_topk", "ms_refine", "ms_fallback", "ms_total")},
    "uncertified_rows_resolved_exactly": st["fallback_rows"],
    "cpu_baseline": {"value": len(rows) * n / tc, "unit": "cell-pair distances/s", "cores": cores, "kind": "port",
                     "sample": "%d targets x %d refs, d=%d, OpenMP" % (len(rows), n, d)},
    "parity": "128 sampled rows bit-equal to the oracle (indices and distances)"}))
