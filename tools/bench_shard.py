"""One rank's share of the N-way ref-sharded step on ONE GPU: nabo_index_query_candidates of shard 0 (n/N reference
rows, candidates_per_shard entries) against all m target rows.    python tools/bench_shard.py [N] [m n g k] [name=value,...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nabo_amd import _knn  # noqa: E402
from nabo_amd._sharded import shard_bounds, candidates_per_shard  # noqa: E402
from nabo_amd._synth import pca_like  # noqa: E402

opts = {}
for a in [a for a in sys.argv[1:] if "=" in a]:
    sys.argv.remove(a)
    opts.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.split(",")})
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m, n, g, k = (int(a) for a in (sys.argv[2:6] if len(sys.argv) > 5 else (1000000, 1000000, 50, 15)))
Y = pca_like(n, g, seed=1003)
X = pca_like(m, g, seed=2003)
dx = _knn.DeviceBuffer(X.nbytes).upload(X)
lo, hi = shard_bounds(n, N, 0)
ix = _knn.KnnIndex(hi - lo, g, metric=0, ref_index_base=lo, options=opts).set_ref(Y[lo:hi])
nc = candidates_per_shard(k, N, m)
if "cand_slack" not in opts:
    ix.set_option("cand_slack", 3 if nc >= k else 0)       # (the sharded query's rule: sharded.hip, phase 1)
di, dd, db = _knn.DeviceBuffer(m * nc * 8), _knn.DeviceBuffer(m * nc * 8), _knn.DeviceBuffer(m * 8)
ts, ks = [], []
for it in range(6):
    t0 = time.perf_counter()
    ix.query_candidates_device(dx.ptr, m, nc, di.ptr, dd.ptr, db.ptr)
    ts.append((time.perf_counter() - t0) * 1e3)
    ks.append(ix.last_stats())
best = min(range(1, 6), key=lambda i: ts[i])
print(json.dumps({"N": N, "options": opts, "n_cand": nc, "ms": ts[best], "ms_topk": ks[best]["ms_topk"], "ms_refine": ks[best]["ms_refine"],
                  "ms_pack": ks[best]["ms_pack"], "kernel": ix.last_kernel()}))
ix.close()
