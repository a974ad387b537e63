"""Per-rank cost of the ref-sharded query on ONE GPU: for N in (2,4,8) time one shard's
nabo_index_query (local certification, kk entries) and nabo_index_query_candidates (global
certification, candidates_per_shard entries) against n/N reference rows, all m target rows.
    python tools/bench_shard.py [m n g k]"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from nabo_amd import _knn, _lib
from nabo_amd._synth import pca_like
from nabo_amd._sharded import shard_bounds, candidates_per_shard
candidates_per_shard = ShardedKnn.candidates_per_shard

m, n, g, k = (int(a) for a in (sys.argv[1:5] or (1000000, 1000000, 50, 15)))
kk = k + 1
Y = pca_like(n, g, seed=1)
X = Y if m == n else pca_like(m, g, seed=2)
dx = _knn.DeviceBuffer(X.nbytes).upload(X)
out = {}
for N in (1, 2, 4, 8):
    lo, hi = shard_bounds(n, N, 0)
    ix = _knn.KnnIndex(hi - lo, g, metric=0, ref_index_base=lo).set_ref(Y[lo:hi])
    di, dd, db = _knn.DeviceBuffer(m * 32 * 8), _knn.DeviceBuffer(m * 32 * 8), _knn.DeviceBuffer(m * 8)
    res = {}
    for name in ("local", "cand"):
        if name == "cand" and N == 1:
            continue
        nc = candidates_per_shard(kk, N, m)
        ts = []
        for it in range(4):
            t0 = time.perf_counter()
            if name == "local":
                ix.query_device(dx.ptr, m, kk, False, di.ptr, dd.ptr)
            else:
                ix.query_candidates_device(dx.ptr, m, nc, di.ptr, dd.ptr, db.ptr)
            ts.append((time.perf_counter() - t0) * 1e3)
        st = ix.last_stats()
        res[name] = {"ms": min(ts[1:]), "stats": st, "n_cand": nc if name == "cand" else kk}
        if name == "cand":
            b = db.download((m,), np.float64)
            res[name]["unknown_bound_rows"] = int((b == -np.inf).sum())
    out[N] = res
    print(N, json.dumps(res), flush=True)
    ix.close()
