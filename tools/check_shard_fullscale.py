"""The 8-way ref-sharded protocol at the bench's full size, emulated on ONE GPU (shards one after the other, the
exchange is a host stack): how many rows does the global certificate accept, and do they equal the unsharded
answer?    python tools/check_shard_fullscale.py [N] [m n d k]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import nabo_amd  # noqa: E402
from nabo_amd import _knn  # noqa: E402
from nabo_amd._sharded import shard_bounds, candidates_per_shard  # noqa: E402
from nabo_amd._synth import pca_like  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m, n, d, k = [int(v) for v in (sys.argv[2:6] if len(sys.argv) > 5 else (1000000, 1000000, 50, 15))]
Y = pca_like(n, d, seed=1003)
X = pca_like(m, d, seed=2003)
ix = nabo_amd.KnnIndex(n, d, metric=0).set_ref(Y)
ri, rd = ix.query(X, k)
ix.close()
Ls = candidates_per_shard(k, N, m)
dx = _knn.DeviceBuffer(X.nbytes).upload(X)
pi = np.empty((N, m, Ls), dtype=np.int64)
pd = np.empty((N, m, Ls), dtype=np.float64)
pb = np.empty((N, m), dtype=np.float64)
t_shard = []
for r in range(N):
    lo, hi = shard_bounds(n, N, r)
    sx = nabo_amd.KnnIndex(hi - lo, d, metric=0, ref_index_base=lo).set_ref(Y[lo:hi])
    di, dd, db = _knn.DeviceBuffer(m * Ls * 8), _knn.DeviceBuffer(m * Ls * 8), _knn.DeviceBuffer(m * 8)
    sx.query_candidates_device(dx.ptr, m, Ls, di.ptr, dd.ptr, db.ptr)
    t0 = time.perf_counter()
    sx.query_candidates_device(dx.ptr, m, Ls, di.ptr, dd.ptr, db.ptr)
    t_shard.append((time.perf_counter() - t0) * 1e3)
    pi[r], pd[r], pb[r] = di.download((m, Ls), np.int64), dd.download((m, Ls), np.float64), db.download((m,), np.float64)
    sx.close()
    di.free(); dd.free(); db.free()
dpi, dpd = _knn.DeviceBuffer(pi.nbytes).upload(pi), _knn.DeviceBuffer(pd.nbytes).upload(pd)
doi, dod = _knn.DeviceBuffer(m * k * 8), _knn.DeviceBuffer(m * k * 8)
_knn.merge_topk_device(dpi.ptr, dpd.ptr, N, m, Ls, k, False, doi.ptr, dod.ptr)
mi, md = doi.download((m, k), np.int64), dod.download((m, k), np.float64)
dk = md[:, k - 1]
ok = (mi[:, k - 1] >= 0) & (dk * dk * (1 + 1e-12) < pb.min(axis=0))
same = np.array_equal(mi[ok], ri[ok]) and np.array_equal(md[ok], rd[ok])
print("N=%d, %d candidates per shard: %d of %d rows certified globally (%d need the second round); certified rows equal "
      "the unsharded answer: %s; per-shard candidate query %.1f ms (min %.1f, max %.1f)"
      % (N, Ls, int(ok.sum()), m, int((~ok).sum()), same, float(np.mean(t_shard)), min(t_shard), max(t_shard)))
sys.exit(0 if same else 1)
