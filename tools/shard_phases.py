"""Phases of one shard's candidate query (ms): python tools/shard_phases.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nabo_amd  # noqa: E402
from nabo_amd import _knn  # noqa: E402
from nabo_amd._sharded import shard_bounds, candidates_per_shard  # noqa: E402
from nabo_amd._synth import pca_like  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m, n, d, k = 1000000, 1000000, 50, 15
Y = pca_like(n, d, seed=1003)
X = pca_like(m, d, seed=2003)
Ls = candidates_per_shard(k, N, m)
dx = _knn.DeviceBuffer(X.nbytes).upload(X)
lo, hi = shard_bounds(n, N, 0)
sx = nabo_amd.KnnIndex(hi - lo, d, metric=0, ref_index_base=lo).set_ref(Y[lo:hi])
di, dd, db = _knn.DeviceBuffer(m * Ls * 8), _knn.DeviceBuffer(m * Ls * 8), _knn.DeviceBuffer(m * 8)
for rep in range(3):
    sx.query_candidates_device(dx.ptr, m, Ls, di.ptr, dd.ptr, db.ptr)
    st = sx.last_stats()
print("N=%d Ls=%d" % (N, Ls), {k2: round(v, 2) for k2, v in st.items() if k2.startswith("ms_")}, sx.last_kernel(), "splits", st.get("splits"), "workgroups", st.get("workgroups"))
