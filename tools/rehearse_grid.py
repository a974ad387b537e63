"""One rank's share of the 8-GPU step under the 2-D layout (R reference pieces x 8/R target slices), measured on ONE GPU:
the candidate query of piece 0 for slice 0 -- n/R references x m/(8/R) target rows.     python tools/rehearse_grid.py [R [N]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nabo_amd  # noqa: E402
from nabo_amd import _knn  # noqa: E402
from nabo_amd._sharded import shard_bounds, candidates_per_shard  # noqa: E402
from nabo_amd._synth import pca_like  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
m, n, d, k = 1000000, 1000000, 50, 15
Y = pca_like(n, d, seed=1003)
X = pca_like(m, d, seed=2003)
ms = m // (N // R)
Ls = candidates_per_shard(k, R, m)
dx = _knn.DeviceBuffer(X[:ms].nbytes).upload(X[:ms])
lo, hi = shard_bounds(n, R, 0)
sx = nabo_amd.KnnIndex(hi - lo, d, metric=0, ref_index_base=lo).set_ref(Y[lo:hi])
di, dd, db = _knn.DeviceBuffer(ms * Ls * 8), _knn.DeviceBuffer(ms * Ls * 8), _knn.DeviceBuffer(ms * 8)
for rep in range(3):
    sx.query_candidates_device(dx.ptr, ms, Ls, di.ptr, dd.ptr, db.ptr)
    st = sx.last_stats()
print("N=%d as %d reference pieces x %d target slices: a rank queries %d references x %d rows, Ls=%d:" % (N, R, N // R, hi - lo, ms, Ls),
      {k2: round(v, 2) for k2, v in st.items() if k2.startswith("ms_")})
