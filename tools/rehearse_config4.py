"""BASELINE configs[4] (5M ref x 5M target, d=100, k=50, cosine, 8 GPUs + 1000-permutation null) rehearsed on ONE GPU:
  (a) one rank's share of a step -- its 625k-reference shard against all 5M targets (candidate lists + bounds), timed;
  (b) the whole 8-shard protocol for a sample of targets, shard after shard, merged and certified exactly as
      nabo_amd/_dist.py does it, compared with the unsharded index over all 5M references;
  (c) one rank's share of the permutation null: 125 of the 1000 permutations over the 5M x 50 edges of the mapping.
    python tools/rehearse_config4.py [n m d k world sample batch]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import nabo_amd  # noqa: E402
from nabo_amd import _knn  # noqa: E402
from nabo_amd._sharded import shard_bounds, candidates_per_shard  # noqa: E402
from nabo_amd._synth import pca_like  # noqa: E402

a = [int(v) for v in sys.argv[1:]]
n, m, d, k, N, sample, batch = a + [5000000, 5000000, 100, 50, 8, 20000, 1000000][len(a):]
MET = nabo_amd.COSINE
t0 = time.perf_counter()
Y = pca_like(n, d, seed=1004)
X = pca_like(m, d, seed=2004)
print("synthetic PCA embeddings: %.0f s" % (time.perf_counter() - t0), flush=True)
Ls = candidates_per_shard(k, N, m)
out = {"config": {"workload": "%d ref x %d target, d=%d, k=%d, cosine, rank 0 of %d" % (n, m, d, k, N)},
       "candidates_per_shard": Ls}

# (a) rank 0's shard against all targets, in batches (what bench.py's step does at N=8)
lo, hi = shard_bounds(n, N, 0)
sx = nabo_amd.KnnIndex(hi - lo, d, metric=MET, ref_index_base=lo).set_ref(Y[lo:hi])
di, dd, db = _knn.DeviceBuffer(batch * Ls * 8), _knn.DeviceBuffer(batch * Ls * 8), _knn.DeviceBuffer(batch * 8)
dx = _knn.DeviceBuffer(batch * d * 8)
ms = []
for b0 in range(0, m, batch):
    b1 = min(m, b0 + batch)
    xb = np.ascontiguousarray(X[b0:b1])
    dx.upload(xb)
    if b0 == 0:
        sx.query_candidates_device(dx.ptr, b1 - b0, Ls, di.ptr, dd.ptr, db.ptr)      # warm-up
    t0 = time.perf_counter()
    sx.query_candidates_device(dx.ptr, b1 - b0, Ls, di.ptr, dd.ptr, db.ptr)
    ms.append((time.perf_counter() - t0) * 1e3)
    print("  targets %d..%d: %.1f ms" % (b0, b1, ms[-1]), flush=True)
sx.close()
for b in (di, dd, db, dx):
    b.free()
pairs = float(hi - lo) * m
out["shard_step"] = {"ms": sum(ms), "pairs_per_s": pairs / (sum(ms) * 1e-3), "refs": hi - lo, "targets": m,
                     "batches": len(ms)}

# (b) all shards for a sample of targets vs the unsharded index
rng = np.random.default_rng(5)
rows = np.sort(rng.choice(m, sample, replace=False))
Xs = np.ascontiguousarray(X[rows])
ix = nabo_amd.KnnIndex(n, d, metric=MET).set_ref(Y)
t0 = time.perf_counter()
ri, rd = ix.query(Xs, k)
out["unsharded_sample_ms"] = (time.perf_counter() - t0) * 1e3
ix.close()
dx = _knn.DeviceBuffer(Xs.nbytes).upload(Xs)
pi = np.empty((N, sample, Ls), dtype=np.int64)
pd = np.empty((N, sample, Ls), dtype=np.float64)
pb = np.empty((N, sample), dtype=np.float64)
for r in range(N):
    lo, hi = shard_bounds(n, N, r)
    sx = nabo_amd.KnnIndex(hi - lo, d, metric=MET, ref_index_base=lo).set_ref(Y[lo:hi])
    di, dd, db = _knn.DeviceBuffer(sample * Ls * 8), _knn.DeviceBuffer(sample * Ls * 8), _knn.DeviceBuffer(sample * 8)
    sx.query_candidates_device(dx.ptr, sample, Ls, di.ptr, dd.ptr, db.ptr)
    pi[r], pd[r], pb[r] = (di.download((sample, Ls), np.int64), dd.download((sample, Ls), np.float64),
                           db.download((sample,), np.float64))
    sx.close()
    for b in (di, dd, db):
        b.free()
dpi, dpd = _knn.DeviceBuffer(pi.nbytes).upload(pi), _knn.DeviceBuffer(pd.nbytes).upload(pd)
doi, dod = _knn.DeviceBuffer(sample * k * 8), _knn.DeviceBuffer(sample * k * 8)
_knn.merge_topk_device(dpi.ptr, dpd.ptr, N, sample, Ls, k, False, doi.ptr, dod.ptr)
mi, md = doi.download((sample, k), np.int64), dod.download((sample, k), np.float64)
dk = md[:, k - 1]
ok = (mi[:, k - 1] >= 0) & (dk * dk * (1 + 1e-12) < pb.min(axis=0))       # nabo_amd/_dist.py:_query_certified
out["protocol_sample"] = {"rows": sample, "certified": int(ok.sum()), "second_round": int((~ok).sum()),
                          "certified_equal_unsharded": bool(np.array_equal(mi[ok], ri[ok]) and np.array_equal(md[ok], rd[ok]))}

# (c) this rank's share of the permutation null over the mapping's edges
P = 1000 // N
del X
group = (rng.random(m) < 0.4).astype(np.uint8)
# edges of the whole 5M x k mapping: the sampled rows' neighbour lists tiled over all targets (weights as the
# reference's shared-neighbour weights would be: w in {s/(2(k-1)-s)})
reps = -(-m // sample)
e_r = np.tile(ri.reshape(-1), reps)[: m * k]
e_t = np.repeat(np.arange(m, dtype=np.int64), k)
w = rng.choice(np.round(np.arange(1, 11) / (2.0 * (k - 1) - np.arange(1, 11)), 2), m * k)
nabo_amd.mapping_score_null(e_t[:1000], e_r[:1000] % 100, w[:1000], group[:100], 100, n_perm=8)
t0 = time.perf_counter()
res = nabo_amd.mapping_score_null(e_t, e_r, w, group, n, n_perm=P, seed=1)
dt = time.perf_counter() - t0
keep = group[e_t] != 0
sc = nabo_amd.mapping_score_from_edges(n, e_r[keep], w[keep], int(group.sum()))
out["null_obs_equals_mapping_score"] = bool(np.allclose(res["obs"], sc, rtol=1e-11, atol=0))
out["null_share"] = {"permutations": P, "edges": int(e_t.shape[0]), "seconds": dt,
                     "edge_permutations_per_s": e_t.shape[0] * P / dt, "smallest_pvalue": float(res["pvalue"].min())}
print(json.dumps(out))
sys.exit(0 if out["protocol_sample"]["certified_equal_unsharded"] and out["null_obs_equals_mapping_score"] else 1)
