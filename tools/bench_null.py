"""Permutation-null mapping scores (BASELINE configs[4] extension) on one MI355X: wall time of
nabo_amd.mapping_score_null for a synthetic bipartite graph, with the oracle timed on a small sample.
    python tools/bench_null.py [n_ref n_targets edges_per_target n_perm]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import nabo_amd  # noqa: E402
from oracle import oracle as orc  # noqa: E402

n_ref, n_t, k, P = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (1000000, 1000000, 15, 1000))]
rng = np.random.default_rng(0)
edge_t = np.repeat(np.arange(n_t, dtype=np.int64), k)
centre = rng.integers(0, n_ref, n_t)
edge_r = (centre[:, None] + rng.integers(-50, 50, (n_t, k))).reshape(-1) % n_ref
w = rng.choice(np.round(np.arange(1, 11) / (20.0 - np.arange(1, 11)), 2), n_t * k)
group = (rng.random(n_t) < 0.4).astype(np.uint8)
nabo_amd.mapping_score_null(edge_t[:1000], edge_r[:1000] % 100, w[:1000], group[:100], 100, n_perm=8)   # warm up
t0 = time.perf_counter()
res = nabo_amd.mapping_score_null(edge_t, edge_r, w, group, n_ref, n_perm=P, seed=1)
dt = time.perf_counter() - t0
# oracle on a sub-graph small enough for plain loops
ns_ref, ns_t = 2000, 4000
sel = (edge_t < ns_t)
t0 = time.perf_counter()
ref = orc.score_null(edge_t[sel], edge_r[sel] % ns_ref, w[sel], group[:ns_t], ns_ref, P, seed=1)
tc = time.perf_counter() - t0
chk = nabo_amd.mapping_score_null(edge_t[sel], edge_r[sel] % ns_ref, w[sel], group[:ns_t], ns_ref, n_perm=P, seed=1)
same = bool(np.array_equal(chk["n_ge"], ref["n_ge"]) and np.array_equal(chk["obs"], ref["obs"]) and
            np.array_equal(chk["sizes"], ref["sizes"]))
E = edge_t.shape[0]
print(json.dumps({
    "metric": "permuted edge-label evaluations/s (mapping-score permutation null)", "value": E * P / dt,
    "unit": "edge-permutations/s", "seconds": dt,
    "config": {"workload": "%d ref nodes, %d pooled target cells, %d edges, %d permutations" % (n_ref, n_t, E, P)},
    "bytes_algorithmic": {"label_bits": n_t * ((P + 32) // 32) * 4, "reduction_reads": E * (((P + 32) // 32) * 4 + 16)},
    "smallest_pvalue": float(res["pvalue"].min()), "nodes_p_below_0.01": int((res["pvalue"] < 0.01).sum()),
    "cpu_oracle": {"edge_permutations_per_s": int(sel.sum()) * P / tc, "sample": "%d edges x %d permutations, numpy loops"
                   % (int(sel.sum()), P)},
    "parity": "sample sub-graph: n_ge, obs, sizes equal to the oracle: %s" % same}))
