"""End-to-end `Mapping` (HDF5 in, HDF5 out) timing with per-phase breakdown -- the reference-style
line of SURVEY section 8d.  Needs h5py (run with /opt/conda/bin/python3.9 in this image).
    /opt/conda/bin/python3.9 tools/bench_mapping.py [n_ref n_target d k layout [dense|per_cell [graph_layout]]]"""
import os, sys, time, tempfile, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import h5py
import nabo_amd
from nabo_amd import _mapping
from nabo_amd._synth import pca_like

n_ref, n_tgt, d, k = (int(a) for a in (sys.argv[1:5] + ["3000", "3000", "30", "11"][len(sys.argv) - 1:]))
layout = sys.argv[5] if len(sys.argv) > 5 else "per_cell"
dense = len(sys.argv) > 6 and sys.argv[6] == "dense"
graph_layout = sys.argv[7] if len(sys.argv) > 7 else "per_node"
tmp = tempfile.mkdtemp(prefix="nabo_bm_")
T = {}


def timed(obj, name):
    f = getattr(obj, name)
    def w(*a, **kw):
        t0 = time.perf_counter()
        r = f(*a, **kw)
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, w)


def write_pca(fn, prefix, Z):
    if dense:
        nabo_amd.write_dense_pca(fn, "data", ["%s%07d" % (prefix, i) for i in range(Z.shape[0])], Z)
        return
    with h5py.File(fn, "w") as h5:
        g = h5.create_group("data")
        for i in range(Z.shape[0]):
            g.create_dataset("%s%07d" % (prefix, i), data=Z[i])


t0 = time.perf_counter()
write_pca(os.path.join(tmp, "ref.h5"), "R", pca_like(n_ref, d + 5, seed=1001))
write_pca(os.path.join(tmp, "tgt.h5"), "T", pca_like(n_tgt, d + 5, seed=2001))
t_gen = time.perf_counter() - t0

for name in ("_read_group_matrix",):
    timed(_mapping, name)
for name in ("calc_dist", "calc_snn", "_dump_graph", "_store_knn", "_repair_round"):
    timed(_mapping.Mapping, name)
timed(_mapping, "snn_edges")

t0 = time.perf_counter()
m = nabo_amd.Mapping(os.path.join(tmp, "map.h5"), "WT", os.path.join(tmp, "ref.h5"), "data", overwrite=True, layout=layout, graph_layout=graph_layout)
m.set_parameters(d, k, 0.25, 500)
t1 = time.perf_counter()
m.make_ref_graph()
t2 = time.perf_counter()
m.map_target("ME", os.path.join(tmp, "tgt.h5"), "data")
t3 = time.perf_counter()
out = {"n_ref": n_ref, "n_target": n_tgt, "d": d, "k": k, "layout": layout, "input": "dense" if dense else "per_cell", "graph_layout": graph_layout, "s_write_inputs": round(t_gen, 2),
       "s_init": round(t1 - t0, 2), "s_make_ref_graph": round(t2 - t1, 2), "s_map_target": round(t3 - t2, 2),
       "phases_s": {a: round(b, 3) for a, b in T.items()},
       "mapping_h5_MB": round(os.path.getsize(os.path.join(tmp, "map.h5")) / 1e6, 1)}
print(json.dumps(out))
import shutil
shutil.rmtree(tmp)
