"""Reference-STYLE end-to-end CPU run of BASELINE configs[0] (3k ref x 3k target, d=30, k=11): the data flow of
nabo/_mapping.py with the C oracle's kernels in place of numba's.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and nothing else runs it).  It exists because
BASELINE.md section 3 promises the number: what the reference's own layout costs on the host cores of the GPU box,
next to nabo_amd.Mapping on the same files.  It is a restatement, not the reference: no tqdm, no networkx, no graph
repair (nabo/_mapping.py:203-249 is not timed), single thread like the reference's numba kernels (:16,:29).

  reference step                                         here
  per-cell PCA datasets, `[:use_comps]` (:105,:113)      same layout, read per cell
  tile loop over chunk_size x chunk_size (:98-130)       same loop; tile kernel = oracle.pairwise (1 thread)
  row scatter into N_t dense (N_r,) float64 datasets     same (fancy-index write per target row and tile)
  masked full argsort per row, `[1:]` for ref (:135-146) numpy.ma argsort, one dataset per cell
  SNN with Python sets over per-cell datasets (:186-198) same, edges kept in a dict of dicts
  one (n,2) dataset per node (:252-273)                  same

    /opt/conda/bin/python3.9 oracle/c1_end_to_end.py [n_ref n_target d k chunk] [--gpu]

Prints one JSON line: seconds per phase for make_ref_graph + map_target; with --gpu also nabo_amd.Mapping on the same
input files (needs an MI355X).
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import h5py  # noqa: E402

import oracle  # noqa: E402
from nabo_amd._synth import pca_like  # noqa: E402


def write_cells(fn, names, data):
    with h5py.File(fn, "w") as h5:
        g = h5.create_group("data")
        for c, v in zip(names, data):
            g.create_dataset(c, data=v)


def distances(out, t_fn, r_fn, ref_cells, dist_grp, order_grp, use_comps, chunk, intra, f, tm):
    t0 = time.perf_counter()
    with h5py.File(t_fn, "r") as th, h5py.File(r_fn, "r") as rh:
        tg, rg = th["data"], rh["data"]
        t_cells = list(tg)
        dd = out.create_group(dist_grp)
        for c in t_cells:
            dd.create_dataset(c, shape=(len(ref_cells),), dtype=np.float64)
        for a in range(0, len(t_cells), chunk):
            tc = t_cells[a:a + chunk]
            tx = np.array([tg[c][:use_comps] for c in tc])
            for b in range(0, len(ref_cells), chunk):
                cols = list(range(b, min(b + chunk, len(ref_cells))))
                ry = np.array([rg[ref_cells[j]][:use_comps] for j in cols])
                t1 = time.perf_counter()
                tile = oracle.pairwise(tx, ry, oracle.EUCLIDEAN if intra else oracle.MOD_CANBERRA, f, nthreads=1)
                tm["kernel"] += time.perf_counter() - t1
                for row, c in zip(tile, tc):
                    dd[c][cols] = row
    tm["dist_total"] += time.perf_counter() - t0
    t0 = time.perf_counter()
    og = out.create_group(order_grp)
    mask = np.zeros(len(ref_cells), dtype=bool)
    for c in dd:
        o = np.argsort(np.ma.array(dd[c][:], mask=mask))
        og.create_dataset(c, data=o[1:] if intra else o)
    tm["sort"] += time.perf_counter() - t0
    return t_cells


def snn(out, order_grp, ref_order_grp, ref_cells, t_cells, suffix, k, graph_grp, tm):
    t0 = time.perf_counter()
    og, rg = out[order_grp], out[ref_order_grp]
    adj = {c + "_" + suffix: {} for c in t_cells}
    factor = 2 * (k - 1)
    n_edges = 0
    for c in t_cells:
        a = set(og[c][:k])
        for j in a:
            s = len(a.intersection(rg[ref_cells[j]][:k]))
            if s > 0:
                w = round(s / (factor - s), 2)
                adj[c + "_" + suffix][ref_cells[j] + "_WT"] = w
                if suffix == "WT":
                    adj[ref_cells[j] + "_WT"][c + "_WT"] = w
                n_edges += 1
    tm["snn"] += time.perf_counter() - t0
    t0 = time.perf_counter()
    gg = out.create_group(graph_grp)
    for node, nb in adj.items():
        gg.create_dataset(node, data=[(v.encode("ascii"), w) for v, w in nb.items()])
    tm["dump"] += time.perf_counter() - t0
    return n_edges


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_ref, n_t, d, k, chunk = [int(v) for v in args] + [3000, 3000, 30, 11, 500][len(args):]
    ref, tgt = pca_like(n_ref, d, 1001), pca_like(n_t, d, 2001)
    rn, tn = ["R%04d" % i for i in range(n_ref)], ["T%04d" % i for i in range(n_t)]
    res = {"workload": "%d ref x %d target, d=%d, k=%d, chunk_size=%d, dist_factor=0.25 (BASELINE configs[0])"
                       % (n_ref, n_t, d, k, chunk), "cores": 1}
    with tempfile.TemporaryDirectory() as td:
        r_fn, t_fn = os.path.join(td, "ref.h5"), os.path.join(td, "tgt.h5")
        write_cells(r_fn, rn, ref)
        write_cells(t_fn, tn, tgt)
        tm = {"kernel": 0.0, "dist_total": 0.0, "sort": 0.0, "snn": 0.0, "dump": 0.0}
        t0 = time.perf_counter()
        with h5py.File(os.path.join(td, "cpu_mapping.h5"), "w") as out:
            with h5py.File(r_fn, "r") as rh:
                ref_cells = list(rh["data"])
            rc = distances(out, r_fn, r_fn, ref_cells, "r_dist", "r_order", d, chunk, True, 0.25, tm)
            e1 = snn(out, "r_order", "r_order", ref_cells, rc, "WT", k, "r_graph", tm)
            t_ref = time.perf_counter() - t0
            tc = distances(out, t_fn, r_fn, ref_cells, "t_dist", "t_order", d, chunk, False, 0.25, tm)
            e2 = snn(out, "t_order", "r_order", ref_cells, tc, "ME", k, "t_graph", tm)
        total = time.perf_counter() - t0
        res["cpu_reference_style"] = {"seconds": total, "make_ref_graph_s": t_ref, "map_target_s": total - t_ref,
                                      "phases_s": {a: round(b, 3) for a, b in tm.items()},
                                      "pairs_per_s": (n_ref * n_ref + n_t * n_ref) / total,
                                      "edges": [e1, e2],
                                      "file_mb": os.path.getsize(os.path.join(td, "cpu_mapping.h5")) / 1e6}
        if "--gpu" in sys.argv:
            import io
            from contextlib import redirect_stdout
            import nabo_amd
            buf = io.StringIO()
            t0 = time.perf_counter()
            with redirect_stdout(buf):
                m = nabo_amd.Mapping(os.path.join(td, "gpu_mapping.h5"), "WT", r_fn, "data", overwrite=True)
                m.set_parameters(d, k, 0.25, chunk)
                m.make_ref_graph()
                t_ref = time.perf_counter() - t0
                m.map_target("ME", t_fn, "data")
            total = time.perf_counter() - t0
            res["gpu_mapping"] = {"seconds": total, "make_ref_graph_s": t_ref, "map_target_s": total - t_ref,
                                  "pairs_per_s": (n_ref * n_ref + n_t * n_ref) / total,
                                  "file_mb": os.path.getsize(os.path.join(td, "gpu_mapping.h5")) / 1e6,
                                  "note": "nabo_amd.Mapping on the same files, graph repair included"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
