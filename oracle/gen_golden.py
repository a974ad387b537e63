#!/opt/conda/bin/python3.9
"""Golden-vector generator: runs the REFERENCE's own nabo/_mapping.py + nabo/_graph.py.

TEST INFRASTRUCTURE ONLY.  Executed in the build container (where /root/reference is
mounted) with /opt/conda/bin/python3.9 (h5py 3.3, numpy 1.26, networkx 2.6):

    /opt/conda/bin/python3.9 oracle/gen_golden.py [--only kernels,mapping_small,dup,c1,minis]

It never travels to / runs on the GPU box; only its outputs (tests/golden/*.npz: inputs
and expected outputs, no reference source) are committed.

Loader recipe: the reference modules are loaded BY FILE PATH
(nabo/__init__.py drags in seaborn/natsort).  Two ordinary compatibility stand-ins:
  * `numba` is absent/ABI-broken here -> a stand-in module whose `jit` is the identity
    decorator, i.e. the reference's kernel bodies (`_euclidean_dist`
    nabo/_mapping.py:16-26, `_mod_canberra_dist` :29-45) run in the interpreter on
    IEEE doubles in the written order;
  * `np.float` (used at nabo/_mapping.py:118) was removed in numpy>=1.24 -> alias to float.
"""
import argparse
import importlib.util
import os
import sys
import tempfile
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("NABO_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from nabo_amd._synth import pca_like, digest  # noqa: E402


def load_reference():
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (a[0] if len(a) == 1 and callable(a[0]) and not k else (lambda f: f))
    sys.modules["numba"] = nb
    if not hasattr(np, "float"):
        np.float = float
    mods = {}
    for name in ("_mapping", "_graph"):
        spec = importlib.util.spec_from_file_location("nabo_ref" + name, os.path.join(REF, "nabo", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["_mapping"], mods["_graph"]


def write_pca_h5(fn, grp, names, data):
    """The a12 input contract: one 1-D float64 dataset per cell (nabo/_dataset.py:1028)."""
    import h5py
    with h5py.File(fn, "w") as h5:
        g = h5.create_group(grp)
        for n, v in zip(names, data):
            g.create_dataset(n, data=v)


def read_rows(h5, uid, cells, n_keep_idx, n_keep_dist):
    """order rows [:n_keep_idx] and the distances at the first n_keep_dist of them."""
    idx = np.stack([h5[uid + "_sortedDist"][c][:n_keep_idx] for c in cells])
    dist = np.stack([h5[uid + "_dist"][c][:][h5[uid + "_sortedDist"][c][:n_keep_dist]] for c in cells])
    return idx, dist


def tie_flags(h5, uid, cells, k_chk):
    """rows whose first k_chk+1 sorted distances contain an exact tie (reference order unstable there)."""
    out = np.zeros(len(cells), dtype=bool)
    for i, c in enumerate(cells):
        o = h5[uid + "_sortedDist"][c][:k_chk + 1]
        d = h5[uid + "_dist"][c][:][o]
        out[i] = bool(np.any(d[1:] == d[:-1]))
    return out


def read_graph(h5, uid, node_names):
    """edge list of a <uid>_graph group (the a10 wire format, nabo/_mapping.py:252-273)."""
    src, dst, w, wraw = [], [], [], []
    grp = h5[uid + "_graph"]
    nodes = [n for n in grp]
    for n in nodes:
        for row in grp[n]:
            src.append(n)
            dst.append(row[0].decode("ascii"))
            wraw.append(row[1].decode("ascii"))
            w.append(float(row[1].decode("ascii")))
    return (np.array(nodes), np.array(src), np.array(dst), np.array(w, dtype=np.float64), np.array(wraw))


def gen_kernels(mp):
    """(1) literal a1 / a2 outputs on small tiles, several g and dist_factor."""
    out = {}
    for d in (7, 30, 50):
        x = pca_like(40, d, seed=11 + d)
        y = pca_like(56, d, seed=29 + d)
        out["x_%d" % d] = x
        out["y_%d" % d] = y
        de = np.empty((40, 56), dtype=np.float64)
        mp._euclidean_dist(x, y, de)
        out["euclid_%d" % d] = de
        for f in (0.1, 0.25, 1.0):
            dc = np.empty((40, 56), dtype=np.float64)
            mp._mod_canberra_dist(x, y, dc, f)
            out["canberra_%d_%s" % (d, str(f).replace(".", "p"))] = dc
    np.savez_compressed(os.path.join(OUT, "kernels.npz"), **out)
    print("kernels.npz written")


def run_mapping_case(mp, gr, tag, ref, ref_names, targets, params, n_keep_idx, n_keep_dist,
                     store_inputs, idx_dtype=np.int32, with_scores=True, seeds=None):
    """Full Mapping.make_ref_graph + map_target runs; harvest what the hot path produced.

    targets: list of (name, names, data, ignore_ref_cells)
    """
    use_comps, k, dist_factor, chunk = params
    res = {"params": np.array([use_comps, k, chunk], dtype=np.int64), "dist_factor": np.float64(dist_factor)}
    import h5py
    with tempfile.TemporaryDirectory() as td:
        ref_fn = os.path.join(td, "ref_pca.h5")
        write_pca_h5(ref_fn, "data", ref_names, ref)
        map_fn = os.path.join(td, "mapping.h5")
        t0 = time.time()
        m = mp.Mapping(map_fn, "WT", ref_fn, "data", overwrite=True)
        m.set_parameters(use_comps, k, dist_factor, chunk)
        m.make_ref_graph()
        print("  [%s] make_ref_graph %.1fs" % (tag, time.time() - t0))
        res["ref_cells"] = np.array(m.refCells)
        for (tname, tnames, tdata, ignore) in targets:
            t0 = time.time()
            tfn = os.path.join(td, "t_%s.h5" % tname)
            write_pca_h5(tfn, "data", tnames, tdata)
            m.map_target(tname, tfn, "data", ignore_ref_cells=ignore)
            print("  [%s] map_target %s %.1fs" % (tag, tname, time.time() - t0))
        with h5py.File(map_fn, "r") as h5:
            uid = h5["name_stash/ref_name"][1].decode()
            cells = list(m.refCells)
            idx, dist = read_rows(h5, uid, cells, n_keep_idx, n_keep_dist)
            res["ref_idx"] = idx.astype(idx_dtype)
            res["ref_dist"] = dist
            res["ref_ties"] = tie_flags(h5, uid, cells, k)
            nodes, s, d_, w, wraw = read_graph(h5, uid, None)
            res["ref_graph_nodes"], res["ref_graph_src"], res["ref_graph_dst"] = nodes, s, d_
            res["ref_graph_w"], res["ref_graph_wraw"] = w, wraw
            tn = {r[0].decode(): r[1].decode() for r in h5["name_stash/target_names"][:]} if targets else {}
            for (tname, tnames, tdata, ignore) in targets:
                tuid = tn[tname]
                tcells = [c for c in h5[tuid + "_sortedDist"]]
                res["t_%s_cells" % tname] = np.array(tcells)
                idx, dist = read_rows(h5, tuid, tcells, n_keep_idx, n_keep_dist)
                res["t_%s_idx" % tname] = idx.astype(idx_dtype)
                res["t_%s_dist" % tname] = dist
                res["t_%s_ties" % tname] = tie_flags(h5, tuid, tcells, k)
                res["t_%s_ignore" % tname] = np.array(ignore if ignore else [], dtype="U32")
                # where the ignored refs ended up in the full order row of the first target cell
                if ignore:
                    full = h5[tuid + "_sortedDist"][tcells[0]][:]
                    ign_idx = [cells.index(c) for c in ignore]
                    res["t_%s_ignore_pos" % tname] = np.array(sorted(int(np.where(full == i)[0][0]) for i in ign_idx))
                    res["t_%s_order_len" % tname] = np.int64(len(full))
                nodes, s, d_, w, wraw = read_graph(h5, tuid, None)
                res["t_%s_graph_nodes" % tname], res["t_%s_graph_src" % tname] = nodes, s
                res["t_%s_graph_dst" % tname], res["t_%s_graph_w" % tname] = d_, w
                res["t_%s_graph_wraw" % tname] = wraw
            res["ref_order_len"] = np.int64(len(h5[uid + "_sortedDist"][cells[0]]))
        if with_scores and targets:
            g = gr.Graph()
            g.load_from_h5(map_fn, "WT", "reference")
            for (tname, _, _, _) in targets:
                g.load_from_h5(map_fn, tname, "target")
                sc = g.get_mapping_score(tname)
                keys = sorted(sc.keys())
                res["score_%s_nodes" % tname] = np.array(keys)
                res["score_%s_vals" % tname] = np.array([sc[x] for x in keys], dtype=np.float64)
            res["graph_n_nodes"] = np.int64(g.number_of_nodes())
            res["graph_n_edges"] = np.int64(g.number_of_edges())
    if store_inputs:
        res["ref"] = ref
        res["ref_names"] = np.array(ref_names)
        for (tname, tnames, tdata, _) in targets:
            res["t_%s_data" % tname] = tdata
            res["t_%s_names" % tname] = np.array(tnames)
    else:
        res["ref_digest"] = np.array(digest(ref))
        for (tname, tnames, tdata, _) in targets:
            res["t_%s_digest" % tname] = np.array(digest(tdata))
    if seeds is not None:
        res["seeds"] = np.array(seeds, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print("%s.npz written" % tag)


def gen_mapping_small(mp, gr):
    """(2)(3)(5)(6): 400 ref x {300,250} targets, 20 comps stored, use_comps=15, k=11, chunk=64
    (non-multiple chunking), un-padded cell names (pins lexicographic HDF5 name order)."""
    ref = pca_like(400, 20, seed=1001)
    t1 = pca_like(300, 20, seed=2001)
    t2 = pca_like(250, 20, seed=2002)
    rn = ["R%d" % i for i in range(400)]
    n1 = ["A%d" % i for i in range(300)]
    n2 = ["B%d" % i for i in range(250)]
    ignore = ["R5", "R10", "R77", "R399", "R123"]
    run_mapping_case(mp, gr, "mapping_small", ref, rn,
                     [("ME", n1, t1, None), ("IG", n2, t2, ignore)],
                     (15, 11, 0.25, 64), 32, 32, store_inputs=True)


def gen_dup(mp, gr):
    """(4) duplicate reference cells: documents the positional `[1:]` self-drop
    (nabo/_mapping.py:142) and an exact-tie row."""
    ref = pca_like(60, 12, seed=1003)
    ref[7] = ref[3]
    ref[41] = ref[40]
    rn = ["R%03d" % i for i in range(60)]
    t = pca_like(30, 12, seed=2003)
    t[4] = ref[3]          # a target identical to a duplicated ref
    tn = ["T%03d" % i for i in range(30)]
    run_mapping_case(mp, gr, "dup", ref, rn, [("TG", tn, t, ["R005", "R010"])],
                     (10, 5, 0.25, 16), 59, 59, store_inputs=True)


def gen_c1(mp, gr):
    """BASELINE.json configs[0]: 3k ref x 3k target, d=30, k=11, chunk 500, dist_factor .25."""
    ref = pca_like(3000, 30, seed=1001)
    tgt = pca_like(3000, 30, seed=2001)
    rn = ["R%04d" % i for i in range(3000)]
    tn = ["T%04d" % i for i in range(3000)]
    run_mapping_case(mp, gr, "c1_3k", ref, rn, [("ME", tn, tgt, None)],
                     (30, 11, 0.25, 500), 16, 12, store_inputs=False, idx_dtype=np.int16,
                     with_scores=True, seeds=[1001, 2001])


MINIS = [
    # tag, n_ref, n_target, comps stored, use_comps, k, dist_factor, chunk_size, n_ignored, name style
    ("mini_0", 150, 120, 12, 12, 5, 0.25, 50, 0, "pad"),
    ("mini_1", 180, 90, 25, 9, 7, 0.1, 37, 6, "plain"),
    ("mini_2", 120, 140, 8, 8, 11, 1.0, 1000, 3, "pad"),
    ("mini_3", 200, 60, 30, 30, 3, 0.5, 64, 0, "mixed"),
    ("mini_4", 90, 200, 16, 5, 15, 0.25, 16, 10, "plain"),
    ("mini_5", 160, 100, 40, 22, 9, 2.0, 33, 2, "pad"),
]


def gen_minis(mp, gr):
    """Small full runs over a spread of parameters (use_comps < stored comps, k from 3 to 15, dist_factor from
    0.1 to 2, chunk sizes that do and do not divide the cell counts, ignore lists, three naming styles)."""
    for i, (tag, nr, nt, comps, uc, k, f, chunk, nign, style) in enumerate(MINIS):
        ref = pca_like(nr, comps, seed=5000 + i)
        tgt = pca_like(nt, comps, seed=6000 + i)
        if style == "pad":
            rn, tn = ["R%04d" % j for j in range(nr)], ["T%04d" % j for j in range(nt)]
        elif style == "plain":
            rn, tn = ["R%d" % j for j in range(nr)], ["c%d" % j for j in range(nt)]
        else:
            rn = [("cell-%d" % j if j % 3 else "AAC%dGT" % j) for j in range(nr)]
            tn = ["t.%d" % (j * 7) for j in range(nt)]
        rng = np.random.default_rng(7000 + i)
        ignore = [rn[j] for j in sorted(rng.choice(nr, nign, replace=False))] if nign else None
        run_mapping_case(mp, gr, tag, ref, rn, [("TG", tn, tgt, ignore)], (uc, k, f, chunk), min(32, nr - 1), min(32, nr - 1),
                         store_inputs=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="kernels,mapping_small,dup")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    mp, gr = load_reference()
    todo = a.only.split(",")
    if "kernels" in todo:
        gen_kernels(mp)
    if "mapping_small" in todo:
        gen_mapping_small(mp, gr)
    if "dup" in todo:
        gen_dup(mp, gr)
    if "c1" in todo:
        gen_c1(mp, gr)
    if "minis" in todo:
        gen_minis(mp, gr)


if __name__ == "__main__":
    main()
