#!/opt/conda/bin/python3.9
"""Golden vectors for the two consumers of the mapping file: `Graph.get_mapping_score` with its options
 and `Graph.load_from_h5` reading a file written by THIS build's writer (a10).

TEST INFRASTRUCTURE ONLY; build container only (imports the reference from /root/reference by file path,
same loader as oracle/gen_golden.py; needs no GPU):

    /opt/conda/bin/python3.9 oracle/gen_golden_graph.py

Writes tests/golden/score_options.npz, tests/golden/score_by_cluster.npz and tests/golden/dump_roundtrip.npz (data
only, no reference source).

1. score_options: the reference's Mapping is run on the `mapping_small` inputs (400 refs, targets ME and IG),
   the reference's Graph loads the file and `get_mapping_score` is called with a spread of options; every call
   and its result is stored as JSON.
2. dump_roundtrip: nabo_amd.Mapping._dump_graph writes the reference graph and the ME graph from the same edges
   (no GPU involved: edge lists come from the golden order rows through the host half of calc_snn and the C
   oracle's shared-neighbour counts); the reference's own Graph.load_from_h5 (nabo/_graph.py:31-116) then reads
   OUR file and the REFERENCE's file; node lists, edge sets, weights and adjacency order are compared and the
   verdict is stored together with a digest of the datasets our writer produced (tests re-create the file and
   compare the digest, so the fixture stays tied to the writer it judged).
"""
import hashlib
import io
import json
import os
import sys
import tempfile
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
OUT = os.path.join(REPO, "tests", "golden")

from gen_golden import load_reference, write_pca_h5  # noqa: E402

SCORE_CALLS = [
    {},
    {"min_weight": 0.1},
    {"min_weight": 0.1, "weighted": False},
    {"min_score": 5.0},
    {"min_score": 5.0, "all_nodes": False},
    {"score_multiplier": 1},
    {"sorted_names_only": True},
    {"sorted_names_only": True, "min_score": 5.0},
    {"sorted_names_only": True, "top_n_only": 25},
    {"sorted_names_only": True, "top_n_only": 25, "min_score": 1e9},
    {"sorted_names_only": True, "remove_suffix": True, "top_n_only": 10},
    {"remove_suffix": True, "all_nodes": False, "min_score": 5.0},
    {"ignore_nodes": "IGN"},
    {"ignore_nodes": "IGN", "sorted_names_only": True, "top_n_only": 30},
    {"include_nodes": "INC"},
    {"include_nodes": "INC", "all_nodes": False, "min_score": 2.0},
    {"include_nodes": "INC_BIG"},
    {"include_nodes": "INC_BIG", "weighted": False, "sorted_names_only": True, "min_score": 3.0},
]


def node_sets(tnodes, rng):
    """named node subsets used by SCORE_CALLS (incl. names that do not exist: the reference drops them)"""
    ign = [tnodes[i] for i in sorted(rng.choice(len(tnodes), 40, replace=False))] + ["nope_ME", "R5_WT"]
    inc = [tnodes[i] for i in sorted(rng.choice(len(tnodes), 35, replace=False))] + ["ghost_ME"]
    big = [tnodes[i] for i in sorted(rng.choice(len(tnodes), 260, replace=False))]
    return {"IGN": ign, "INC": inc, "INC_BIG": big}


def graph_digest(h5, grp):
    """sha256 over (node name, dataset dtype, raw bytes) in HDF5 name order"""
    h = hashlib.sha256()
    for n in h5[grp]:
        d = h5[grp][n]
        h.update(n.encode())
        h.update(d.dtype.str.encode())
        h.update(np.ascontiguousarray(d[()]).tobytes())
    return h.hexdigest()


def graph_state(g):
    """what a loaded reference Graph holds: nodes in order, adjacency in order with weights"""
    return {"nodes": list(g.nodes()), "adj": {n: [(v, d["weight"]) for v, d in g.adj[n].items()] for n in g.nodes()}}


def main():
    import h5py
    mp, gr = load_reference()
    gold = np.load(os.path.join(OUT, "mapping_small.npz"))
    uc, k, chunk = [int(v) for v in gold["params"]]
    f = float(gold["dist_factor"])
    ref, rn = gold["ref"], [str(x) for x in gold["ref_names"]]
    with tempfile.TemporaryDirectory() as td:
        ref_fn = os.path.join(td, "ref.h5")
        write_pca_h5(ref_fn, "data", rn, ref)
        map_fn = os.path.join(td, "mapping.h5")
        buf = io.StringIO()
        with redirect_stdout(buf):
            m = mp.Mapping(map_fn, "WT", ref_fn, "data", overwrite=True)
            m.set_parameters(uc, k, f, chunk)
            m.make_ref_graph()
            for t in ("ME", "IG"):
                tfn = os.path.join(td, "t_%s.h5" % t)
                write_pca_h5(tfn, "data", [str(x) for x in gold["t_%s_names" % t]], gold["t_%s_data" % t])
                ign = [str(x) for x in gold["t_%s_ignore" % t]]
                m.map_target(t, tfn, "data", ignore_ref_cells=ign if ign else None)
        g = gr.Graph()
        g.load_from_h5(map_fn, "WT", "reference")
        g.load_from_h5(map_fn, "ME", "target")
        g.load_from_h5(map_fn, "IG", "target")
        # ---- 1. score options ------------------------------------------------------------
        sets = node_sets(list(g.targetNodes["ME"]), np.random.default_rng(77))
        calls = []
        for kw in SCORE_CALLS:
            real = {a: (sets[b] if isinstance(b, str) and b in sets else b) for a, b in kw.items()}
            with redirect_stdout(buf):
                res = g.get_mapping_score("ME", **real)
            calls.append({"kwargs": kw, "result": res})
        errors = []
        for kw, exc in (({"ignore_nodes": ["a"], "include_nodes": ["b"]}, "ValueError"),
                        ({"sorted_names_only": True, "top_n_only": 401}, "ValueError")):
            try:
                g.get_mapping_score("ME", **kw)
                errors.append({"kwargs": kw, "raises": None})
            except Exception as e:      # noqa: BLE001
                errors.append({"kwargs": kw, "raises": type(e).__name__})
                assert type(e).__name__ == exc
        try:
            g.get_mapping_score("NOPE")
        except Exception as e:          # noqa: BLE001
            errors.append({"target": "NOPE", "raises": type(e).__name__})
        np.savez_compressed(os.path.join(OUT, "score_options.npz"),
                            calls=np.array(json.dumps(calls)), node_sets=np.array(json.dumps(sets)),
                            errors=np.array(json.dumps(errors)), ref_nodes=np.array(list(g.refNodes)))
        print("score_options.npz written (%d calls)" % len(calls))
        # ---- 1b. by_cluster (nabo/_graph.py:602-606,655-671): a fresh Graph (no node has a cluster), then clusters
        # imported through import_clusters (:334-356) for part of the reference nodes, unknown names included
        rng = np.random.default_rng(78)
        rnodes = list(g.refNodes)
        cdict = {rnodes[i]: int(rng.integers(1, 6)) for i in sorted(rng.choice(len(rnodes), 300, replace=False))}
        cdict["ghost_WT"] = 3
        bc = []
        with redirect_stdout(buf):
            bc.append({"clusters": None, "kwargs": {}, "result": g.get_mapping_score("ME", by_cluster=True)})
            g.import_clusters(cdict)
            for kw in ({}, {"min_weight": 0.1, "weighted": False}, {"include_nodes": "INC_BIG", "score_multiplier": 1}):
                real = {a: (sets[b] if isinstance(b, str) and b in sets else b) for a, b in kw.items()}
                bc.append({"clusters": "CDICT", "kwargs": kw, "result": g.get_mapping_score("ME", by_cluster=True, **real)})
        np.savez_compressed(os.path.join(OUT, "score_by_cluster.npz"), calls=np.array(json.dumps(bc)),
                            cdict=np.array(json.dumps(cdict)), node_sets=np.array(json.dumps(sets)))
        print("score_by_cluster.npz written (%d calls)" % len(bc))
        if "--by-cluster-only" in sys.argv:
            return

        # ---- 2. our writer, the reference's reader ---------------------------------------
        import nabo_amd
        import oracle
        from nabo_amd._mapping import snn_edges_from_counts
        ref_state = graph_state(g)
        with h5py.File(map_fn, "r") as h5:
            ruid = h5["name_stash/ref_name"][1].decode()
            tuids = {r[0].decode(): r[1].decode() for r in h5["name_stash/target_names"][:]}
            digest_reference_file = {"WT": graph_digest(h5, ruid + "_graph"), "ME": graph_digest(h5, tuids["ME"] + "_graph"),
                                     "IG": graph_digest(h5, tuids["IG"] + "_graph")}
        our_fn = os.path.join(td, "ours.h5")
        with redirect_stdout(buf):
            om = nabo_amd.Mapping(our_fn, "WT", ref_fn, "data", overwrite=True)
            om.set_parameters(uc, k, f, chunk)
        assert list(om.refCells) == [str(c) for c in gold["ref_cells"]]
        r_idx = gold["ref_idx"][:, :k].astype(np.int64)

        def counts(t_idx):
            ot, oj, w = oracle.snn_edges(t_idx, r_idx, k)          # edges in slot order: rebuild the [m,k] table
            cnt = np.zeros(t_idx.shape, dtype=np.int32)
            tab = {round(s / (2 * (k - 1) - s), 2): s for s in range(1, k + 1)}
            for t, j, ww in zip(ot, oj, w):
                cnt[t, int(np.nonzero(t_idx[t] == j)[0][0])] = tab[float(ww)]
            return cnt

        fixw = 0.5 / ((2 * (k - 1)) - 0.5)
        pos = {c + "_WT": i for i, c in enumerate(om.refCells)}
        extra, seen = [], set()
        for s_, d_, w_ in zip(gold["ref_graph_src"], gold["ref_graph_dst"], gold["ref_graph_w"]):
            if float(w_) == fixw:
                a, b = pos[str(s_)], pos[str(d_)]
                if (min(a, b), max(a, b)) not in seen:
                    seen.add((min(a, b), max(a, b)))
                    extra.append((a, b, fixw))
        et, ej, ew = snn_edges_from_counts(r_idx, counts(r_idx), k)
        om._dump_graph(om._refGraphGrpName, list(om.refCells), "WT", True, et, ej, ew, extra)
        uids = {}
        for t in ("ME", "IG"):
            t_idx = gold["t_%s_idx" % t][:, :k].astype(np.int64)
            et, ej, ew = snn_edges_from_counts(t_idx, counts(t_idx), k)
            om._stash_target_name(t)
            uids[t] = om._nameStash[t]
            om._dump_graph(uids[t] + "_graph", [str(c) for c in gold["t_%s_cells" % t]], t, False, et, ej, ew, [])
        g2 = gr.Graph()
        g2.load_from_h5(our_fn, "WT", "reference")
        g2.load_from_h5(our_fn, "ME", "target")
        g2.load_from_h5(our_fn, "IG", "target")
        ours_state = graph_state(g2)
        ties = {t: {str(c) + "_" + t for c, tt in zip(gold["t_%s_cells" % t], gold["t_%s_ties" % t]) if tt} for t in ("ME", "IG")}
        tied = ties["ME"] | ties["IG"]
        same_nodes = ours_state["nodes"] == ref_state["nodes"]
        same_adj = all(ours_state["adj"][n] == ref_state["adj"][n] for n in ref_state["nodes"] if n not in tied)
        sc_ref = g.get_mapping_score("ME")
        sc_ours = g2.get_mapping_score("ME")
        with h5py.File(our_fn, "r") as h5:
            digest_ours = {"WT": graph_digest(h5, om._refGraphGrpName), "ME": graph_digest(h5, uids["ME"] + "_graph"),
                           "IG": graph_digest(h5, uids["IG"] + "_graph")}
        np.savez_compressed(os.path.join(OUT, "dump_roundtrip.npz"),
                            reference_reads_same_nodes=np.bool_(same_nodes),
                            reference_reads_same_adjacency_in_order=np.bool_(same_adj),
                            reference_scores_equal=np.bool_(sc_ref == sc_ours),
                            n_nodes=np.int64(g2.number_of_nodes()), n_edges=np.int64(g2.number_of_edges()),
                            n_nodes_reference_file=np.int64(g.number_of_nodes()),
                            n_edges_reference_file=np.int64(g.number_of_edges()),
                            digest_ours=np.array(json.dumps(digest_ours)),
                            digest_reference_file=np.array(json.dumps(digest_reference_file)),
                            rows_with_ties=np.array(sorted(tied)),
                            extra=np.array(extra, dtype=np.float64))
        print("dump_roundtrip.npz written: same nodes %s, same adjacency (order + weights) %s, scores equal %s, "
              "dataset digests equal: %s" % (same_nodes, same_adj, sc_ref == sc_ours,
                                             {t: digest_ours[t] == digest_reference_file[t] for t in digest_ours}))


if __name__ == "__main__":
    main()
