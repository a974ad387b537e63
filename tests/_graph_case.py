"""CPU-side checks of the two consumers of the mapping file, against fixtures the
REFERENCE produced (oracle/gen_golden_graph.py): run by test_graph_consumers.py under an interpreter with h5py.

  * nabo_amd.Mapping._dump_graph, fed the edges the golden order rows imply (shared-neighbour counts from the C
    oracle -- test infrastructure; the product counts them on the GPU), must write datasets whose names, dtypes
    and raw bytes hash to the digest of the file the REFERENCE's Mapping wrote for the same inputs, and to the
    digest of the file the reference's own Graph.load_from_h5 was shown to read identically;
  * nabo_amd.get_mapping_score with every option must return what the reference's Graph.get_mapping_score
    returned on that file.
"""
import hashlib
import io
import json
import os
import sys
import tempfile
from contextlib import redirect_stdout

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import h5py  # noqa: E402

import nabo_amd  # noqa: E402
import oracle  # noqa: E402
from nabo_amd._mapping import snn_edges_from_counts  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


def graph_digest(h5, grp):
    h = hashlib.sha256()
    for n in h5[grp]:
        d = h5[grp][n]
        h.update(n.encode())
        h.update(d.dtype.str.encode())
        h.update(np.ascontiguousarray(d[()]).tobytes())
    return h.hexdigest()


def write_pca(fn, grp, names, data):
    with h5py.File(fn, "w") as h5:
        g = h5.create_group(grp)
        for n, v in zip(names, data):
            g.create_dataset(str(n), data=v)


def build_file(td, gold, graph_layout="per_node", tag="ours"):
    """the mapping file of `mapping_small`, written by this build's writer without a GPU"""
    uc, k, chunk = [int(v) for v in gold["params"]]
    ref_fn = os.path.join(td, "ref.h5")
    write_pca(ref_fn, "data", [str(x) for x in gold["ref_names"]], gold["ref"])
    fn = os.path.join(td, tag + ".h5")
    buf = io.StringIO()
    with redirect_stdout(buf):
        om = nabo_amd.Mapping(fn, "WT", ref_fn, "data", overwrite=True, graph_layout=graph_layout)
        om.set_parameters(uc, k, float(gold["dist_factor"]), chunk)
    r_idx = gold["ref_idx"][:, :k].astype(np.int64)

    def counts(t_idx):
        ot, oj, w = oracle.snn_edges(t_idx, r_idx, k)
        cnt = np.zeros(t_idx.shape, dtype=np.int32)
        tab = {round(s / (2 * (k - 1) - s), 2): s for s in range(1, k + 1)}
        for t, j, ww in zip(ot, oj, w):
            cnt[t, int(np.nonzero(t_idx[t] == j)[0][0])] = tab[float(ww)]
        return cnt

    rt = np.load(os.path.join(GOLD, "dump_roundtrip.npz"))
    extra = [(int(a), int(b), float(w)) for a, b, w in rt["extra"]]
    et, ej, ew = snn_edges_from_counts(r_idx, counts(r_idx), k)
    om._dump_graph(om._refGraphGrpName, list(om.refCells), "WT", True, et, ej, ew, extra)
    uids = {"WT": om._nameStash["WT"]}
    for t in ("ME", "IG"):
        t_idx = gold["t_%s_idx" % t][:, :k].astype(np.int64)
        et, ej, ew = snn_edges_from_counts(t_idx, counts(t_idx), k)
        om._stash_target_name(t)
        uids[t] = om._nameStash[t]
        om._dump_graph(uids[t] + "_graph", [str(c) for c in gold["t_%s_cells" % t]], t, False, et, ej, ew, [])
    return fn, uids, rt


def same(a, b):
    """reference result vs ours: lists equal; dict values within 1e-12 relative (the reference sums a node's
    weights in networkx adjacency order, which for `include_nodes` subsets follows a hash-ordered set)"""
    if isinstance(a, list):
        return a == b
    if set(a) != set(b):
        return False
    return all(abs(a[k] - b[k]) <= 1e-12 * max(1.0, abs(a[k])) for k in a)


def main():
    gold = np.load(os.path.join(GOLD, "mapping_small.npz"))
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fn, uids, rt = build_file(td, gold)
        with h5py.File(fn, "r") as h5:
            mine = {t: graph_digest(h5, uids[t] + "_graph") for t in uids}
        out["digest_equals_fixture"] = mine == json.loads(str(rt["digest_ours"]))
        out["digest_equals_reference_file"] = mine == json.loads(str(rt["digest_reference_file"]))
        out["reference_reader_verdict"] = bool(rt["reference_reads_same_nodes"] and rt["reference_scores_equal"] and
                                               rt["reference_reads_same_adjacency_in_order"] and
                                               int(rt["n_nodes"]) == int(rt["n_nodes_reference_file"]) and
                                               int(rt["n_edges"]) == int(rt["n_edges_reference_file"]))
        so = np.load(os.path.join(GOLD, "score_options.npz"))
        sets = json.loads(str(so["node_sets"]))
        bad = []
        for i, call in enumerate(json.loads(str(so["calls"]))):
            kw = {a: (sets[b] if isinstance(b, str) and b in sets else b) for a, b in call["kwargs"].items()}
            got = nabo_amd.get_mapping_score(fn, "WT", "ME", **kw)
            if type(got) is not type(call["result"]) or not same(call["result"], got):
                bad.append((i, call["kwargs"]))
        out["score_calls_checked"] = len(json.loads(str(so["calls"])))
        out["score_calls_differ"] = bad
        errs = []
        for e in json.loads(str(so["errors"])):
            try:
                if "target" in e:
                    nabo_amd.get_mapping_score(fn, "WT", e["target"])
                else:
                    nabo_amd.get_mapping_score(fn, "WT", "ME", **e["kwargs"])
                errs.append((e, None))
            except Exception as ex:     # noqa: BLE001
                if type(ex).__name__ != e["raises"]:
                    errs.append((e, type(ex).__name__))
        out["score_errors_differ"] = errs
        # by_cluster (nabo/_graph.py:655-671): {cluster: [scores of its reference nodes, in refNodes order]}
        bc = np.load(os.path.join(GOLD, "score_by_cluster.npz"))
        cdict = json.loads(str(bc["cdict"]))
        bsets = json.loads(str(bc["node_sets"]))
        bad = []
        for i, call in enumerate(json.loads(str(bc["calls"]))):
            kw = {a: (bsets[b] if isinstance(b, str) and b in bsets else b) for a, b in call["kwargs"].items()}
            got = nabo_amd.get_mapping_score(fn, "WT", "ME", by_cluster=True, clusters=cdict if call["clusters"] else None, **kw)
            want = call["result"]
            ok = set(got) == set(want) and all(len(got[c]) == len(want[c]) and all(
                abs(a - b) <= 1e-12 * max(1.0, abs(a)) for a, b in zip(want[c], got[c])) for c in want)
            if not ok:
                bad.append((i, call["kwargs"]))
        # columnar graph layout (opt-in extension): same scores straight from the columnar file; expand_graph() rewrites it
        # in the reference's per-node wire format, same digest as the default writer (and thus as the reference's file)
        fn2, uids2, _ = build_file(td, gold, "columnar", "ours_columnar")
        sc1, sc2 = nabo_amd.get_mapping_score(fn, "WT", "ME"), nabo_amd.get_mapping_score(fn2, "WT", "ME")
        out["columnar_graph_same_scores"] = list(sc1) == list(sc2) and all(sc1[n] == sc2[n] for n in sc1)
        for name in ("WT", "ME", "IG"):
            nabo_amd.expand_graph(fn2, name)
        with h5py.File(fn2, "r") as h5:
            mine2 = {t: graph_digest(h5, uids2[t] + "_graph") for t in uids2}
        out["columnar_graph_expands_to_same_digest"] = mine2 == mine
        out["by_cluster_calls_checked"] = len(json.loads(str(bc["calls"])))
        out["by_cluster_calls_differ"] = bad
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
