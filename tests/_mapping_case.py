"""Runs nabo_amd.Mapping end to end on HDF5 inputs built from the golden fixtures and compares
everything the reference wrote for the same inputs.  Executed by test_mapping.py, in-process
when h5py is importable, otherwise under an interpreter that has it.

    python tests/_mapping_case.py validate     # API / validation rules only (no GPU needed)
    python tests/_mapping_case.py small|dup|c1|mini_0..mini_5   # full runs (GPU)
"""
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
from nabo_amd._synth import pca_like  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


def write_pca(fn, grp, names, data):
    with h5py.File(fn, "w") as h5:
        g = h5.create_group(grp)
        for n, v in zip(names, data):
            g.create_dataset(str(n), data=v)


def read_graph_like_reference(fn, name, kind):
    """What Graph.load_from_h5 does with the file (nabo/_graph.py:62-107), as plain sets."""
    with h5py.File(fn, "r") as h5:
        if kind == "reference":
            assert h5["name_stash/ref_name"][0].decode("UTF-8") == name
            uid = h5["name_stash/ref_name"][1].decode("UTF-8")
        else:
            uid = None
            for i in h5["name_stash/target_names"][:]:
                if i[0].decode("UTF-8") == name:
                    uid = i[1].decode("UTF-8")
            assert uid is not None
        grp = h5[uid + "_graph"]
        nodes, edges = [], set()
        for node in grp:
            nodes.append(node)
            for j in grp[node]:
                edges.add((node, j[0].decode("UTF-8"), float(j[1].decode("UTF-8"))))
        return nodes, edges, uid


def read_graph_raw(fn, uid):
    """nodes in HDF5 order and, per node, its rows as stored: (neighbour bytes, weight bytes) in row order"""
    with h5py.File(fn, "r") as h5:
        grp = h5[uid + "_graph"]
        return [(n, [(r[0], r[1]) for r in grp[n][()].tolist()] if grp[n].shape != (0,) else []) for n in grp]


def golden_raw(g, prefix):
    """the same from a golden: what the REFERENCE's file held (oracle/gen_golden.py read_graph keeps file order)"""
    rows = {str(n): [] for n in g[prefix + "_nodes"]}
    for s, d, w in zip(g[prefix + "_src"], g[prefix + "_dst"], g[prefix + "_wraw"]):
        rows[str(s)].append((str(d).encode("ascii"), str(w).encode("ascii")))
    return [(str(n), rows[str(n)]) for n in g[prefix + "_nodes"]]


def golden_edges(g, prefix):
    return {(str(s), str(d), float(w)) for s, d, w in zip(g[prefix + "_src"], g[prefix + "_dst"], g[prefix + "_w"])}


def mapping_score(ref_nodes, t_nodes, t_edges, score_multiplier=1000):
    """get_mapping_score defaults (nabo/_graph.py:632-653): weighted degree of target edges."""
    sc = {n: 0.0 for n in ref_nodes}
    for a, b, w in t_edges:
        if w > 0:
            sc[b] += w
    return {k: score_multiplier * v / len(t_nodes) for k, v in sc.items()}


def run_case(tag):
    g = np.load(os.path.join(GOLD, tag + ".npz"))
    uc, k, chunk = [int(v) for v in g["params"]]
    f = float(g["dist_factor"])
    if tag == "c1_3k":
        ref = pca_like(3000, 30, 1001)
        rn = ["R%04d" % i for i in range(3000)]
        targets = [("ME", ["T%04d" % i for i in range(3000)], pca_like(3000, 30, 2001), None)]
    else:
        ref, rn = g["ref"], list(g["ref_names"])
        targets = []
        for key in g.files:
            if key.startswith("t_") and key.endswith("_data"):
                t = key[2:-5]
                ign = [str(x) for x in g["t_%s_ignore" % t]]
                targets.append((t, list(g["t_%s_names" % t]), g[key], ign if ign else None))
    out = {}
    with tempfile.TemporaryDirectory() as td:
        ref_fn = os.path.join(td, "ref.h5")
        write_pca(ref_fn, "data", rn, ref)
        map_fn = os.path.join(td, "mapping.h5")
        buf = io.StringIO()
        with redirect_stdout(buf):
            m = nabo_amd.Mapping(map_fn, "WT", ref_fn, "data", overwrite=True)
            m.set_parameters(uc, k, f, chunk)
            m.make_ref_graph()
        out["ref_cells_equal"] = list(m.refCells) == [str(c) for c in g["ref_cells"]]
        for (tn, names, data, ign) in targets:
            tfn = os.path.join(td, "t_%s.h5" % tn)
            write_pca(tfn, "data", names, data)
            with redirect_stdout(buf):
                m.map_target(tn, tfn, "data", ignore_ref_cells=ign)
        out["log"] = buf.getvalue()
        # stored neighbour lists == first k entries of the reference's order rows
        with h5py.File(map_fn, "r") as h5:
            uid = h5["name_stash/ref_name"][1].decode()
            idx = np.stack([h5[uid + "_sortedDist"][c][:] for c in m.refCells])
            dist = np.stack([h5[uid + "_dist"][c][:] for c in m.refCells])
        ties = g["ref_ties"]
        out["ref_idx_equal"] = bool(np.array_equal(idx[~ties], g["ref_idx"][~ties][:, :k]))
        out["ref_dist_equal"] = bool(np.array_equal(dist, g["ref_dist"][:, :k]))
        nodes, edges, _ = read_graph_like_reference(map_fn, "WT", "reference")
        ge = golden_edges(g, "ref_graph")
        out["ref_graph_nodes_equal"] = sorted(nodes) == sorted(str(x) for x in g["ref_graph_nodes"])
        out["ref_graph_edges_equal"] = edges == ge
        out["ref_graph_missing"] = sorted(ge - edges)[:5]
        out["ref_graph_extra"] = sorted(edges - ge)[:5]
        # the datasets themselves: node order, per-node ROW order, neighbour names and weight STRINGS as stored
        out["ref_graph_raw_equal"] = read_graph_raw(map_fn, uid) == golden_raw(g, "ref_graph")
        fixw = 0.5 / ((2 * (k - 1)) - 0.5)
        out["n_repair_edges"] = len({e for e in ge if e[2] == fixw}) // 2
        for (tn, names, data, ign) in targets:
            tnodes, tedges, tuid = read_graph_like_reference(map_fn, tn, "target")
            gte = golden_edges(g, "t_%s_graph" % tn)
            tt = g["t_%s_ties" % tn]
            out["t_%s_graph_nodes_equal" % tn] = sorted(tnodes) == sorted(str(x) for x in g["t_%s_graph_nodes" % tn])
            if not tt.any():
                out["t_%s_graph_edges_equal" % tn] = tedges == gte
            else:       # rows with exact ties inside the first k+1: reference order is unstable there
                tied = {str(c) + "_" + tn for c, t in zip(g["t_%s_cells" % tn], tt) if t}
                out["t_%s_graph_edges_equal" % tn] = ({e for e in tedges if e[0] not in tied} ==
                                                      {e for e in gte if e[0] not in tied})
            raw, graw = read_graph_raw(map_fn, tuid), golden_raw(g, "t_%s_graph" % tn)
            tied_n = {str(c) + "_" + tn for c, t in zip(g["t_%s_cells" % tn], tt) if t}
            out["t_%s_graph_raw_equal" % tn] = ([x for x in raw if x[0] not in tied_n] ==
                                               [x for x in graw if x[0] not in tied_n])
            if "score_%s_nodes" % tn in g.files and not tt.any():
                sc = mapping_score(nodes, tnodes, tedges)
                gs = dict(zip([str(x) for x in g["score_%s_nodes" % tn]], g["score_%s_vals" % tn]))
                out["t_%s_score_maxerr" % tn] = float(max(abs(sc[n] - gs[n]) for n in gs))
                sc2 = nabo_amd.get_mapping_score(map_fn, "WT", tn)          # the product's own function
                out["t_%s_score_api_maxerr" % tn] = float(max(abs(sc2[n] - gs[n]) for n in gs))
        # permutation null read straight from the mapping file (needs two mapped samples): its observed score is
        # the reference's mapping score of the sample of interest
        if len(targets) >= 2:
            ta, tb = targets[0][0], targets[1][0]
            nul = nabo_amd.get_mapping_score_null(map_fn, "WT", ta, tb, n_perm=64, seed=5)
            sc_a = nabo_amd.get_mapping_score(map_fn, "WT", ta)
            out["null_obs_is_mapping_score"] = bool(all(abs(nul[n][0] - sc_a[n]) <= 1e-9 * max(1.0, abs(sc_a[n])) for n in sc_a))
            out["null_pvalues_in_range"] = bool(all(1.0 / 65 <= v[1] <= 1.0 for v in nul.values()))
        # use_stored_distances: rebuild the graphs from the stored lists only
        with redirect_stdout(buf):
            m2 = nabo_amd.Mapping(map_fn, "WT", ref_fn, "data")
            m2.set_parameters(uc, k, f, chunk)
            m2.make_ref_graph(use_stored_distances=True)
        _, edges2, _ = read_graph_like_reference(map_fn, "WT", "reference")
        out["stored_distances_same_graph"] = edges2 == edges
        # store_k: distances computed at a SMALLER k keep max(k, store_k) entries per row, so a later, larger k is served
        # from the stored lists like the reference's full rows serve it (nabo/_mapping.py:537-541); without store_k
        # the lists are too short and calc_snn says so
        k_small = k - 3 if k - 3 >= 3 else k          # (k = 2 divides by zero in the reference's weight, :194)
        for layout in ("per_cell", "columnar"):
            map7 = os.path.join(td, "mapping_storek_%s.h5" % layout)
            with redirect_stdout(buf):
                m7 = nabo_amd.Mapping(map7, "WT", ref_fn, "data", overwrite=True, store_k=k + 6, layout=layout)
                m7.set_parameters(uc, k_small, f, chunk)
                m7.make_ref_graph()
                m7 = nabo_amd.Mapping(map7, "WT", ref_fn, "data", store_k=k + 6, layout=layout)      # a later session
                m7.set_parameters(uc, k, f, chunk)
                m7.make_ref_graph(use_stored_distances=True)
            _, edges7, _ = read_graph_like_reference(map7, "WT", "reference")
            out["store_k_serves_larger_k_%s" % layout] = edges7 == edges
        map8 = os.path.join(td, "mapping_nostorek.h5")
        with redirect_stdout(buf):
            m8 = nabo_amd.Mapping(map8, "WT", ref_fn, "data", overwrite=True)
            m8.set_parameters(uc, k_small, f, chunk)
            m8.make_ref_graph()
            m8.set_parameters(uc, k, f, chunk)
        try:
            m8.make_ref_graph(use_stored_distances=True)
            out["no_store_k_raises"] = k_small == k              # (nothing to raise when k could not be lowered)
        except ValueError:
            out["no_store_k_raises"] = k_small < k
        # columnar GRAPH layout (opt-in): same scores straight from the columnar file, and expand_graph() rewrites it in
        # the reference's per-node wire format -- the same datasets, byte for byte, as the default writer's
        map9 = os.path.join(td, "mapping_gcol.h5")
        tn9, names9, data9, ign9 = targets[0]
        with redirect_stdout(buf):
            m9 = nabo_amd.Mapping(map9, "WT", ref_fn, "data", overwrite=True, graph_layout="columnar")
            m9.set_parameters(uc, k, f, chunk)
            m9.make_ref_graph()
            m9.map_target(tn9, os.path.join(td, "t_%s.h5" % tn9), "data", ignore_ref_cells=ign9)
        sc_a = nabo_amd.get_mapping_score(map_fn, "WT", tn9)
        sc_b = nabo_amd.get_mapping_score(map9, "WT", tn9)
        out["columnar_graph_same_scores"] = (list(sc_a) == list(sc_b)) and all(sc_a[n] == sc_b[n] for n in sc_a)
        if len(targets) >= 2:
            with redirect_stdout(buf):
                m9.map_target(targets[1][0], os.path.join(td, "t_%s.h5" % targets[1][0]), "data", ignore_ref_cells=targets[1][3])
            na = nabo_amd.get_mapping_score_null(map_fn, "WT", targets[0][0], targets[1][0], n_perm=32, seed=5)
            nb_ = nabo_amd.get_mapping_score_null(map9, "WT", targets[0][0], targets[1][0], n_perm=32, seed=5)
            out["columnar_graph_same_null"] = na == nb_
        nabo_amd.expand_graph(map9, "WT")
        nabo_amd.expand_graph(map9, tn9)
        _, _, ruid9 = read_graph_like_reference(map9, "WT", "reference")
        _, _, tuid9 = read_graph_like_reference(map9, tn9, "target")
        _, _, ruid0 = read_graph_like_reference(map_fn, "WT", "reference")
        _, _, tuid0 = read_graph_like_reference(map_fn, tn9, "target")
        out["columnar_graph_expands_to_the_wire_format"] = (read_graph_raw(map9, ruid9) == read_graph_raw(map_fn, ruid0) and
                                                            read_graph_raw(map9, tuid9) == read_graph_raw(map_fn, tuid0))
        # columnar layout gives the same graph
        map2 = os.path.join(td, "mapping_col.h5")
        with redirect_stdout(buf):
            m3 = nabo_amd.Mapping(map2, "WT", ref_fn, "data", overwrite=True, layout="columnar")
            m3.set_parameters(uc, k, f, chunk)
            m3.make_ref_graph()
        _, edges3, _ = read_graph_like_reference(map2, "WT", "reference")
        out["columnar_same_graph"] = edges3 == edges
        # reference rows sharded over "devices" (here 3 shard-ranks on the one GPU, loopback transport): same graphs
        map5 = os.path.join(td, "mapping_sharded.h5")
        with redirect_stdout(buf):
            m6 = nabo_amd.Mapping(map5, "WT", ref_fn, "data", overwrite=True, devices=[0, 0, 0], shard_transport="loopback")
            m6.set_parameters(uc, k, f, chunk)
            m6.make_ref_graph()
            tn0_, names0_, data0_, ign0_ = targets[0]
            m6.map_target(tn0_, os.path.join(td, "t_%s.h5" % tn0_), "data", ignore_ref_cells=ign0_)
        _, edges6, _ = read_graph_like_reference(map5, "WT", "reference")
        _, tedges6, _ = read_graph_like_reference(map5, tn0_, "target")
        _, tedges0, _ = read_graph_like_reference(map_fn, tn0_, "target")
        out["sharded_devices_same_graphs"] = (edges6 == edges) and (tedges6 == tedges0)
        # ... and the same devices as target slices (ref_shards = 1: every rank holds all the references; Euclidean for the
        # reference graph, the modified-Canberra default for the target -- the layout serves every metric)
        map7 = os.path.join(td, "mapping_slices.h5")
        with redirect_stdout(buf):
            m7 = nabo_amd.Mapping(map7, "WT", ref_fn, "data", overwrite=True, devices=[0, 0, 0], shard_transport="loopback",
                                  ref_shards=1)
            m7.set_parameters(uc, k, f, chunk)
            m7.make_ref_graph()
            m7.map_target(tn0_, os.path.join(td, "t_%s.h5" % tn0_), "data", ignore_ref_cells=ign0_)
        _, edges7, _ = read_graph_like_reference(map7, "WT", "reference")
        _, tedges7, _ = read_graph_like_reference(map7, tn0_, "target")
        out["target_slices_same_graphs"] = (edges7 == edges) and (tedges7 == tedges0)
        # dense [N, n_comps] input (rows deliberately NOT in name order) gives the same graph
        ref_dense = os.path.join(td, "ref_dense.h5")
        perm = np.random.default_rng(3).permutation(len(rn))
        nabo_amd.write_dense_pca(ref_dense, "data", [str(rn[i]) for i in perm], np.asarray(ref)[perm])
        map3 = os.path.join(td, "mapping_dense.h5")
        with redirect_stdout(buf):
            m4 = nabo_amd.Mapping(map3, "WT", ref_dense, "data", overwrite=True)
            m4.set_parameters(uc, k, f, chunk)
            m4.make_ref_graph()
        _, edges4, _ = read_graph_like_reference(map3, "WT", "reference")
        out["dense_input_same_graph"] = (edges4 == edges) and (list(m4.refCells) == list(m.refCells))
        # opt-in target metric (extension): the stored lists are the oracle's Euclidean order rows
        import oracle
        tn0, names0, data0, ign0 = targets[0]
        map4 = os.path.join(td, "mapping_euc.h5")
        with redirect_stdout(buf):
            m5 = nabo_amd.Mapping(map4, "WT", ref_fn, "data", overwrite=True, target_metric="euclidean", layout="columnar")
            m5.set_parameters(uc, k, f, chunk)
            m5.make_ref_graph()
            m5.map_target(tn0, os.path.join(td, "t_%s.h5" % tn0), "data")
        with h5py.File(map4, "r") as h5:
            tuid = [i[1].decode() for i in h5["name_stash/target_names"][:] if i[0].decode() == tn0][0]
            t_idx = h5[tuid + "_sortedDist/__knn_idx"][:]
            t_cells = [x.decode() for x in h5[tuid + "_sortedDist/__knn_cells"][:]]
        order_t = np.argsort(np.array([str(x) for x in names0]))         # HDF5 name order of the target cells
        Xo = np.asarray(data0)[order_t][:, :uc]
        Yo = np.stack([np.asarray(ref)[list(map(str, rn)).index(c)][:uc] for c in m.refCells])
        oi, _ = oracle.knn(Xo, Yo, k, 0, nthreads=4)
        out["target_metric_euclidean"] = bool(t_cells == [str(names0[i]) for i in order_t] and np.array_equal(t_idx, oi))
    return out


def run_validate():
    """Validation rules and exception types of the reference API (no distance computation)."""
    out = {}

    def raises(exc, fn, *a, **kw):
        try:
            fn(*a, **kw)
        except exc:
            return True
        except Exception as e:      # wrong type
            return "wrong exception %r" % (e,)
        return False

    with tempfile.TemporaryDirectory() as td:
        ref_fn = os.path.join(td, "ref.h5")
        names = ["c%d" % i for i in range(12)]
        write_pca(ref_fn, "data", names, pca_like(12, 6, 1))
        map_fn = os.path.join(td, "m.h5")
        M = nabo_amd.Mapping
        out["ref_name_double_underscore"] = raises(ValueError, M, map_fn, "a__b", ref_fn, "data")
        out["same_in_out_file"] = raises(ValueError, M, ref_fn, "WT", ref_fn, "data")
        out["missing_file"] = raises(ValueError, M, map_fn, "WT", os.path.join(td, "nope.h5"), "data")
        out["missing_group"] = raises(ValueError, M, map_fn, "WT", ref_fn, "nogroup")
        m = M(map_fn, "WT", ref_fn, "data", overwrite=True)
        out["ref_cells_name_order"] = m.refCells == sorted(names)
        out["calc_dist_needs_parameters"] = raises(ValueError, m.calc_dist, ref_fn, "data", "a", "b", [])
        out["calc_snn_needs_parameters"] = raises(ValueError, m.calc_snn, "a", "WT", "g")
        out["dist_factor_zero"] = raises(ValueError, m.set_parameters, 5, 3, 0, 10)
        out["dist_factor_str"] = raises(ValueError, m.set_parameters, 5, 3, "x", 10)
        m.set_parameters(5, 3, 0.25, 10)
        out["target_same_as_ref"] = raises(ValueError, m.map_target, "T", ref_fn, "data")
        out["target_is_mapping_file"] = raises(ValueError, m.map_target, "T", map_fn, "data")
        out["target_named_like_ref"] = raises(ValueError, m.map_target, "WT", os.path.join(td, "t.h5"), "data")
        out["target_double_underscore"] = raises(ValueError, m.map_target, "T__1", os.path.join(td, "t.h5"), "data")
        out["calc_snn_missing_group"] = raises(KeyError, m.calc_snn, "nogrp_sortedDist", "T", "g")
        # metadata persists and is validated on reopen (nabo/_mapping.py:357-392)
        with h5py.File(map_fn, "r") as h5:
            out["name_stash_layout"] = (h5["name_stash/ref_name"][0] == b"WT" and
                                        len(h5["name_stash/ref_name"][1]) == 30 and
                                        [x.decode() for x in h5["ref_cells/ref_cells"][:]] == sorted(names))
        out["different_ref_name_on_reopen"] = raises(ValueError, M, map_fn, "OTHER", ref_fn, "data")
        ref2 = os.path.join(td, "ref2.h5")
        write_pca(ref2, "data", names[:-1] + ["zz"], pca_like(12, 6, 1))
        out["different_cells_on_reopen"] = raises(ValueError, M, map_fn, "WT", ref2, "data")
        m2 = M(map_fn, "WT", ref_fn, "data")
        out["reopen_keeps_uid"] = m2._refGraphGrpName == m._refGraphGrpName
        # host I/O helpers: dense input == per-cell input; too-short vectors are refused
        from nabo_amd import _mapping as MM
        Z = pca_like(12, 6, 1)
        dense_fn = os.path.join(td, "dense.h5")
        perm = np.random.default_rng(0).permutation(12)
        nabo_amd.write_dense_pca(dense_fn, "data", [names[i] for i in perm], Z[perm])
        c1, A = MM._read_group_matrix(ref_fn, "data", None, 5)
        c2, B = MM._read_group_matrix(dense_fn, "data", None, 5)
        out["dense_equals_per_cell"] = bool(c1 == c2 and np.array_equal(A, B) and
                                            np.array_equal(A, Z[np.argsort(names)][:, :5]))
        out["too_many_comps_per_cell"] = raises(ValueError, MM._read_group_matrix, ref_fn, "data", None, 7)
        out["too_many_comps_dense"] = raises(ValueError, MM._read_group_matrix, dense_fn, "data", None, 7)
        # the vectorised graph writer against a plain dict-of-dicts statement of nabo/_mapping.py:252-273
        rg = np.random.default_rng(4)

        def expect(t_cells, target_name, is_ref, et, ej, ew, extra):
            ref_nodes = [c + "_WT" for c in m.refCells]
            t_nodes = [c + "_" + target_name for c in t_cells]
            adj = [dict() for _ in t_cells]
            if is_ref:
                pos = {c: i for i, c in enumerate(t_cells)}
                rpos = [pos[c] for c in m.refCells]
                for t, j, w in zip(et, ej, ew):
                    adj[t][ref_nodes[j]] = w
                    adj[rpos[j]][t_nodes[t]] = w
                for a, b, w in extra:
                    adj[rpos[a]][ref_nodes[b]] = w
                    adj[rpos[b]][ref_nodes[a]] = w
            else:
                for t, j, w in zip(et, ej, ew):
                    adj[t][ref_nodes[j]] = w
            return {n: [(k.encode(), repr(float(v)).encode()) for k, v in d.items()] for n, d in zip(t_nodes, adj)}

        ok_dump = True
        for is_ref in (False, True):
            t_cells = list(reversed(m.refCells)) if is_ref else ["t%02d" % i for i in range(9)]
            nt = len(t_cells)
            et = rg.integers(0, nt - 1, 40); ej = rg.integers(0, 12, 40); ew = rg.choice([0.05, 0.11, 1.0, 0.33], 40)
            extra = [(1, 2, 0.03), (7, 9, 0.03)] if is_ref else []
            tn = "WT" if is_ref else "T1"
            m._dump_graph("dump_test", t_cells, tn, is_ref, et, ej, ew, extra)
            want = expect(t_cells, tn, is_ref, et, ej, ew, extra)
            with h5py.File(map_fn, "r") as h5:
                got = {n: [tuple(r) for r in h5["dump_test"][n][:].tolist()] if h5["dump_test"][n].shape != (0,) else []
                       for n in h5["dump_test"]}
                dts = {h5["dump_test"][n].dtype.str for n in h5["dump_test"] if h5["dump_test"][n].shape != (0,)}
            ok_dump = ok_dump and got == want and dts <= {"|S32"}
        out["dump_graph_matches_dict_statement"] = bool(ok_dump)
        with h5py.File(os.path.join(td, "rows.h5"), "w") as h5:
            gg = h5.create_group("g")
            I = np.arange(36, dtype=np.int64).reshape(12, 3)
            MM._write_rows(gg, names, I)
            out["rows_roundtrip"] = bool(np.array_equal(MM._read_rows(gg, names, 3, np.int64), I) and
                                         np.array_equal(np.stack([gg[c][:] for c in names]), I) and
                                         np.array_equal(MM._read_rows(gg, names, 2, np.int64), I[:, :2]))
    return out


if __name__ == "__main__":
    mode = sys.argv[1]
    res = run_validate() if mode == "validate" else run_case({"small": "mapping_small", "dup": "dup",
                                                                "c1": "c1_3k"}.get(mode, mode))
    print("RESULT " + json.dumps(res))
