"""Mapping score: the weighted degree of incident target nodes on reference nodes.

Array / HDF5 restatement of the default behaviour of the reference's
`Graph.get_mapping_score` (nabo/_graph.py:555-697): for every reference node, the sum of the
weights (> min_weight) of its edges to nodes of ONE target sample, times
`score_multiplier / n_target_nodes` (nabo/_graph.py:644-653).  The reference builds a networkx
sub-graph and loops over Python dicts; here it is one segmented sum over the target's edge list
as stored in the mapping file (`<uid>_graph`, nabo/_mapping.py:252-273).
"""
import ctypes as C

import numpy as np


def mapping_score_from_edges(n_ref, edge_ref_idx, edge_weight, n_target_nodes, min_weight=0.0,
                             min_score=0.0, weighted=True, score_multiplier=1000):
    """edge_ref_idx[e] = reference-cell position of target edge e; returns float64 [n_ref]."""
    edge_ref_idx = np.asarray(edge_ref_idx, dtype=np.int64)
    w = np.asarray(edge_weight, dtype=np.float64)
    if weighted:
        keep = w > min_weight
        sc = np.bincount(edge_ref_idx[keep], weights=w[keep], minlength=n_ref)
    else:
        sc = np.bincount(edge_ref_idx, minlength=n_ref).astype(np.float64)
    sc = score_multiplier * sc / float(n_target_nodes)
    sc[sc < min_score] = 0.0
    return sc


def _target_uid(h5, target):
    uid = None
    for i in h5["name_stash/target_names"][:]:
        if i[0].decode("UTF-8") == target:
            uid = i[1].decode("UTF-8")
    if uid is None:
        raise ValueError("ERROR: %s not present in graph" % target)
    return uid


def get_mapping_score(mapping_h5_fn, ref_name, target, min_weight=0, min_score=0, weighted=True, by_cluster=False,
                      sorted_names_only=False, top_n_only=None, all_nodes=True, score_multiplier=1000,
                      ignore_nodes=None, include_nodes=None, remove_suffix=False, verbose=False, clusters=None):
    """Mapping score of one mapped target, read straight from the mapping file: same parameters, defaults,
    return shapes and errors as nabo.Graph.get_mapping_score (nabo/_graph.py:555-697), which needs the whole
    networkx graph in memory; `mapping_h5_fn, ref_name` stand for the Graph object (what `load_from_h5`
    would have been given, nabo/_graph.py:31-116).

      * nodes are named `<cell>_<sample>`; reference nodes come in the order `load_from_h5` meets them
        (HDF5 name order of the reference's `<uid>_graph` group, nabo/_graph.py:93-107) -- that order decides
        ties in `sorted_names_only` (a stable ascending sort, reversed: :678-684);
      * `ignore_nodes` / `include_nodes` (target NODE names; unknown names are dropped, :611-628) restrict the
        target nodes whose edges count, and the denominator is the number of nodes that remain (:629,652-653);
      * `top_n_only` (with `sorted_names_only`) ignores `min_score` (:677-681); `all_nodes=False` keeps only
        scores >= min_score, otherwise smaller scores are reset to 0 (:689-692);
      * `remove_suffix=True` returns a LIST of cell names in both the sorted and the dict form -- the
        reference iterates the dict there (:693-694), kept as is;
      * `by_cluster=True` returns {cluster: [scores of its reference nodes]} (:655-671).  The Graph object keeps the
        clusters as node attributes, set by `make_clusters` (out of scope: SURVEY.md section 2) or `import_clusters`;
        here they arrive as `clusters` = the dict `import_clusters` takes ({reference node: cluster}, :334-356: values
        become strings, reference nodes the dict does not name get 'NA').  Without `clusters` no node has one and
        every score lands in 'NA' -- what the reference returns on a fresh Graph (its guard at :603 compares a set
        with a string and never fires).
    """
    import h5py
    with h5py.File(mapping_h5_fn, "r") as h5:
        if h5["name_stash/ref_name"][0].decode("UTF-8") != ref_name:
            raise KeyError("ERROR: The reference is not named %s in the mapping file" % ref_name)
        if "target_names" not in h5["name_stash"]:
            raise ValueError("ERROR: %s not present in graph" % target)
        uid = _target_uid(h5, target)
        if ignore_nodes is not None and include_nodes is not None:
            raise ValueError("ERROR: PLease provide only one of either 'ignore_nodes' or 'include_nodes' at a time")
        ref_uid = h5["name_stash/ref_name"][1].decode("UTF-8")
        if ref_uid + "_graph" in h5 and "__graph_ptr" in h5[ref_uid + "_graph"]:
            # columnar reference graph: the per-node group would list its nodes in HDF5 (byte) name order
            ref_nodes = sorted(x.decode("UTF-8") for x in h5[ref_uid + "_graph/__graph_nodes"][:])
        elif ref_uid + "_graph" in h5:
            ref_nodes = [n for n in h5[ref_uid + "_graph"]]            # load order of Graph.refNodes
        else:
            ref_nodes = sorted(x.decode("UTF-8") + "_" + ref_name for x in h5["ref_cells/ref_cells"][:])
        pos = {n: i for i, n in enumerate(ref_nodes)}
        grp = h5[uid + "_graph"]
        from ._mapping import _G_PTR, read_graph_csr
        if _G_PTR in grp:
            # columnar graph (Mapping(graph_layout="columnar")): neighbour positions refer to ref_cells/ref_cells
            cells_pos = np.array([pos[x.decode("UTF-8") + "_" + ref_name] for x in h5["ref_cells/ref_cells"][:]], dtype=np.int64)
            t_nodes, ptr, nbr, wts = read_graph_csr(grp, None)
            nbr = cells_pos[nbr]
        else:
            t_nodes, ptr, nbr, wts = None, None, None, None
        if t_nodes is None:
            t_nodes = [n for n in grp]
        tset = {n: None for n in t_nodes}
        ign = set(n for n in (ignore_nodes or []) if n in tset)
        inc = list(t_nodes) if include_nodes is None else [n for n in include_nodes if n in tset]
        inc = set(inc).difference(ign)
        ridx, w = [], []
        n_iso_t = 0
        if ptr is not None:
            keep = np.array([n in inc for n in t_nodes], dtype=bool)
            deg = np.diff(ptr)
            n_iso_t = int(((deg == 0) & keep).sum())
            sel = np.repeat(keep, deg)
            ridx, w = nbr[sel].tolist(), wts[sel].tolist()
        else:
            for node in t_nodes:
                if node not in inc:
                    continue
                rows = grp[node]
                if rows.shape[0] == 0:
                    n_iso_t += 1
                for row in rows:
                    ridx.append(pos[row[0].decode("UTF-8")])
                    w.append(float(row[1].decode("UTF-8")))
    if verbose:
        hit = np.zeros(len(ref_nodes), dtype=bool)
        hit[np.asarray(ridx, dtype=np.int64)] = True
        print("INFO: The bipartite graph has %d edges" % len(ridx))
        print("INFO: Mapping calculated against %d %s nodes" % (len(inc), target))
        print("INFO: %d reference nodes do not connect with any target node" % int((~hit).sum()))
        print("INFO: %d target nodes do not connect with any reference node" % n_iso_t)
    if len(inc) == 0:
        raise ZeroDivisionError("division by zero")                      # :652-653 with no target node left
    # raw scores (no min_score reset yet: the sorted forms filter on the raw value)
    sc = mapping_score_from_edges(len(ref_nodes), ridx, w, len(inc), min_weight, -np.inf, weighted, score_multiplier)
    score = dict(zip(ref_nodes, sc.tolist()))
    if by_cluster:
        cluster_dict = {} if clusters is None else {n: str(clusters[n]) if n in clusters else "NA" for n in ref_nodes}
        cluster_values = {x: [] for x in set(cluster_dict.values())}
        na_cluster_score = []
        for node in score:
            if node in cluster_dict:
                cluster_values[cluster_dict[node]].append(score[node])
            else:
                na_cluster_score.append(score[node])
        if len(na_cluster_score) > 0:
            if "NA" not in cluster_values:
                cluster_values["NA"] = []
            else:
                print("WARNING: 'NA' cluster already exists. Appending value to it")
            cluster_values["NA"].extend(na_cluster_score)
        return cluster_values
    if sorted_names_only:
        if top_n_only is not None:
            if top_n_only > len(score):
                raise ValueError("ERROR: Value of top_n_only should be less than total number of nodes in "
                                 "reference graph")
            retval = [x[0] for x in sorted(score.items(), key=lambda x: x[1])][::-1][:top_n_only]
        else:
            ms = {k: v for k, v in score.items() if v >= min_score}
            retval = [x[0] for x in sorted(ms.items(), key=lambda x: x[1])][::-1]
        return [x.rsplit("_", 1)[0] for x in retval] if remove_suffix else retval
    if not all_nodes:
        retval = {k: v for k, v in score.items() if v >= min_score}
    else:
        retval = {k: v if v >= min_score else 0 for k, v in score.items()}
    return [x.rsplit("_", 1)[0] for x in retval] if remove_suffix else retval


# ---- permutation null (EXTENSION: BASELINE.json configs[4]; the reference has no permutation test) ------------
def mapping_score_null(edge_t, edge_ref_idx, edge_weight, group, n_ref, n_perm=1000, seed=0, score_multiplier=1000,
                       key_bits=64, device=0):
    """Label-permutation null of the mapping score on the GPU (nabo_score_null, include/nabo_knn.h).

    edge_t[e] / edge_ref_idx[e] / edge_weight[e]: target->reference edges of POOLED target cells (e.g. two
    mapped samples); group[t] != 0 marks the cells of the sample of interest.  Every permutation relabels the
    pooled cells (same group size) and recomputes the score; returns a dict of float64 / int64 arrays [n_ref]:
    obs (== the reference's mapping score of the sample of interest), n_ge (#permutations with score >= obs),
    pvalue = (1 + n_ge) / (1 + n_perm), null_mean, null_sd, and sizes [n_perm] (realised group sizes)."""
    from . import _lib
    edge_t = np.ascontiguousarray(edge_t, dtype=np.int64)
    edge_r = np.ascontiguousarray(edge_ref_idx, dtype=np.int64)
    w = np.ascontiguousarray(edge_weight, dtype=np.float64)
    group = np.ascontiguousarray(np.asarray(group) != 0, dtype=np.uint8)
    if not (edge_t.shape == edge_r.shape == w.shape) or edge_t.ndim != 1:
        raise ValueError("ERROR: edge arrays must be 1-D and of equal length")
    obs = np.empty(n_ref); mean = np.empty(n_ref); sd = np.empty(n_ref)
    nge = np.empty(n_ref, dtype=np.int64)
    sizes = np.empty(int(n_perm), dtype=np.int64)
    L = _lib.lib()
    # the edge list goes to the device as it is: the CSR by reference node (edge order kept inside a row) is built
    # there by a stable sort -- at 250M edges the host argsort alone took 25 s
    L.nabo_score_null_edges.argtypes = [C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.c_double, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
    _lib.check(L.nabo_score_null_edges(int(device), int(n_ref), int(edge_r.shape[0]), edge_r.ctypes.data,
                                       edge_t.ctypes.data, w.ctypes.data, int(group.shape[0]), group.ctypes.data,
                                       int(n_perm), int(seed) & (2 ** 64 - 1), int(key_bits), float(score_multiplier),
                                       obs.ctypes.data, nge.ctypes.data, mean.ctypes.data, sd.ctypes.data,
                                       sizes.ctypes.data))
    return {"obs": obs, "n_ge": nge, "pvalue": (1.0 + nge) / (1.0 + n_perm), "null_mean": mean, "null_sd": sd,
            "sizes": sizes}


def _read_target_edges(h5, ref_name, target, pos):
    uid = None
    for i in h5["name_stash/target_names"][:]:
        if i[0].decode("UTF-8") == target:
            uid = i[1].decode("UTF-8")
    if uid is None:
        raise ValueError("ERROR: %s not present in graph" % target)
    from ._mapping import read_graph_csr
    nodes, ptr, nbr, w = read_graph_csr(h5[uid + "_graph"], pos)           # either graph layout
    t = np.repeat(np.arange(len(nodes), dtype=np.int64), np.diff(ptr))
    return len(nodes), t, nbr, w


def get_mapping_score_null(mapping_h5_fn, ref_name, target, background, n_perm=1000, seed=0, score_multiplier=1000,
                           remove_suffix=False, device=0):
    """Permutation null for `target`'s mapping scores against the pooled cells of `target` + `background`
    (another mapped sample, or a list of them), read from the mapping file.  Returns {reference node name:
    (score, pvalue, null_mean, null_sd)}."""
    import h5py
    if isinstance(background, str):
        background = [background]
    with h5py.File(mapping_h5_fn, "r") as h5:
        if h5["name_stash/ref_name"][0].decode("UTF-8") != ref_name:
            raise KeyError("ERROR: The reference is not named %s in the mapping file" % ref_name)
        ref_cells = [x.decode("UTF-8") for x in h5["ref_cells/ref_cells"][:]]
        pos = {c + "_" + ref_name: i for i, c in enumerate(ref_cells)}
        parts = [_read_target_edges(h5, ref_name, s, pos) for s in [target] + list(background)]
    off, ts, rs, ws, grp = 0, [], [], [], []
    for i, (n_nodes, t, r, w) in enumerate(parts):
        ts.append(t + off); rs.append(r); ws.append(w)
        grp.append(np.full(n_nodes, 1 if i == 0 else 0, dtype=np.uint8))
        off += n_nodes
    res = mapping_score_null(np.concatenate(ts), np.concatenate(rs), np.concatenate(ws), np.concatenate(grp),
                             len(ref_cells), n_perm, seed, score_multiplier, device=device)
    names = ref_cells if remove_suffix else [c + "_" + ref_name for c in ref_cells]
    return {n: (float(res["obs"][i]), float(res["pvalue"][i]), float(res["null_mean"][i]), float(res["null_sd"][i]))
            for i, n in enumerate(names)}
