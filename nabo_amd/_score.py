"""Mapping score: the weighted degree of incident target nodes on reference nodes.

Array / HDF5 restatement of the default behaviour of the reference's
`Graph.get_mapping_score` (nabo/_graph.py:555-697): for every reference node, the sum of the
weights (> min_weight) of its edges to nodes of ONE target sample, times
`score_multiplier / n_target_nodes` (nabo/_graph.py:644-653).  The reference builds a networkx
sub-graph and loops over Python dicts; here it is one segmented sum over the target's edge list
as stored in the mapping file (`<uid>_graph`, nabo/_mapping.py:252-273).
"""
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


def get_mapping_score(mapping_h5_fn, ref_name, target, min_weight=0, min_score=0, weighted=True,
                      score_multiplier=1000, remove_suffix=False):
    """{reference node name: score} for one mapped target, read straight from the mapping file
    (same defaults and meaning as nabo.Graph.get_mapping_score with all_nodes=True)."""
    import h5py
    with h5py.File(mapping_h5_fn, "r") as h5:
        if h5["name_stash/ref_name"][0].decode("UTF-8") != ref_name:
            raise KeyError("ERROR: The reference is not named %s in the mapping file" % ref_name)
        uid = None
        for i in h5["name_stash/target_names"][:]:
            if i[0].decode("UTF-8") == target:
                uid = i[1].decode("UTF-8")
        if uid is None:
            raise ValueError("ERROR: %s not present in graph" % target)
        ref_cells = [x.decode("UTF-8") for x in h5["ref_cells/ref_cells"][:]]
        pos = {c + "_" + ref_name: i for i, c in enumerate(ref_cells)}
        grp = h5[uid + "_graph"]
        ridx, w = [], []
        n_nodes = 0
        for node in grp:
            n_nodes += 1
            for row in grp[node]:
                ridx.append(pos[row[0].decode("UTF-8")])
                w.append(float(row[1].decode("UTF-8")))
    sc = mapping_score_from_edges(len(ref_cells), ridx, w, n_nodes, min_weight, min_score, weighted, score_multiplier)
    names = ref_cells if remove_suffix else [c + "_" + ref_name for c in ref_cells]
    return dict(zip(names, sc.tolist()))
