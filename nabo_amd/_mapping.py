"""`Mapping` -- drop-in for the reference's nabo.Mapping (nabo/_mapping.py:280-621) whose
distance + neighbour-selection work runs on the MI355X through libnabo_knn.so.

Kept verbatim from the reference: constructor / method signatures, parameter validation
rules and exception types, reference-cell ordering (HDF5 name order, :404), the
`name_stash` / `ref_cells` metadata (:340-355, :543-555), metric dispatch (Euclidean for
ref<->ref, modified Canberra for target<->ref, :433-440), `ignore_ref_cells` masking
(:135-138), the positional `[1:]` drop (:142), SNN edge weights (:185-198), the
disconnected-graph repair (:203-249, :476-492) and the `<uid>_graph` wire format that
`Graph.load_from_h5` reads (nabo/_graph.py:89-107).

Deliberately different (DESIGN.md "stored layout"): the reference materialises the dense
N_t x N_r float64 matrix and full int64 order rows on disk (:102-103, :145; 16 TB at 1M
cells).  Only the first k entries of each order row are ever consumed (:190,:193), so this
class stores exactly those: `<uid>_sortedDist/<cell>` holds the first k order entries and
`<uid>_dist/<cell>` their distances (same positions), per cell like the reference, or as
two [N,k] arrays with `layout='columnar'`.
"""
import ctypes as C
import os
import random
import string

import numpy as np

from . import _knn, _lib
from ._lib import EUCLIDEAN, MOD_CANBERRA, COSINE

__all__ = ["Mapping", "write_dense_pca"]

_COL_IDX, _COL_DIST, _COL_CELLS = "__knn_idx", "__knn_dist", "__knn_cells"


def _h5py():
    try:
        import h5py
    except ImportError as e:  # pragma: no cover - depends on the interpreter
        raise ImportError("nabo_amd.Mapping needs h5py for Nabo's HDF5 files (the array-level "
                          "nabo_amd.knn / KnnIndex API does not)") from e
    return h5py


def _uid(n=30):
    return "".join(random.choice(string.ascii_lowercase) for _ in range(n))


_DENSE_MAT, _DENSE_CELLS = "__pca_matrix", "__pca_cells"


def write_dense_pca(fn, grp, cells, Z):
    """Dense input layout: one [N, n_comps] float64 matrix + the cell names, in
    place of the one-dataset-per-cell group `Dataset.transform_pca` writes (nabo/_dataset.py:1028).
    `Mapping` reads either; cells are still taken in name order, so results do not depend on the layout."""
    h5py = _h5py()
    Z = np.ascontiguousarray(Z, dtype=np.float64)
    if Z.ndim != 2 or Z.shape[0] != len(cells):
        raise ValueError("ERROR: Z must be [len(cells), n_comps]")
    with h5py.File(fn, mode="a") as h5:
        if grp in h5:
            del h5[grp]
        g = h5.create_group(grp)
        g.create_dataset(_DENSE_MAT, data=Z)
        g.create_dataset(_DENSE_CELLS, data=np.array([c.encode("ascii") for c in cells]))


def _group_cells(g):
    """Cell names of a PCA group in the reference's order (HDF5 name order, nabo/_mapping.py:79,404)."""
    if _DENSE_MAT in g:
        return sorted(x.decode("UTF-8") for x in g[_DENSE_CELLS][:])
    return [x for x in g]


def _read_rows(g, names, width, dtype):
    """[len(names), width] array of the first `width` entries of the 1-D datasets `names` of group g.
    Uses h5py's low-level API: the high-level `g[name][:w]` costs ~190 us per dataset, this ~35 us."""
    h5py = _h5py()
    gid = g.id
    out = np.empty((len(names), width), dtype=dtype)
    buf = np.empty(max(width, 1), dtype=dtype)
    for i, c in enumerate(names):
        ds = h5py.h5d.open(gid, c.encode("utf-8"))
        shp = ds.shape
        if len(shp) != 1 or shp[0] < width:
            raise ValueError("ERROR: dataset %s has shape %s, need at least %d entries" % (c, shp, width))
        if shp[0] != buf.shape[0]:
            buf = np.empty(shp[0], dtype=dtype)
        ds.read(h5py.h5s.ALL, h5py.h5s.ALL, buf)
        out[i] = buf[:width]
    return out


def _write_rows(g, names, arr):
    """One 1-D dataset per row of `arr` (the reference's per-cell layout), low-level API."""
    h5py = _h5py()
    arr = np.ascontiguousarray(arr)
    gid = g.id
    tid = h5py.h5t.py_create(arr.dtype)
    space = h5py.h5s.create_simple((arr.shape[1],))
    for i, c in enumerate(names):
        ds = h5py.h5d.create(gid, c.encode("utf-8"), tid, space)
        ds.write(h5py.h5s.ALL, h5py.h5s.ALL, arr[i])


def _read_group_matrix(fn, grp, cells, use_comps):
    """Gather the per-cell float64 vectors `[:use_comps]` (nabo/_mapping.py:105,113) into one
    dense C-contiguous [n_cells, use_comps] array -- once, instead of once per tile.  A group written
    by `write_dense_pca` is read as one matrix."""
    h5py = _h5py()
    with h5py.File(fn, mode="r") as h5:
        g = h5[grp]
        if cells is None:
            cells = _group_cells(g)
        if _DENSE_MAT in g:
            names = [x.decode("UTF-8") for x in g[_DENSE_CELLS][:]]
            Z = g[_DENSE_MAT]
            if Z.shape[1] < use_comps:
                raise ValueError("ERROR: the PCA matrix has only %d components, use_comps=%d" % (Z.shape[1], use_comps))
            pos = {c: i for i, c in enumerate(names)}
            rows = np.array([pos[c] for c in cells], dtype=np.int64)
            out = np.ascontiguousarray(Z[:, :use_comps][rows] if len(rows) else np.empty((0, use_comps)))
            return cells, out
        try:
            out = _read_rows(g, cells, use_comps, np.float64)
        except ValueError as e:
            raise ValueError("ERROR: use_comps=%d: %s" % (use_comps, e)) from None
    return cells, out


def snn_weight_table(k):
    """round(snn / (2*(k-1) - snn), 2) for snn = 0..k (nabo/_mapping.py:185,194); entries whose
    denominator is zero are NaN and raise if they are ever needed."""
    factor = 2 * (k - 1)
    tab = np.full(k + 1, np.nan)
    for s in range(k + 1):
        if factor - s != 0:
            tab[s] = round(s / (factor - s), 2)
    return tab


def pyset_iteration_order(rows):
    """For every row of DISTINCT non-negative ints: the permutation of its columns in which CPython iterates
    `set(row)` -- the order in which the reference's `for j in a` (nabo/_mapping.py:190-191) visits a cell's
    neighbours, hence the order of every node's rows in the `<uid>_graph` datasets (networkx keeps insertion
    order).  `nabo_pyset_order` (csrc/host_graph.hip) restates CPython's open-addressing set (Objects/setobject.c:
    8-slot table, hash(int) = int, 9 linear probes, perturb shift 5, growth to the first power of two > 4*used once
    fill*5 >= mask*3; the same in 3.7 .. 3.12) on the host's cores.  Pinned against the interpreter's own `set` by
    tests/test_host_logic.py and against the reference's files by the `*_graph_dst` goldens."""
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    n, k = rows.shape
    perm = np.zeros((n, k), dtype=np.int32)
    if n == 0 or k == 0:
        return perm.astype(np.int64)
    _lib.check(_lib.lib().nabo_pyset_order(rows.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int32(k),
                                           perm.ctypes.data_as(C.c_void_p)))
    return perm.astype(np.int64)


def snn_edges_from_counts(t_idx, cnt, k):
    """Host half of `snn_edges`: shared-neighbour counts [m,k] -> (t, j, weight) edge list, a cell's edges in
    the order the reference's set iteration yields them (`pyset_iteration_order`)."""
    t_idx = np.asarray(t_idx)[:, :k]
    cnt = np.asarray(cnt)[:, :k]
    tab = snn_weight_table(k)
    perm = pyset_iteration_order(t_idx)
    t_idx = np.take_along_axis(t_idx, perm, axis=1)
    cnt = np.take_along_axis(cnt, perm, axis=1)
    tt, ss = np.nonzero(cnt > 0)
    w = tab[cnt[tt, ss]]
    if np.isnan(w).any() or (k == 1 and cnt.size):
        raise ZeroDivisionError("division by zero")        # what round(snn/(factor-snn)) does at :194 (k = 1: 0/0)
    return tt.astype(np.int64), t_idx[tt, ss].astype(np.int64), w


def snn_edges(t_idx, r_idx, k, device=0):
    """(t, j, weight) for every j in order_t[:k] sharing >= 1 neighbour with order_ref_j[:k]
    (nabo/_mapping.py:186-198).  Counting runs on the GPU (nabo_snn_counts)."""
    t_idx = np.asarray(t_idx)[:, :k]
    return snn_edges_from_counts(t_idx, _knn.snn_counts(t_idx, r_idx, k, device=device), k)


def _component_labels(n, a, b):
    """Connected-component label (= smallest member index) of every node of an undirected edge list
    (`nabo_component_labels`, csrc/host_graph.hip: union-find, the larger root hooked under the smaller)."""
    a = np.ascontiguousarray(a, dtype=np.int64)
    b = np.ascontiguousarray(b, dtype=np.int64)
    lab = np.empty(n, dtype=np.int64)
    _lib.check(_lib.lib().nabo_component_labels(C.c_int64(n), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                                C.c_int64(a.shape[0]), lab.ctypes.data_as(C.c_void_p)))
    return lab


def group_edges(n_nodes, node, nb, w):
    """(node, neighbour, weight) rows in insertion order -> (starts [n_nodes+1], neighbours, weights) grouped by node: a
    node's neighbours in the order they were first added, a repeated pair keeps its first position and takes its last
    weight -- what networkx's adjacency holds when the reference dumps it (nabo/_mapping.py:252-273).
    `nabo_group_edges` (csrc/host_graph.hip): a stable counting sort by node + a per-node pass over the repeats."""
    node = np.ascontiguousarray(node, dtype=np.int64)
    nb = np.ascontiguousarray(nb, dtype=np.int64)
    w = np.ascontiguousarray(w, dtype=np.float64)
    rows = node.shape[0]
    starts = np.zeros(n_nodes + 1, dtype=np.int64)
    nb_out = np.empty(rows, dtype=np.int64)
    w_out = np.empty(rows, dtype=np.float64)
    _lib.check(_lib.lib().nabo_group_edges(C.c_int64(n_nodes), C.c_int64(rows), node.ctypes.data_as(C.c_void_p),
                                           nb.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p),
                                           starts.ctypes.data_as(C.c_void_p), nb_out.ctypes.data_as(C.c_void_p),
                                           w_out.ctypes.data_as(C.c_void_p)))
    kept = int(starts[-1])
    return starts, nb_out[:kept], w_out[:kept]


# ---- columnar graph layout (opt-in extension; the reference's per-node layout is the default) ---------------------
_G_NODES, _G_PTR, _G_NBR, _G_W = "__graph_nodes", "__graph_ptr", "__graph_nbr", "__graph_w"


def write_columnar_graph(grp, node_names, ptr, nbr, w):
    """CSR by node: node i has neighbours nbr[ptr[i]:ptr[i+1]] (positions in `ref_cells/ref_cells`) with weights w[...],
    rows in the order the per-node datasets would list them."""
    grp.create_dataset(_G_NODES, data=np.array([n.encode("ascii") for n in node_names]))
    grp.create_dataset(_G_PTR, data=np.asarray(ptr, dtype=np.int64))
    grp.create_dataset(_G_NBR, data=np.asarray(nbr, dtype=np.int32))
    grp.create_dataset(_G_W, data=np.asarray(w, dtype=np.float64))


def read_graph_csr(grp, ref_pos):
    """(node names, ptr int64 [n+1], neighbour reference positions int64 [E], weights float64 [E]) of a `<uid>_graph`
    group in either layout; ref_pos maps a reference NODE name to its position in `ref_cells/ref_cells`."""
    if _G_PTR in grp:
        # nodes in HDF5 (byte) name order, as iterating a per-node group yields them: float64 sums over a reference
        # node's edges (mapping scores) then add up in the same order whichever layout the file uses
        names = grp[_G_NODES][:]
        ptr, nbr, w = grp[_G_PTR][:].astype(np.int64), grp[_G_NBR][:].astype(np.int64), grp[_G_W][:]
        order = np.argsort(names, kind="stable")
        deg = np.diff(ptr)[order]
        optr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        idx = np.arange(int(optr[-1]), dtype=np.int64) - np.repeat(optr[:-1], deg) + np.repeat(ptr[:-1][order], deg)
        return [x.decode("UTF-8") for x in names[order]], optr, nbr[idx], w[idx]
    nodes, ptr, nbr, w = [], [0], [], []
    for node in grp:
        nodes.append(node)
        for row in grp[node]:
            nbr.append(ref_pos[row[0].decode("UTF-8")])
            w.append(float(row[1].decode("UTF-8")))      # the weight as the reference parses it (nabo/_graph.py:105-106)
        ptr.append(len(nbr))
    return nodes, np.array(ptr, dtype=np.int64), np.array(nbr, dtype=np.int64), np.array(w, dtype=np.float64)


def expand_graph(mapping_h5_fn, name):
    """Rewrite the columnar graph of `name` (the reference's or a target's name) in the reference's per-node wire format
    (nabo/_mapping.py:252-273), in place, so that `Graph.load_from_h5` (nabo/_graph.py:31-116) can read it."""
    h5py = _h5py()
    with h5py.File(mapping_h5_fn, mode="a") as h5:
        ref_name = h5["name_stash/ref_name"][0].decode("UTF-8")
        uid = h5["name_stash/ref_name"][1].decode("UTF-8") if name == ref_name else None
        if uid is None:
            for i in h5["name_stash/target_names"][:]:
                if i[0].decode("UTF-8") == name:
                    uid = i[1].decode("UTF-8")
        if uid is None:
            raise ValueError("ERROR: %s not present in graph" % name)
        grp = h5[uid + "_graph"]
        if _G_PTR not in grp:
            return
        nodes, ptr, nbr, w = read_graph_csr(grp, None)
        ref_nodes = np.array([(x.decode("UTF-8") + "_" + ref_name).encode("ascii") for x in h5["ref_cells/ref_cells"][:]])
        del h5[uid + "_graph"]
        out = h5.create_group(uid + "_graph")
        for i, node in enumerate(nodes):
            a, b = int(ptr[i]), int(ptr[i + 1])
            if a == b:
                out.create_dataset(node, data=np.empty((0,), dtype=np.float64))
                continue
            width = max(32, max(len(x) for x in ref_nodes[nbr[a:b]]))
            rows = np.empty((b - a, 2), dtype="S%d" % width)
            rows[:, 0] = ref_nodes[nbr[a:b]]
            rows[:, 1] = [repr(float(x)).encode("ascii") for x in w[a:b]]
            out.create_dataset(node, data=rows)


class Mapping:
    """
    Cell mapping on the GPU.  Same constructor and methods as the reference class
    (nabo/_mapping.py:280-621).

    :param mapping_h5_fn: Output HDF5 file (results; existing data may be re-used)
    :param ref_name: Label for reference samples
    :param ref_pca_fn: HDF5 file with the reference PCA data (one dataset per cell)
    :param ref_pca_grp_name: Group inside ref_pca_fn holding the data
    :param overwrite: start from scratch, deleting everything saved in mapping_h5_fn
    Extensions (keyword only, defaults = reference behaviour): device, devices, layout, target_metric, store_k.
    `devices=[0, 1, ...]` shards the reference rows over several GPUs of this node (one host thread and one RCCL
    communicator per GPU behind nabo_sharded_query, include/nabo_knn.h; results equal the one-GPU run bit for bit as
    long as at least k (+1 for the reference graph) reference cells are not ignored -- with fewer the one-GPU path
    continues a row with the ignored cells, as numpy.ma does, and the sharded one raises a ValueError).
    `ref_shards` lays the devices out as that many pieces of the references x len(devices) / ref_shards slices of the
    target rows (`nabo_comm_set_ref_shards`); 1 = every device holds all the references and answers its own slice of
    the rows, the cheapest layout whenever the references fit one GPU.
    `graph_layout="columnar"`: the SNN graphs as four arrays per graph (node names, row pointers, neighbour positions,
    weights) instead of the reference's one dataset per node (nabo/_mapping.py:252-273) -- at 1M cells the per-node
    format IS the run time (23 of 28 s); `nabo_amd.get_mapping_score*` read either, `nabo_amd.expand_graph` rewrites a
    columnar graph in the reference's format for `Graph.load_from_h5`.
    `store_k`: keep max(k, store_k) entries of every order row, so that `use_stored_distances=True` still works after
    `set_parameters` RAISED k up to store_k -- the reference keeps full rows (nabo/_mapping.py:139-145) and therefore
    serves any later k (:537-541, :596-607); the filter's candidate lists already hold more than k entries, so a
    store_k of about 2k costs no kernel time.
    """

    def __init__(self, mapping_h5_fn, ref_name, ref_pca_fn, ref_pca_grp_name, overwrite=False, *,
                 device=0, devices=None, layout="per_cell", target_metric=None, shard_transport="rccl", store_k=None,
                 graph_layout="per_node", ref_shards=None):
        self._h5Fn = mapping_h5_fn
        if ref_name.find("__") != -1:
            raise ValueError("ERROR: Underscores are not allowed in the value for `ref_name` parameter")
        self.refName = ref_name
        self._refPcaFn = ref_pca_fn
        self._refPcaGrp = ref_pca_grp_name
        if self._h5Fn == self._refPcaFn:
            raise ValueError("ERROR: Input HDF5 and output HDF5 file cannot be same")
        if layout not in ("per_cell", "columnar"):
            raise ValueError("ERROR: layout must be 'per_cell' or 'columnar'")
        if target_metric not in (None, "mod_canberra", "euclidean", "cosine"):
            raise ValueError("ERROR: target_metric must be None, 'mod_canberra', 'euclidean' or 'cosine'")
        # None / 'mod_canberra' = the reference's target<->reference metric (nabo/_mapping.py:122-124)
        self._targetMetric = {None: MOD_CANBERRA, "mod_canberra": MOD_CANBERRA, "euclidean": EUCLIDEAN,
                              "cosine": COSINE}[target_metric]
        self._devices = [int(d) for d in devices] if devices else [int(device)]
        self._device = self._devices[0]
        self._shardTransport = shard_transport       # "rccl", or "loopback" (device-to-device copies; devices may repeat)
        # layout of the devices: None = one piece of the references per device (BASELINE's form); R = R pieces x
        # len(devices) / R slices of the target rows; 1 = every device holds all the references and answers its own slice
        self._refShards = None if ref_shards is None else int(ref_shards)
        self._layout = layout
        if graph_layout not in ("per_node", "columnar"):
            raise ValueError("ERROR: graph_layout must be 'per_node' or 'columnar'")
        self._graphLayout = graph_layout
        if store_k is not None and int(store_k) < 1:
            raise ValueError("ERROR: store_k must be a positive integer")
        self._storeK = None if store_k is None else int(store_k)
        self._check_h5(self._refPcaFn, self._refPcaGrp)
        self.refCells = []
        self._nameStash = {}
        self._check_preload(overwrite)
        self._refDistGrp = self._nameStash[self.refName] + "_dist"
        self._refSortedDistGrp = self._nameStash[self.refName] + "_sortedDist"
        self._refGraphGrpName = self._nameStash[self.refName] + "_graph"
        self._useComps = None
        self._k = None
        self._distFactor = None
        self._chunkSize = None
        self._refMatrix = None          # cached dense reference [:use_comps]
        self._knnCache = {}             # sorted-dist group -> (cells, idx[N,k]) written by this object

    # ---- metadata (nabo/_mapping.py:322-406) ---------------------------------------------
    @staticmethod
    def _check_h5(fn, group):
        h5py = _h5py()
        if os.path.exists(fn) is False:
            raise ValueError("File %s doesn't exist" % fn)
        with h5py.File(fn, mode="r") as h5:
            if group not in h5:
                raise ValueError("Group %s does not exist in file %s" % (group, fn))
        return True

    def _load_ref_cells(self):
        with _h5py().File(self._refPcaFn, mode="r") as h5:
            return _group_cells(h5[self._refPcaGrp])         # HDF5 name order (:404)

    def _create_metadata(self, h5):
        for key in list(h5.keys()):
            del h5[key]
        grp = h5.create_group("name_stash")
        rs = _uid(30)
        self._nameStash[self.refName] = rs
        grp.create_dataset("ref_name", data=[self.refName.encode("ascii"), rs.encode("ascii")])
        self.refCells = self._load_ref_cells()
        grp = h5.create_group("ref_cells")
        grp.create_dataset("ref_cells", data=[x.encode("ascii") for x in self.refCells])
        return True

    def _check_preload(self, overwrite):
        with _h5py().File(self._h5Fn, mode="a") as h5:
            have = ("ref_cells" in h5 and "ref_cells" in h5["ref_cells"] and
                    "name_stash" in h5 and "ref_name" in h5["name_stash"])
            if overwrite is True or not have:
                self._create_metadata(h5)
                return
            ref_name = h5["name_stash/ref_name"][0].decode("UTF-8")
            if ref_name != self.refName:
                raise ValueError("ERROR: A different ref_name was used before for this mapping file. "
                                 "Please set overwrite=True if you want to overwrite all the saved data.")
            if "target_names" in h5["name_stash"]:
                for i in h5["name_stash/target_names"]:
                    self._nameStash[i[0].decode("UTF-8")] = i[1].decode("UTF-8")
            self._nameStash[self.refName] = h5["name_stash/ref_name"][1].decode("UTF-8")
            saved_cells = [x.decode("UTF-8") for x in h5["ref_cells/ref_cells"][:]]
            pca_cells = self._load_ref_cells()
            if len(saved_cells) == len(pca_cells) == len(set(saved_cells).intersection(pca_cells)):
                self.refCells = saved_cells
            else:
                raise ValueError("ERROR: Cell names in PCA file does not match those used before in this "
                                 "mapping file. Please set overwrite=True if you want to overwrite all the "
                                 "saved data.")

    def _stash_target_name(self, target):
        with _h5py().File(self._h5Fn, mode="a") as h5:
            if "target_names" in h5["name_stash"]:
                del h5["name_stash/target_names"]
            self._nameStash[target] = _uid(30)
            stash = [[n.encode("ascii"), u.encode("ascii")] for n, u in self._nameStash.items()
                     if n != self.refName]
            h5["name_stash"].create_dataset("target_names", data=stash)

    # ---- parameters (nabo/_mapping.py:495-524) -------------------------------------------
    def set_parameters(self, use_comps, k, dist_factor, chunk_size):
        """use_comps: leading input dimensions to use; k: neighbours; dist_factor: window of the
        modified Canberra distance (> 0); chunk_size: kept for API compatibility -- the GPU
        path streams tiles from HBM and does not need a host-side chunk size.

        Storage note: calc_dist keeps the first max(k, store_k) entries of every order row (and their distances), not
        the reference's full rows (nabo/_mapping.py:102-103,145 -- 16 TB at 1M x 1M).  `use_stored_distances=True`
        therefore works for any later k up to that length (`Mapping(..., store_k=...)`); beyond it call
        make_ref_graph() / map_target(..., overwrite=True) without use_stored_distances to recompute (calc_snn
        raises a ValueError that says so when the stored lists are too short)."""
        self._useComps = use_comps
        self._k = k
        try:
            float(dist_factor)
            assert dist_factor > 0
        except (ValueError, AssertionError, TypeError):
            raise ValueError('ERROR: "dist_factor" must be a non-zero float value')
        self._distFactor = dist_factor
        self._chunkSize = chunk_size
        self._refMatrix = None
        return None

    # ---- distances + neighbour selection (nabo/_mapping.py:408-444 -> :48-148) -------------
    def _ref_matrix(self):
        if self._refMatrix is None or self._refMatrix.shape[1] != self._useComps:
            _, self._refMatrix = _read_group_matrix(self._refPcaFn, self._refPcaGrp, self.refCells, self._useComps)
        return self._refMatrix

    def _store_knn(self, h5, dist_grp, sorted_dist_grp, cells, idx, dist):
        for g in (dist_grp, sorted_dist_grp):
            if g in h5:
                del h5[g]
        dg = h5.create_group(dist_grp)
        sg = h5.create_group(sorted_dist_grp)
        if self._layout == "columnar":
            sg.create_dataset(_COL_IDX, data=idx)
            sg.create_dataset(_COL_CELLS, data=[c.encode("ascii") for c in cells])
            dg.create_dataset(_COL_DIST, data=dist)
        else:
            _write_rows(sg, cells, idx)
            _write_rows(dg, cells, dist)
        # what calc_snn reads next; the file stays the hand-off for later sessions (use_stored_distances)
        self._knnCache[sorted_dist_grp] = (list(cells), idx)

    def _load_knn(self, h5, sorted_dist_grp):
        if sorted_dist_grp in self._knnCache:
            return self._knnCache[sorted_dist_grp]
        sg = h5[sorted_dist_grp]
        if _COL_IDX in sg:
            cells = [x.decode("UTF-8") for x in sg[_COL_CELLS][:]]
            return cells, sg[_COL_IDX][:]
        cells = [x for x in sg]
        if not cells:
            return cells, np.empty((0, 0), dtype=np.int64)
        width = min(sg[c].shape[0] for c in (cells[0], cells[-1]))
        width = min(width, self._k) if self._k is not None else width
        return cells, _read_rows(sg, cells, width, np.int64)

    def calc_dist(self, target_fn, target_grp, dist_grp, sorted_dist_grp, ignore_ref_cells):
        """Euclidean (reference vs itself) or modified Canberra (target vs reference) distances,
        then the first k entries of every order row, on the GPU."""
        if self._useComps is None or self._chunkSize is None or self._distFactor is None:
            raise ValueError('ERROR: Please set the parameters first using "set_parameters" method')
        if ignore_ref_cells is None:
            ignore_ref_cells = []
        intra_ref = (target_fn == self._refPcaFn and target_grp == self._refPcaGrp)
        ref = self._ref_matrix()
        cells, X = _read_group_matrix(target_fn, target_grp, None, self._useComps)   # HDF5 name order (:79)
        mask = None
        if len(ignore_ref_cells) > 0:
            ign = set(ignore_ref_cells)
            mask = np.array([c in ign for c in self.refCells], dtype=np.uint8)
        drop = 1 if intra_ref else 0
        n_ref = ref.shape[0]
        k_store = min(max(self._k if self._k is not None else 1, self._storeK or 1), n_ref - drop)
        if k_store < 1:
            raise ValueError("ERROR: not enough reference cells")
        metric = EUCLIDEAN if intra_ref else self._targetMetric
        if len(self._devices) > 1:
            # reference rows sharded over the GPUs (nabo/_mapping.py:441-444 is the call site this replaces)
            from ._sharded import ShardedGroup
            n_ign = int(mask.sum()) if mask is not None else 0
            if n_ref - n_ign < k_store + drop:
                # one GPU continues such rows with the ignored cells by index (numpy.ma's NaN fill, :135-146); shards
                # cannot (an ignored cell of one shard would enter the merge as a neighbour): absent entries instead
                raise ValueError("ERROR: only %d reference cells are not ignored, fewer than the %d neighbours to keep: "
                                 "map on one device (devices=None) or ignore fewer cells" % (n_ref - n_ign, k_store + drop))
            grp = ShardedGroup(self._devices, n_ref, self._useComps, metric, ref, dist_factor=float(self._distFactor),
                               ref_mask=mask, transport=self._shardTransport, ref_shards=self._refShards)
            try:
                idx, dist = grp.set_ref().query(X, k_store, drop_first=bool(drop))
            finally:
                grp.close()
        else:
            idx, dist = _knn.knn(X, ref, k_store, metric=metric, dist_factor=float(self._distFactor), ref_mask=mask,
                                 drop_first=bool(drop), device=self._device)
        with _h5py().File(self._h5Fn, mode="a") as h5:
            self._store_knn(h5, dist_grp, sorted_dist_grp, cells, idx, dist)

    # ---- SNN graph (nabo/_mapping.py:446-493 -> :151-273) ----------------------------------
    def _repair_round(self, lab, weight, index):
        """One repair round (nabo/_mapping.py:203-249) on component labels `lab`: every component that has a
        strictly larger one gets an edge from its member closest (Euclidean) to any cell of those larger
        components.  The reference walks full order rows; here it is a masked 1-NN query against the resident
        reference index (only the mask changes between components).  Returns the new edges."""
        comps, inv, sizes = np.unique(lab, return_inverse=True, return_counts=True)
        if len(comps) == 1:
            return []
        ref = self._ref_matrix()
        size_of_node = sizes[inv]
        smax = sizes.max()
        # components of one size share their mask (allowed = cells of strictly larger components): one mask upload and
        # one query per DISTINCT size serve all of them; a component's edge is its best row by (distance, member, neighbour)
        best_of = {}
        for sz in np.unique(sizes):
            if sz == smax:                                          # no strictly larger component
                continue
            members = np.nonzero(size_of_node == sz)[0]             # ascending cell index
            index.set_mask((size_of_node <= sz).astype(np.uint8))
            idx, dist = index.query(ref[members], 1)
            ci = inv[members]
            o = np.lexsort((idx[:, 0], members, dist[:, 0], ci))
            firsts = o[np.concatenate([[True], ci[o][1:] != ci[o][:-1]])]
            for f in firsts:
                best_of[int(ci[f])] = (int(members[f]), int(idx[f, 0]), weight)
        return [best_of[c] for c in sorted(best_of)]               # in component order, as the reference adds them

    @staticmethod
    def _merge_labels(lab, new_edges):
        """Component labels (= smallest member index) after adding a few edges: union-find over the labels."""
        if not new_edges:
            return lab
        parent = {}

        def find(c):
            while parent.setdefault(c, c) != c:
                parent[c] = parent[parent[c]]
                c = parent[c]
            return c

        for a, b, _ in new_edges:
            ra, rb = find(int(lab[a])), find(int(lab[b]))
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)
        lut = np.arange(lab.shape[0], dtype=np.int64)
        for c in list(parent):
            lut[c] = find(c)
        return lut[lab]

    def calc_snn(self, target_sorted_dist_grp, target_name, graph_grp, fix_graph_attempts=5, fix_weight=None):
        """Shared-nearest-neighbour graph from the stored neighbour lists; written in the layout
        Graph.load_from_h5 reads."""
        if self._k is None:
            raise ValueError("ERROR: Set parameters first")
        k = self._k
        with _h5py().File(self._h5Fn, mode="r") as h5:
            if self._refSortedDistGrp not in h5:
                raise KeyError("ERROR: Please make sure that the distances between reference cells has "
                               "already been calculated")
            if target_sorted_dist_grp not in h5:
                raise KeyError("ERROR: Please make sure that the distances between reference and target "
                               "cells has already been calculated")
            t_cells, t_idx = self._load_knn(h5, target_sorted_dist_grp)
            r_cells, r_idx = self._load_knn(h5, self._refSortedDistGrp)
        if t_idx.shape[1] < k or r_idx.shape[1] < k:
            raise ValueError("ERROR: stored neighbour lists are shorter than k=%d; recompute the distances" % k)
        order = {c: i for i, c in enumerate(r_cells)}
        if r_cells != list(self.refCells):
            r_idx = r_idx[[order[c] for c in self.refCells]]
        et, ej, ew = snn_edges(t_idx, r_idx, k, device=self._device)
        is_ref = (target_name == self.refName)
        extra = []
        if is_ref:
            n = len(self.refCells)
            tpos = np.array([order[c] for c in t_cells]) if t_cells != list(self.refCells) else np.arange(n)
            if fix_weight is None:
                fix_weight = 0.5 / ((2 * (k - 1)) - 0.5)
            # components once from all SNN edges; repair rounds then only merge labels (a round adds few edges)
            lab = _component_labels(n, tpos[et], ej)
            n_comp = len(np.unique(lab))
            if n_comp > 1:
                print("INFO: Reference graph is disconnected. Trying to fix..")
                index = _knn.KnnIndex(n, self._useComps, metric=EUCLIDEAN, device=self._device).set_ref(self._ref_matrix())
                try:
                    new = self._repair_round(lab, fix_weight, index)
                    for _ in range(fix_graph_attempts):             # same schedule as nabo/_mapping.py:476-490
                        extra.extend(new)
                        lab = self._merge_labels(lab, new)
                        n_comp = len(np.unique(lab))
                        if n_comp == 1:
                            print("INFO: Reference graph is no longer disconnected.")
                            break
                        new = self._repair_round(lab, fix_weight, index)
                finally:
                    index.close()
            if n_comp > 1:
                print("WARNING: Output graph is disconnected.")
        self._dump_graph(graph_grp, t_cells, target_name, is_ref, et, ej, ew, extra)

    def _dump_graph(self, out_grp, t_cells, target_name, is_ref, et, ej, ew, extra):
        """The a10 wire format (nabo/_mapping.py:252-273): per node one dataset of
        (neighbour name, weight) rows coerced to byte strings.  Rows of a node keep the order in which
        the reference's graph would have gained them (edge insertion order, first occurrence wins).
        Built as ONE [rows,2] byte-string array sliced per node; datasets are created through h5py's
        low-level API (one create+write per node is what the format costs)."""
        h5py = _h5py()
        n_t, n_r = len(t_cells), len(self.refCells)
        et = np.asarray(et, dtype=np.int64)
        ej = np.asarray(ej, dtype=np.int64)
        ew = np.asarray(ew, dtype=np.float64)
        ref_nodes = np.array([(c + "_" + self.refName).encode("ascii") for c in self.refCells])
        t_nodes_s = [c + "_" + target_name for c in t_cells]
        E = et.shape[0]
        if is_ref:
            # neighbour ids: position in refCells.  Each SNN edge (t, j) lists j under t and t under j;
            # repair edges (reference-cell positions) follow, in both directions.
            pos = {c: i for i, c in enumerate(t_cells)}
            rpos = np.array([pos[c] for c in self.refCells], dtype=np.int64)       # refCells position -> t position
            tref = np.empty(n_t, dtype=np.int64)
            tref[rpos] = np.arange(n_r)                                             # t position -> refCells position
            xa = np.array([a for a, _, _ in extra], dtype=np.int64)
            xb = np.array([b for _, b, _ in extra], dtype=np.int64)
            xw = np.array([w for _, _, w in extra], dtype=np.float64)
            node = np.concatenate([np.stack([et, rpos[ej]], 1).reshape(-1), np.stack([rpos[xa], rpos[xb]], 1).reshape(-1)])
            nb = np.concatenate([np.stack([ej, tref[et]], 1).reshape(-1), np.stack([xb, xa], 1).reshape(-1)])
            w = np.concatenate([np.repeat(ew, 2), np.repeat(xw, 2)])
        else:
            node, nb, w = et, ej, ew
        # group by node, keep insertion order inside a node, drop repeated (node, neighbour) pairs: a dict keeps the
        # first position and the last value, as networkx's adjacency does (group_edges)
        starts, nb_k, w_keep = group_edges(n_t, node, nb, w)
        counts = np.diff(starts)
        if self._graphLayout == "columnar":
            with h5py.File(self._h5Fn, mode="a") as h5:
                if out_grp in h5:
                    del h5[out_grp]
                write_columnar_graph(h5.create_group(out_grp), t_nodes_s, starts, nb_k, w_keep)
            return
        # weights take few distinct values: format each once (numpy's float -> bytes coercion is repr)
        uw, winv = np.unique(w_keep, return_inverse=True)
        wtxt = np.array([repr(float(x)).encode("ascii") for x in uw]) if len(uw) else np.empty(0, dtype="S1")
        width = max(32, ref_nodes.dtype.itemsize if n_r else 1)
        rows = np.empty((nb_k.shape[0], 2), dtype="S%d" % width)
        if nb_k.shape[0]:
            rows[:, 0] = ref_nodes[nb_k]
            rows[:, 1] = wtxt[winv]
        name_len = np.char.str_len(ref_nodes) if n_r else np.zeros(0, dtype=np.int64)
        uniform = (n_r == 0) or int(name_len.max()) <= 32       # else the S-width differs per node (max over its rows)
        with h5py.File(self._h5Fn, mode="a") as h5:
            if out_grp in h5:
                del h5[out_grp]
            out = h5.create_group(out_grp)
            gid = out.id
            tid = h5py.h5t.py_create(rows.dtype)
            f64 = h5py.h5t.py_create(np.dtype(np.float64))
            spaces = {}
            empty = np.empty((0,), dtype=np.float64)
            for t in range(n_t):
                c = int(counts[t])
                name = t_nodes_s[t].encode("utf-8")
                if c == 0:
                    ds = h5py.h5d.create(gid, name, f64, h5py.h5s.create_simple((0,)))
                    continue
                blk = rows[starts[t]:starts[t] + c]
                if not uniform:
                    wd = max(32, int(name_len[nb_k[starts[t]:starts[t] + c]].max()))
                    if wd != width:
                        out.create_dataset(t_nodes_s[t], data=blk.astype("S%d" % wd))
                        continue
                sp = spaces.get(c)
                if sp is None:
                    sp = spaces[c] = h5py.h5s.create_simple((c, 2))
                ds = h5py.h5d.create(gid, name, tid, sp)
                ds.write(h5py.h5s.ALL, h5py.h5s.ALL, blk)

    # ---- entry points (nabo/_mapping.py:526-541, :557-621) ----------------------------------
    def make_ref_graph(self, use_stored_distances=False):
        if use_stored_distances is False:
            self.calc_dist(self._refPcaFn, self._refPcaGrp, self._refDistGrp, self._refSortedDistGrp, [])
        self.calc_snn(self._refSortedDistGrp, self.refName, self._refGraphGrpName)

    def map_target(self, target_name, target_pca_fn, target_pca_grp_name, ignore_ref_cells=None,
                   use_stored_distances=False, overwrite=False):
        if target_pca_fn == self._refPcaFn and target_pca_grp_name == self._refPcaGrp:
            raise ValueError("ERROR: Target PCA file name and group name can not be same as that of reference")
        if target_pca_fn == self._h5Fn:
            raise ValueError("ERROR: Input HDF5 and output HDF5 file cannot be same")
        if target_name == self.refName:
            raise ValueError("ERROR: Target name cannot be same as reference name. Please provide a "
                             "different name.")
        if target_name.find("__") != -1:
            raise ValueError("ERROR: Underscores are not allowed in the value for `target_name` parameter")
        if ignore_ref_cells is None:
            ignore_ref_cells = []
        if use_stored_distances is True:
            if target_name not in self._nameStash:
                print("WARNING: Target data not saved. use_stored_distances will have no effect")
            else:
                if overwrite is True:
                    print("WARNING: overwrite has no effect as use_stored_distances is set to True")
                self.calc_snn(self._nameStash[target_name] + "_sortedDist", target_name,
                              self._nameStash[target_name] + "_graph")
                return None
        else:
            if overwrite is False and target_name in self._nameStash:
                raise ValueError("ERROR: Data with this target name exists. Please set overwrite=True if "
                                 "you want to map this target again.")
        self._stash_target_name(target_name)
        self._check_h5(target_pca_fn, target_pca_grp_name)
        uid = self._nameStash[target_name]
        self.calc_dist(target_pca_fn, target_pca_grp_name, uid + "_dist", uid + "_sortedDist", ignore_ref_cells)
        self.calc_snn(uid + "_sortedDist", target_name, uid + "_graph")
        return None
