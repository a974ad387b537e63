"""numpy restatements of the host-side graph helpers (the forms the product used before csrc/host_graph.hip): test
infrastructure only -- tests/test_host_logic.py checks the C-ABI entry points against them and against CPython's own set."""
import numpy as np


def pyset_iteration_order_numpy(rows):
    """For every row of DISTINCT non-negative ints: the permutation of its columns in which CPython iterates
    `set(row)` -- the order in which the reference's `for j in a` (nabo/_mapping.py:190-191) visits a cell's
    neighbours, hence the order of every node's rows in the `<uid>_graph` datasets (networkx keeps insertion
    order).  Restates CPython's open-addressing set (Objects/setobject.c: 8-slot table, hash(int) = int,
    9 linear probes, perturb shift 5, growth to the first power of two > 4*used once fill*5 >= mask*3; the
    same in 3.7 .. 3.12), vectorised over the rows.  Pinned against the interpreter's own `set` by
    tests/test_host_logic.py and against the reference's files by the `*_graph_dst` goldens."""
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    n, k = rows.shape
    if n == 0 or k == 0:
        return np.zeros((n, k), dtype=np.int64)
    if rows.min() < 0:
        raise ValueError("ERROR: neighbour indices must be non-negative")

    def place(table, rsel, cols, vals, mask):
        """One insertion per selected row: column id cols[q] (hash vals[q]) into row rsel[q] of `table`
        [n, mask+1] (-1 = empty): set_add_entry / set_insert_clean without the equality tests (keys differ)."""
        i = vals & mask
        perturb = vals.copy()
        live = np.arange(rsel.size)
        while live.size:
            r, ii = rsel[live], i[live]
            free = table[r, ii] < 0
            table[r[free], ii[free]] = cols[live[free]]
            live, r, ii = live[~free], r[~free], ii[~free]
            if not live.size:
                break
            placed = np.zeros(live.size, dtype=bool)
            lin = ii + 9 <= mask
            for j in range(1, 10):
                cand = np.nonzero(lin & ~placed)[0]
                if not cand.size:
                    break
                fr = table[r[cand], ii[cand] + j] < 0
                hit = cand[fr]
                table[r[hit], ii[hit] + j] = cols[live[hit]]
                placed[hit] = True
            live, ii = live[~placed], ii[~placed]
            perturb[live] >>= 5
            i[live] = (ii * 5 + 1 + perturb[live]) & mask

    mask = 7
    table = np.full((n, mask + 1), -1, dtype=np.int64)
    every = np.arange(n)
    for c in range(k):
        place(table, every, np.full(n, c, dtype=np.int64), rows[:, c].copy(), mask)
        fill = c + 1
        if fill * 5 >= mask * 3:                               # set_table_resize(used > 50000 ? used*2 : used*4)
            minused = fill * 2 if fill > 50000 else fill * 4
            newsize = 8
            while newsize <= minused:
                newsize <<= 1
            old, mask = table, newsize - 1
            table = np.full((n, newsize), -1, dtype=np.int64)
            for s_ in range(old.shape[1]):                      # old entries re-inserted in table order
                sub = np.nonzero(old[:, s_] >= 0)[0]
                if sub.size:
                    cc = old[sub, s_]
                    place(table, sub, cc, rows[sub, cc].copy(), mask)
    return table[table >= 0].reshape(n, k)                      # occupied slots of every row, left to right


def component_labels_numpy(n, a, b):
    """Connected-component label (= smallest member index) of every node of an undirected edge
    list: label hooking + pointer jumping, vectorised.  Labels are member indices and never increase,
    the smallest member keeps its own, so the fixed point is the component minimum."""
    lab = np.arange(n, dtype=np.int64)
    a = np.asarray(a, dtype=np.int64)
    b = np.asarray(b, dtype=np.int64)
    if a.size == 0:
        return lab
    while True:
        la, lb = lab[a], lab[b]
        lo, hi = np.minimum(la, lb), np.maximum(la, lb)
        cut = hi != lo
        if not cut.any():
            return lab
        a, b = a[cut], b[cut]                # edges inside one component stay there: drop them
        new = lab.copy()
        new[hi[cut]] = lo[cut]               # hook the larger root under the smaller one: every write is a
                                             # strict decrease, so whichever of several writers wins is fine
        while True:                          # flatten
            nn = new[new]
            if np.array_equal(nn, new):
                break
            new = nn
        if np.array_equal(new, lab):
            return lab
        lab = new


def group_edges_numpy(n_nodes, node, nb, w):
    """Rows grouped by node in insertion order; a repeated (node, neighbour) pair keeps its first position and its last
    weight (two stable sorts of composite keys)."""
    node = np.asarray(node, dtype=np.int64)
    nb = np.asarray(nb, dtype=np.int64)
    w = np.asarray(w, dtype=np.float64)
    o = np.argsort((node.astype(np.uint64) << np.uint64(32)) | nb.astype(np.uint64), kind="stable")
    node_o, nb_o = node[o], nb[o]
    first = np.ones(o.shape[0], dtype=bool)
    first[1:] = (node_o[1:] != node_o[:-1]) | (nb_o[1:] != nb_o[:-1])
    grp_id = np.cumsum(first) - 1
    last_of = np.zeros(int(first.sum()), dtype=np.int64)
    last_of[grp_id] = np.arange(o.shape[0])
    keep_first = o[first]
    w_keep = w[o[last_of]]
    node_k, nb_k = node[keep_first], nb[keep_first]
    o2 = np.argsort((node_k.astype(np.uint64) << np.uint64(32)) | keep_first.astype(np.uint64))
    node_k, nb_k, w_keep = node_k[o2], nb_k[o2], w_keep[o2]
    counts = np.bincount(node_k, minlength=n_nodes)
    return np.concatenate([[0], np.cumsum(counts)]), nb_k, w_keep
