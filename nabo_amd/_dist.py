"""Reference-row sharding across the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI).  Rank r holds
reference rows [base_r, base_r + n_r) resident in its HBM; every rank sees all target rows.
    1. local k-NN on the shard -> first k' = k + drop_first entries per target, GLOBAL indices,
       exact float64 distances (nabo_index_query with ref_index_base = base_r);
    2. ONE exchange: all_to_all of the [m, k'] lists so that rank r owns target rows
       [r*m/N, (r+1)*m/N) from every shard (each GPU receives N*k' candidates for m/N rows --
       1/N of the bytes an all-gather of full lists would move over each xGMI link);
    3. k-way merge by the canonical (distance, index) order on the GPU (nabo_merge_topk),
       positional drop applied AFTER the merge (nabo/_mapping.py:142 is positional);
    4. all_gather of the merged [m/N, k] slices -> every rank holds the full result.
The merge is deterministic, so N shards == 1 shard bit for bit.

GLOBAL certification (default when the shard can emit candidates, `local_cand`): asking every shard
for its exact local top-k' wastes most of the work -- on average only k'/N of a shard's neighbours
survive the merge, yet each shard maintains a k'+8 deep list for every target and re-evaluates it in
float64.  Instead each shard emits its first Ls < k' order-row entries (exact distances) plus a BOUND
on the squared distance of everything it did not emit (nabo_index_query_candidates); the owner of a
target row merges the N*Ls entries and accepts the k'-th merged distance d when d^2 < min over shards
of the bound -- then no unreported reference anywhere can enter or tie.  Rows that fail (a shard held
more than Ls of the global top-k': Ls is chosen so that < 0.1 rows of a batch are expected to) are re-solved
exactly in a second, tiny round with the certified local query.  Same bits, ~half the list maintenance per shard.

torch is used for the process group, the collectives and (on GPU) tensor memory only; the
compute goes through the C ABI with raw pointers.  The two compute steps are injectable so
the exchange/merge plumbing can be exercised with the gloo backend on CPU (tests/).
"""
import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous, balanced reference-row partition: rows [lo, hi) for `rank`."""
    return (n * rank) // world, (n * (rank + 1)) // world


def merge_numpy(parts_idx, parts_dist, k, drop_first):
    """Host statement of nabo_merge_topk (used by the gloo tests; parts are [P, m, kp])."""
    P, m, kp = parts_idx.shape
    idx = np.transpose(parts_idx, (1, 0, 2)).reshape(m, P * kp)
    dist = np.transpose(parts_dist, (1, 0, 2)).reshape(m, P * kp)
    out_i = np.empty((m, k), dtype=np.int64)
    out_d = np.empty((m, k), dtype=np.float64)
    d0 = 1 if drop_first else 0
    for r in range(m):
        valid = idx[r] >= 0
        key_d = np.where(valid, dist[r], np.inf)
        key_i = np.where(valid, idx[r], np.iinfo(np.int64).max)
        o = np.lexsort((key_i, key_d))[d0:d0 + k]
        out_i[r] = np.where(valid[o], idx[r][o], -1)
        out_d[r] = np.where(valid[o], dist[r][o], np.nan)
    return out_i, out_d


class ShardedKnn:
    """k-NN of replicated targets against row-sharded references.

    local_knn(X, kk) -> (idx [m,kk] int64 GLOBAL, dist [m,kk] float64) as torch tensors on `device`
    merge(parts_idx [N,mr,kk], parts_dist, k, drop_first) -> (idx [mr,k], dist [mr,k]) torch tensors
    """

    def __init__(self, dist_module, local_knn, merge, device, local_cand=None):
        self.dist = dist_module
        self.local_knn = local_knn
        self.merge = merge
        self.local_cand = local_cand      # (X, n_cand) -> (idx [m,n_cand], dist [m,n_cand], bound [m]) or None
        self.device = device
        self.last_uncertified = 0
        self.world = dist_module.get_world_size() if dist_module.is_initialized() else 1
        self.rank = dist_module.get_rank() if dist_module.is_initialized() else 0
        # rehearsal mode: device tensors but a CPU-only backend (gloo) -> stage collectives through host
        self.stage = (dist_module.is_initialized() and dist_module.get_backend() == "gloo"
                      and str(device).startswith("cuda"))

    def _a2a(self, recv, send):
        if self.stage:
            r, s = recv.cpu(), send.cpu()
            self.dist.all_to_all_single(r, s)
            recv.copy_(r)
        else:
            self.dist.all_to_all_single(recv, send)

    def _gather(self, full, part):
        if self.stage:
            f, p = full.cpu(), part.cpu()
            self.dist.all_gather_into_tensor(f, p)
            full.copy_(f)
        else:
            self.dist.all_gather_into_tensor(full, part)

    def _allreduce_max(self, t):
        if self.stage:
            c = t.cpu()
            self.dist.all_reduce(c, op=self.dist.ReduceOp.MAX)
            t.copy_(c)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)

    @staticmethod
    def candidates_per_shard(kk, world, m=None):
        """Entries each shard emits.  A row needs the second round when some shard holds at least Ls of its global
        top-k'; for exchangeable shards that is world * P[Bin(k', 1/world) >= Ls] per row.  With `m` (target rows)
        given, Ls is the smallest length that leaves an expected < 0.1 such rows in the whole batch -- never more
        than k'+1 (a shard cannot hold more than k' of the top k') -- because a longer list costs ~0.7 ms per
        entry and step while the second round costs ~3.5 ms plus three more collectives.  Without `m`: the share
        k'/N with 50 % head room, +6."""
        cap = min(kk + 1, 32)
        if m is None:
            ls = max(8, -(-3 * kk // (2 * world)) + 6, -(-kk // world))
            return min(ls, kk + 8, 32)
        from math import comb
        p = 1.0 / world
        tail = 0.0
        ls = cap
        for j in range(kk, 0, -1):                       # tail = P[Bin(kk, p) >= j]
            tail += comb(kk, j) * p ** j * (1.0 - p) ** (kk - j)
            if tail * world * m >= 0.1:
                ls = j + 1
                break
            ls = j
        return max(min(ls, cap), -(-kk // world), 1)

    def query(self, X, m, k, drop_first=False):
        import torch
        kk = k + (1 if drop_first else 0)
        N = self.world
        import os
        force = os.environ.get("NABO_DIST_FORCE_CERT") == "1"          # experiments: protocol overhead at N = 1
        if self.local_cand is not None and (N > 1 or force) and -(-kk // N) <= 32:
            return self._query_certified(X, m, k, drop_first)
        idx, dst = self.local_knn(X, kk)
        if N == 1:
            oi, od = self.merge(idx.view(1, m, kk), dst.view(1, m, kk), k, drop_first)
            return oi, od
        mr = (m + N - 1) // N                      # target rows owned per rank
        m_pad = mr * N
        if m_pad != m:                             # ragged tail: pad with absent entries
            pi = torch.full((m_pad, kk), -1, dtype=torch.int64, device=self.device)
            pd = torch.full((m_pad, kk), float("inf"), dtype=torch.float64, device=self.device)
            pi[:m] = idx
            pd[:m] = dst
            idx, dst = pi, pd
        recv_i = torch.empty((N, mr, kk), dtype=torch.int64, device=self.device)
        recv_d = torch.empty((N, mr, kk), dtype=torch.float64, device=self.device)
        self._a2a(recv_i.view(-1), idx.contiguous().view(-1))
        self._a2a(recv_d.view(-1), dst.contiguous().view(-1))
        oi, od = self.merge(recv_i, recv_d, k, drop_first)
        full_i = torch.empty((m_pad, k), dtype=torch.int64, device=self.device)
        full_d = torch.empty((m_pad, k), dtype=torch.float64, device=self.device)
        self._gather(full_i.view(-1), oi.contiguous().view(-1))
        self._gather(full_d.view(-1), od.contiguous().view(-1))
        return full_i[:m], full_d[:m]


    def _query_certified(self, X, m, k, drop_first):
        import torch
        d0 = 1 if drop_first else 0
        kk = k + d0
        N, dev = self.world, self.device
        Ls = self.candidates_per_shard(kk, N, m)
        ci, cd, cb = self.local_cand(X, Ls)
        mr = (m + N - 1) // N
        m_pad = mr * N
        if m_pad != m:
            pi = torch.full((m_pad, Ls), -1, dtype=torch.int64, device=dev)
            pd = torch.full((m_pad, Ls), float("inf"), dtype=torch.float64, device=dev)
            pb = torch.full((m_pad,), float("inf"), dtype=torch.float64, device=dev)
            pi[:m], pd[:m], pb[:m] = ci, cd, cb
            ci, cd, cb = pi, pd, pb
        recv_i = torch.empty((N, mr, Ls), dtype=torch.int64, device=dev)
        recv_d = torch.empty((N, mr, Ls), dtype=torch.float64, device=dev)
        recv_b = torch.empty((N, mr), dtype=torch.float64, device=dev)
        self._a2a(recv_i.view(-1), ci.contiguous().view(-1))
        self._a2a(recv_d.view(-1), cd.contiguous().view(-1))
        self._a2a(recv_b.view(-1), cb.contiguous().view(-1))
        mi, md = self.merge(recv_i, recv_d, kk, False)                     # [mr, kk], positional drop later
        dk = md[:, kk - 1]
        ok = (mi[:, kk - 1] >= 0) & (dk * dk * (1.0 + 1e-12) < recv_b.min(dim=0).values)
        row0 = self.rank * mr
        ok |= (torch.arange(mr, device=dev) + row0) >= m                  # padding rows
        bad = torch.nonzero(~ok).view(-1) + row0                          # global row ids I own and could not certify
        cnt = torch.tensor([bad.numel()], dtype=torch.int64, device=dev)
        self._allreduce_max(cnt)
        nb_max = int(cnt.item())
        self.last_uncertified = 0
        if nb_max > 0:
            # second round: exact local top-k' of the uncertified rows on every shard, merged by the owners
            ids = torch.full((nb_max,), -1, dtype=torch.int64, device=dev)
            ids[:bad.numel()] = bad
            all_ids = torch.empty((N * nb_max,), dtype=torch.int64, device=dev)
            self._gather(all_ids, ids)
            all_ids = all_ids.view(N, nb_max)
            sel = all_ids[all_ids >= 0]                                     # rank-major order, identical everywhere
            nb = int(sel.numel())
            self.last_uncertified = nb
            Xb = X.index_select(0, sel).contiguous()
            bi, bd = self.local_knn(Xb, kk)
            gi = torch.empty((N, nb, kk), dtype=torch.int64, device=dev)
            gd = torch.empty((N, nb, kk), dtype=torch.float64, device=dev)
            self._gather(gi.view(-1), bi.contiguous().view(-1))
            self._gather(gd.view(-1), bd.contiguous().view(-1))
            mine = torch.nonzero((sel >= row0) & (sel < row0 + mr)).view(-1)
            if mine.numel() > 0:
                fi, fd = self.merge(gi[:, mine, :].contiguous(), gd[:, mine, :].contiguous(), kk, False)
                loc = sel[mine] - row0
                mi[loc] = fi
                md[loc] = fd
        oi = mi[:, d0:d0 + k].contiguous()
        od = md[:, d0:d0 + k].contiguous()
        full_i = torch.empty((m_pad, k), dtype=torch.int64, device=dev)
        full_d = torch.empty((m_pad, k), dtype=torch.float64, device=dev)
        self._gather(full_i.view(-1), oi.view(-1))
        self._gather(full_d.view(-1), od.view(-1))
        return full_i[:m], full_d[:m]


def gpu_callables(index, device_index):
    """local_knn / merge bound to a nabo_amd.KnnIndex and nabo_merge_topk (torch CUDA tensors)."""
    import torch
    from . import _knn
    dev = torch.device("cuda", device_index)

    def local_knn(X, kk):
        m = X.shape[0]
        oi = torch.empty((m, kk), dtype=torch.int64, device=dev)
        od = torch.empty((m, kk), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        index.query_device(X.data_ptr(), m, kk, False, oi.data_ptr(), od.data_ptr())
        return oi, od

    def merge(pi, pd, k, drop_first):
        P, mr, kk = pi.shape
        oi = torch.empty((mr, k), dtype=torch.int64, device=dev)
        od = torch.empty((mr, k), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        _knn.merge_topk_device(pi.data_ptr(), pd.data_ptr(), P, mr, kk, k, drop_first, oi.data_ptr(), od.data_ptr(),
                               device=device_index)
        return oi, od

    def local_cand(X, n_cand):
        m = X.shape[0]
        oi = torch.empty((m, n_cand), dtype=torch.int64, device=dev)
        od = torch.empty((m, n_cand), dtype=torch.float64, device=dev)
        ob = torch.empty((m,), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        index.query_candidates_device(X.data_ptr(), m, n_cand, oi.data_ptr(), od.data_ptr(), ob.data_ptr())
        return oi, od, ob

    return local_knn, merge, local_cand
