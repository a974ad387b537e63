"""The sharded protocol's EXECUTABLE SPECIFICATION over torch.distributed -- TEST HARNESS ONLY.

The product's multi-GPU path is nabo_sharded_query in libnabo_knn.so (nabo_amd/csrc/sharded.hip: RCCL through the C
ABI, no torch; nabo_amd/_sharded.py binds it).  This file states the same protocol in a few lines of torch with the
two compute steps injected, so that its plumbing -- shard bounds, global index bases, the exchange layout, ragged
target counts, the owner's certificate, the second round, the positional drop after the merge, the final gather --
can be run with world_size 2 and 3 on the gloo backend without a GPU (tests/test_dist_gloo.py), and so that the GPU
tests have an independent statement to compare the C implementation with.

Protocol (rank r holds reference rows [base_r, base_r + n_r); every rank sees all target rows):
    1. local k-NN on the shard -> first k' = k + drop_first entries per target, GLOBAL indices, exact float64
       distances;
    2. ONE exchange: all_to_all of the [m, k'] lists so that rank r owns target rows [r*m/N, (r+1)*m/N) from every
       shard;
    3. k-way merge by the canonical (distance, index) order, positional drop applied AFTER the merge
       (nabo/_mapping.py:142 is positional);
    4. all_gather of the merged [m/N, k] slices -> every rank holds the full result.
GLOBAL certification: each shard emits its first Ls < k' order-row entries plus a BOUND on the squared distance of
everything it did not emit; the owner of a target row merges the N*Ls entries and accepts the k'-th merged distance d
when d^2 < min over shards of the bound; rows that fail are re-solved exactly in a second, small round.
"""
import numpy as np


from nabo_amd._sharded import shard_bounds, candidates_per_shard as _candidates_per_shard  # noqa: E402,F401


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

    def __init__(self, dist_module, local_knn, merge, device, local_cand=None, ref_shards=None):
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
        # 2-D layout (nabo_comm_set_ref_shards): R reference pieces x world / R target slices; rank r holds piece r % R
        # (local_knn / local_cand answer for THAT piece) and exchanges inside the group of R ranks of its slice
        self.R = self.world if ref_shards is None else int(ref_shards)
        assert self.world % self.R == 0
        self.group = None
        if self.R != self.world:
            for t in range(self.world // self.R):          # every rank creates every group, in the same order
                grp = dist_module.new_group(list(range(t * self.R, (t + 1) * self.R)))
                if t == self.rank // self.R:
                    self.group = grp

    def _a2a(self, recv, send):
        if self.stage:
            r, s = recv.cpu(), send.cpu()
            self.dist.all_to_all_single(r, s, group=self.group)
            recv.copy_(r)
        else:
            self.dist.all_to_all_single(recv, send, group=self.group)

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

    candidates_per_shard = staticmethod(_candidates_per_shard)

    def query(self, X, m, k, drop_first=False):
        import torch
        kk = k + (1 if drop_first else 0)
        N = self.world
        import os
        force = os.environ.get("NABO_DIST_FORCE_CERT") == "1"          # experiments: protocol overhead at N = 1
        if self.local_cand is not None and (N > 1 or force) and -(-kk // self.R) <= 32:
            return self._query_certified(X, m, k, drop_first)
        assert self.R == N, "the 2-D layout needs the global-certification protocol"
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
        N, dev, R = self.world, self.device, self.R
        Ls = self.candidates_per_shard(kk, R, m)
        mr = (m + N - 1) // N
        m_pad = mr * N
        # my group's slice of the target rows: [s0, s0 + R mr), ms of them exist (R = N: the whole batch)
        gfirst = (self.rank // R) * R
        s0, ms_pad = gfirst * mr, R * mr
        ms = max(0, min(m - s0, ms_pad))
        ci = torch.full((ms_pad, Ls), -1, dtype=torch.int64, device=dev)
        cd = torch.full((ms_pad, Ls), float("inf"), dtype=torch.float64, device=dev)
        cb = torch.full((ms_pad,), float("inf"), dtype=torch.float64, device=dev)
        if ms > 0:
            ci[:ms], cd[:ms], cb[:ms] = self.local_cand(X[s0:s0 + ms], Ls)
        recv_i = torch.empty((R, mr, Ls), dtype=torch.int64, device=dev)
        recv_d = torch.empty((R, mr, Ls), dtype=torch.float64, device=dev)
        recv_b = torch.empty((R, mr), dtype=torch.float64, device=dev)
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
                # every rank re-solved every refused row on ITS piece: the R parts of my group cover all pieces
                fi, fd = self.merge(gi[gfirst:gfirst + R][:, mine, :].contiguous(), gd[gfirst:gfirst + R][:, mine, :].contiguous(), kk, False)
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
    from nabo_amd import _knn
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
