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

    def __init__(self, dist_module, local_knn, merge, device):
        self.dist = dist_module
        self.local_knn = local_knn
        self.merge = merge
        self.device = device
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

    def query(self, X, m, k, drop_first=False):
        import torch
        kk = k + (1 if drop_first else 0)
        idx, dst = self.local_knn(X, kk)
        N = self.world
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

    return local_knn, merge
