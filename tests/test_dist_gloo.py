"""World-size-2 (and 3) runs of the reference-row-sharded k-NN over the gloo backend on CPU.
The two compute steps are injected (oracle for the per-shard k-NN, merge_numpy for the merge), so
what is exercised is exactly the product's N>1 plumbing in tests/_dist_spec.py (the executable specification of nabo_sharded_query): shard bounds, global
index bases, the all_to_all exchange layout, ragged target counts, the positional drop after the
merge, and the final all_gather -- and that N shards == 1 shard bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, m, n, g, k, drop, metric, out_dir, certified=False, sorted_refs=False, ref_shards=None):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import oracle
    from _dist_spec import ShardedKnn, merge_numpy, shard_bounds
    from nabo_amd._synth import pca_like
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Y = pca_like(n, g, seed=11)
    X = pca_like(m, g, seed=12)
    if sorted_refs:                 # neighbours concentrate in one shard: forces the second (exact) round
        Y = Y[np.argsort(Y[:, 0])]
    R = world if ref_shards is None else ref_shards          # 2-D layout: rank r holds piece r % R of R
    lo, hi = shard_bounds(n, R, rank % R)

    def local_cand(Xt, ncand):
        """What nabo_index_query_candidates returns, stated with the oracle: the shard's first ncand
        order-row entries and (as the bound) the squared distance of the next one."""
        kq = min(ncand + 1, hi - lo)
        i, d = oracle.knn(Xt.numpy(), Y[lo:hi], kq, 0)
        mm = i.shape[0]
        oi = np.full((mm, ncand), -1, dtype=np.int64)
        od = np.full((mm, ncand), np.inf)
        take = min(ncand, kq)
        oi[:, :take] = i[:, :take] + lo
        od[:, :take] = d[:, :take]
        ob = d[:, ncand] ** 2 if kq > ncand else np.full(mm, np.inf)
        return torch.from_numpy(oi), torch.from_numpy(od), torch.from_numpy(np.ascontiguousarray(ob))

    def local_knn(Xt, kk):
        i, d = oracle.knn(Xt.numpy(), Y[lo:hi], kk, metric)
        return torch.from_numpy(i + lo), torch.from_numpy(d)

    def merge(pi, pd, kq, dropq):
        i, d = merge_numpy(pi.numpy(), pd.numpy(), kq, dropq)
        return torch.from_numpy(i), torch.from_numpy(d)

    sk = ShardedKnn(dist, local_knn, merge, torch.device("cpu"), local_cand=local_cand if certified else None,
                    ref_shards=ref_shards)
    oi, od = sk.query(torch.from_numpy(X), m, k, drop)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), idx=oi.numpy(), dist=od.numpy(), unc=sk.last_uncertified)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,m,n,drop,metric,certified,sorted_refs,ref_shards", [
    (2, 101, 600, False, 0, False, False, None), (2, 64, 501, True, 0, False, False, None),
    (3, 50, 400, False, 1, False, False, None),
    (2, 101, 600, True, 0, True, False, None),        # global certification, ragged m, positional drop
    (3, 77, 900, False, 0, True, False, None),
    (3, 60, 900, False, 0, True, True, None),         # one shard holds the neighbours -> second round
    (4, 101, 900, True, 0, True, False, 2),           # 2-D layout: 2 reference pieces x 2 target slices, ragged, drop
    (4, 61, 900, False, 0, True, True, 2),            # ... with the second round
])
def test_sharded_equals_unsharded(tmp_path, world, m, n, drop, metric, certified, sorted_refs, ref_shards):
    import torch.multiprocessing as mp
    import oracle
    from nabo_amd._synth import pca_like
    g, k = 12, 7
    if certified:
        k = 15
    port = _free_port()
    mp.spawn(_worker, args=(world, port, m, n, g, k, drop, metric, str(tmp_path), certified, sorted_refs, ref_shards),
             nprocs=world, join=True)
    Y = pca_like(n, g, seed=11)
    if sorted_refs:
        Y = Y[np.argsort(Y[:, 0])]
    X = pca_like(m, g, seed=12)
    oi, od = oracle.knn(X, Y, k, metric, drop_first=drop)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        assert np.array_equal(z["idx"], oi) and np.array_equal(z["dist"], od)
        if sorted_refs:
            assert int(z["unc"]) > 0         # the exact second round really ran


def test_shard_bounds_cover_everything():
    from nabo_amd._sharded import shard_bounds
    for n in (1, 7, 1000, 1000003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_merge_numpy_matches_oracle_order():
    from _dist_spec import merge_numpy
    rng = np.random.default_rng(0)
    pi = np.stack([np.arange(0, 6)[None].repeat(4, 0), np.arange(6, 12)[None].repeat(4, 0)])
    pd = np.sort(rng.integers(0, 4, size=(2, 4, 6)).astype(np.float64), axis=2)      # many exact ties
    i, d = merge_numpy(pi, pd, 5, True)
    for r in range(4):
        allp = sorted(zip(np.concatenate([pd[0, r], pd[1, r]]), np.concatenate([pi[0, r], pi[1, r]])))
        assert [x[1] for x in allp[1:6]] == list(i[r]) and [x[0] for x in allp[1:6]] == list(d[r])
