#!/usr/bin/env python
"""bench.py -- k-NN mapping throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--targets M --refs N --dims D --neighbors K]

A step = one full k-NN build of the workload: pack the (sharded) references, fused
distance + top-k on the MFMA pipe, float64 refine, and for N>1 the RCCL exchange + merge.
Inputs (float64 PCA-like embeddings) are already resident in HBM when the timed region starts.
Default workload: BASELINE.json configs[2], 1M ref x 1M target, d=50, k=15, Euclidean.
For N>1 the driver launches this file under torch.distributed.run (one rank per GPU); reference
rows are sharded, total work is fixed ("strong" scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz


def cpu_baseline(d, k, metric=0):
    """The oracle (bit-equal port of the reference's CPU arithmetic + selection), all host
    cores, on a bounded sample of the same workload."""
    import oracle
    from nabo_amd._synth import pca_like
    cores = max(1, min(os.cpu_count() or 1, 16))      # the GPU box's CPU share for one GPU
    n_s, m_s = 100000, 1024
    Y = pca_like(n_s, d, seed=1003)
    X = pca_like(m_s, d, seed=2003)
    oracle.knn(X[:64], Y, k, metric, nthreads=cores)          # warm up threads / page in
    t0 = time.perf_counter()
    oracle.knn(X, Y, k, metric, nthreads=cores)
    dt = time.perf_counter() - t0
    # scale the sample towards ~10-20 s of CPU work
    reps = int(max(1, min(16, 12.0 / max(dt, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        oracle.knn(X, Y, k, metric, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": reps * m_s * n_s / dt, "unit": "cell-pair distances/s", "cores": cores, "kind": "port",
            "sample": "%d x (%d targets x %d refs), d=%d, k=%d, float64 sequential + ordered top-k, OpenMP"
                      % (reps, m_s, n_s, d, k)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--targets", dest="m", type=int, default=1000000)
    ap.add_argument("--refs", dest="n", type=int, default=1000000)
    ap.add_argument("--dims", dest="d", type=int, default=50)
    ap.add_argument("--neighbors", dest="k", type=int, default=15)
    ap.add_argument("--metric", choices=["euclidean", "cosine"], default="euclidean",
                    help="cosine is an extension (BASELINE configs[4]); the headline workload is euclidean")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = world > 1 or os.environ.get("NABO_BENCH_FORCE_DIST") == "1"    # rehearsal of the N>1 code on 1 GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world
    if use_dist:
        # torch bundles its own HIP runtime: it must be loaded BEFORE libnabo_knn.so so that the
        # process ends up with ONE libamdhip64 (the library then binds to torch's copy)
        import torch  # noqa: F401
    import nabo_amd
    from nabo_amd import _knn
    from nabo_amd._dist import shard_bounds
    from nabo_amd._synth import pca_like
    if nabo_amd.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")

    m, n, d, k = a.m, a.n, a.d, a.k
    backend = os.environ.get("NABO_BENCH_BACKEND", "nccl")     # "gloo": N ranks rehearsed on ONE GPU
    dev = local_rank if (world > 1 and backend == "nccl") else 0
    lo, hi = shard_bounds(n, world, rank)
    Y = pca_like(n, d, seed=1003)[lo:hi]
    X = pca_like(m, d, seed=2003)

    if use_dist:
        import torch
        import torch.distributed as dist
        from nabo_amd._dist import ShardedKnn, gpu_callables
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
        # build the communicators before anything is timed (RCCL creates them lazily, per collective kind)
        cdev = ("cuda:%d" % dev) if backend == "nccl" else "cpu"
        w_ = dist.get_world_size()
        a2a_s, a2a_r = torch.zeros(w_ * 4, dtype=torch.float64, device=cdev), torch.zeros(w_ * 4, dtype=torch.float64, device=cdev)
        dist.all_to_all_single(a2a_r, a2a_s)
        ag = torch.zeros(w_ * 4, dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(ag, torch.zeros(4, dtype=torch.int64, device=cdev))
        dist.all_reduce(torch.zeros(1, dtype=torch.int64, device=cdev), op=dist.ReduceOp.MAX)
        dist.barrier()
        tX = torch.from_numpy(X).to("cuda:%d" % dev)
        tY = torch.from_numpy(np.ascontiguousarray(Y)).to("cuda:%d" % dev)
        torch.cuda.synchronize()
        x_ptr, y_ptr = tX.data_ptr(), tY.data_ptr()
    else:
        dX = _knn.DeviceBuffer(X.nbytes, dev).upload(X)
        dY = _knn.DeviceBuffer(Y.nbytes, dev).upload(np.ascontiguousarray(Y))
        dI = _knn.DeviceBuffer(m * k * 8, dev)
        dD = _knn.DeviceBuffer(m * k * 8, dev)
        x_ptr, y_ptr = dX.ptr, dY.ptr

    metric_id = nabo_amd.COSINE if a.metric == "cosine" else nabo_amd.EUCLIDEAN
    index = nabo_amd.KnnIndex(hi - lo, d, metric=metric_id, ref_index_base=lo, device=dev)
    stats = []

    if use_dist:
        lk, mg, lc = gpu_callables(index, dev)
        if os.environ.get("NABO_DIST_LOCAL_CERT") == "1":      # A/B: every shard certifies its own top-k'
            lc = None
        sk = ShardedKnn(dist, lk, mg, torch.device("cuda", dev), local_cand=lc)

        def step():
            index.set_ref(y_device_ptr=y_ptr)
            out = sk.query(tX, m, k, False)
            stats.append(index.last_stats())
            return out

        def sync():
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
    else:
        def step():
            index.set_ref(y_device_ptr=y_ptr)
            index.query_device(x_ptr, m, k, False, dI.ptr, dD.ptr)
            stats.append(index.last_stats())

        def sync():
            _knn._lib.check(_knn._lib.lib().nabo_dev_synchronize(dev))

    for _ in range(a.warmup):
        step()
    stats.clear()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=("cuda:%d" % dev) if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # light self-check outside the timed region (rank 0): sorted rows, valid indices
    if not use_dist and not os.environ.get("NABO_DEBUG_ABLATE"):
        gi = dI.download((m, k), np.int64)
        gd = dD.download((m, k), np.float64)
        assert gi.min() >= 0 and gi.max() < n and (np.diff(gd[:: max(1, m // 4096)], axis=1) >= 0).all()

    if use_dist and os.environ.get("NABO_BENCH_CHECK") == "1":
        # rehearsal check: the sharded result must equal one unsharded index on the same data
        full = pca_like(n, d, seed=1003)
        ref_ix = nabo_amd.KnnIndex(n, d, metric=metric_id, device=dev).set_ref(full)
        ri, rd = ref_ix.query(X, k)
        ref_ix.close()
        oi, od = step()
        same = bool((oi.cpu().numpy() == ri).all() and (od.cpu().numpy() == rd).all())
        print("rank %d sharded == unsharded: %s" % (rank, same), flush=True)
        assert same
    if rank == 0:
        ms_step = dt / a.steps * 1e3
        t_kernel = float(np.mean([s["ms_topk"] for s in stats])) * 1e-3       # HIP events, kernel's own stream
        flops = 2.0 * m * (hi - lo) * d                                       # algorithmic: the -2XY^T term
        achieved = flops / t_kernel / 1e12
        traffic, traffic_note = None, None
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
            wl = "%dk ref x %dk target, d=%d, k=%d, %s, refs sharded %d-way" % (n // 1000, m // 1000, d, k, a.metric, world)
            if wl in tj and not os.environ.get("NABO_L2_MODE"):
                # bytes per step (= the two launches of the dominant kernel), gfx950 FETCH_SIZE correction applied
                traffic = (2.0 * tj[wl]["fetch_kb"] + tj[wl]["write_kb"]) * 1024
                traffic_note = "bytes per step from " + tj[wl]["source"]
        except Exception:
            traffic, traffic_note = None, None
        line = {
            "metric": "cell-pair distances/s (k-NN build, 1Mx1M d=50 k=15)" if (m, n, d, k) == (1000000, 1000000, 50, 15)
                      else "cell-pair distances/s (k-NN build)",
            "value": m * n * a.steps / dt, "unit": "cell-pair distances/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "knn_build_s": dt / a.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dk ref x %dk target, d=%d, k=%d, %s, refs sharded %d-way"
                                   % (n // 1000, m // 1000, d, k, a.metric, world),
                       "parallelism": "ref-shard%d" % world,
                       "arithmetic": "fp32 MFMA score filter, float64 re-evaluation: indices and distances equal the "
                                     "reference's float64 path"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes": 4.0 * d * (m + (hi - lo)) + 12.0 * k * m,      # fp32 operands + (i32, f64) results
                         "kernel": "l2_topk_kernel (v_mfma_f32_32x32x2_f32)", "kernel_ms": t_kernel * 1e3},
            "phases_ms": {key: float(np.mean([s[key] for s in stats])) for key in
                          ("ms_pack", "ms_topk", "ms_refine", "ms_fallback", "ms_total")},
            "fallback_rows": int(np.max([s["fallback_rows"] for s in stats])),
        }
        if world == 1 and not use_dist and not os.environ.get("NABO_L2_MODE") and not os.environ.get("NABO_DEBUG_ABLATE") \
                and os.environ.get("NABO_BENCH_ALT", "1") == "1" and d <= 64 and a.metric == "euclidean":
            # informational, NOT the headline: the same step with the filter on the f16 matrix pipe
            # (3-product hi/lo split), outside the timed region; results must be the same bits
            os.environ["NABO_L2_MODE"] = "f16x3"
            alt = nabo_amd.KnnIndex(n, d, metric=nabo_amd.EUCLIDEAN, device=dev)
            del os.environ["NABO_L2_MODE"]
            aI, aD = _knn.DeviceBuffer(m * k * 8, dev), _knn.DeviceBuffer(m * k * 8, dev)
            ts = []
            for _ in range(3):
                sync()
                t0 = time.perf_counter()
                alt.set_ref(y_device_ptr=y_ptr)
                alt.query_device(x_ptr, m, k, False, aI.ptr, aD.ptr)
                sync()
                ts.append(time.perf_counter() - t0)
            st = alt.last_stats()
            same = bool(np.array_equal(aI.download((m, k), np.int64), gi) and np.array_equal(aD.download((m, k), np.float64), gd))
            alt.close()
            line["alt_f16x3"] = {"ms_per_step": min(ts[1:]) * 1e3, "value": m * n / min(ts[1:]), "kernel_ms": st["ms_topk"],
                                 "fallback_rows": st["fallback_rows"], "same_bits_as_f32_path": same,
                                 "note": "opt-in NABO_L2_MODE=f16x3; 3x v_mfma_f32_32x32x16_f16 per K-slab"}
        if not a.no_cpu_baseline and world == 1 and not use_dist:
            line["cpu_baseline"] = cpu_baseline(d, k, metric_id)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
