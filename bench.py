#!/usr/bin/env python
"""bench.py -- k-NN mapping throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--targets M --refs N --dims D --neighbors K]
                    [--metric euclidean|cosine|canberra]

A step = one full k-NN build of the workload: pack the (sharded) references, fused distance + top-k filter on the
matrix pipe, float64 refine + certification, and for N>1 the RCCL exchange + merge (nabo_sharded_query).  Inputs
(float64 PCA-like embeddings) are already resident in HBM when the timed region starts.
Default workload: BASELINE.json configs[2], 1M ref x 1M target, d=50, k=15, Euclidean.

For N>1 the driver launches this file under torch.distributed.run, one rank per GPU; only the launcher is torch: the
ranks read RANK / LOCAL_RANK / WORLD_SIZE from the environment, rank 0 hands the RCCL unique id to the others through
a file (single node), and every collective -- the data path's and the barrier / max-over-ranks of the timing -- goes
through libnabo_knn.so's C ABI (nabo_comm_*).  Reference rows are sharded, total work is fixed ("strong" scaling).

The default N=1 line also carries (outside the timed region): `canberra` -- the reference's default target<->reference
metric (nabo/_mapping.py:122-124) on the same 1M x 1M workload, checked against the oracle on sampled rows;
`alt` -- the other Euclidean filter kernel on the same step (same bits required); `cpu_baseline` -- the oracle on this
box's host cores (OpenMP and single thread) and the reference-style end-to-end run of BASELINE configs[0].
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# MI355X_MICROARCH.md: 256 CU x 4 SIMD x 2.4 GHz
PEAK_F32_MFMA_TFLOPS = 157.3        # v_mfma_f32_32x32x2_f32: 64 flop/clk/SIMD
# v_mfma_f32_32x32x16_f16 / 16x16x32_f16: 1024 flop/clk/SIMD (dense) at the nominal 2.4 GHz.  The part holds that clock
# only on all-zero operands; a bare MFMA loop on random data sustains 1.6 PF (32x32x16) / 1.9-2.0 PF (16x16x32) --
# tools/mfma_clock_lab.hip, profiles/r2_mfma_clock_lab.txt, DESIGN.md 4.1b.  The roofline is priced against the nominal peak.
PEAK_F16_MFMA_TFLOPS = 2516.6
# vector ISSUE peak for the mod-Canberra counting pass, MEASURED on this part by tools/issue_lab.hip for the pass's own
# instruction mix (v_sub_u32, v_sub_u32, v_bitop3_b32, v_bcnt_u32_b32 on independent chains, 8 waves per SIMD):
# 0.263 wave-instructions per clock and SIMD = 646.9 G/s (v_sub / v_bitop3 alone issue at 0.43 per clock, v_bcnt and
# the packed-f16 kinds at 0.24; the nominal "one per 4 clocks" would be 614.4) -- profiles/r2_issue_lab.txt
PEAK_VALU_GINST = 646.9


def _time_knn(oracle, X, Y, k, metric, threads, budget_s):
    oracle.knn(X[:32], Y, k, metric, 0.25, nthreads=threads)
    t0 = time.perf_counter()
    oracle.knn(X, Y, k, metric, 0.25, nthreads=threads)
    dt = time.perf_counter() - t0
    reps = int(max(1, min(256, budget_s / max(dt, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        oracle.knn(X, Y, k, metric, 0.25, nthreads=threads)
    dt = time.perf_counter() - t0
    return reps, dt


def cpu_baseline(d, k, metric=0):
    """The oracle (bit-equal port of the reference's CPU arithmetic + ordered selection, kind "port") on a bounded
    sample of the same workload: all host cores of this box's share (OpenMP), and ONE thread -- the reference itself is
    single-core numba (nabo/_mapping.py:16,29).  ~12 s of CPU work each."""
    import oracle
    from nabo_amd._synth import pca_like
    cores = max(1, min(os.cpu_count() or 1, 16))      # the GPU box's CPU share for one GPU
    n_s = 100000
    Y = pca_like(n_s, d, seed=1003)
    X = pca_like(1024, d, seed=2003)
    reps, dt = _time_knn(oracle, X, Y, k, metric, cores, 12.0)
    out = {"value": reps * X.shape[0] * n_s / dt, "unit": "cell-pair distances/s", "cores": cores, "kind": "port",
           "sample": "%d x (%d targets x %d refs), d=%d, k=%d, float64 sequential + ordered top-k, OpenMP; %.1f s of CPU work"
                     % (reps, X.shape[0], n_s, d, k, dt)}
    X1 = X[:128]
    reps, dt = _time_knn(oracle, X1, Y, k, metric, 1, 12.0)
    out["single_thread"] = {"value": reps * X1.shape[0] * n_s / dt, "unit": "cell-pair distances/s", "cores": 1, "kind": "port",
                            "sample": "%d x (%d targets x %d refs), same arithmetic, 1 thread (the reference's numba kernels "
                                      "are single-core); %.1f s of CPU work" % (reps, X1.shape[0], n_s, dt)}
    out["extrapolated_1Mx1M_seconds"] = {"cores_%d" % cores: 1e12 / out["value"], "cores_1": 1e12 / out["single_thread"]["value"],
                                         "note": "extrapolated from the samples, never measured"}
    return out


def c1_end_to_end(gpu):
    """BASELINE configs[0] in the reference's own per-cell HDF5 layout on the host (oracle/c1_end_to_end.py) next to
    nabo_amd.Mapping on the same files; needs an interpreter with h5py."""
    for py in (sys.executable, "/opt/conda/bin/python3.9", "/opt/conda/bin/python"):
        if not os.path.exists(py):
            continue
        if subprocess.run([py, "-c", "import h5py, numpy"], stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode:
            continue
        cmd = [py, os.path.join(REPO, "oracle", "c1_end_to_end.py")] + (["--gpu"] if gpu else [])
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
        if r.returncode == 0:
            return json.loads(r.stdout.strip().splitlines()[-1])
        return {"error": r.stderr[-400:]}
    return {"skipped": "no interpreter with h5py"}


def pmc_record(kind, workload, so_digest):
    """HBM-side bytes / instruction counts of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc.json, collected with tools/pmc_round.sh in runs of their own).  Only a record taken on THIS build of
    the library (same .so digest) and workload is reported as a number; anything else is a pointer, never a value."""
    try:
        rec = json.load(open(os.path.join(REPO, "profiles", "pmc.json")))[kind][workload]
    except Exception:       # noqa: BLE001
        return None
    return rec if rec.get("so_digest") == so_digest else {"stale": True, "source": rec.get("source"),
                                                          "so_digest_of_record": rec.get("so_digest")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--targets", dest="m", type=int, default=1000000)
    ap.add_argument("--refs", dest="n", type=int, default=1000000)
    ap.add_argument("--dims", dest="d", type=int, default=50)
    ap.add_argument("--neighbors", dest="k", type=int, default=15)
    ap.add_argument("--metric", choices=["euclidean", "cosine", "canberra"], default="euclidean",
                    help="cosine is an extension (BASELINE configs[4]); canberra is the reference's target<->reference "
                         "metric; the headline workload is euclidean")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the canberra / alt / C1 blocks (profiling runs)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        a.gpus = world
    import nabo_amd
    from nabo_amd import _knn, _lib, _sharded
    from nabo_amd._sharded import shard_bounds
    from nabo_amd._synth import pca_like
    if nabo_amd.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")

    m, n, d, k = a.m, a.n, a.d, a.k
    dev = local_rank if world > 1 else 0
    loop = int(os.environ.get("NABO_BENCH_LOOPBACK", "0"))     # rehearsal: N shard-ranks as threads on ONE GPU
    metric_id = {"euclidean": nabo_amd.EUCLIDEAN, "cosine": nabo_amd.COSINE, "canberra": nabo_amd.MOD_CANBERRA}[a.metric]
    Yfull = pca_like(n, d, seed=1003)
    X = pca_like(m, d, seed=2003)
    force_comm = os.environ.get("NABO_BENCH_FORCE_COMM") == "1"      # rehearsal: the N>1 code path with one rank
    comm = _sharded.Comm.from_env(dev) if (world > 1 or force_comm) else None
    # Layout of the N ranks: R reference pieces x N / R target slices.  Every row fills a candidate list on
    # EVERY piece, so that part of a rank's work does not shrink with the piece; a 1M x 50 reference set (0.4 GB of 288)
    # has no need to be cut eight ways.  Default: two pieces from four ranks on (the exchange / merge / certificate of the
    # prescribed ref-sharded form inside each pair, the gather over all ranks); NABO_REF_SHARDS=<N> is the 1-D form.
    # (Modified Canberra shards exchange certified lists: 1-D only.)
    ranks = max(world, loop, 1)
    R = int(os.environ.get("NABO_REF_SHARDS", "0"))
    if R <= 0:
        R = 2 if (ranks >= 4 and ranks % 2 == 0 and a.metric != "canberra") else ranks
    if ranks % R:
        raise SystemExit("NABO_REF_SHARDS must divide the number of ranks")
    if comm is not None and R != world:
        comm.set_ref_shards(R)
    lo, hi = shard_bounds(n, R if world > 1 else 1, rank % R if world > 1 else 0)
    dX = _knn.DeviceBuffer(X.nbytes, dev).upload(X)
    dY = _knn.DeviceBuffer((hi - lo) * d * 8, dev).upload(np.ascontiguousarray(Yfull[lo:hi]))
    dI = _knn.DeviceBuffer(m * k * 8, dev)
    dD = _knn.DeviceBuffer(m * k * 8, dev)
    index = nabo_amd.KnnIndex(hi - lo, d, metric=metric_id, dist_factor=0.25, ref_index_base=lo, device=dev)
    stats, xstats = [], []

    if loop > 1:
        group = _sharded.LoopbackGroup(loop, dev, n, d, metric_id, Yfull, ref_shards=R)

        def step():
            group.set_ref()
            group.query_device(dX.ptr, m, k, False, dI.ptr, dD.ptr)
            stats.append(group.indices[0].last_stats())
            xstats.append(group.last_stats(0))
    elif comm is not None:
        sk = _sharded.ShardedIndex(comm, index, os.environ.get("NABO_BENCH_PROTOCOL", "global" if force_comm and a.metric != "canberra" else "auto"))

        def step():
            index.set_ref(y_device_ptr=dY.ptr)
            sk.query_device(dX.ptr, m, k, False, dI.ptr, dD.ptr)
            stats.append(index.last_stats())
            xstats.append(sk.last_stats())
    else:
        def step():
            index.set_ref(y_device_ptr=dY.ptr)
            index.query_device(dX.ptr, m, k, False, dI.ptr, dD.ptr)
            stats.append(index.last_stats())

    def sync():
        _lib.check(_lib.lib().nabo_dev_synchronize(dev))
        if comm is not None:
            comm.barrier()                     # RCCL all-reduce on the communicator's stream + host wait
            _lib.check(_lib.lib().nabo_dev_synchronize(dev))

    for _ in range(a.warmup):
        step()
    stats.clear()
    xstats.clear()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    rank_max = {}
    if comm is not None:
        dt = comm.allreduce_max(dt)            # MAX over ranks
        # the slowest rank's phases (every rank enters these collectives; rank 0 prints): a first N>1 record must
        # show where the time went without a second run
        for key, src in (("ms_topk", stats), ("ms_total", stats), ("ms_local", xstats), ("ms_exchange", xstats),
                         ("ms_merge", xstats), ("ms_second", xstats), ("ms_gather", xstats)):
            rank_max[key] = comm.allreduce_max(float(np.mean([x[key] for x in src])))

    # light self-check outside the timed region: sorted rows, valid indices
    gi = gd = None
    if not os.environ.get("NABO_DEBUG_ABLATE"):
        gi = dI.download((m, k), np.int64)
        gd = dD.download((m, k), np.float64)
        assert gi.min() >= 0 and gi.max() < n and (np.diff(gd[:: max(1, m // 4096)], axis=1) >= 0).all()
    if os.environ.get("NABO_BENCH_CHECK") == "1" and (comm is not None or loop > 1):
        # rehearsal check: the sharded result must equal one unsharded index on the same data
        ref_ix = nabo_amd.KnnIndex(n, d, metric=metric_id, dist_factor=0.25, device=dev).set_ref(Yfull)
        ri, rd = ref_ix.query(X, k)
        ref_ix.close()
        same = bool(np.array_equal(gi, ri) and np.array_equal(gd, rd))
        print("rank %d sharded == unsharded: %s" % (rank, same), flush=True)
        assert same

    if rank == 0:
        shards, slices = R, ranks // R
        so = _lib.so_digest()
        kern = index.last_kernel() if loop <= 1 else group.indices[0].last_kernel()
        ms_step = dt / a.steps * 1e3
        t_kernel = float(np.mean([s["ms_topk"] for s in stats])) * 1e-3       # HIP events on the kernel's own stream
        n_shard = (hi - lo) if loop <= 1 else shard_bounds(n, R, 0)[1]
        workload = "%dk ref x %dk target, d=%d, k=%d, %s, refs sharded %d-way" % (n // 1000, m // 1000, d, k, a.metric, shards)
        if slices > 1:
            workload += " x %d target slices" % slices
        line = {
            "metric": "cell-pair distances/s (k-NN build, 1Mx1M d=50 k=15)" if (m, n, d, k) == (1000000, 1000000, 50, 15)
                      else "cell-pair distances/s (k-NN build)",
            "value": m * n * a.steps / dt, "unit": "cell-pair distances/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "knn_build_s": dt / a.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if "f32_32x32x2" in kern else ("f16" if "f16" in kern else "f64"), "data": "synthetic",
            "config": {"workload": workload, "parallelism": "ref-shard%d" % shards + ("-slice%d" % slices if slices > 1 else ""),
                       "arithmetic": "low-precision score filter on the matrix pipe, float64 re-evaluation + certification: "
                                     "indices and distances equal the reference's float64 path"},
            "phases_ms": {key: float(np.mean([s[key] for s in stats])) for key in
                          ("ms_pack", "ms_topk", "ms_refine", "ms_fallback", "ms_total")},
            "fallback_rows": int(np.max([s["fallback_rows"] for s in stats])),
            "so_digest": so,
        }
        if a.metric != "canberra":
            flops = 2.0 * (m / slices) * n_shard * d                          # algorithmic (this rank): the -2XY^T term
            achieved = flops / t_kernel / 1e12
            f16 = "f16" in kern
            peak = PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS
            rec = pmc_record("traffic", workload, so)
            line["roofline"] = {
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "peak_dtype": "f16 dense MFMA" if f16 else "f32 MFMA",
                "frac_of_f32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS, "frac_of_f16_mfma_peak": achieved / PEAK_F16_MFMA_TFLOPS,
                "traffic": (rec or {}).get("bytes_per_step"), "traffic_record": rec,
                "algorithmic_bytes": 4.0 * d * (m + n_shard) + 12.0 * k * m,      # fp32 operands + (i32, f64) results
                "kernel": kern, "kernel_ms": t_kernel * 1e3,
                "matrix_pipe_busy": (rec or {}).get("matrix_pipe_busy"),
                "clock_ghz_held": (rec or {}).get("clock_ghz_held")}
        else:
            rec = pmc_record("canberra", workload, so)
            line["roofline"] = canberra_roofline(rec, t_kernel, kern)
        if xstats:
            line["sharded"] = {"world": world, "loopback_ranks": loop if loop > 1 else None,
                               "layout": {"ref_shards": shards, "target_slices": slices,
                                          "note": "NABO_REF_SHARDS=%d is the 1-D form (one piece per rank)" % ranks},
                               "exchange_ms": float(np.mean([x["ms_exchange"] for x in xstats])),
                               "merge_ms": float(np.mean([x["ms_merge"] for x in xstats])),
                               "gather_ms": float(np.mean([x["ms_gather"] for x in xstats])),
                               "second_round_ms": float(np.mean([x["ms_second"] for x in xstats])),
                               "local_query_ms": float(np.mean([x["ms_local"] for x in xstats])),
                               "candidates_per_shard": int(xstats[-1]["candidates"]),
                               "last_uncertified": int(max(x["uncertified"] for x in xstats)),
                               "rank0_ms_topk": t_kernel * 1e3,
                               "max_over_ranks_ms": rank_max or None}
        extras = comm is None and loop <= 1 and not a.no_extras and not os.environ.get("NABO_DEBUG_ABLATE")
        if extras and a.metric == "euclidean" and not os.environ.get("NABO_L2_MODE"):
            line["alt"] = alt_block(nabo_amd, _knn, index, kern, dev, n, m, d, k, dY, dX, gi, gd, sync)
        if extras and a.metric == "euclidean" and (m, n) == (1000000, 1000000):
            line["canberra"] = canberra_block(nabo_amd, _knn, dev, n, m, d, k, dY, dX, X, Yfull, so, sync)
        if not a.no_cpu_baseline and comm is None and loop <= 1:
            line["cpu_baseline"] = cpu_baseline(d, k, metric_id)
            if extras:
                line["cpu_baseline"]["c1_end_to_end"] = c1_end_to_end(gpu=True)
        print(json.dumps(line), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()


def canberra_roofline(rec, t_kernel, kern):
    """The mod-Canberra filter is vector-ALU work (no contraction): the bound is the chip's vector ISSUE rate, and what
    is priced against it is the number of vector instructions the kernel ACTUALLY issued (SQ_INSTS_VALU of a
    rocprofv3 --pmc pass on this build), so the fraction cannot exceed 1."""
    insts = (rec or {}).get("valu_insts_per_step")
    ach = insts / t_kernel / 1e9 if insts else None
    return {"bound": "valu-issue", "achieved": ach, "peak": PEAK_VALU_GINST, "unit": "G wave-instructions/s",
            "frac": ach / PEAK_VALU_GINST if ach else None, "valu_insts_per_step": insts, "pmc_record": rec,
            "kernel": kern, "kernel_ms": t_kernel * 1e3}


def canberra_block(nabo_amd, _knn, dev, n, m, d, k, dY, dX, X, Yfull, so, sync):
    """The reference's default target<->reference metric on the same workload (outside the headline's timed region)."""
    import oracle
    ix = nabo_amd.KnnIndex(n, d, metric=nabo_amd.MOD_CANBERRA, dist_factor=0.25, device=dev)
    cI, cD = _knn.DeviceBuffer(m * k * 8, dev), _knn.DeviceBuffer(m * k * 8, dev)
    ts, st = [], None
    for it in range(3):
        sync()
        t0 = time.perf_counter()
        ix.set_ref(y_device_ptr=dY.ptr)
        ix.query_device(dX.ptr, m, k, False, cI.ptr, cD.ptr)
        sync()
        ts.append(time.perf_counter() - t0)
        st = ix.last_stats()
    kern = ix.last_kernel()
    ix.close()
    gi, gd = cI.download((m, k), np.int64), cD.download((m, k), np.float64)
    cI.free(); cD.free()
    rows = np.random.default_rng(11).choice(m, 32, replace=False)
    oi, od = oracle.knn(X[rows], Yfull, k, oracle.MOD_CANBERRA, 0.25, nthreads=max(1, min(os.cpu_count() or 1, 16)))
    same = bool(np.array_equal(gi[rows], oi) and np.array_equal(gd[rows], od))
    best = min(ts[1:])
    workload = "%dk ref x %dk target, d=%d, k=%d, canberra, refs sharded 1-way" % (n // 1000, m // 1000, d, k)
    return {"workload": "1M ref x 1M target, d=50, k=15, modified Canberra (nabo/_mapping.py:29-45), dist_factor 0.25",
            "ms_per_step": best * 1e3, "value": m * n / best, "unit": "cell-pair distances/s",
            "phases_ms": {key: st[key] for key in ("ms_pack", "ms_topk", "ms_refine", "ms_fallback", "ms_total")},
            "fallback_rows": st["fallback_rows"], "sampled_rows_equal_oracle": same,
            "roofline": canberra_roofline(pmc_record("canberra", workload, so), st["ms_topk"] * 1e-3, kern)}


def alt_block(nabo_amd, _knn, index, kern, dev, n, m, d, k, dY, dX, gi, gd, sync):
    """The same step with the OTHER Euclidean filter kernel (fp32 MFMA <-> f16x3 split); results must be the same bits."""
    other = "f32" if "f16" in kern else "f16x3"
    os.environ["NABO_L2_MODE"] = other
    try:
        alt = nabo_amd.KnnIndex(n, d, metric=nabo_amd.EUCLIDEAN, device=dev)
    finally:
        del os.environ["NABO_L2_MODE"]
    aI, aD = _knn.DeviceBuffer(m * k * 8, dev), _knn.DeviceBuffer(m * k * 8, dev)
    ts = []
    for _ in range(3):
        sync()
        t0 = time.perf_counter()
        alt.set_ref(y_device_ptr=dY.ptr)
        alt.query_device(dX.ptr, m, k, False, aI.ptr, aD.ptr)
        sync()
        ts.append(time.perf_counter() - t0)
    st = alt.last_stats()
    akern = alt.last_kernel()
    same = bool(np.array_equal(aI.download((m, k), np.int64), gi) and np.array_equal(aD.download((m, k), np.float64), gd))
    alt.close()
    aI.free(); aD.free()
    return {"mode": other, "kernel": akern, "ms_per_step": min(ts[1:]) * 1e3, "value": m * n / min(ts[1:]),
            "kernel_ms": st["ms_topk"], "fallback_rows": st["fallback_rows"], "same_bits_as_headline_path": same}


if __name__ == "__main__":
    main()
