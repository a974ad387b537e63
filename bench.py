#!/usr/bin/env python
"""bench.py -- k-NN mapping throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--targets M --refs N --dims D --neighbors K]
                    [--metric euclidean|cosine|canberra]

A step = one full k-NN build of the workload: pack the (sharded) references, fused distance + top-k filter on the
matrix pipe, float64 refine + certification, and for N>1 the RCCL exchange + merge (nabo_sharded_query).  Inputs
(float64 PCA-like embeddings) are already resident in HBM when the timed region starts.
Default workload: BASELINE.json configs[2], 1M ref x 1M target, d=50, k=15, Euclidean; with N>1 GPUs BASELINE configs[3]:
the same workload with the reference rows sharded N ways + RCCL exchange ("strong" scaling: total work is fixed).

N>1 runs however the file is launched:
  * under a launcher (python -m torch.distributed.run ... bench.py --gpus N: WORLD_SIZE / RANK / LOCAL_RANK in the
    environment) every process is one rank on its own GPU; only the launcher is torch -- rank 0 hands the RCCL unique id
    to the others through a file, every collective (the data path's and the barrier / max-over-ranks of the timing) goes
    through libnabo_knn.so's C ABI (nabo_comm_*);
  * plain `python bench.py --gpus N` (no launcher environment) drives the N devices from THIS process: one communicator
    per GPU (nabo_comm_create_all = ncclCommInitAll) and one host thread per rank (nabo_amd.ShardedGroup).  Fewer than
    N visible GPUs is an error (exit code 2), never a silent one-GPU run.
The N>1 headline is the layout BASELINE configs[3] names -- references sharded N ways, one piece per rank; the
2 x N/2 layout (two reference pieces x N/2 target slices; N >= 4) and pure target slicing (every
rank holds all the references and answers its own m/N rows: no exchange) are measured in the same invocation as the
`alt_layout` / `alt_layout_target_slices` blocks and must give the same bits.  (NABO_BENCH_LOOPBACK=N rehearses all of
it with N shard-ranks on ONE GPU through the loopback transport.)

The default N=1 line also carries (outside the timed region): `canberra` -- the reference's default target<->reference
metric (nabo/_mapping.py:122-124) on the same 1M x 1M workload, checked against the oracle on sampled rows;
`alt`, `alt_f16x3` -- the fp32-MFMA filter / the f16x3 split as the first pass on the same step (same bits required); `cpu_baseline` -- the oracle on this
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
# the packed-f16 kinds at 0.24) -- profiles/r2_issue_lab.txt.  The NOMINAL rate, one instruction per 4 clocks and SIMD
# (1024 SIMD x 2.4 GHz / 4), is 614.4 G/s: both fractions are reported.
PEAK_VALU_GINST = 646.9
NOMINAL_VALU_GINST = 614.4
# the bit-sliced counting pass (canberra_bits.hip) issues v_bitop3 / v_and / v_xor / v_add almost exclusively: kinds that
# issue at 0.43 per clock and SIMD on this part (same microbenchmark, same file) = 1056.8 G/s
PEAK_VALU_GINST_BITS = 1056.8


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


def pmc_record(kind, workload, digest):
    """HBM-side bytes / instruction counts of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc.json, collected in runs of their own: tools/r3_pmc.sh).  A record is reported as a number only when it
    was taken on THESE SOURCES of the kernel (a digest of the kernel's source files + compiler flags -- reproducible,
    unlike the bytes of a .so) and on this workload; anything else is a pointer, never a value.  Either way the fields
    are counters of an EARLIER run of the same code, not of this one: `measured_in_this_run` is false."""
    try:
        rec = json.load(open(os.path.join(REPO, "profiles", "pmc.json")))[kind][workload]
    except Exception:       # noqa: BLE001
        return None
    if rec.get("src_digest") == digest:
        return dict(rec, measured_in_this_run=False)
    return {"stale": True, "source": rec.get("source"), "src_digest_of_record": rec.get("src_digest"),
            "measured_in_this_run": False}


class Layout:
    """One way of running the step: `ranks` shard-ranks as R reference pieces x ranks/R target slices.
    kind "single": one GPU, no communicator; "launcher": this process is one rank of `world` (one process per GPU);
    "threads": this process drives `ranks` GPUs, one host thread per rank (ShardedGroup over RCCL);
    "loopback": `ranks` shard-ranks on ONE GPU (rehearsal)."""

    def __init__(self, a, kind, ranks, R, comm, rank, dev, X, Yfull, metric_id):
        import nabo_amd
        from nabo_amd import _knn, _lib, _sharded
        self.a, self.kind, self.ranks, self.R, self.comm, self.rank, self.dev = a, kind, ranks, R, comm, rank, dev
        self._lib, self._knn = _lib, _knn
        m, n, d, k = a.m, a.n, a.d, a.k
        self.group = self.index = self.sk = None
        self.stats, self.xstats = [], []
        if kind in ("threads", "loopback"):
            devices = [dev] * ranks if kind == "loopback" else list(range(ranks))
            self.group = _sharded.ShardedGroup(devices, n, d, metric_id, Yfull, ref_shards=R,
                                               transport="loopback" if kind == "loopback" else "rccl")
            uniq = sorted(set(devices))
            self.dX = {dv: _knn.DeviceBuffer(X.nbytes, dv).upload(X) for dv in uniq}
            self.dI = {dv: _knn.DeviceBuffer(m * k * 8, dv) for dv in uniq}
            self.dD = {dv: _knn.DeviceBuffer(m * k * 8, dv) for dv in uniq}
            self.devices = devices
            self.n_shard = _sharded.shard_bounds(n, R, 0)[1]
        else:
            lo, hi = _sharded.shard_bounds(n, R if comm is not None else 1, rank % R if comm is not None else 0)
            self.n_shard = hi - lo
            self.dXb = _knn.DeviceBuffer(X.nbytes, dev).upload(X)
            self.dY = _knn.DeviceBuffer((hi - lo) * d * 8, dev).upload(np.ascontiguousarray(Yfull[lo:hi]))
            self.dIb, self.dDb = _knn.DeviceBuffer(m * k * 8, dev), _knn.DeviceBuffer(m * k * 8, dev)
            self.index = nabo_amd.KnnIndex(hi - lo, d, metric=metric_id, dist_factor=0.25, ref_index_base=lo, device=dev)
            if comm is not None:
                comm.set_ref_shards(R)
                force = os.environ.get("NABO_BENCH_FORCE_COMM") == "1"
                self.sk = _sharded.ShardedIndex(comm, self.index, os.environ.get(
                    "NABO_BENCH_PROTOCOL", "global" if force and a.metric != "canberra" else "auto"))

    def step(self):
        a = self.a
        if self.group is not None:
            g = self.group
            g.set_ref()
            g.query_device([self.dX[dv].ptr for dv in self.devices], a.m, a.k, False,
                           [self.dI[dv].ptr for dv in self.devices], [self.dD[dv].ptr for dv in self.devices])
            self.stats.append([ix.last_stats() for ix in g.indices])
            self.xstats.append([g.last_stats(r) for r in range(self.ranks)])
        elif self.sk is not None:
            self.index.set_ref(y_device_ptr=self.dY.ptr)
            self.sk.query_device(self.dXb.ptr, a.m, a.k, False, self.dIb.ptr, self.dDb.ptr)
            self.stats.append([self.index.last_stats()])
            self.xstats.append([self.sk.last_stats()])
        else:
            self.index.set_ref(y_device_ptr=self.dY.ptr)
            self.index.query_device(self.dXb.ptr, a.m, a.k, False, self.dIb.ptr, self.dDb.ptr)
            self.stats.append([self.index.last_stats()])

    def sync(self):
        L = self._lib.lib()
        for dv in (sorted(set(self.devices)) if self.group is not None else [self.dev]):
            self._lib.check(L.nabo_dev_synchronize(dv))
        if self.comm is not None:
            self.comm.barrier()                     # RCCL all-reduce on the communicator's stream + host wait
            self._lib.check(L.nabo_dev_synchronize(self.dev))

    def run(self):
        """W untimed steps, then exactly K steps between barrier + device sync on both sides; MAX over ranks."""
        for _ in range(self.a.warmup):
            self.step()
        self.stats, self.xstats = [], []
        self.sync()
        t0 = time.perf_counter()
        for _ in range(self.a.steps):
            self.step()
        self.sync()
        dt = time.perf_counter() - t0
        if self.comm is not None:
            dt = self.comm.allreduce_max(dt)
        return dt

    def rank_stats(self):
        """mean over the timed steps, per rank: this process' ranks, or (launcher) the MAX over all ranks through RCCL"""
        keys_i = ("ms_pack", "ms_topk", "ms_refine", "ms_fallback", "ms_total", "seeded_pass_rows", "second_pass_rows", "wide_list_rows")
        keys_x = ("ms_local", "ms_topk_local", "ms_exchange", "ms_merge", "ms_second", "ms_gather", "ms_total")
        per = []
        for r in range(len(self.stats[0])):
            e = {"rank": self.rank + r, "index": {kk: float(np.mean([s[r][kk] for s in self.stats])) for kk in keys_i}}
            if self.xstats:
                e["sharded"] = {kk: float(np.mean([s[r][kk] for s in self.xstats])) for kk in keys_x}
            per.append(e)
        mx = {}
        for kk in keys_i:
            mx[kk] = max(e["index"][kk] for e in per)
        for kk in keys_x if self.xstats else ():
            mx[kk if kk != "ms_total" else "ms_sharded_total"] = max(e["sharded"][kk] for e in per)
        if self.comm is not None:              # every rank enters these collectives; rank 0 prints
            mx = {kk: self.comm.allreduce_max(v) for kk, v in sorted(mx.items())}
        return per, mx

    def result(self):
        a = self.a
        if self.group is not None:
            dv = self.devices[0]
            return self.dI[dv].download((a.m, a.k), np.int64), self.dD[dv].download((a.m, a.k), np.float64)
        return self.dIb.download((a.m, a.k), np.int64), self.dDb.download((a.m, a.k), np.float64)

    def kernel(self):
        return (self.group.indices[0] if self.group is not None else self.index).last_kernel()

    def close(self):
        if self.group is not None:
            self.group.close()
            for b in list(self.dX.values()) + list(self.dI.values()) + list(self.dD.values()):
                b.free()
        else:
            self.index.close()
            for b in (self.dXb, self.dY, self.dIb, self.dDb):
                b.free()


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
    ap.add_argument("--no-extras", action="store_true", help="skip the canberra / alt / alt_layout / C1 blocks (profiling runs)")
    a = ap.parse_args()

    # ---- how was this file launched?  (decided BEFORE anything touches a GPU) --------------------------------------
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    loop = int(os.environ.get("NABO_BENCH_LOOPBACK", "0"))     # rehearsal: N shard-ranks as threads on ONE GPU
    force_comm = os.environ.get("NABO_BENCH_FORCE_COMM") == "1"      # rehearsal: the launcher code path with one rank
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if env_world > 1:
        if a.gpus not in (1, env_world):
            print("bench.py: --gpus %d contradicts WORLD_SIZE=%d of the launcher" % (a.gpus, env_world), file=sys.stderr)
            sys.exit(2)
        kind, ranks = "launcher", env_world
    elif loop > 1:
        kind, ranks = "loopback", loop
    elif a.gpus > 1 or os.environ.get("NABO_BENCH_FORCE_THREADS") == "1":      # (rehearsal: that path with the one rank a GPU allows)
        kind, ranks = "threads", a.gpus
    else:
        kind, ranks = ("launcher" if force_comm else "single"), 1
    import nabo_amd
    from nabo_amd import _knn, _lib, _sharded
    from nabo_amd._synth import pca_like
    have = nabo_amd.device_count()
    if have < 1:
        print("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    need = ranks if kind == "threads" else (local_rank + 1 if kind == "launcher" else 1)
    if have < need:
        print("bench.py: --gpus %d needs %d visible GPUs, this process sees %d -- refusing to run (and report) a smaller job"
              % (a.gpus, need, have), file=sys.stderr)
        sys.exit(2)
    n_gpus = 1 if kind in ("single", "loopback") else ranks

    m, n, d, k = a.m, a.n, a.d, a.k
    dev = local_rank if kind == "launcher" and env_world > 1 else 0
    metric_id = {"euclidean": nabo_amd.EUCLIDEAN, "cosine": nabo_amd.COSINE, "canberra": nabo_amd.MOD_CANBERRA}[a.metric]
    Yfull = pca_like(n, d, seed=1003)
    X = pca_like(m, d, seed=2003)
    comm = _sharded.Comm.from_env(dev) if kind == "launcher" else None
    # Layout of the N ranks.  Headline: BASELINE configs[3] -- the references sharded N ways, one piece per
    # rank.  alt_layout: R = 2 pieces x N / 2 target slices (every row fills a candidate list on EVERY piece, so that
    # part of a rank's work does not shrink with the piece).  NABO_REF_SHARDS=<R> pins the headline's R (experiments).
    R = int(os.environ.get("NABO_REF_SHARDS", "0")) or ranks
    if ranks % R:
        raise SystemExit("NABO_REF_SHARDS must divide the number of ranks")
    lay = Layout(a, kind, ranks, R, comm, rank, dev, X, Yfull, metric_id)
    dt = lay.run()
    per_rank, rank_max = lay.rank_stats()
    kern = lay.kernel()
    ablate = bool(os.environ.get("NABO_DEBUG_ABLATE"))
    head = {"fallback_rows": int(max(s["fallback_rows"] for st in lay.stats for s in st)),
            "candidates": int(lay.xstats[-1][0]["candidates"]) if lay.xstats else None,
            "second_round_rows": int(max(x[0]["uncertified"] for x in lay.xstats)) if lay.xstats else None}

    # self-check outside the timed region: valid, sorted rows; a row sample against the oracle
    gi = gd = None
    sampled = None
    if not ablate:
        gi, gd = lay.result()
        assert gi.min() >= 0 and gi.max() < n and (np.diff(gd[:: max(1, m // 4096)], axis=1) >= 0).all()
        if rank == 0:
            import oracle
            rows = np.random.default_rng(12).choice(m, min(32, m), replace=False)
            oi, od = oracle.knn(X[rows], Yfull, k, metric_id, 0.25, nthreads=max(1, min(os.cpu_count() or 1, 16)))
            sampled = bool(np.array_equal(gi[rows], oi) and np.array_equal(gd[rows], od))
    if os.environ.get("NABO_BENCH_CHECK") == "1" and kind != "single":
        # rehearsal check: the sharded result must equal one unsharded index on the same data
        ref_ix = nabo_amd.KnnIndex(n, d, metric=metric_id, dist_factor=0.25, device=dev).set_ref(Yfull)
        ri, rd = ref_ix.query(X, k)
        ref_ix.close()
        same = bool(np.array_equal(gi, ri) and np.array_equal(gd, rd))
        print("rank %d sharded == unsharded: %s" % (rank, same), flush=True)
        assert same

    # the other layout of the same ranks, same invocation, same bits required (every rank takes part)
    alt_layout = None
    want_alt = (ranks >= 2 and R == ranks and a.metric != "canberra" and not a.no_extras and not ablate
                and not os.environ.get("NABO_REF_SHARDS"))
    alt_slices = None
    if want_alt:
        def other_layout(R2):
            """the same ranks laid out as R2 reference pieces x ranks / R2 target slices: timed like the headline, same bits required"""
            lay2 = Layout(a, kind, ranks, R2, comm, rank, dev, X, Yfull, metric_id)
            dt2 = lay2.run()
            per2, max2 = lay2.rank_stats()
            ai, ad = lay2.result()
            out = {"layout": {"ref_shards": R2, "target_slices": ranks // R2},
                   "workload": "refs sharded %d-way x %d target slices" % (R2, ranks // R2),
                   "ms_per_step": dt2 / a.steps * 1e3, "value": m * n * a.steps / dt2,
                   "same_bits_as_headline_layout": bool(np.array_equal(ai, gi) and np.array_equal(ad, gd)),
                   "candidates_per_shard": int(lay2.xstats[-1][0]["candidates"]),
                   "second_round_rows": int(max(x[0]["uncertified"] for x in lay2.xstats)),
                   "max_over_ranks_ms": max2, "per_rank_ms": per2}
            return lay2, out
        if ranks >= 4 and ranks % 2 == 0:
            lay.close()
            lay, alt_layout = other_layout(2)
        # ... and pure target slicing (every rank holds all the references and certifies its own slice: no exchange, one
        # all-gather) -- not BASELINE configs[3]'s layout, reported beside it because it is the cheapest one at every N
        lay.close()
        lay, alt_slices = other_layout(1)

    if rank == 0:
        shards, slices = R, ranks // R
        ms_step = dt / a.steps * 1e3
        # HIP events on the kernel's own stream (rank 0's index; for a shard the candidate query's kernel -- the index's
        # own record may belong to a second-round query)
        t_kernel = (per_rank[0]["sharded"]["ms_topk_local"] if "sharded" in per_rank[0] else per_rank[0]["index"]["ms_topk"]) * 1e-3
        workload = "%dk ref x %dk target, d=%d, k=%d, %s, refs sharded %d-way" % (n // 1000, m // 1000, d, k, a.metric, shards)
        if slices > 1:
            workload += " x %d target slices" % slices
        line = {
            "metric": "cell-pair distances/s (k-NN build, 1Mx1M d=50 k=15)" if (m, n, d, k) == (1000000, 1000000, 50, 15)
                      else "cell-pair distances/s (k-NN build)",
            "value": m * n * a.steps / dt, "unit": "cell-pair distances/s",
            "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "knn_build_s": dt / a.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if "f32_32x32x2" in kern else ("f16" if "f16" in kern else "f64"), "data": "synthetic",
            "config": {"workload": workload, "parallelism": "ref-shard%d" % shards + ("-slice%d" % slices if slices > 1 else ""),
                       "launch": {"single": "one process, one GPU", "launcher": "one process per GPU (launcher environment)",
                                  "threads": "one process, one host thread + one RCCL communicator per GPU",
                                  "loopback": "%d shard-ranks on ONE GPU, loopback transport (rehearsal)" % ranks}[kind],
                       "arithmetic": "low-precision score filter on the matrix pipe (one f16 product per pair as a rigorous lower "
                                     "bound; rows it cannot certify go on to the seeded pass / the f16x3 pass / the exact kernels), "
                                     "float64 re-evaluation + certification: indices and distances equal the reference's float64 path"},
            "phases_ms": {kk: v for kk, v in per_rank[0]["index"].items() if kk.startswith("ms_")},
            "rows_by_pass": {"seeded_one_product_pass": int(per_rank[0]["index"]["seeded_pass_rows"]),
                             "f16x3_pass": int(per_rank[0]["index"]["second_pass_rows"]),
                             "wide_lists": int(per_rank[0]["index"]["wide_list_rows"]), "exact_float64": head["fallback_rows"]},
            "fallback_rows": head["fallback_rows"],
            "sampled_rows_equal_oracle": sampled,
            "so_digest": _lib.so_digest(), "src_digest": _lib.src_digest(),
        }
        if a.metric != "canberra":
            dig = _lib.src_digest(_lib.KERNEL_SOURCES["euclid"])
            flops = 2.0 * (m / slices) * lay_n_shard(n, shards) * d                  # algorithmic (one rank): the -2XY^T term
            achieved = flops / t_kernel / 1e12
            f16 = "f16" in kern
            peak = PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS
            rec = pmc_record("traffic", workload, dig)
            line["roofline"] = {
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "peak_dtype": "f16 dense MFMA" if f16 else "f32 MFMA",
                "frac_of_f32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS, "frac_of_f16_mfma_peak": achieved / PEAK_F16_MFMA_TFLOPS,
                "traffic": (rec or {}).get("bytes_per_step"), "traffic_record": rec,
                "algorithmic_bytes": 4.0 * d * (m / slices + lay_n_shard(n, shards)) + 12.0 * k * (m / slices),
                "kernel": kern, "kernel_ms": t_kernel * 1e3, "kernel_src_digest": dig,
                "executed_over_algorithmic_flops": executed_flops_factor(kern, d),
                "matrix_pipe_busy": (rec or {}).get("matrix_pipe_busy"),
                "clock_ghz_held": (rec or {}).get("clock_ghz_held")}
        else:
            dig = _lib.src_digest(_lib.KERNEL_SOURCES["canberra"])
            line["roofline"] = canberra_roofline(pmc_record("canberra", workload, dig), t_kernel, kern, dig, float(m) * n * d)
        if kind != "single":
            hx = [s for s in per_rank if "sharded" in s]
            # (what the TRANSPORT reports, not what was asked for: ncclCommCount of this rank's communicator)
            tr = (comm.transport_ranks() if comm is not None else lay.group.comms[0].transport_ranks() if lay.group is not None else None)
            line["sharded"] = {"world": ranks if kind != "loopback" else 1, "rccl_world": tr if kind in ("launcher", "threads") else None,
                               "ranks_requested": ranks,
                               "loopback_ranks": ranks if kind == "loopback" else None,
                               "layout": {"ref_shards": shards, "target_slices": slices},
                               "candidates_per_shard": head["candidates"], "second_round_rows": head["second_round_rows"],
                               "max_over_ranks_ms": rank_max, "per_rank_ms": hx if kind != "launcher" else hx[:1]}
        if alt_layout is not None:
            line["alt_layout"] = alt_layout
        if alt_slices is not None:
            line["alt_layout_target_slices"] = alt_slices
        extras = kind == "single" and not a.no_extras and not ablate
        if extras and a.metric == "euclidean" and not os.environ.get("NABO_L2_MODE"):
            line["alt"] = alt_block(nabo_amd, _knn, "f32" if "f16" in kern else "f16x3", dev, n, m, d, k, lay.dY, lay.dXb, gi, gd, lay.sync)
            if "one-product" in kern:           # ... and the f16x3 split as the first pass (round 2's default)
                line["alt_f16x3"] = alt_block(nabo_amd, _knn, "f16x3", dev, n, m, d, k, lay.dY, lay.dXb, gi, gd, lay.sync)
        if extras and a.metric == "euclidean" and (m, n) == (1000000, 1000000):
            line["canberra"] = canberra_block(nabo_amd, _knn, _lib, dev, n, m, d, k, lay.dY, lay.dXb, X, Yfull, lay.sync)
        if extras and a.metric == "euclidean" and (m, n) == (1000000, 1000000) and os.environ.get("NABO_BENCH_CONFIG4", "1") != "0":
            line["config4_one_gpu"] = config4_block()
        if not a.no_cpu_baseline and kind == "single":
            line["cpu_baseline"] = cpu_baseline(d, k, metric_id)
            if extras:
                line["cpu_baseline"]["c1_end_to_end"] = c1_end_to_end(gpu=True)
        print(json.dumps(line), flush=True)
    lay.close()
    if comm is not None:
        comm.barrier()
        comm.close()


def config4_block():
    """BASELINE configs[4] run WHOLE on this one GPU (tools/config4_one_gpu.py: 5M x 5M, d=100, k=50, cosine -- unsharded, then
    through nabo_sharded_query with 8 loopback shard-ranks, all rows compared -- the SNN graph and the 1000-permutation null
    of the mapping scores on its ~190M edges).  Both are EXTENSIONS (no reference output exists for either): the times are
    reported, parity is this build's own oracle's (tests/test_configs_gpu.py).  ~40 s; NABO_BENCH_CONFIG4=0 skips it."""
    import importlib.util
    try:
        spec = importlib.util.spec_from_file_location("config4_one_gpu", os.path.join(REPO, "tools", "config4_one_gpu.py"))
        c4 = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(c4)
        rep, _ = c4.run(5000000, 5000000, 100, 50, ranks=8, perms=1000, keep=False)
        return rep
    except Exception as e:                      # (never costs the headline line)
        return {"error": "%s: %s" % (type(e).__name__, e)}


def lay_n_shard(n, shards):
    from nabo_amd._sharded import shard_bounds
    return shard_bounds(n, shards, 0)[1]


def canberra_roofline(rec, t_kernel, kern, dig=None, pair_dims=None):
    """The mod-Canberra filter is vector-ALU work (no contraction): the bound is the chip's vector ISSUE rate, and what
    is priced against it is the number of vector instructions the kernel ACTUALLY issued (SQ_INSTS_VALU of a
    rocprofv3 --pmc pass on this kernel's sources), so the fraction cannot exceed 1.  `frac` is against the issue peak
    measured for the counting pass's own instruction mix, `frac_of_nominal_issue_peak` against one instruction per
    4 clocks and SIMD."""
    insts = (rec or {}).get("valu_insts_per_step")
    ach = insts / t_kernel / 1e9 if insts else None
    peak = PEAK_VALU_GINST_BITS if "cbb_filter" in kern else PEAK_VALU_GINST
    return {"bound": "valu-issue", "achieved": ach, "peak": peak, "unit": "G wave-instructions/s",
            "frac": ach / peak if ach else None,
            "peak_note": "issue rate measured for this pass's instruction kinds (tools/issue_lab.hip, profiles/r2_issue_lab.txt)",
            "nominal_peak": NOMINAL_VALU_GINST, "frac_of_nominal_issue_peak": ach / NOMINAL_VALU_GINST if ach else None,
            "valu_insts_per_step": insts, "pmc_record": rec, "kernel": kern, "kernel_ms": t_kernel * 1e3,
            "kernel_src_digest": dig,
            # (the issue fraction does not move when a change removes instructions and time alike -- rounds 3 and 4 both read
            # 0.35 while the pass went 786 -> 504 ms: the work rate says what the kernel got faster by.  1M x 1M x d pair-
            # dimensions per step; lane-instructions = 64 per vector wave-instruction.)
            "pair_dimensions_per_s": pair_dims / t_kernel if pair_dims else None,
            "vector_lane_instructions_per_pair_dimension": insts * 64.0 / pair_dims if insts and pair_dims else None}


def canberra_block(nabo_amd, _knn, _lib, dev, n, m, d, k, dY, dX, X, Yfull, sync):
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
    dig = _lib.src_digest(_lib.KERNEL_SOURCES["canberra"])
    return {"workload": "1M ref x 1M target, d=50, k=15, modified Canberra (nabo/_mapping.py:29-45), dist_factor 0.25",
            "ms_per_step": best * 1e3, "value": m * n / best, "unit": "cell-pair distances/s",
            "phases_ms": {key: st[key] for key in ("ms_pack", "ms_topk", "ms_refine", "ms_fallback", "ms_total")},
            "fallback_rows": st["fallback_rows"], "sampled_rows_equal_oracle": same,
            "roofline": canberra_roofline(pmc_record("canberra", workload, dig), st["ms_topk"] * 1e-3, kern, dig, float(m) * n * d)}


def executed_flops_factor(kern, d):
    """MFMA flops the filter kernel executes per algorithmic flop (2 m n d): K slots per pair / d."""
    import re
    mt = re.match(r"l2c_topk_kernel<(\d+),", kern)
    if mt:
        return 32.0 * int(mt.group(1)) / d                 # one-product operands: g + 3 slots in steps of 32
    mt = re.match(r"l2[qhs]_topk_kernel<(\d+)", kern)
    if mt:
        return 16.0 * int(mt.group(1)) / d                 # K-concatenated f16 operands in steps of 16 (f16x3: 3 (g + 1) slots)
    mt = re.match(r"l2_topk_kernel<(\d+),", kern)
    if mt:
        return 2.0 * int(mt.group(1)) / d                  # fp32: k-steps of 2
    return None


def alt_block(nabo_amd, _knn, other, dev, n, m, d, k, dY, dX, gi, gd, sync):
    """The same step with ANOTHER Euclidean filter (fp32 MFMA, or the f16x3 split as the first pass); results must be the same bits."""
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
