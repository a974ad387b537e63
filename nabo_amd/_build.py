"""Builds libnabo_knn.so (HIP, gfx950) in-tree with hipcc.  No torch, no cmake.

    python -m nabo_amd._build [--force] [--verbose]
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libnabo_knn.so")
SOURCES = ["api.hip", "pack.hip", "l2_topk.hip", "l2q_topk.hip", "l2c_topk.hip", "refine.hip", "canberra.hip", "canberra_f32.hip", "canberra_bits.hip", "score_null.hip", "csr_build.hip", "sharded.hip", "multi.hip", "host_graph.hip"]
# kernels of the experiments build only (-DNABO_EXPERIMENTS, tools/ab: measured slower than the product's and kept for A/B runs):
# the f16x3 split on the 32x32x16 MFMA shape per wave / with LDS-shared tiles, locality-ordered streaming
EXPERIMENT_SOURCES = ["l2h_topk.hip", "l2s_topk.hip", "order.hip"]
# per-file extra flags: -fno-honor-nans for the fp32 score kernel (scores are finite or +inf by construction; without it
# every fminf tree starts with two v_max canonicalisations, and on gfx950 the fp32 MFMA cannot overlap vector-ALU work);
# (NOT for l2h_topk.hip: its masked / padding cells carry an inf - inf = NaN low part, and the filter relies on NaN
# comparing false -- with the flag 4 of 1000 rows lose their certificate);
# canberra_f32.hip with LLVM's iterative-ilp scheduler: the counting loop is four independent packed-f16 chains per
# dimension pair, and the default scheduler leaves 156 hazard s_nop in it (none with this one): kernel 45.4 -> 40.8 ms at
# 100k x 100k, 3.79 -> 3.43 s at 1M x 1M;
# the same scheduler for l2_topk.hip (hit path and filter are plain vector code around the fenced MFMA chains): 1M x 1M
# kernel 755 -> 735 ms; no effect on l2h_topk.hip (372 ms either way);
# keep MFMA accumulators in arch VGPRs so the C-in (||y||^2 block) needs no
# v_accvgpr_write and the filter reads the scores without v_accvgpr_read (see l2_topk.hip)
FILE_FLAGS = {"l2_topk.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"], "l2h_topk.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "l2q_topk.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
              # l2c_topk.hip: its scores are finite or +inf by construction (pack_ctiles_kernel<.,.,1>), so the filter's minimum tree
              # needs no NaN canonicalisation (two v_max per row-block otherwise)
              "l2c_topk.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"], "l2s_topk.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
              "canberra_f32.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]}
HEADERS = [os.path.join(CSRC, "knn_common.h"), os.path.join(CSRC, "topk_lists.h"), os.path.join(HERE, "..", "include", "nabo_knn.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-inline-asm"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libnabo_knn.so cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(args):
    src, obj, verbose, extra = args
    cmd = [hipcc()] + FLAGS + extra + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    if r.returncode != 0 or verbose:
        sys.stderr.write(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s" % src)
    return obj


def build(force=False, verbose=False, extra=None, out=None):
    """out: build a VARIANT of the library (extra compiler flags) into another file, with its own object directory --
    tools/ab/*.sh load it through NABO_KNN_SO; the product's libnabo_knn.so is never touched by experiments."""
    extra = list(extra or [])
    global SO
    so_saved = SO
    objdir = os.path.join(CSRC, "build")
    if out:
        SO = os.path.abspath(out)
        objdir = os.path.join(CSRC, "build_" + os.path.splitext(os.path.basename(out))[0])
    try:
        return _build(force, verbose, extra, objdir)
    finally:
        SO = so_saved


def _build(force, verbose, extra, objdir):
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    for s in SOURCES + (EXPERIMENT_SOURCES if "-DNABO_EXPERIMENTS" in extra else []):
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS + [os.path.abspath(__file__)]):
            jobs.append((src, obj, verbose, extra + FILE_FLAGS.get(s, [])))
    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(jobs), 6)) as ex:
            list(ex.map(_compile, jobs))
    if jobs or force or _stale(SO, objs):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", SO] + objs + ["-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    # python -m nabo_amd._build [--force] [--verbose] [--out tools/ab/x.so -DFLAG ... [-- more hipcc flags]]
    args = sys.argv[1:]
    out = args[args.index("--out") + 1] if "--out" in args else None
    extra = [a for a in args if a.startswith("-D")] + (args[args.index("--") + 1:] if "--" in args else [])
    print(build(force="--force" in args, verbose="--verbose" in args, extra=extra, out=out))
