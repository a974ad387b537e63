"""Turn a tools/r3_pmc.sh (round 2: tools/r2_callB.sh) output directory into profiles/pmc.json, the record bench.py's roofline reads:
HBM-side bytes and matrix-pipe occupancy of the Euclidean filter kernel, vector instructions of the mod-Canberra
filter, each keyed by workload and by a digest of the kernel's SOURCES (nabo_amd/_lib.py: src_digest(KERNEL_SOURCES[...]) --
reproducible after a rebuild, unlike the bytes of the .so).
    python tools/pmc_to_json.py gpurun_out/<tag> profiles/<tag>_   (copies the summaries next to it)"""
import csv
import json
import os
import shutil
import sys

src, dst_prefix = sys.argv[1], sys.argv[2]


def summary(fn):
    out = {}
    for r in csv.DictReader(open(fn)):
        out.setdefault(r["kernel"], {})[r["counter"]] = float(r["sum"])
    return out


def bench_line(fn):
    return json.loads(open(fn).read().strip().splitlines()[-1])


rec = {"traffic": {}, "canberra": {}}
be = bench_line(os.path.join(src, "pmc_euclid_pass1.json"))
eu = summary(os.path.join(src, "pmc_euclid_summary.csv"))
kern = [k for k in eu if "topk_kernel" in k and "l2" in k]
k0 = max(kern, key=lambda k: eu[k].get("SQ_INSTS_MFMA", 0))
c = eu[k0]
gui = c["GRBM_GUI_ACTIVE"]
rec["traffic"][be["config"]["workload"]] = {
    "so_digest": be["so_digest"], "src_digest": be["roofline"]["kernel_src_digest"], "kernel": k0,
    "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"],
    # FETCH_SIZE tallies 128-byte requests at 64 bytes for wide streaming reads on gfx950 (MI355X_MICROARCH.md, HBM): x2
    "bytes_per_step": 2.0 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024,
    "matrix_pipe_busy": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 1024.0),
    # GRBM_GUI_ACTIVE sums the 8 XCDs: cycles the kernel took, and the clock the part held under it (DVFS)
    "cycles_per_simd": gui / 8.0, "kernel_ms_in_that_pass": be["roofline"]["kernel_ms"],
    "clock_ghz_held": gui / 8.0 / (be["roofline"]["kernel_ms"] * 1e6),
    "insts": {k: c[k] for k in c if k.startswith("SQ_INSTS")},
    "wave_cycles": {k: c[k] for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU") if k in c},
    "source": dst_prefix + "pmc_euclid_summary.csv (rocprofv3 --pmc, one counter group per pass, tools/r3_pmc.sh)"}
bc = bench_line(os.path.join(src, "pmc_canberra_pass1.json"))
ca = summary(os.path.join(src, "pmc_canberra_summary.csv"))
kc = max([k for k in ca if "cbf_filter_kernel" in k or "cbb_filter_kernel" in k], key=lambda k: ca[k].get("SQ_INSTS_VALU", 0))
rec["canberra"][bc["config"]["workload"]] = {
    "so_digest": bc["so_digest"], "src_digest": bc["roofline"]["kernel_src_digest"], "kernel": kc, "valu_insts_per_step": ca[kc]["SQ_INSTS_VALU"],
    "insts": {k: ca[kc][k] for k in ca[kc] if k.startswith("SQ_INSTS")},
    "wave_cycles": {k: ca[kc][k] for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU") if k in ca[kc]},
    "source": dst_prefix + "pmc_canberra_summary.csv"}
for f in ("pmc_euclid_summary.csv", "pmc_canberra_summary.csv", "issue_lab.txt", "ablate.txt", "shard_fullscale.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), dst_prefix + f)
json.dump(rec, open(os.path.join(os.path.dirname(dst_prefix) or ".", "pmc.json"), "w"), indent=1)
print(json.dumps(rec, indent=1)[:1500])
