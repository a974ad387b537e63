# round 2: cycles, held clock and matrix-pipe occupancy of the f16x3 kernels on the current build, one --pmc pass each
set -e
TAG=${TAG:-r2w}
O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B="--no-extras --no-cpu-baseline"
pm() {
  name=$1; shift
  ( export "$@" _X=1; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc_$name/pass1 -- python3 bench.py $B --steps 1 --warmup 0 > $O/pmc_$name.json 2> $O/pmc_$name.err ) || echo "$name failed"
  python tools/pmc_summary.py $O/pmc_$name > $O/pmc_${name}_summary.csv
  python - <<PY >> $O/ab.txt
import csv, json
d = json.loads(open("$O/pmc_$name.json").read().strip().splitlines()[-1])
c = {}
for r in csv.DictReader(open("$O/pmc_${name}_summary.csv")):
    if "topk_kernel" in r["kernel"]: c[r["counter"]] = c.get(r["counter"], 0) + float(r["sum"])
ms = d["roofline"]["kernel_ms"]
cyc = c["GRBM_GUI_ACTIVE"] / 8
print("%-24s %-16s kernel_ms %.1f  cycles/SIMD %.3e  clock held %.3f GHz  matrix pipe busy %.1f %%  VALU insts (incl. MFMA) %.3e  fallback rows %s  digest %s"
      % ("$name", d["roofline"]["kernel"].split("<")[0], ms, cyc, cyc / ms / 1e6, 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), c["SQ_INSTS_VALU"], d.get("fallback_rows"), d.get("so_digest")))
PY
}
pm default_16x16x32
pm default_16x16x32_nohit NABO_DEBUG_ABLATE=1
pm f16x3h_32x32x16 NABO_L2_MODE=f16x3h
pm f16x3h_32x32x16_nohit NABO_L2_MODE=f16x3h NABO_DEBUG_ABLATE=1
pm f16x3s_shared_tiles NABO_L2_MODE=f16x3s
pm f32 NABO_L2_MODE=f32
cat $O/ab.txt
