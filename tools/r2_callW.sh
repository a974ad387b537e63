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
print("%-12s kernel_ms %.1f  cycles/SIMD %.3e  clock %.3f GHz  matrix pipe busy %.1f %%  VALU insts (incl. MFMA) %.3e  fallback %s"
      % ("$name", ms, cyc, cyc / ms / 1e6, 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), c["SQ_INSTS_VALU"], d.get("fallback_rows")))
PY
}
pm l2q NABO_L2_MODE=f16x3q
pm l2q_nohit NABO_L2_MODE=f16x3q NABO_DEBUG_ABLATE=1
pm l2h
pm l2h_nohit NABO_DEBUG_ABLATE=1
cat $O/ab.txt
NABO_KNN_SO=$PWD/tools/ab/prof.so NABO_L2_MODE=f16x3q python bench.py $B --steps 1 --warmup 0 2>&1 >/dev/null | grep "lists prof" | tail -1
NABO_KNN_SO=$PWD/tools/ab/prof.so python bench.py $B --steps 1 --warmup 0 2>&1 >/dev/null | grep "lists prof" | tail -1
pm l2q_l1win NABO_L2_MODE=f16x3q NABO_DEBUG_ABLATE=5
pm l2q_l2win NABO_L2_MODE=f16x3q NABO_DEBUG_ABLATE=3
tail -2 $O/ab.txt
