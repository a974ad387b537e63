# round 2, GPU call R: matrix-pipe busy fraction and held clock (GRBM_GUI_ACTIVE) of the f16x3 kernel and of its
# compile-time ablations; MFMA clock lab; issue lab with the integer (SWAR) instruction kinds.
set -e
TAG=${TAG:-r2r}
O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B="--no-extras --no-cpu-baseline"
tools/mfma_clock_lab.bin > $O/mfma_clock_lab.txt 2>&1 || true
tools/issue_lab.bin > $O/issue_lab.txt 2>&1 || true
echo labs done
pm() {   # name, env assignments...
  name=$1; shift
  env "$@" true
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
print("%-10s kernel_ms %.1f  cycles/SIMD %.3e  clock %.3f GHz  matrix pipe busy %.1f %%  VALU insts (incl. MFMA) %.3e  per MFMA %.2f"
      % ("$name", ms, cyc, cyc / ms / 1e6, 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), c["SQ_INSTS_VALU"], c["SQ_INSTS_VALU"] / 9.77e9))
PY
  echo "$name done"
}
pm full
pm nohit NABO_DEBUG_ABLATE=1
pm nofilter NABO_DEBUG_ABLATE=1 NABO_KNN_SO=$PWD/tools/ab/nofilter.so
pm noreload NABO_DEBUG_ABLATE=1 NABO_KNN_SO=$PWD/tools/ab/noreload.so
pm bare NABO_DEBUG_ABLATE=1 NABO_KNN_SO=$PWD/tools/ab/bare.so
pm l2window NABO_DEBUG_ABLATE=3
cat $O/ab.txt $O/mfma_clock_lab.txt
tail -16 $O/issue_lab.txt
