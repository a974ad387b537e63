# round 2, GPU call S: parity suite + PMC summary of the f16x3 kernel after a kernel change
set -e
TAG=${TAG:-r2s}
O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B="--no-extras --no-cpu-baseline"
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
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
print("%-10s kernel_ms %.1f  cycles/SIMD %.3e  clock %.3f GHz  matrix pipe busy %.1f %%  VALU insts (incl. MFMA) %.3e  fallback %s"
      % ("$name", ms, cyc, cyc / ms / 1e6, 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), c["SQ_INSTS_VALU"], d.get("fallback_rows")))
PY
}
pm full
pm nohit NABO_DEBUG_ABLATE=1
pm l2window NABO_DEBUG_ABLATE=3
for i in 1 2; do python bench.py $B --steps 5 --warmup 2 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench kernel_ms', round(d['roofline']['kernel_ms'],1), 'ms_per_step', round(d['ms_per_step'],1), 'fallback', d.get('fallback_rows'))" >> $O/ab.txt; done
python bench.py $B --targets 100000 --refs 100000 --steps 10 --warmup 2 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('100k kernel_ms', round(d['roofline']['kernel_ms'],2), 'ms_per_step', round(d['ms_per_step'],2))" >> $O/ab.txt
cat $O/ab.txt
