set -e
O=$PWD/gpurun_out/${TAG:-r2g}; mkdir -p $O
B="--no-extras --no-cpu-baseline"
run() { timeout -k 5 200 python bench.py $B --steps 3 --warmup 1 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'ms_per_step', round(d['ms_per_step'],1), d['roofline']['kernel'][:24], 'fallback', d['fallback_rows'])" | tee -a $O/ab.txt; }
for v in "$@"; do
  export NABO_KNN_SO=$PWD/tools/ab/$v.so
  NABO_L2_MODE=f16x3h timeout -k 5 120 python tools/r2_smoke.py > $O/smoke_$v.txt 2>&1 || { tail -5 $O/smoke_$v.txt; exit 1; }
  NABO_L2_MODE=f16x3h run ${v}_l2h
  NABO_L2_MODE=f16x3s run ${v}_l2s
  NABO_L2_MODE=f32 run ${v}_f32
done
unset NABO_KNN_SO
