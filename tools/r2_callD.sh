set -e
O=$PWD/gpurun_out/${TAG:-r2d}; mkdir -p $O
B="--no-extras --no-cpu-baseline"
timeout -k 5 120 python tools/r2_smoke.py 2>&1 | tee $O/smoke.txt
run() { python bench.py $B --steps 3 --warmup 1 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'ms_per_step', round(d['ms_per_step'],1), d['roofline']['kernel'][:24], 'fallback', d['fallback_rows'])" | tee -a $O/ab.txt; }
NABO_L2S_SYNC=1 run sync1
NABO_L2S_SYNC=0 run sync0
NABO_L2S_SYNC=1 NABO_DEBUG_ABLATE=1 run sync1_nohit
NABO_L2S_SYNC=0 NABO_DEBUG_ABLATE=1 run sync0_nohit
NABO_L2_MODE=f16x3h run l2h
( time timeout -k 10 600 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
