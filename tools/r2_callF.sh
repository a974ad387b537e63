set -e
O=$PWD/gpurun_out/${TAG:-r2f}; mkdir -p $O
B="--no-extras --no-cpu-baseline"
NABO_L2_MODE=f16x3s timeout -k 5 120 python tools/r2_smoke.py 2>&1 | tee $O/smoke_s.txt
NABO_L2_MODE=f16x3h timeout -k 5 120 python tools/r2_smoke.py 2>&1 | tee $O/smoke_h.txt
NABO_L2_MODE=f32 timeout -k 5 120 python tools/r2_smoke.py 2>&1 | tee $O/smoke_f.txt
run() { timeout -k 5 200 python bench.py $B --steps 3 --warmup 1 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'ms_per_step', round(d['ms_per_step'],1), d['roofline']['kernel'][:24], 'fallback', d['fallback_rows'])" | tee -a $O/ab.txt; }
NABO_L2_MODE=f16x3h run l2h
NABO_L2_MODE=f16x3h NABO_DEBUG_ABLATE=1 run l2h_nohit
NABO_L2_MODE=f16x3s run l2s_sync2
NABO_L2_MODE=f16x3s NABO_DEBUG_ABLATE=1 run l2s_sync2_nohit
NABO_L2_MODE=f16x3s NABO_L2S_SYNC=1 run l2s_sync1
NABO_L2_MODE=f32 run f32
NABO_L2_MODE=f32 NABO_DEBUG_ABLATE=1 run f32_nohit
( time timeout -k 10 600 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python bench.py $B --metric canberra --steps 2 --warmup 1 > $O/bench_canberra.json 2>> $O/err.txt
python -c "import json; d=json.loads(open('$O/bench_canberra.json').read().strip().splitlines()[-1]); print('canberra ms_per_step', d['ms_per_step'], d['phases_ms'], d['fallback_rows'])"
