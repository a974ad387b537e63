set -e
O=$PWD/gpurun_out/${TAG:-r2l}; mkdir -p $O
tools/xor_lane_test.bin | tail -9 | tee $O/xor_lane_test.txt
B="--no-extras --no-cpu-baseline"
for md in f16x3h f16x3s f32; do NABO_L2_MODE=$md timeout -k 5 120 python tools/r2_smoke.py > $O/smoke_$md.txt 2>&1 || { tail -8 $O/smoke_$md.txt; exit 1; }; done
grep -c "same=True" $O/smoke_*.txt
run() { timeout -k 5 200 python bench.py $B --steps 3 --warmup 1 $2 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'ms_per_step', round(d['ms_per_step'],1), d['roofline']['kernel'][:24], 'fallback', d['fallback_rows'])" | tee -a $O/ab.txt; }
NABO_L2_MODE=f16x3h run l2h
NABO_L2_MODE=f16x3s run l2s
NABO_L2_MODE=f32 run f32
NABO_L2_MODE=f16x3h run l2h_100k "--targets 100000 --refs 100000"
NABO_KNN_SO=$PWD/tools/ab/prof.so NABO_L2_MODE=f16x3h timeout -k 5 200 python bench.py $B --steps 1 --warmup 0 2>&1 >/dev/null | tail -1 | tee $O/prof.txt
( time timeout -k 10 600 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for n in 8 4 2; do python tools/check_shard_fullscale.py $n | tail -1 | cut -c1-260; done | tee $O/shard_fullscale.txt
python bench.py $B --metric canberra --steps 2 --warmup 1 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('canberra ms_per_step', d['ms_per_step'], d['phases_ms'], d['fallback_rows'])"
