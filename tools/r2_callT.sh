set -e
TAG=${TAG:-r2t}
O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
B="--no-extras --no-cpu-baseline"
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() { n=$1; shift; ( export "$@" _X=1; python bench.py $B --steps 4 --warmup 2 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'ms_per_step', round(d['ms_per_step'],1), 'fallback', d.get('fallback_rows'), d['roofline']['kernel'][:28])" | tee -a $O/ab.txt ); }
run default
run nohit NABO_DEBUG_ABLATE=1
run f32 NABO_L2_MODE=f32
run l2s NABO_L2_MODE=f16x3s
${EXTRA:-true}
