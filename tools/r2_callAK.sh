set -e
O=gpurun_out/r2ak; rm -rf $O; mkdir -p $O
B="--no-extras --no-cpu-baseline"
timeout -k 10 600 python -m pytest tests/test_knn_gpu.py tests/test_sharded.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() { n=$1; shift; ( export "$@" _X=1; timeout -k 10 120 python bench.py $B --steps 4 --warmup 2 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'ms_per_step', round(d['ms_per_step'],1), 'fallback', d.get('fallback_rows'), d['roofline']['kernel'][:28])" | tee -a $O/ab.txt ); }
run default
run default2
python tools/shard_phases.py 8
python tools/shard_phases.py 2
