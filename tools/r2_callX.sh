set -e
TAG=${TAG:-r2x}
O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
B="--no-extras --no-cpu-baseline"
NABO_L2H_R=2 NABO_L2Q_NB=4 timeout -k 10 600 python -m pytest tests/test_knn_gpu.py -x -q -m gpu -k "both_filter or tail_round or nan or no_cliff or duplicates" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() { n=$1; shift; ( export "$@" _X=1; timeout -k 10 120 python bench.py $B --steps 4 --warmup 2 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'ms_per_step', round(d['ms_per_step'],1), 'fallback', d.get('fallback_rows'), d['roofline']['kernel'][:28])" | tee -a $O/ab.txt ); }
run l2h_r2 NABO_L2H_R=2
run l2h_r2_nohit NABO_L2H_R=2 NABO_DEBUG_ABLATE=1
run l2q_nb4 NABO_L2_MODE=f16x3q NABO_L2Q_NB=4
run l2q_nb4_nohit NABO_L2_MODE=f16x3q NABO_L2Q_NB=4 NABO_DEBUG_ABLATE=1
run l2h
run l2q NABO_L2_MODE=f16x3q
