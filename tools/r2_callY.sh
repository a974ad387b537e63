set -e
TAG=${TAG:-r2y}
O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
B="--no-extras --no-cpu-baseline"
run() { n=$1; shift; ( export "$@" _X=1; timeout -k 10 120 python bench.py $B --steps 4 --warmup 2 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'ms_per_step', round(d['ms_per_step'],1), 'fallback', d.get('fallback_rows'), d['roofline']['kernel'][:28])" | tee -a $O/ab.txt ); }
export NABO_L2_MODE=f16x3q
run l2q
run l2q_nohit NABO_DEBUG_ABLATE=1
run qseq NABO_KNN_SO=$PWD/tools/ab/qseq.so
run qseq_nohit NABO_KNN_SO=$PWD/tools/ab/qseq.so NABO_DEBUG_ABLATE=1
run qlate NABO_KNN_SO=$PWD/tools/ab/qlate.so
run qlate_nohit NABO_KNN_SO=$PWD/tools/ab/qlate.so NABO_DEBUG_ABLATE=1
run l2q_again
