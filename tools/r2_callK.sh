set -e
O=$PWD/gpurun_out/${TAG:-r2k}; mkdir -p $O
B="--no-extras --no-cpu-baseline"
run() { timeout -k 5 200 python bench.py $B --steps 3 --warmup 1 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'ms_per_step', round(d['ms_per_step'],1), d['roofline']['kernel'][:24])" | tee -a $O/ab.txt; }
export NABO_L2_MODE=f16x3h NABO_DEBUG_ABLATE=1
for v in nofilter noreload bare ilp; do NABO_KNN_SO=$PWD/tools/ab/$v.so run ${v}_nohit; done
unset NABO_DEBUG_ABLATE
NABO_KNN_SO=$PWD/tools/ab/ilp.so run ilp_full
