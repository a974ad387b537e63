set -e
O=$PWD/gpurun_out/${TAG:-r2p}; mkdir -p $O
B="--no-extras --no-cpu-baseline"
run() { timeout -k 5 200 python bench.py $B --steps 3 --warmup 1 $2 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'ms_per_step', round(d['ms_per_step'],1), 'refine', round(d['phases_ms']['ms_refine'],1), 'fallback_ms', round(d['phases_ms']['ms_fallback'],2), d['roofline']['kernel'][:24], 'fallback', d['fallback_rows'])" | tee -a $O/ab.txt; }
run l2h
NABO_DEBUG_ABLATE=1 run l2h_nohit
for lk in 17 19 21; do NABO_LKEEP=$lk run l2h_lkeep$lk; done
NABO_L2_MODE=f16x3s NABO_LKEEP=19 run l2s_lkeep19
run l2h_cosine_d50 "--metric cosine"
run l2h_k10 "--neighbors 10"
run l2h_d30 "--dims 30"
