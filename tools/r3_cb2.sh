set -e
O=$PWD/gpurun_out/${1:-r3k}; mkdir -p $O
python -m pytest tests/test_knn_gpu.py -q -m gpu -k "canberra" > $O/pytest_cb.log 2>&1 && echo "canberra parity: $(tail -1 $O/pytest_cb.log)" || { echo "CANBERRA PARITY FAILED"; tail -40 $O/pytest_cb.log; }
run() {
  local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --metric canberra --no-extras --no-cpu-baseline --steps 2 --warmup 1 2> $O/err_$label.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label 1M: ms_per_step %.1f kernel_ms %.1f pack %.2f refine %.2f fallback %d oracle_rows %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['phases_ms']['ms_pack'], d['phases_ms']['ms_refine'], d['fallback_rows'], d['sampled_rows_equal_oracle']))" || { echo "$label bench failed"; tail -5 $O/err_$label.txt; }
  env "$@" timeout -k 10 300 python bench.py --metric canberra --targets 100000 --refs 100000 --no-extras --no-cpu-baseline --steps 5 --warmup 1 2>> $O/err_$label.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label 100k: ms_per_step %.2f kernel_ms %.2f pack %.2f refine %.2f fallback %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['phases_ms']['ms_pack'], d['phases_ms']['ms_refine'], d['fallback_rows']))" || echo "$label 100k failed"
}
run bits64 NABO_CANBERRA_MODE=bits
run bits32 NABO_CANBERRA_MODE=bits NABO_KNN_SO=$PWD/tools/ab/cbb32.so
run swar NABO_CANBERRA_MODE=swar
