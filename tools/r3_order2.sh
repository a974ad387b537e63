# order on/off + home pre-pass length variants, same box
set -e
TAG=${1:-r3e}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
B="--no-extras --no-cpu-baseline"
run() {   # label, env...
  local label=$1; shift
  env "$@" python bench.py $B --steps 5 --warmup 2 > $O/bench_$label.json 2> $O/bench_$label.err || echo "bench $label FAILED"
  python -c "
import json
d=json.loads(open('$O/bench_$label.json').read().strip().splitlines()[-1])
print('$label: ms_per_step %.2f kernel_ms %.2f pack %.2f refine %.2f fallback %d oracle_rows %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['phases_ms']['ms_pack'], d['phases_ms']['ms_refine'], d['fallback_rows'], d['sampled_rows_equal_oracle']))" | tee -a $O/summary.txt
  env "$@" python tools/bench_shard.py 8 2>> $O/bench_$label.err | tail -1 | sed "s/^/$label shard: /" | cut -c1-200 | tee -a $O/summary.txt
  env "$@" python bench.py $B --targets 100000 --refs 100000 --steps 10 --warmup 2 2>> $O/bench_$label.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label 100k x 100k: ms_per_step %.3f kernel_ms %.3f pack %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['phases_ms']['ms_pack']))" | tee -a $O/summary.txt
}
run order0 NABO_L2Q_ORDER=0
run home8 NABO_L2Q_ORDER=1
run home4 NABO_KNN_SO=$PWD/tools/ab/home4.so
run home16 NABO_KNN_SO=$PWD/tools/ab/home16.so
run order0_again NABO_L2Q_ORDER=0
for ord in 1 0; do
  NABO_L2Q_ORDER=$ord NABO_KNN_SO=$PWD/tools/ab/prof.so python bench.py $B --steps 1 --warmup 0 > /dev/null 2> $O/prof_order$ord.txt || true
  grep "lists prof" $O/prof_order$ord.txt | head -1 | sed "s/^/order=$ord 1M main launch: /" | tee -a $O/summary.txt
done
python -m pytest tests/test_knn_gpu.py tests/test_sharded.py -q -m gpu > $O/pytest.log 2>&1 && echo "parity (order on): $(tail -1 $O/pytest.log)" | tee -a $O/summary.txt || { echo "PARITY FAILED (order on)" | tee -a $O/summary.txt; tail -30 $O/pytest.log; }
