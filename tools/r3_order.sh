# round 3: locality-ordered streaming (order.hip) on / off -- NABO_L2Q_ORDER is a run-time switch of the product library.
# 1M x 1M step, one shard of eight, list-update counts from a -DNABO_LISTS_PROF build (tools/ab/prof.so), parity suites.
set -e
TAG=${1:-r3d}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
B="--no-extras --no-cpu-baseline"
for ord in 1 0; do
  export NABO_L2Q_ORDER=$ord
  python bench.py $B --steps 5 --warmup 2 > $O/bench_order$ord.json 2> $O/bench_order$ord.err || echo "bench order=$ord FAILED"
  python -c "
import json
d=json.loads(open('$O/bench_order$ord.json').read().strip().splitlines()[-1])
print('order=$ord: ms_per_step %.2f kernel_ms %.2f pack %.2f refine %.2f fallback %d oracle_rows %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['phases_ms']['ms_pack'], d['phases_ms']['ms_refine'], d['fallback_rows'], d['sampled_rows_equal_oracle']))" | tee -a $O/summary.txt
  python tools/bench_shard.py 8 2>> $O/bench_order$ord.err | tail -1 | sed "s/^/order=$ord shard: /" | tee -a $O/summary.txt
  python bench.py $B --targets 100000 --refs 100000 --steps 10 --warmup 2 2>> $O/bench_order$ord.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order=$ord 100k x 100k: ms_per_step %.3f kernel_ms %.3f pack %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['phases_ms']['ms_pack']))" | tee -a $O/summary.txt
  if [ -f tools/ab/prof.so ]; then
    NABO_KNN_SO=$PWD/tools/ab/prof.so python bench.py $B --steps 1 --warmup 0 > /dev/null 2> $O/prof_order$ord.txt || true
    grep "lists prof" $O/prof_order$ord.txt | tail -2 | sed "s/^/order=$ord 1M: /" | tee -a $O/summary.txt
    NABO_KNN_SO=$PWD/tools/ab/prof.so python tools/bench_shard.py 8 > /dev/null 2> $O/prof_shard_order$ord.txt || true
    grep "lists prof" $O/prof_shard_order$ord.txt | tail -1 | sed "s/^/order=$ord shard (cumulative over 6 queries): /" | tee -a $O/summary.txt
  fi
done
unset NABO_L2Q_ORDER
python -m pytest tests/test_knn_gpu.py tests/test_sharded.py tests/test_configs_gpu.py -q -m gpu > $O/pytest.log 2>&1 && echo "parity (order on): $(tail -1 $O/pytest.log)" | tee -a $O/summary.txt || { echo "PARITY FAILED (order on)" | tee -a $O/summary.txt; tail -30 $O/pytest.log; }
cat $O/summary.txt
