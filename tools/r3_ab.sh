# round 3 A/B harness: the 1M x 1M step and one shard of eight on every library variant given (names of tools/ab/<v>.so;
# "product" = nabo_amd/libnabo_knn.so), same box, plus a parity run of the Euclidean GPU tests on each variant.
#   bash tools/r3_ab.sh TAG product ring3 spread2
set -e
TAG=$1; shift
O=$PWD/gpurun_out/$TAG; mkdir -p $O
B="--no-extras --no-cpu-baseline"
for v in "$@"; do
  if [ "$v" = product ]; then unset NABO_KNN_SO; else export NABO_KNN_SO=$PWD/tools/ab/$v.so; fi
  python bench.py $B --steps 5 --warmup 2 > $O/bench_$v.json 2> $O/bench_$v.err || echo "bench $v FAILED"
  python -c "
import json,sys
d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1])
print('$v: ms_per_step %.2f kernel_ms %.2f refine %.2f oracle_rows %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['phases_ms']['ms_refine'], d['sampled_rows_equal_oracle']))" | tee -a $O/summary.txt
  python tools/bench_shard.py 8 2>> $O/bench_$v.err | tail -1 | sed "s/^/$v shard: /" | tee -a $O/summary.txt
  if [ "$v" != product ]; then
    python -m pytest tests/test_knn_gpu.py -q -x -m gpu -k "not canberra" > $O/pytest_$v.log 2>&1 && echo "$v parity ok: $(tail -1 $O/pytest_$v.log)" | tee -a $O/summary.txt || { echo "$v PARITY FAILED" | tee -a $O/summary.txt; tail -20 $O/pytest_$v.log; }
  fi
done
unset NABO_KNN_SO
cat $O/summary.txt
