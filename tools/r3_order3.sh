set -e
TAG=${1:-r3f}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
B="--no-extras --no-cpu-baseline"
for f in 0 1 2 3 7 0; do
  NABO_L2Q_ORDER=$f python bench.py $B --steps 4 --warmup 2 2> $O/err_$f.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('flags=$f: ms_per_step %.2f kernel_ms %.2f pack %.2f refine %.2f oracle_rows %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['phases_ms']['ms_pack'], d['phases_ms']['ms_refine'], d['sampled_rows_equal_oracle']))" | tee -a $O/summary.txt
done
