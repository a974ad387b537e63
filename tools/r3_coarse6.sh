O=$PWD/gpurun_out/r3coarse6; mkdir -p $O
B="--no-extras --no-cpu-baseline --steps 3 --warmup 1"
export NABO_COARSE_SLACK=${SLACK:-0}
run() { # name env...
  n=$1; shift
  env "$@" python bench.py $B > $O/$n.json 2> $O/$n.err
  python -c "
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
print('$n: kernel_ms %.2f ms_per_step %.2f oracle %s %s' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['sampled_rows_equal_oracle'], d['phases_ms']))"
}
run q_plain NABO_COARSE_KERNEL_Q=1
run q_ord7 NABO_L2Q_ORDER=7
run q_ord3 NABO_L2Q_ORDER=3
run q_ord2 NABO_L2Q_ORDER=2
run q_ord1 NABO_L2Q_ORDER=1
