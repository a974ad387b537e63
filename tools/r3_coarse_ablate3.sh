O=$PWD/gpurun_out/r3coarse_ab3; mkdir -p $O
B="--no-extras --no-cpu-baseline --steps 3 --warmup 1"
export NABO_COARSE_SLACK=${SLACK:-0}
run() { # name env...
  n=$1; shift
  env "$@" python bench.py $B > $O/$n.json 2> $O/$n.err
  python -c "
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
print('$n: kernel_ms %.2f ms_per_step %.2f oracle %s' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['sampled_rows_equal_oracle']))"
}
for v in exp nofilter noreload nofr; do
run ${v}_nohit NABO_KNN_SO=$PWD/tools/ab/$v.so NABO_DEBUG_ABLATE=1
run ${v}_nohit_h NABO_KNN_SO=$PWD/tools/ab/$v.so NABO_DEBUG_ABLATE=1 NABO_L2_MODE=f16x1h
done
