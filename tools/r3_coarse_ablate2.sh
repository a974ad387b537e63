O=$PWD/gpurun_out/r3coarse_ab2; mkdir -p $O
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
run base NABO_KNN_SO=$PWD/tools/ab/exp.so
run nohit NABO_KNN_SO=$PWD/tools/ab/exp.so NABO_DEBUG_ABLATE=1
run h_base NABO_KNN_SO=$PWD/tools/ab/exp.so NABO_L2_MODE=f16x1h
run h_nohit NABO_KNN_SO=$PWD/tools/ab/exp.so NABO_L2_MODE=f16x1h NABO_DEBUG_ABLATE=1
