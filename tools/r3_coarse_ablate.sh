# where the one-product pass's kernel time goes: no-hit / L2-resident / L1-resident ablations (experiment build) + list counters
O=$PWD/gpurun_out/r3coarse_ab; mkdir -p $O
B="--no-extras --no-cpu-baseline --steps 3 --warmup 1"
export NABO_COARSE_SLACK=${SLACK:-0}
run() { # name env...
  n=$1; shift
  env "$@" python bench.py $B > $O/$n.json 2> $O/$n.err
  python -c "
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
print('$n: kernel_ms %.2f ms_per_step %.2f' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
}
run base NABO_KNN_SO=$PWD/tools/ab/exp.so
run nohit NABO_KNN_SO=$PWD/tools/ab/exp.so NABO_DEBUG_ABLATE=1
run l2res NABO_KNN_SO=$PWD/tools/ab/exp.so NABO_DEBUG_ABLATE=2
run l1res NABO_KNN_SO=$PWD/tools/ab/exp.so NABO_DEBUG_ABLATE=4
run nohit_l1 NABO_KNN_SO=$PWD/tools/ab/exp.so NABO_DEBUG_ABLATE=5
run prof NABO_KNN_SO=$PWD/tools/ab/prof.so
grep "lists prof" $O/prof.err | tail -2
run f16x3_base NABO_KNN_SO=$PWD/tools/ab/exp.so NABO_L2_MODE=f16x3
run f16x3_nohit NABO_KNN_SO=$PWD/tools/ab/exp.so NABO_L2_MODE=f16x3 NABO_DEBUG_ABLATE=1
