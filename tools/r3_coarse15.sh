O=$PWD/gpurun_out/r3c15; mkdir -p $O
B="--no-extras --no-cpu-baseline --steps 3 --warmup 1"
run() { n=$1; shift
  env "$@" python bench.py $B > $O/$n.json 2> $O/$n.err
  python -c "
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
print('$n: kernel_ms %.2f ms_per_step %.2f' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
}
export NABO_KNN_SO=$PWD/tools/ab/exp.so
run b_base
run b_nohit NABO_DEBUG_ABLATE=1
run b_nohit_l1 NABO_DEBUG_ABLATE=5
run a_nohit NABO_DEBUG_ABLATE=1 NABO_L2C_GEO=a
