O=$PWD/gpurun_out/r3c16; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_knn_gpu.py tests/test_sharded.py -m gpu -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -1 $O/pytest.log | cut -c1-200)"; if [ $rc != 0 ]; then exit 1; fi
B="--no-extras --no-cpu-baseline --steps 5 --warmup 2"
run() { n=$1; shift
  env "$@" python bench.py $B > $O/$n.json 2> $O/$n.err
  python -c "
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
print('$n: ms_per_step %.2f oracle %s fallback %s' % (d['ms_per_step'], d['sampled_rows_equal_oracle'], d['fallback_rows']), {k:round(v,2) for k,v in d['phases_ms'].items()}, d['roofline']['kernel'][:36])"
}
run geo_b
run geo_a NABO_L2C_GEO=a
run f16x3 NABO_L2_MODE=f16x3
run f32 NABO_L2_MODE=f32
