O=$PWD/gpurun_out/r3c13; mkdir -p $O
python -m pytest tests/test_knn_gpu.py tests/test_mapping.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
B="--no-extras --no-cpu-baseline --steps 5 --warmup 2"
run() { n=$1; shift
  env "$@" python bench.py $B > $O/$n.json 2> $O/$n.err
  python -c "
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
print('$n: ms_per_step %.2f oracle %s fallback %s' % (d['ms_per_step'], d['sampled_rows_equal_oracle'], d['fallback_rows']), {k:round(v,2) for k,v in d['phases_ms'].items()})"
}
run default
run noseed NABO_SEEDED_PASS=0
run slackm2 NABO_COARSE_SLACK=-2
run slackm4 NABO_COARSE_SLACK=-4
run slackm6 NABO_COARSE_SLACK=-6
