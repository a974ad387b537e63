# one-product first pass: parity of the Euclidean suite, then the step time against the f16x3 first pass and over the list slack
O=$PWD/gpurun_out/r3coarse; mkdir -p $O
python -m pytest tests/test_knn_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for sl in 5 0 9; do
  NABO_COARSE_SLACK=$sl python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_slack$sl.json 2> $O/bench_slack$sl.err; echo "slack $sl rc=$?"
done
NABO_L2_MODE=f16x3 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_f16x3.json 2> $O/bench_f16x3.err; echo "f16x3 rc=$?"
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('O','gpurun_out/r3coarse')+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), round(d['ms_per_step'],1), d['phases_ms'], d.get('fallback_rows'), d.get('second_pass_rows'), d.get('sampled_rows_equal_oracle'), d['roofline']['kernel'][:40])
    except Exception as e: print(f, 'ERR', e)
PY
