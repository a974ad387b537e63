# round 4, call 14: the tail of a one-round plan beside the main launch (second stream)
O=$PWD/gpurun_out/${TAG:-r4c14}; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "cut_launches or tail_round or reference_splits or config2 or row_pass" > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for shape in "100000 100000" "76000 50000" "120000 1000000" "110000 30000" "60000 200000"; do
  timeout -k 10 300 python tools/sweep_plan.py $shape 50 15 default one_round=0 > $O/sweep_$(echo $shape | tr ' ' 'x').txt 2>&1
  echo "== $shape"; cut -c1-250 $O/sweep_$(echo $shape | tr ' ' 'x').txt
done
timeout -k 10 200 python bench.py --targets 100000 --refs 100000 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_100k.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_100k.json").read().strip().splitlines()[-1]); print("bench_100k ms/step %.2f kernel %.2f"%(d["ms_per_step"], d["roofline"]["kernel_ms"]), d["phases_ms"], d["sampled_rows_equal_oracle"])
PY
