# round 4, call 13: one round of workgroups for queries with fewer column-workgroups than slots (uniform splits + a tail launch),
# lists merged in the first pass only: parity suite, then the sweeps and the bench shapes against the round-4 start (base.so).
O=$PWD/gpurun_out/${TAG:-r4c13}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for shape in "100000 100000" "30000 30000" "49152 100000" "150000 100000" "20000 100000" "3000 3000" "76000 50000" "120000 1000000"; do
  timeout -k 10 300 python tools/sweep_plan.py $shape 50 15 default one_round=0 > $O/sweep_$(echo $shape | tr ' ' 'x').txt 2>&1
  echo "== $shape"; cut -c1-250 $O/sweep_$(echo $shape | tr ' ' 'x').txt
done
for so in tools/ab/base.so ""; do
  tag=$( [ -z "$so" ] && echo product || basename $so .so )
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_1M_$tag.json 2> $O/bench_1M_$tag.err
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --targets 100000 --refs 100000 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_100k_$tag.json 2>> $O/bench_1M_$tag.err
  python - <<PY
import json
for f in ("bench_1M","bench_100k"):
    try:
        d=json.loads(open("$O/%s_$tag.json"%f).read().strip().splitlines()[-1]); print("$tag",f,"ms/step %.2f kernel %.2f"%(d["ms_per_step"], d["roofline"]["kernel_ms"]), d["phases_ms"], d["sampled_rows_equal_oracle"])
    except Exception as e: print("$tag",f,"ERR",e)
PY
done
