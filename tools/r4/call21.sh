# round 4, call 21: the tile refill requested in the middle of a step instead of at its top (behind pair 1 / pair 2)
O=$PWD/gpurun_out/${TAG:-r4c21}; mkdir -p $O
for so in tools/ab/load1.so tools/ab/load2.so; do
NABO_KNN_SO=$so timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "one_product or euclidean_knn or cosine_knn or tournament or config2" > $O/pytest_$(basename $so .so).log 2>&1; echo "pytest $so rc=$? $(tail -1 $O/pytest_$(basename $so .so).log)"
done
for so in "" tools/ab/load1.so tools/ab/load2.so ""; do
  tag=$( [ -z "$so" ] && echo product || basename $so .so )
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_1M_$tag.json 2> $O/bench_$tag.err
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --metric cosine --dims 100 --neighbors 50 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_cosine_$tag.json 2>> $O/bench_$tag.err
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --targets 100000 --refs 100000 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_100k_$tag.json 2>> $O/bench_$tag.err
  python - <<PY
import json
for f in ("bench_1M","bench_cosine","bench_100k"):
    try:
        d=json.loads(open("$O/%s_$tag.json"%f).read().strip().splitlines()[-1]); print("$tag",f,"ms/step %.2f kernel %.2f"%(d["ms_per_step"], d["roofline"]["kernel_ms"]), d["sampled_rows_equal_oracle"])
    except Exception as e: print("$tag",f,"ERR",e)
PY
  NABO_KNN_SO=$so timeout -k 10 200 python tools/bench_shard.py 8 2>/dev/null | tail -1 | cut -c1-120
done
