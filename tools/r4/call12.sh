# round 4, call 12: why is a launch cut into pieces slower than the split grid?  kernel traces of 49 152 x 100 000 both ways;
# parity of the merged lists (first pass only); Canberra back at its round-4 best.
O=$PWD/gpurun_out/${TAG:-r4c12}; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for v in pieces=1 pieces=0; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 tools/sweep_plan.py 49152 100000 50 15 $v > $O/trace_$v.txt 2> $O/trace_$v.err
  find $O/prof_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/stats_$v.csv
  find $O/prof_$v -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/ktrace_$v.csv
  rm -rf $O/prof_$v
  echo "== $v"; cut -c1-200 $O/trace_$v.txt; cut -d, -f1-4 $O/stats_$v.csv | head -8
done
for so in tools/ab/base.so ""; do
  tag=$( [ -z "$so" ] && echo product || basename $so .so )
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --metric canberra --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/canberra_$tag.json 2> $O/canberra_$tag.err
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_1M_$tag.json 2> $O/bench_1M_$tag.err
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --targets 100000 --refs 100000 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_100k_$tag.json 2>> $O/bench_1M_$tag.err
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --metric cosine --dims 100 --neighbors 50 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_cosine_$tag.json 2>> $O/bench_1M_$tag.err
  NABO_KNN_SO=$so timeout -k 10 200 python tools/bench_shard.py 8 2>> $O/bench_1M_$tag.err | tail -1 > $O/shard_$tag.txt
  python - <<PY
import json
for f in ("canberra","bench_1M","bench_100k","bench_cosine"):
    try:
        d=json.loads(open("$O/%s_$tag.json"%f).read().strip().splitlines()[-1]); print("$tag",f,"ms/step %.2f kernel %.2f"%(d["ms_per_step"], d["roofline"]["kernel_ms"]), d["phases_ms"], d["sampled_rows_equal_oracle"])
    except Exception as e: print("$tag",f,"ERR",e)
PY
  cut -c1-200 $O/shard_$tag.txt
done
