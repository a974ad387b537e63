# round 4, call 17: pack kernel with its rows staged through LDS, tournament length 90, row-pass test; list-length sweep at 1M
O=$PWD/gpurun_out/${TAG:-r4c18}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_1M.json 2> $O/bench.err
timeout -k 10 200 python bench.py --targets 100000 --refs 100000 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_100k.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --metric cosine --dims 100 --neighbors 50 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_cosine.json 2>> $O/bench.err
python - <<PY
import json
for f in ("bench_1M","bench_100k","bench_cosine"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1]); print(f,"ms/step %.2f kernel %.2f"%(d["ms_per_step"], d["roofline"]["kernel_ms"]), d["phases_ms"], d["rows_by_pass"], d["sampled_rows_equal_oracle"])
    except Exception as e: print(f,"ERR",e)
PY
for o in "" prepass=70; do timeout -k 10 200 python tools/bench_shard.py 8 $o 2>> $O/bench.err | tail -1 | cut -c1-200; done

