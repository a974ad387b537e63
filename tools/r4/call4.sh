# round 4, call 4: after the library cleanup (options instead of environment knobs, plan / execute split, experimental kernels out
# of the product build), geometry B as two 4-wave workgroups per CU, the pipelined row reads of the Canberra counting pass.
O=$PWD/gpurun_out/${TAG:-r4c4}; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_1M.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --targets 100000 --refs 100000 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_100k.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --metric cosine --dims 100 --neighbors 50 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_cosine.json 2>> $O/bench.err
timeout -k 10 200 python tools/bench_shard.py 8 2>> $O/bench.err | tail -1 > $O/shard_share.txt
timeout -k 10 300 python tools/bench_canberra.py 1000000 1000000 50 15 > $O/canberra_1M.json 2>> $O/bench.err
NABO_BENCH_LOOPBACK=8 NABO_BENCH_CHECK=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_loopback8.json 2> $O/bench_loopback8.err; echo "loopback rc=$?"
python - <<'PY'
import json,glob,os
O=os.environ.get("TAG","r4c4")
for f in sorted(glob.glob("gpurun_out/%s/bench_*.json"%O)):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms/step %.2f kernel %.2f frac %.3f"%(d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]), d["phases_ms"], d["rows_by_pass"], d["sampled_rows_equal_oracle"])
        if "sharded" in d: print("   sharded", d["sharded"]["max_over_ranks_ms"], "alt", {k:(d[k]["ms_per_step"], d[k]["same_bits_as_headline_layout"]) for k in d if k.startswith("alt_layout")})
    except Exception as e:
        print(f, "ERR", e)
d=json.loads(open("gpurun_out/%s/canberra_1M.json"%O).read().strip().splitlines()[-1]); print("canberra", d["ms_per_step"], d["phases_ms"])
PY
cat $O/shard_share.txt | cut -c1-220
