# round 4, call 1: the parity suite on the new build (row-pass record, tournament seeds, configs[4] whole), then same-box A/Bs
# of the tournament seeds (NABO_PREPASS = percent of the planned length, 0 = off) on the headline, a shard of eight, the long
# lists (cosine d=100 k=50) and configs[1].
O=$PWD/gpurun_out/${TAG:-r4c1}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for pre in 100 0 50 200; do
  NABO_PREPASS=$pre timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_pre$pre.json 2> $O/bench_pre$pre.err; echo "bench pre=$pre rc=$?"
done
for pre in 100 0 200; do
  NABO_PREPASS=$pre timeout -k 10 200 python tools/bench_shard.py 8 2>> $O/shard.err | tail -1 | sed "s/^/pre=$pre /" >> $O/shard_share.txt
  NABO_PREPASS=$pre timeout -k 10 200 python bench.py --metric cosine --dims 100 --neighbors 50 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_cosine_pre$pre.json 2>> $O/shard.err
  NABO_PREPASS=$pre timeout -k 10 200 python bench.py --targets 100000 --refs 100000 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_100k_pre$pre.json 2>> $O/shard.err
done
python - <<'PY'
import json,glob,os
O=os.environ.get("TAG","r4c1")
for f in sorted(glob.glob("gpurun_out/%s/bench_*.json"%O)):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms/step %.2f kernel %.2f frac %.3f"%(d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]), d["phases_ms"], d["rows_by_pass"], d["sampled_rows_equal_oracle"])
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/shard_share.txt | cut -c1-200
