# round 4 final GPU call: parity suite, smoke, the driver's bench line, the 8-rank loopback rehearsal, the other shapes, one rank's
# share, kernel traces + counter passes (tools/r4/pmc.sh).
O=$PWD/gpurun_out/${TAG:-r4final}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
NABO_BENCH_LOOPBACK=8 NABO_BENCH_CHECK=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_loopback8.json 2> $O/bench_loopback8.err; echo "loopback rc=$?"
timeout -k 10 200 python bench.py --targets 100000 --refs 100000 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_100k.json 2>> $O/bench_default.err
timeout -k 10 200 python bench.py --metric cosine --dims 100 --neighbors 50 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_cosine.json 2>> $O/bench_default.err
for n in 8 4 2; do timeout -k 10 200 python tools/bench_shard.py $n 2>> $O/bench_default.err | tail -1 >> $O/shard_share.txt; done
TAG=${TAG:-r4final}_pmc bash tools/r4/pmc.sh > $O/pmc.log 2>&1; echo "pmc rc=$?"
tail -3 $O/pmc.log | cut -c1-200; cut -c1-200 $O/shard_share.txt
python - <<PY
import json
for f in ("bench_default","bench_loopback8","bench_100k","bench_cosine"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1]); print(f,"ms/step %.2f kernel %.2f frac %.3f"%(d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"] or 0), d["phases_ms"], d["sampled_rows_equal_oracle"], {k:(round(v["ms_per_step"],1) if isinstance(v,dict) and "ms_per_step" in v else None) for k,v in d.items() if k in ("canberra","alt","alt_f16x3","alt_layout","alt_layout_target_slices")}, (d.get("config4_one_gpu") or {}).get("target_knn",{}).get("gpu_ms"))
    except Exception as e: print(f,"ERR",e)
PY
