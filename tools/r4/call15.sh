# round 4, call 15: the seeded pass's lists merged to 64 entries; Canberra with 4 / 2 DMA waves; parity suite
O=$PWD/gpurun_out/${TAG:-r4c15}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for so in "" tools/ab/cbb_dmaw4.so tools/ab/cbb_dmaw2.so ""; do
  tag=$( [ -z "$so" ] && echo product || basename $so .so )
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --metric canberra --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/canberra_$tag.json 2> $O/canberra_$tag.err
  python -c "
import json; d=json.loads(open('$O/canberra_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1), round(d['roofline']['kernel_ms'],1), d['phases_ms'], d['sampled_rows_equal_oracle'])"
done
for so in tools/ab/base.so ""; do
  tag=$( [ -z "$so" ] && echo product || basename $so .so )
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_1M_$tag.json 2> $O/bench_1M_$tag.err
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --targets 100000 --refs 100000 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_100k_$tag.json 2>> $O/bench_1M_$tag.err
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --metric cosine --dims 100 --neighbors 50 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_cosine_$tag.json 2>> $O/bench_1M_$tag.err
  NABO_KNN_SO=$so timeout -k 10 200 python tools/bench_shard.py 8 2>> $O/bench_1M_$tag.err | tail -1 > $O/shard_$tag.txt
  python - <<PY
import json
for f in ("bench_1M","bench_100k","bench_cosine"):
    try:
        d=json.loads(open("$O/%s_$tag.json"%f).read().strip().splitlines()[-1]); print("$tag",f,"ms/step %.2f kernel %.2f"%(d["ms_per_step"], d["roofline"]["kernel_ms"]), d["phases_ms"], d["rows_by_pass"], d["sampled_rows_equal_oracle"])
    except Exception as e: print("$tag",f,"ERR",e)
PY
  cut -c1-200 $O/shard_$tag.txt
done
