# round 4, call 20: tail launch with up to 12 splits (lists merged), refine gather 8 / 12 / 16 rows at a time, Canberra DMA waves
# at raised priority -- same box
O=$PWD/gpurun_out/${TAG:-r4c20}; mkdir -p $O
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "tail_round or config3 or cut_launches or row_pass or canberra_bit_sliced" > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for so in tools/ab/base.so "" tools/ab/sb12.so tools/ab/sb16.so tools/ab/base.so ""; do
  tag=$( [ -z "$so" ] && echo product || basename $so .so )
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_1M_$tag.json 2> $O/bench_1M_$tag.err
  python - <<PY
import json
d=json.loads(open("$O/bench_1M_$tag.json").read().strip().splitlines()[-1]); print("$tag","ms/step %.2f kernel %.2f"%(d["ms_per_step"], d["roofline"]["kernel_ms"]), d["phases_ms"], d["sampled_rows_equal_oracle"])
PY
done
for so in tools/ab/base.so tools/ab/prio2.so ""; do
  tag=$( [ -z "$so" ] && echo product || basename $so .so )
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --metric canberra --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/canberra_$tag.json 2> $O/canberra_$tag.err
  python -c "
import json; d=json.loads(open('$O/canberra_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1), round(d['roofline']['kernel_ms'],1), d['sampled_rows_equal_oracle'])"
done
for so in tools/ab/base.so tools/ab/sb12.so; do NABO_KNN_SO=$so timeout -k 10 200 python tools/bench_shard.py 8 2>/dev/null | tail -1 | cut -c1-170; done
