O=$PWD/gpurun_out/r3c12; mkdir -p $O
python -m pytest tests/test_sharded.py tests/test_configs_gpu.py tests/test_knn_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
python tools/bench_small_query.py 125000 2>&1 | cut -c1-150 | head -4
NABO_BENCH_LOOPBACK=8 NABO_BENCH_CHECK=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/lb8.json 2> $O/lb8.err; echo "loopback rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r3c12/lb8.json').read().strip().splitlines()[-1])
s=d['sharded']
print(d['ms_per_step'], d['sampled_rows_equal_oracle'], 'second_round', s['second_round_rows'], {k:round(v,2) for k,v in s['max_over_ranks_ms'].items()})
a=d['alt_layout']
print('   alt', a['ms_per_step'], a['same_bits_as_headline_layout'], a['second_round_rows'])
PY
python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench.json 2>$O/bench.err; python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['phases_ms'], d['sampled_rows_equal_oracle'])"
