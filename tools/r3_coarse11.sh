O=$PWD/gpurun_out/r3c11; mkdir -p $O
for sl in 0 1; do
NABO_COARSE_CAND_SLACK=$sl NABO_BENCH_LOOPBACK=8 NABO_BENCH_CHECK=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/lb8_$sl.json 2> $O/lb8_$sl.err; echo "loopback slack $sl rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r3c11/lb8_$sl.json').read().strip().splitlines()[-1])
s=d['sharded']
print('slack $sl', d['ms_per_step'], d['sampled_rows_equal_oracle'], 'second_round', s['second_round_rows'], {k:round(v,2) for k,v in s['max_over_ranks_ms'].items()})
a=d['alt_layout']
print('   alt', a['ms_per_step'], a['same_bits_as_headline_layout'], a['second_round_rows'])
PY
done
