mkdir -p gpurun_out/r3lb24
for N in 2; do
NABO_BENCH_LOOPBACK=$N NABO_BENCH_CHECK=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3lb24/N$N.json 2> gpurun_out/r3lb24/N$N.err; echo "N=$N rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r3lb24/N$N.json').read().strip().splitlines()[-1])
s=d['sharded']
print('N=$N', round(d['ms_per_step'],1), d['sampled_rows_equal_oracle'], d['config']['workload'], 'second', s['second_round_rows'], 'cand', s['candidates_per_shard'], [(k, round(d[k]['ms_per_step'],1), d[k]['same_bits_as_headline_layout'], d[k]['second_round_rows']) for k in ('alt_layout','alt_layout_target_slices') if k in d])
PY
done
