# all layouts of eight loopback ranks on one GPU (ref_shards = 1, 2, 4, 8): GPU time of the whole protocol, second-round rows, slowest rank's phases
mkdir -p gpurun_out/r3lb
for R in 1 2 4 8; do
NABO_REF_SHARDS=$R NABO_BENCH_LOOPBACK=8 NABO_BENCH_CHECK=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3lb/R$R.json 2> gpurun_out/r3lb/R$R.err; echo "R=$R rc=$?"
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r3lb/R$R.json').read().strip().splitlines()[-1])
    s=d['sharded']; mx=s['max_over_ranks_ms']
    print('R=$R', round(d['ms_per_step'],1), d['sampled_rows_equal_oracle'], 'second_round', s['second_round_rows'], 'cand', s['candidates_per_shard'], {k:round(v,2) for k,v in mx.items() if k in ('ms_topk_local','ms_local','ms_exchange','ms_merge','ms_second','ms_gather')})
except Exception as e: print('R=$R', 'ERR', e, open('gpurun_out/r3lb/R$R.err').read()[-300:])
PY
done
