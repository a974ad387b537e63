O=$PWD/gpurun_out/r3c10; mkdir -p $O
python -m pytest tests/test_sharded.py tests/test_configs_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for n in 8 4 2; do python tools/bench_shard.py $n 2>> $O/err.txt | tail -1 | cut -c1-200; done
for n in 8; do NABO_COARSE_CAND_SLACK=0 python tools/bench_shard.py $n 2>> $O/err.txt | tail -1 | cut -c1-200; done
for n in 8; do NABO_L2_MODE=f16x3 python tools/bench_shard.py $n 2>> $O/err.txt | tail -1 | cut -c1-200; done
NABO_BENCH_LOOPBACK=8 NABO_BENCH_CHECK=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_loopback8.json 2> $O/bench_loopback8.err; echo "loopback rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3c10/bench_loopback8.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('ms_per_step','sampled_rows_equal_oracle') if k in d}, d.get('sharded'), {k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('ms_per_step','same_bits_as_headline_layout','second_round_rows')}) for k,v in d.items() if k=='alt_layout'})
PY
