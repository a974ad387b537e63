set -e
B="--no-extras --no-cpu-baseline --steps 2 --warmup 1"
for L in 8 4 2; do
NABO_BENCH_LOOPBACK=$L NABO_BENCH_CHECK=1 timeout -k 10 300 python bench.py $B | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('loopback $L', round(d['ms_per_step'],1), d['config']['parallelism'], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d['sharded'].items() if k not in ('max_over_ranks_ms',)})"
done
NABO_REF_SHARDS=8 NABO_BENCH_LOOPBACK=8 NABO_BENCH_CHECK=1 timeout -k 10 300 python bench.py $B | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('loopback 8 one-dimensional', round(d['ms_per_step'],1), d['config']['parallelism'], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d['sharded'].items() if k not in ('max_over_ranks_ms',)})"
