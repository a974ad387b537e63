# round 2, GPU call A: the whole -m gpu suite, the default bench line, the counter list (run from the repo root on the GPU box)
set -e
O=$PWD/gpurun_out/${TAG:-r2a}_a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( time timeout -k 10 850 python -m pytest tests -m gpu -q --durations=15 ) > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -25 $O/pytest.log
echo "pytest done"
timeout -k 10 400 python bench.py > $O/bench_1Mx1M.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python -c "
import json,sys
d=json.loads(open('$O/bench_1Mx1M.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','fallback_rows','so_digest')}); print(d['roofline']); print(d.get('alt')); print(d.get('canberra')); print(d.get('cpu_baseline'))"
echo "bench done"
rocprofv3 -L > $O/counters.txt 2>&1 || true
echo "counter list done"
