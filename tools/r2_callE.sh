set -e
O=$PWD/gpurun_out/${TAG:-r2e}; mkdir -p $O
( time timeout -k 10 600 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python bench.py --no-extras --no-cpu-baseline --metric canberra --steps 2 --warmup 1 > $O/bench_canberra.json 2> $O/err.txt
python -c "import json; d=json.loads(open('$O/bench_canberra.json').read().strip().splitlines()[-1]); print('canberra ms_per_step', d['ms_per_step'], d['phases_ms'], d['fallback_rows'])"
python bench.py --no-extras --no-cpu-baseline --metric canberra --targets 100000 --refs 100000 --steps 5 --warmup 1 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('canberra 100k ms_per_step', d['ms_per_step'], d['phases_ms'])"
