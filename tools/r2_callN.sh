set -e
O=$PWD/gpurun_out/${TAG:-r2n}; mkdir -p $O
B="--no-extras --no-cpu-baseline"
( time timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "canberra or mapping or golden or smoke or sweep" ) > $O/pytest_canberra.log 2>&1 || { tail -30 $O/pytest_canberra.log; exit 1; }
tail -3 $O/pytest_canberra.log
python bench.py $B --metric canberra --steps 2 --warmup 1 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('canberra ms_per_step', d['ms_per_step'], d['phases_ms'], d['fallback_rows'])"
NABO_DEBUG_ABLATE=4 python bench.py $B --metric canberra --steps 1 --warmup 0 2>&1 >/dev/null | grep "nabo debug" | tail -2
python bench.py $B --metric canberra --targets 100000 --refs 100000 --steps 5 --warmup 1 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('canberra 100k ms_per_step', d['ms_per_step'], d['phases_ms'], d['fallback_rows'])"
( time timeout -k 10 600 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
