O=$PWD/gpurun_out/r3wide; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -1 $O/pytest.log | cut -c1-200)"; if [ $rc != 0 ]; then tail -30 $O/pytest.log; exit 1; fi
python bench.py --metric cosine --dims 100 --neighbors 50 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_cosine.json 2> $O/err.txt
NABO_L2_MODE=f32 python bench.py --metric cosine --dims 100 --neighbors 50 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_cosine_f32.json 2>> $O/err.txt
python bench.py --dims 100 --neighbors 15 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_d100.json 2>> $O/err.txt
python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_default.json 2>> $O/err.txt
python - <<'PY'
import json
for f in ('bench_cosine','bench_cosine_f32','bench_d100','bench_default'):
    d=json.loads(open('gpurun_out/r3wide/%s.json'%f).read().strip().splitlines()[-1])
    print(f, round(d['ms_per_step'],1), d['sampled_rows_equal_oracle'], d['rows_by_pass'], {k:round(v,1) for k,v in d['phases_ms'].items()}, d['roofline']['kernel'][:40])
PY
