# Canberra parity subset + the 1M x 1M step on the product library and on tools/ab variants:   bash tools/r3_cb3.sh product [variant ...]
O=$PWD/gpurun_out/r3cb3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_knn_gpu.py tests/test_mapping.py -m gpu -q -x -k "canberra or mapping" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -1 $O/pytest.log | cut -c1-200)"; if [ $rc != 0 ]; then tail -30 $O/pytest.log; exit 1; fi
for v in "$@"; do
  if [ "$v" = product ]; then unset NABO_KNN_SO; else export NABO_KNN_SO=$PWD/tools/ab/$v.so; fi
  python bench.py --metric canberra --steps 3 --warmup 1 --no-cpu-baseline > $O/cb_$v.json 2> $O/cb_$v.err
  python -c "
import json
d=json.loads(open('$O/cb_$v.json').read().strip().splitlines()[-1])
print('$v', round(d['ms_per_step'],1), d['sampled_rows_equal_oracle'], d['fallback_rows'], {k:round(v,1) for k,v in d['phases_ms'].items() if k.startswith('ms_')}, d['roofline']['kernel'][:30])"
done
