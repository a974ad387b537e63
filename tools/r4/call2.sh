# round 4, call 2: the rewritten mod-Canberra counting pass (four words per lane, LDS-DMA, four-step carry-save):
# parity first (every canberra test + the Mapping end-to-end cases), then same-box A/B against the round-3 kernel
# (tools/ab/r4_base.so = this tree's library before the rewrite), then the whole -m gpu suite.
O=$PWD/gpurun_out/${TAG:-r4c2}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_knn_gpu.py tests/test_mapping.py -m gpu -x -q -k "canberra or mapping or golden or random_small or ignore_mask or row_pass" > $O/pytest_canberra.log 2>&1; rc=$?; echo "canberra pytest rc=$rc $(tail -1 $O/pytest_canberra.log)"
if [ $rc -ne 0 ]; then tail -40 $O/pytest_canberra.log; fi
for so in "" tools/ab/r4_base.so; do
  tag=$( [ -z "$so" ] && echo new || echo base )
  NABO_KNN_SO=$so timeout -k 10 300 python tools/bench_canberra.py 1000000 1000000 50 15 > $O/canberra_1M_$tag.json 2> $O/canberra_1M_$tag.err; echo "canberra 1M $tag rc=$?"
  NABO_KNN_SO=$so NABO_CANBERRA_MODE=bits timeout -k 10 300 python tools/bench_canberra.py 100000 100000 50 15 > $O/canberra_100k_bits_$tag.json 2>> $O/canberra_1M_$tag.err
  NABO_KNN_SO=$so timeout -k 10 300 python tools/bench_canberra.py 300000 300000 30 11 > $O/canberra_300k_d30_$tag.json 2>> $O/canberra_1M_$tag.err
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/%s/canberra_*.json"%os.environ.get("TAG","r4c2"))):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms/step %.1f"%d["ms_per_step"], d["phases_ms"], d["uncertified_rows_resolved_exactly"])
    except Exception as e:
        print(f, "ERR", e)
PY
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
