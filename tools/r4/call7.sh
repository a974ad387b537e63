# round 4, call 7: the L2 prefetch of the Canberra counting pass (parity first, then same-box A/B of the prefetch distance / roles),
# and the whole-run time of configs[4] on one GPU.
O=$PWD/gpurun_out/${TAG:-r4c7}; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "canberra or Canberra or mapping" > $O/pytest_canberra.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest_canberra.log)"
for so in "" tools/ab/cbb_pf0.so tools/ab/cbb_pf2.so tools/ab/cbb_pf5.so tools/ab/cbb_v2.so tools/ab/cbb_v2d6.so tools/ab/cbb_pf3_nodma.so ""; do
  tag=$( [ -z "$so" ] && echo product || basename $so .so )
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --metric canberra --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/canberra_$tag.json 2> $O/canberra_$tag.err
  python -c "
import json; d=json.loads(open('$O/canberra_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1), round(d['roofline']['kernel_ms'],1), d['sampled_rows_equal_oracle'])"
done
( time timeout -k 10 500 python tools/config4_one_gpu.py > $O/config4.json 2> $O/config4.err ) 2>&1 | grep real
tail -c 1500 $O/config4.json
