# round 4, call 8: Canberra counting pass with the one-pointer fetch (immediate-offset pieces), the carry-chain comparator and
# constant-offset row-address reads: parity (which immediate-offset form is right), then same-box A/B; how 100k x 100k should be cut.
O=$PWD/gpurun_out/${TAG:-r4c8}; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "canberra or Canberra" > $O/pytest_canberra.log 2>&1; echo "pytest product rc=$? $(tail -1 $O/pytest_canberra.log)"
NABO_KNN_SO=tools/ab/cbb_m0per.so timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "canberra or Canberra" > $O/pytest_canberra_m0per.log 2>&1; echo "pytest m0per rc=$? $(tail -1 $O/pytest_canberra_m0per.log)"
for so in tools/ab/cbb_base.so "" tools/ab/cbb_m0per.so tools/ab/cbb_dmaw8.so tools/ab/cbb_base.so ""; do
  tag=$( [ -z "$so" ] && echo product || basename $so .so )
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --metric canberra --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/canberra_$tag.json 2> $O/canberra_$tag.err
  python -c "
import json; d=json.loads(open('$O/canberra_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1), round(d['roofline']['kernel_ms'],1), d['sampled_rows_equal_oracle'])"
done
timeout -k 10 300 python tools/sweep_plan.py 100000 100000 50 15 default splits=2 splits=3 splits=4 l2c_geo=0 l2c_geo=0,splits=2 l2c_geo=0,splits=3 prepass=0 prepass=50 > $O/sweep_100k.txt 2>&1
cut -c1-260 $O/sweep_100k.txt
timeout -k 10 300 python tools/sweep_plan.py 30000 30000 50 15 default splits=4 splits=8 l2c_geo=0,splits=4 > $O/sweep_30k.txt 2>&1
cut -c1-260 $O/sweep_30k.txt
