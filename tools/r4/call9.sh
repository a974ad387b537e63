# round 4, call 9: is the Canberra counting pass bound by its table DMA?  (every piece requested twice: same results);
# one round of workgroups at two waves per SIMD for ~100k rows (98 304 = 256 x 384); list counters of the long-list kernel.
O=$PWD/gpurun_out/${TAG:-r4c9}; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "canberra or Canberra" > $O/pytest_canberra.log 2>&1; echo "pytest product rc=$? $(tail -1 $O/pytest_canberra.log)"
for so in "" tools/ab/cbb_dma2x.so tools/ab/cbb_dmaw16_2x.so tools/ab/cbb_nodma.so ""; do
  tag=$( [ -z "$so" ] && echo product || basename $so .so )
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --metric canberra --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/canberra_$tag.json 2> $O/canberra_$tag.err
  python -c "
import json; d=json.loads(open('$O/canberra_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1), round(d['roofline']['kernel_ms'],1), d['sampled_rows_equal_oracle'])"
done
timeout -k 10 300 python tools/sweep_plan.py 98304 100000 50 15 default splits=2 splits=2,prepass=50 > $O/sweep_98304.txt 2>&1
cut -c1-260 $O/sweep_98304.txt
timeout -k 10 300 python tools/sweep_plan.py 49152 100000 50 15 default splits=2 splits=4 > $O/sweep_49152.txt 2>&1
cut -c1-260 $O/sweep_49152.txt
NABO_KNN_SO=tools/ab/lprof.so timeout -k 10 300 python bench.py --metric cosine --dims 100 --neighbors 50 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $O/lprof_cosine.json 2> $O/lprof_cosine.err
grep "lists prof" $O/lprof_cosine.err | tail -4
NABO_KNN_SO=tools/ab/lprof.so timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $O/lprof_1M.json 2> $O/lprof_1M.err
grep "lists prof" $O/lprof_1M.err | tail -4
