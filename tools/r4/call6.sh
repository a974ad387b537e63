O=$PWD/gpurun_out/${TAG:-r4c6}; mkdir -p $O
for so in "" tools/ab/cbb_b32.so tools/ab/cbb_b32_nodma.so; do
  tag=$( [ -z "$so" ] && echo product || basename $so .so )
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --metric canberra --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/canberra_$tag.json 2> $O/canberra_$tag.err
  python -c "
import json; d=json.loads(open('$O/canberra_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1), round(d['roofline']['kernel_ms'],1), d['phases_ms'], d['sampled_rows_equal_oracle'])"
done
