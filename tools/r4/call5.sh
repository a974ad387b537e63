# round 4, call 5: (a) what bounds the Canberra counting pass: timing variants without the table DMA / without the step barrier
# (garbage results, time only); (b) kernel traces of configs[1] (100k x 100k) and of the headline.
O=$PWD/gpurun_out/${TAG:-r4c5}; mkdir -p $O
export TMPDIR=/tmp
for so in "" tools/ab/cbb_nodma.so tools/ab/cbb_nobar.so tools/ab/cbb_nodma_nobar.so; do
  tag=$( [ -z "$so" ] && echo product || basename $so .so )
  NABO_KNN_SO=$so timeout -k 10 200 python bench.py --metric canberra --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/canberra_$tag.json 2> $O/canberra_$tag.err
  python -c "
import json; d=json.loads(open('$O/canberra_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1), round(d['roofline']['kernel_ms'],1))"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_100k -- python3 bench.py --targets 100000 --refs 100000 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_100k_rocprof.json 2> $O/rocprof.err
find $O/prof_100k -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/100k_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1M -- python3 bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_1M_rocprof.json 2>> $O/rocprof.err
find $O/prof_1M -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/1M_kernel_stats.csv
rm -rf $O/prof_100k $O/prof_1M
cut -d, -f1-4 $O/100k_kernel_stats.csv | head -14; cut -d, -f1-4 $O/1M_kernel_stats.csv | head -12
