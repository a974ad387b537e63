set -e
O=$PWD/gpurun_out/${TAG:-r2h}; mkdir -p $O
export NABO_KNN_SO=$PWD/tools/ab/prof.so
NABO_L2_MODE=f16x3h timeout -k 5 200 python bench.py --no-extras --no-cpu-baseline --steps 1 --warmup 0 > $O/prof_l2h.json 2> $O/prof_l2h.err
tail -4 $O/prof_l2h.err
python -c "import json; d=json.loads(open('$O/prof_l2h.json').read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['ms_per_step'])"
