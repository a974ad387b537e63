set -e
TAG=${TAG:-r2ab}
O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B="--no-extras --no-cpu-baseline"
python bench.py $B --steps 4 --warmup 2 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['phases_ms'], d['ms_per_step'])"
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d $O/pmc_l1/pass1 -- python3 bench.py $B --steps 1 --warmup 0 > $O/pmc_l1.json 2> $O/pmc_l1.err || echo "pmc l1 failed"
python tools/pmc_summary.py $O/pmc_l1 | grep "topk_kernel" || tail -5 $O/pmc_l1.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/pmc_l2/pass1 -- python3 bench.py $B --steps 1 --warmup 0 > $O/pmc_l2.json 2> $O/pmc_l2.err || echo "pmc l2 failed"
python tools/pmc_summary.py $O/pmc_l2 | grep "topk_kernel" || tail -5 $O/pmc_l2.err
