O=$PWD/gpurun_out/r3wide2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_knn_gpu.py tests/test_configs_gpu.py -m gpu -q -x -k "cosine or config or wide_lists or both_filter" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -1 $O/pytest.log | cut -c1-200)"; if [ $rc != 0 ]; then tail -30 $O/pytest.log; exit 1; fi
run() { n=$1; shift
 env "$@" python bench.py --metric cosine --dims 100 --neighbors 50 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/$n.json 2>> $O/err.txt
 python -c "
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
print('$n', round(d['ms_per_step'],1), d['sampled_rows_equal_oracle'], d['rows_by_pass'], {k:round(v,1) for k,v in d['phases_ms'].items()}, d['roofline']['kernel'][:40])"
}
run centred
run uncentred NABO_COSINE_CENTRE=0
run centred_slack6 NABO_COARSE_SLACK=6
run centred_f32 NABO_L2_MODE=f32
