O=$PWD/gpurun_out/r3coarse8; mkdir -p $O
python -m pytest tests/test_knn_gpu.py -m gpu -q -x --deselect "tests/test_knn_gpu.py::test_both_filter_kernels_give_the_same_bits" --deselect "tests/test_knn_gpu.py::test_locality_ordered_streaming_gives_the_same_bits" > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
B="--no-extras --no-cpu-baseline --steps 3 --warmup 1"
export NABO_COARSE_SLACK=${SLACK:-0}
run() { # name env...
  n=$1; shift
  env "$@" python bench.py $B > $O/$n.json 2> $O/$n.err
  python -c "
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
print('$n: kernel_ms %.2f ms_per_step %.2f oracle %s' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['sampled_rows_equal_oracle']))"
}
run product
run exp_nohit NABO_KNN_SO=$PWD/tools/ab/exp.so NABO_DEBUG_ABLATE=1
run blt_base NABO_KNN_SO=$PWD/tools/ab/blt.so
run blt_nohit NABO_KNN_SO=$PWD/tools/ab/blt.so NABO_DEBUG_ABLATE=1
