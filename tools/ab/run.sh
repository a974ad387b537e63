# Same-box A/B of kernel builds: put one libnabo_knn.so per variant into tools/ab/<name>.so (built .so files travel with
# the gpurun snapshot, git ignores them), then on the GPU box:   bash tools/ab/run.sh base variant [variant ...]
# Box-to-box spread of the 1M x 1M kernel is ~1 % (783-792 ms): differences smaller than that need the same box.
for rep in 1 2; do for v in "$@"; do
  export NABO_KNN_SO=$PWD/tools/ab/$v.so      # the product .so is never overwritten (nabo_amd/_lib.py)
  timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > gpurun_out/ab_$v.log 2>&1 || exit 1
  echo "$v $(tail -1 gpurun_out/ab_$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["phases_ms"]["ms_fallback"])')"
done; done
