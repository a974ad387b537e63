for rep in 1 2; do for v in "$@"; do
  export NABO_KNN_SO=$PWD/tools/ab/$v.so      # the product .so is never overwritten (nabo_amd/_lib.py)
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$v.log 2>&1 || exit 1
  echo "$v $(tail -1 gpurun_out/ab_$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["phases_ms"]["ms_fallback"], (d.get("alt_f16x3") or {}).get("kernel_ms"), (d.get("alt_f16x3") or {}).get("same_bits_as_f32_path"))')"
done; done
