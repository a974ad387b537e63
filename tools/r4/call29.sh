# round 4, call 29: where the seeded bitmap count overtakes the SWAR count; parity of the Canberra paths
O=$PWD/gpurun_out/${TAG:-r4c29}; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "canberra or Canberra or mapping or golden or sweep or sharded" > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for shape in "10000 10000" "30000 30000" "100000 30000" "30000 45000" "3000 3000"; do
  for mode in swar bits; do
    NABO_CANBERRA_MODE=$mode timeout -k 10 200 python bench.py --metric canberra --targets ${shape% *} --refs ${shape#* } --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/cb_${mode}_$(echo $shape | tr ' ' 'x').json 2>> $O/err.txt
    python - <<PY
import json
d=json.loads(open("$O/cb_${mode}_$(echo $shape | tr ' ' 'x').json").read().strip().splitlines()[-1]); print("$shape $mode ms/step %.2f total %.2f"%(d["ms_per_step"], d["phases_ms"]["ms_total"]), d["sampled_rows_equal_oracle"])
PY
  done
done
