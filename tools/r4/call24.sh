# round 4, call 24: where the bit-sliced Canberra count overtakes the SWAR count now (the size rule is n >= 262 144 references)
O=$PWD/gpurun_out/${TAG:-r4c24}; mkdir -p $O
for shape in "30000 30000" "100000 60000" "100000 100000" "100000 200000" "30000 200000" "300000 100000"; do
  for mode in swar bits; do
    NABO_CANBERRA_MODE=$mode timeout -k 10 200 python bench.py --metric canberra --targets ${shape% *} --refs ${shape#* } --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/cb_${mode}_$(echo $shape | tr ' ' 'x').json 2>> $O/err.txt
    python - <<PY
import json
d=json.loads(open("$O/cb_${mode}_$(echo $shape | tr ' ' 'x').json").read().strip().splitlines()[-1]); print("$shape $mode ms/step %.2f"%d["ms_per_step"], {k:round(v,2) for k,v in d["phases_ms"].items()}, d["sampled_rows_equal_oracle"])
PY
  done
done
