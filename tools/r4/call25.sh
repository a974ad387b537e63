# round 4, call 26: bitmaps from 98k references on, their sample rows gathered on the device
O=$PWD/gpurun_out/${TAG:-r4c26}; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "canberra or Canberra or mapping or golden" > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for shape in "100000 100000" "100000 200000" "1000000 1000000" "30000 100000"; do
    timeout -k 10 200 python bench.py --metric canberra --targets ${shape% *} --refs ${shape#* } --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/cb_$(echo $shape | tr ' ' 'x').json 2>> $O/err.txt
    python - <<PY
import json
d=json.loads(open("$O/cb_$(echo $shape | tr ' ' 'x').json").read().strip().splitlines()[-1]); print("$shape ms/step %.2f"%d["ms_per_step"], {k:round(v,2) for k,v in d["phases_ms"].items()}, d["roofline"]["kernel"][:20], d["sampled_rows_equal_oracle"])
PY
done
