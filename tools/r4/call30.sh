# round 4, call 30: Canberra lists compacted after 4 / 8 / 12 / 16 pending entries (fresher thresholds against more sorts)
O=$PWD/gpurun_out/${TAG:-r4c30}; mkdir -p $O
for so in "" tools/ab/pend12.so tools/ab/pend8.so tools/ab/pend4.so ""; do
  tag=$( [ -z "$so" ] && echo product || basename $so .so )
  for shape in "100000 100000" "1000000 1000000"; do
    NABO_KNN_SO=$so timeout -k 10 200 python bench.py --metric canberra --targets ${shape% *} --refs ${shape#* } --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/cb_${tag}_$(echo $shape | tr ' ' 'x').json 2>> $O/err.txt
    python - <<PY
import json
d=json.loads(open("$O/cb_${tag}_$(echo $shape | tr ' ' 'x').json").read().strip().splitlines()[-1]); print("$tag $shape ms/step %.2f kernel %.2f"%(d["ms_per_step"], d["roofline"]["kernel_ms"]), d["sampled_rows_equal_oracle"])
PY
  done
done
