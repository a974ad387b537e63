# round 4: counter passes of the modified-Canberra filter at 1M x 1M (one counter group per pass, no tracing beside it)
O=$PWD/gpurun_out/${TAG:-r4cbpmc}; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B="--no-extras --no-cpu-baseline --metric canberra --steps 1 --warmup 0"
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc/pass$i -- python3 bench.py $B > $O/pass$i.json 2> $O/pass$i.err || echo "pass $i failed"
done
python tools/pmc_summary.py $O/pmc > $O/pmc_canberra_summary.csv
rm -rf $O/pmc
grep "cbb_filter" $O/pmc_canberra_summary.csv
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/%s/pass*.json"%os.environ.get("TAG","r4cbpmc"))):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d["ms_per_step"], d["roofline"]["kernel_ms"])
    except Exception as e: print(f,"ERR",e)
PY
