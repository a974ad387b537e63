set -e
O=$PWD/gpurun_out/${1:-r3n}; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B="--no-extras --no-cpu-baseline --metric canberra --steps 1 --warmup 0"
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/pmc/pass$i -- python3 bench.py $B > $O/pass$i.json 2> $O/pass$i.err || echo "pass $i failed"
done
python tools/pmc_summary.py $O/pmc > $O/pmc_canberra_bits_summary.csv
rm -rf $O/pmc
grep "cbb_filter" $O/pmc_canberra_bits_summary.csv
