# round 4: rocprofv3 kernel traces + counter passes of the default Euclidean kernel and of the modified-Canberra filter (counters in
# their own runs, no tracing beside them).  Run from the repo root on the GPU box.
TAG=${TAG:-r4pmc}
O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B="--no-extras --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_euclid -- python3 bench.py $B --steps 3 --warmup 1 > $O/bench_under_rocprof.json 2> $O/rocprof.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_canberra -- python3 bench.py $B --metric canberra --steps 2 --warmup 1 > $O/bench_canberra_under_rocprof.json 2>> $O/rocprof.err
echo "kernel traces done"
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_euclid/pass$i -- python3 bench.py $B --steps 1 --warmup 0 > $O/pmc_euclid_pass$i.json 2> $O/pmc_euclid_pass$i.err || echo "euclid pmc pass $i failed"
done
python tools/pmc_summary.py $O/pmc_euclid > $O/pmc_euclid_summary.csv
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_canberra/pass$i -- python3 bench.py $B --metric canberra --steps 1 --warmup 0 > $O/pmc_canberra_pass$i.json 2> $O/pmc_canberra_pass$i.err || echo "canberra pmc pass $i failed"
done
python tools/pmc_summary.py $O/pmc_canberra > $O/pmc_canberra_summary.csv
echo "pmc done"
find $O/prof_euclid -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/euclid_kernel_stats.csv
find $O/prof_canberra -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/canberra_kernel_stats.csv
rm -rf $O/prof_euclid $O/prof_canberra $O/pmc_euclid $O/pmc_canberra
grep -h "l2c_topk\|cbb_filter" $O/pmc_euclid_summary.csv $O/pmc_canberra_summary.csv | head -50
head -6 $O/euclid_kernel_stats.csv | cut -c1-160
