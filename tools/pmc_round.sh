# rocprofv3 counter passes for the dominant kernel (one bench step each; counters in their own runs, no tracing).
set -e
O=$PWD/gpurun_out/r1d_pmc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp NABO_BENCH_ALT=0
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/pass$i -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pass$i.json 2> $O/pass$i.err
  echo "pass $i done"
done
python tools/pmc_summary.py $O > $O/summary.csv
grep l2_topk $O/summary.csv
