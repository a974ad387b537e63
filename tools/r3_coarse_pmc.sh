# counters of the one-product kernel, with and without hits (experiment build)
O=$PWD/gpurun_out/r3coarse_pmc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
export NABO_KNN_SO=$PWD/tools/ab/exp.so NABO_COARSE_SLACK=${SLACK:-0}
B="--no-extras --no-cpu-baseline --steps 1 --warmup 0"
for ab in 0 1; do
  if [ $ab = 1 ]; then export NABO_DEBUG_ABLATE=1; fi
  i=0
  for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD"; do
    i=$((i+1))
    rocprofv3 --pmc $c --output-format csv -d $O/ab$ab/pass$i -- python3 bench.py $B > $O/ab${ab}_pass$i.json 2> $O/ab${ab}_pass$i.err || echo "pass $i failed"
  done
  python tools/pmc_summary.py $O/ab$ab > $O/ab${ab}_summary.csv
  grep "l2q_topk" $O/ab${ab}_summary.csv
  python -c "
import json
d=json.loads(open('$O/ab${ab}_pass1.json').read().strip().splitlines()[-1]); print('ablate $ab kernel_ms', d['roofline']['kernel_ms'])"
done
rm -rf $O/ab0 $O/ab1
