# round 2, GPU call B: profiles of the default (f16x3) Euclidean kernel and of the modified-Canberra filter,
# ablations, the vector issue-rate microbenchmark.  Run from the repo root on the GPU box.  TAG names the output set.
set -e
TAG=${TAG:-r2b}
O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B="--no-extras --no-cpu-baseline"
tools/issue_lab.bin > $O/issue_lab.txt 2>&1 || true
tools/mfma_clock_lab.bin > $O/mfma_clock_lab.txt 2>&1 || true
echo "issue lab done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_euclid -- python3 bench.py $B --steps 3 --warmup 1 > $O/bench_under_rocprof.json 2> $O/rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_canberra -- python3 bench.py $B --metric canberra --steps 2 --warmup 1 > $O/bench_canberra_under_rocprof.json 2>> $O/rocprof.err
echo "kernel traces done"
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_euclid/pass$i -- python3 bench.py $B --steps 1 --warmup 0 > $O/pmc_euclid_pass$i.json 2> $O/pmc_euclid_pass$i.err
  echo "euclid pmc pass $i done"
done
python tools/pmc_summary.py $O/pmc_euclid > $O/pmc_euclid_summary.csv
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_canberra/pass$i -- python3 bench.py $B --metric canberra --steps 1 --warmup 0 > $O/pmc_canberra_pass$i.json 2> $O/pmc_canberra_pass$i.err || echo "canberra pmc pass $i failed"
  echo "canberra pmc pass $i done"
done
python tools/pmc_summary.py $O/pmc_canberra > $O/pmc_canberra_summary.csv
echo "pmc done"
for ab in 1 2 3; do
  NABO_DEBUG_ABLATE=$ab python bench.py $B --steps 3 --warmup 1 2>> $O/ablate.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate $ab kernel_ms', d['roofline']['kernel_ms'], 'ms_per_step', d['ms_per_step'])" >> $O/ablate.txt
done
NABO_L2_MODE=f32 python bench.py $B --steps 3 --warmup 1 > $O/bench_f32_mode.json 2>> $O/ablate.err
NABO_L2_MODE=f16x3h python bench.py $B --steps 3 --warmup 1 > $O/bench_f16x3h_mode.json 2>> $O/ablate.err
NABO_L2_MODE=f16x3s python bench.py $B --steps 3 --warmup 1 > $O/bench_f16x3s_mode.json 2>> $O/ablate.err
python bench.py $B --targets 100000 --refs 100000 --steps 10 --warmup 2 > $O/bench_100kx100k.json 2>> $O/ablate.err
python bench.py $B --metric cosine --dims 100 --neighbors 50 --steps 2 --warmup 1 > $O/bench_cosine_1M_d100_k50.json 2>> $O/ablate.err
for n in 8 4 2; do python tools/check_shard_fullscale.py $n | tail -1 >> $O/shard_fullscale.txt; done
cat $O/ablate.txt $O/shard_fullscale.txt
grep -h "l2q_topk\|cbf_filter" $O/pmc_euclid_summary.csv $O/pmc_canberra_summary.csv
cat $O/issue_lab.txt $O/mfma_clock_lab.txt
