# round 4, call 22: cosine d=100 k=50 -- the seeded pass's 4 x 64 entries merged to the 128 best (merge_lists=2) or all re-evaluated
O=$PWD/gpurun_out/${TAG:-r4c22}; mkdir -p $O
timeout -k 10 600 python tools/sweep_plan.py 1000000 1000000 100 50 2 default merge_lists=2 > $O/sweep_cosine.txt 2>&1; cut -c1-420 $O/sweep_cosine.txt
timeout -k 10 300 python tools/sweep_plan.py 200000 300000 30 40 0 default merge_lists=2 > $O/sweep_k40.txt 2>&1; cut -c1-420 $O/sweep_k40.txt
