# round 4, call 23: what a few leftover columns cost on their own (100k rows = 256 columns x 2 splits + 5 columns)
O=$PWD/gpurun_out/${TAG:-r4c23}; mkdir -p $O
timeout -k 10 300 python tools/sweep_plan.py 1696 100000 50 15 default splits=16 splits=8 splits=4 prepass=0,splits=16 l2c_geo=0,splits=8 > $O/sweep_1696.txt 2>&1; cut -c1-300 $O/sweep_1696.txt
timeout -k 10 300 python tools/sweep_plan.py 98304 100000 50 15 default > $O/sweep_98304.txt 2>&1; cut -c1-300 $O/sweep_98304.txt
timeout -k 10 300 python tools/sweep_plan.py 100000 100000 50 15 default l2c_geo=0 > $O/sweep_100k.txt 2>&1; cut -c1-300 $O/sweep_100k.txt
