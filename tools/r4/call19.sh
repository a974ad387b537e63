# round 4, call 19: cosine d=100 k=50 -- how much list slack the one-product pass should keep now that the seeded pass is cheap
O=$PWD/gpurun_out/${TAG:-r4c19}; mkdir -p $O
timeout -k 10 120 python -m pytest tests -m gpu -x -q -k "row_pass_record" > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
timeout -k 10 600 python tools/sweep_plan.py 1000000 1000000 100 50 2 default coarse_slack=0 coarse_slack=2 coarse_slack=4 prepass=50 prepass=200 > $O/sweep_cosine.txt 2>&1; cut -c1-330 $O/sweep_cosine.txt
