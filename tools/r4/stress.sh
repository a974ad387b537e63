# round 4: randomised parity sweeps on the final build (shapes, metrics, masks, ties, scales, shards, sequences; the pass chain)
O=$PWD/gpurun_out/${TAG:-r4stress}; mkdir -p $O
( timeout -k 10 500 python tools/stress_sweep.py 4000 401 2>&1 | tail -3
  timeout -k 10 400 python tools/stress_sweep2.py 2500 402 2>&1 | tail -3
  timeout -k 10 300 python tools/stress_large.py 2>&1 | tail -3
  timeout -k 10 500 python tools/stress_sweep3.py 3000 403 2>&1 | tail -3 ) > $O/stress_sweeps.txt 2>&1
cat $O/stress_sweeps.txt | cut -c1-300
