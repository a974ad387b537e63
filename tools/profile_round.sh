# Final measurements of a round on one MI355X (run through gpurun from the repo root); outputs under gpurun_out/r1d/.
set -e
O=$PWD/gpurun_out/r1d; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_1Mx1M.json 2> $O/bench.err
echo "bench done" 
export TMPDIR=/tmp
NABO_BENCH_ALT=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
echo "rocprof done"
python bench.py --targets 100000 --refs 100000 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_100kx100k.json 2>> $O/bench.err
python bench.py --metric cosine --dims 100 --neighbors 50 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cosine_1M_d100_k50.json 2>> $O/bench.err
for n in 8 4 2; do python tools/check_shard_fullscale.py $n | tail -1 >> $O/shard_fullscale.txt; done
echo "shards done"
if [ -n "$NABO_PROFILE_MAPPING" ]; then for cfg in "100000 100000 50 15 per_cell" "100000 100000 50 15 columnar dense" "1000000 200000 50 15 columnar dense"; do /opt/conda/bin/python3.9 tools/bench_mapping.py $cfg | tail -1 >> $O/mapping_end_to_end.jsonl; echo "mapping $cfg done"; done; fi
