# round 3 final GPU call: parity suite, the driver's bench line, the 8-rank loopback rehearsal, kernel traces + counter passes.
O=$PWD/gpurun_out/${TAG:-r3final}; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
NABO_BENCH_LOOPBACK=8 NABO_BENCH_CHECK=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_loopback8.json 2> $O/bench_loopback8.err; echo "loopback rc=$?"
python bench.py --targets 100000 --refs 100000 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_100k.json 2>> $O/bench_default.err
python bench.py --metric cosine --dims 100 --neighbors 50 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_cosine.json 2>> $O/bench_default.err
for n in 8 4 2; do python tools/bench_shard.py $n 2>> $O/bench_default.err | tail -1 >> $O/shard_share.txt; done
TAG=${TAG:-r3final}_pmc bash tools/r3_pmc.sh > $O/pmc.log 2>&1; echo "pmc rc=$?"
/opt/conda/bin/python3.9 tools/bench_mapping.py 1000000 1000000 50 15 columnar dense columnar | tail -1 > $O/mapping_1M_columnar.json
tail -3 $O/pmc.log; cat $O/shard_share.txt | cut -c1-160
