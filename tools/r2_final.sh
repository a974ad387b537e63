# round 2, final GPU call: parity suite, smoke, the default bench line, then the profile pass (tools/r2_callB.sh)
set -e
TAG=${TAG:-r2g}
mkdir -p gpurun_out
timeout -k 10 850 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1 || { tail -30 gpurun_out/${TAG}_pytest.log; exit 1; }
tail -2 gpurun_out/${TAG}_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
tail -c 600 gpurun_out/${TAG}_bench_default.json
TAG=$TAG bash tools/r2_callB.sh > gpurun_out/${TAG}_callB.log 2>&1
cp gpurun_out/${TAG}_bench_default.json gpurun_out/$TAG/bench_default.json
tail -5 gpurun_out/${TAG}_callB.log
