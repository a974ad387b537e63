# Mapping end to end (HDF5 in, HDF5 out) on the round-3 build: 3k / 100k / 1M cells, reference-style per-cell files and the
# opt-in columnar forms.
set -e
O=$PWD/gpurun_out/${1:-r3map}; mkdir -p $O
PY=/opt/conda/bin/python3.9
$PY tools/bench_mapping.py 3000 3000 30 11 per_cell per_cell per_node | tail -1 | tee -a $O/mapping.jsonl
$PY tools/bench_mapping.py 100000 100000 50 15 per_cell per_cell per_node | tail -1 | tee -a $O/mapping.jsonl
$PY tools/bench_mapping.py 100000 100000 50 15 columnar dense columnar | tail -1 | tee -a $O/mapping.jsonl
$PY tools/bench_mapping.py 1000000 1000000 50 15 columnar dense per_node | tail -1 | tee -a $O/mapping.jsonl
$PY tools/bench_mapping.py 1000000 1000000 50 15 columnar dense columnar | tail -1 | tee -a $O/mapping.jsonl
