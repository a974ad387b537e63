set -e
mkdir -p gpurun_out
timeout -k 5 120 python tools/r2_smoke.py 2>&1 | tee gpurun_out/smoke_${TAG:-r2c}.txt
bash tools/r2_callA.sh
TAG=${TAG:-r2c} bash tools/r2_callB.sh
