for v in "$@"; do export NABO_KNN_SO=$PWD/tools/ab/$v.so; echo "$v $(python tools/check_shard_fullscale.py 8 | grep -o 'per-shard.*')"; done
