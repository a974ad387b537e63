for v in "$@"; do cp tools/ab/$v.so nabo_amd/libnabo_knn.so; echo "$v $(python tools/check_shard_fullscale.py 8 | grep -o 'per-shard.*')"; done
