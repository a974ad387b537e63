# A/B of the one-product pass (profiles/r3_coarse_experiments.txt): library variants built on the CPU side with
#   python -m nabo_amd._build --out tools/ab/<name>.so -DNABO_EXPERIMENTS [-DNABO_L2C_BUILTIN | -DNABO_L2C_ABL=1|2|3 |
#                                                                          -DNABO_L2H_NOFILTER | -DNABO_L2H_NORELOAD | -DNABO_LISTS_PROF]
# then on the GPU box:   bash tools/r3_coarse_ab.sh TAG name[:ENV=VAL,...] ...     ("product" = nabo_amd/libnabo_knn.so)
#   e.g.  bash tools/r3_coarse_ab.sh ab exp exp:NABO_DEBUG_ABLATE=1 exp:NABO_L2C_GEO=a exp:NABO_COARSE_KERNEL_Q=1 exp:NABO_L2_MODE=f16x3
TAG=$1; shift
O=$PWD/gpurun_out/$TAG; mkdir -p $O
B="--no-extras --no-cpu-baseline --steps 3 --warmup 1"
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  so=""; [ "$v" != product ] && so="NABO_KNN_SO=$PWD/tools/ab/$v.so"
  tag=$(echo "$spec" | tr ':=,' '___')
  env $so $envs python bench.py $B > $O/$tag.json 2> $O/$tag.err || { echo "$spec FAILED"; tail -3 $O/$tag.err; continue; }
  python -c "
import json
d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1])
print('$spec: kernel_ms %.2f ms_per_step %.2f oracle %s rows_by_pass %s %s' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['sampled_rows_equal_oracle'], d.get('rows_by_pass'), d['roofline']['kernel'][:32]))"
  grep "lists prof" $O/$tag.err | tail -1
done
