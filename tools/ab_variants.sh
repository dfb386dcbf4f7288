# Development aid: time bench.py workloads on library variants built by tools/sweep_build.sh (run on the GPU box).
# usage: bash tools/ab_variants.sh "base lean ..." ["cfg3 --batch 12" "cfg4" ...]
variants=${1:-base}; shift
[ $# -eq 0 ] && set -- "cfg3 --batch 12" "cfg3 --batch 1"
for rep in 1 2; do
for v in $variants; do
  if [ $v = base ]; then unset MGM_HIP_LIB; else export MGM_HIP_LIB=$PWD/mgm_amd/lib/variants/$v/libmgm_hip.so; fi
  for w in "$@"; do
    timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --repeats 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$v','$w',round(d['value'],1),round(d['roofline']['frac'],3),d.get('parity',{}).get('status'))"
  done
done
done
