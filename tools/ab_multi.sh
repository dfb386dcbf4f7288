#!/bin/bash
# Development aid (GPU box): alternate several environment settings on one workload.
#   bash tools/ab_multi.sh "cfg3 1" "MGM_HIP_XCDQ=0" "MGM_HIP_XCDQ=1 MGM_HIP_XCDQ_K=4" ... [REPS=2]
set -- $1 "${@:2}"
w=$1; b=$2; shift 2
for i in $(seq 1 ${REPS:-2}); do
  for setting in "$@"; do
    env $setting timeout 300 python bench.py --workload $w --batch $b --steps 20 --repeats 0 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); a=d['roofline']['avg_launch_ms']
print('$w x$b [$setting]', 'vol/s %.1f'%d['value'], 'K3 %.2f'%a.get('k_pass2',0), 'frac %.3f'%d['roofline']['frac'], flush=True)"
  done
done
