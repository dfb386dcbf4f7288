#!/bin/bash
# Development aid (runs on the GPU box): alternate two settings of a library switch on one workload, N times each.
#   bash tools/ab_env.sh "cfg2 16" MGM_HIP_DEEP 0 1 [reps]
set -- $1 "$2" "$3" "$4" "${5:-4}"
w=$1; b=$2; var=$3; A=$4; B=$5; reps=$6
for i in $(seq 1 $reps); do
  for val in $A $B; do
    env $var=$val timeout 300 python bench.py --workload $w --batch $b --steps 20 --repeats 0 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); a=d['roofline']['avg_launch_ms']
print('$w x$b $var=$val', 'vol/s %.1f'%d['value'], 'K3 %.2f'%a.get('k_pass2',0), 'frac %.3f'%d['roofline']['frac'], flush=True)"
  done
done
