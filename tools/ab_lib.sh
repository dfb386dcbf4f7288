#!/bin/bash
# Development aid (GPU box): alternate library variants (tools/sweep_build.sh) on one workload.  bash tools/ab_lib.sh "cfg3 1" "base dd3 dd6" [reps]
set -- $1 "$2" "${3:-2}"
w=$1; b=$2; libs=$3; reps=$4
for i in $(seq 1 $reps); do
  for v in $libs; do
    if [ $v = base ]; then unset MGM_HIP_LIB; else export MGM_HIP_LIB=$PWD/mgm_amd/lib/variants/$v/libmgm_hip.so; fi
    timeout 300 python bench.py --workload $w --batch $b --steps 20 --repeats 0 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); a=d['roofline']['avg_launch_ms']
print('$w x$b $v', 'vol/s %.1f'%d['value'], 'K3 %.2f'%a.get('k_pass2',0), 'frac %.3f'%d['roofline']['frac'], flush=True)"
  done
done
