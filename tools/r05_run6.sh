#!/bin/bash
O=gpurun_out/r05f; mkdir -p $O
i=0
for cfg in "cfg3 1 order=1,xcdq=2" "cfg3 1 order=1,xcdq_k=2" "cfg2 1 order=1" "cfg2 2 order=1,xcdq=2"; do
  set -- $cfg; i=$((i+1))
  rm -f /tmp/tl.txt
  MGM_HIP_TUNE=$3 MGM_HIP_LIB=$PWD/mgm_amd/lib/variants/tl/libmgm_hip.so MGM_HIP_TIMELINE=/tmp/tl.txt timeout 300 python bench.py --workload $1 --batch $2 --steps 3 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity > /dev/null 2>$O/err$i.txt
  python tools/timeline.py /tmp/tl.txt --csv $O/timeline_$i.csv > $O/timeline_$i.txt 2>&1
  echo "=== $cfg"; cat $O/timeline_$i.txt
done
