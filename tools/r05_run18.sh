#!/bin/bash
O=gpurun_out/r05_rel; mkdir -p $O
for cfg in "cfg3r 1 1" "cfg3hr 1 1" "cfg3r 4 3"; do
  set -- $cfg
  rm -f /tmp/tl.txt
  MGM_HIP_LIB=$PWD/mgm_amd/lib/variants/relph/libmgm_hip.so MGM_HIP_REL=2 MGM_HIP_TUNE=rel_wg=$3 MGM_HIP_TIMELINE=/tmp/tl.txt MGM_BENCH_PLACE_TRIES=0 timeout 300 python bench.py --workload $1 --batch $2 --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity > /dev/null 2>$O/tl_$1_b$2.err
  echo "== $cfg"; python tools/timeline.py /tmp/tl.txt | sed -n 1,9p; python tools/rel_phases.py /tmp/tl.txt
done
