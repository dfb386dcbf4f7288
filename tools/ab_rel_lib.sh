#!/bin/bash
# Development aid (GPU box): A/B of two builds of the library on the range-proportional workloads, alternating runs.
#   bash tools/ab_rel_lib.sh mgm_amd/lib/variants/head/libmgm_hip.so [tune]      (the other one is the in-tree build)
OTHER=$1; TUNE=${2:-rel=1}
for rep in 1 2; do
  for lib in "$OTHER" ""; do
    for cfg in "cfg3r 1" "cfg3r 2" "cfg3r 4" "cfg3hr 1" "cfg3hr 4"; do
      set -- $cfg
      MGM_HIP_LIB=$lib MGM_HIP_TUNE=$TUNE timeout 300 python bench.py --workload $1 --batch $2 --steps 10 --warmup 2 --repeats 0 --no-cpu-baseline --no-parity --extras off 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-in-tree} | $1 x$2', round(d['value'],1), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
    done
  done
done
