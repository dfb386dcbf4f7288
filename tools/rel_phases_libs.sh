#!/bin/bash
# Development aid (GPU box): phase clocks + timeline summary of prebuilt -DMGM_REL_PHASES=1 variants (tools/sweep_build.sh style):
#   bash tools/rel_phases_libs.sh "cfg3r 4" mgm_amd/lib/variants/relph_head/libmgm_hip.so mgm_amd/lib/variants/relph_new/libmgm_hip.so
CFG=$1; shift
for lib in "$@"; do
  set -- $CFG
  rm -f /tmp/tl.txt
  MGM_HIP_LIB=$lib MGM_HIP_TIMELINE=/tmp/tl.txt timeout 300 python bench.py --workload $1 --batch $2 --steps 1 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity --extras off > /dev/null 2>&1
  echo "== $lib | $1 x$2"
  python tools/timeline.py /tmp/tl.txt 2>/dev/null | head -20
  python tools/rel_phases.py /tmp/tl.txt
done
