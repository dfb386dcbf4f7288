#!/bin/bash
# Development aid (GPU box): builds the -DMGM_REL_PHASES=1 variant of the range-proportional pass kernel IN PLACE (hipcc is on the box; the
# snapshot is scratch), runs cfg3r / cfg3hr once per tune and prints the phase clocks and the timeline summary per launch.
MGM_REL_DEFINES="-DMGM_REL_PHASES=1" python mgm_amd/build.py > /dev/null 2>&1 || exit 1
for tune in "${@:-rel_multi=1}"; do
  for cfg in "cfg3r 1" "cfg3r 4"; do
    set -- $cfg
    rm -f /tmp/tl.txt
    MGM_HIP_TUNE=$tune MGM_HIP_TIMELINE=/tmp/tl.txt timeout 300 python bench.py --workload $1 --batch $2 --steps 1 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity --extras off > /dev/null 2>&1
    echo "== $tune | $1 x$2"
    python tools/timeline.py /tmp/tl.txt 2>/dev/null | head -12
    python tools/rel_phases.py /tmp/tl.txt
  done
done
