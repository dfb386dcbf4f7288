#!/bin/bash
# Development aid (GPU box): timeline of one k_pass_rel launch per workload ("cfg3r 1" ...), summary + the chains of some passes
#   bash tools/rel_timeline_run.sh "cfg3r 1" [passes, default "3 0 5"] [every]
CFG=${1:-cfg3r 1}; PASSES=${2:-3 0 5}; EVERY=${3:-6}
set -- $CFG; rm -f /tmp/tl.txt
MGM_HIP_TIMELINE=/tmp/tl.txt timeout 300 python bench.py --workload $1 --batch $2 --steps 1 --warmup 0 --repeats 0 --no-cpu-baseline --no-parity --extras off > /dev/null 2>&1
echo "== $1 x$2"; python tools/timeline.py /tmp/tl.txt 2>/dev/null | tail -n +1 | head -22
for p in $PASSES; do echo "-- pass $p"; python tools/rel_chain.py /tmp/tl.txt $p $EVERY | awk 'NR==1 || NR%2==0'; done
