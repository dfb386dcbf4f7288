#!/bin/bash
# Development aid (GPU box): timeline of one k_pass_rel launch per workload ("cfg3r 1" ...), summary + the chain of one pass
for cfg in "${@:-cfg3r 1}"; do set -- $cfg; rm -f /tmp/tl.txt
MGM_HIP_TIMELINE=/tmp/tl.txt timeout 300 python bench.py --workload $1 --batch $2 --steps 1 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity --extras off > /dev/null 2>&1
echo "== $1 x$2"; python tools/timeline.py /tmp/tl.txt 2>/dev/null | head -22
for p in 3 0 5; do echo "-- pass $p"; python tools/rel_chain.py /tmp/tl.txt $p 6; done; done
