#!/bin/bash
# end-of-round evidence on the final sources: bench matrix + rocprof stats + HBM counters (collect_profiles.sh), timelines, smoke
SKIP_SWEEP=1 bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1
tail -75 gpurun_out/collect.log
O=gpurun_out/r05_final; mkdir -p $O
for wb in "cfg3 1" "cfg3 2" "cfg2 1" "cfg2 2" "cfg3h 1" "cfg1s 2"; do
  set -- $wb
  rm -f /tmp/tl.txt
  MGM_HIP_LIB=$PWD/mgm_amd/lib/variants/tl/libmgm_hip.so MGM_HIP_TIMELINE=/tmp/tl.txt MGM_BENCH_PLACE_TRIES=0 timeout 300 python bench.py --workload $1 --batch $2 --steps 3 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity > /dev/null 2>$O/tl_$1_b$2.err
  python tools/timeline.py /tmp/tl.txt > $O/timeline_$1_b$2.txt 2>&1
done
head -12 $O/timeline_cfg3_b1.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
