#!/bin/bash
# Runs ON THE GPU BOX (round 5, first call): where a single launch's step goes (development build: in-kernel timers of the
# single-counter schedule) and the same-box baseline of the one-pair legs on the product library.
O=gpurun_out/r05a; mkdir -p $O
MGM_STATS_REAL=1 MGM_HIP_TUNE=xcdq=0 timeout 600 python tools/gpu_stats.py cfg3 cfg2 cfg3h > $O/stats.txt 2>&1
STEPS=20 bash tools/bench_matrix.sh $O/base.jsonl "cfg3:1 cfg3:2 cfg2:1 cfg2:2 cfg1s:2 cfg1s:1 cfg3:12" > $O/base.txt 2>&1
tail -n 30 $O/stats.txt; cat $O/base.txt
