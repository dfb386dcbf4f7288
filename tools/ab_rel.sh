#!/bin/bash
# Development aid (GPU box): the range-proportional workloads, one line each (A/B of a kernel change: run before and after)
timeout 900 python -m pytest tests/test_gpu_rel.py -x -q 2>&1 | tail -n 1
for cfg in "cfg3r 1" "cfg3r 2" "cfg3r 4" "cfg3hr 1" "cfg3hr 4"; do
  set -- $cfg
  timeout 300 python bench.py --workload $1 --batch $2 --steps 10 --warmup 2 --repeats 0 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 x$2', round(d['value'],1), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
done
