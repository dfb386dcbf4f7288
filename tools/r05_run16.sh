#!/bin/bash
# four lines per wave in k_pass_rel: parity, then workgroups per CU
timeout 900 python -m pytest tests/test_gpu_rel.py -x -q 2>&1 | tail -3
for cfg in "cfg3r 1 1" "cfg3r 1 2" "cfg3r 1 3" "cfg3r 1 4" "cfg3r 4 2" "cfg3r 4 3" "cfg3r 4 4" "cfg3r 4 6" "cfg3hr 1 1" "cfg3hr 1 2" "cfg3hr 1 3" "cfg3hr 4 2" "cfg3hr 4 3" "cfg3hr 4 4"; do
  set -- $cfg
  MGM_HIP_REL=2 MGM_HIP_TUNE=rel_wg=$3 timeout 300 python bench.py --workload $1 --batch $2 --steps 5 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 x$2 wg=$3', round(d['value'],1), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
done
