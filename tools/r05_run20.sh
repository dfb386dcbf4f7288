#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_rel.py -x -q 2>&1 | tail -2
for lib in default ld2 lead1 ld5old; do
for cfg in "cfg3r 1 1" "cfg3r 4 3" "cfg3hr 1 1" "cfg3hr 4 3"; do
  set -- $cfg
  L=$PWD/mgm_amd/lib/variants/$lib/libmgm_hip.so; [ $lib = default ] && L=$PWD/mgm_amd/lib/libmgm_hip.so
  MGM_HIP_LIB=$L MGM_HIP_REL=2 MGM_HIP_TUNE=rel_wg=$3 timeout 300 python bench.py --workload $1 --batch $2 --steps 5 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib $1 x$2 wg=$3', round(d['value'],1), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
done; done
