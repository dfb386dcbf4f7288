#!/bin/bash
# Development aid (GPU box): A/B of the range-proportional pass kernel's switches (MGM_HIP_TUNE), one line per (tune, workload, batch)
for tune in "${@:-rel_multi=1}"; do
  for cfg in "cfg3r 1" "cfg3r 4" "cfg3hr 1"; do
    set -- $cfg
    MGM_HIP_TUNE=$tune timeout 300 python bench.py --workload $1 --batch $2 --steps 10 --warmup 2 --repeats 0 --no-cpu-baseline --no-parity --extras off 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tune | $1 x$2', round(d['value'],1), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
  done
done
