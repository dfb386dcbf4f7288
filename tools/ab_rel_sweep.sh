#!/bin/bash
# Development aid (GPU box): workgroups per CU x loader lead of the range-proportional pass kernel, per batch size
for b in 1 2 4; do
  for wg in 1 2 3 4; do
    for ld in 2 3; do
      MGM_HIP_TUNE=rel_wg=$wg,rel_ld=$ld timeout 300 python bench.py --workload cfg3r --batch $b --steps 10 --warmup 2 --repeats 0 --no-cpu-baseline --no-parity --extras off 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3r x$b wg $wg ld $ld', round(d['value'],1), round(d['kernel_ms_per_step']['k_pass_rel'],2))"
    done
  done
done
