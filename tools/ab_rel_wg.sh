#!/bin/bash
# Development aid (GPU box): workgroups of k_pass_rel per CU and steps of DMA in flight, per batch size
run() { MGM_HIP_TUNE=$3 timeout 300 python bench.py --workload $1 --batch $2 --steps 8 --warmup 2 --repeats 0 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 x$2 $3', round(d['value'],1), round(d['kernel_ms_per_step']['k_pass_rel'],2))"; }
for w in cfg3r cfg3hr; do
  for t in rel_wg=1 rel_wg=2 rel_wg=3; do run $w 1 $t; done
  for t in rel_wg=2 rel_wg=3 rel_wg=4; do run $w 2 $t; done
  for t in rel_wg=3 rel_wg=4 rel_wg=5 rel_wg=4,rel_ld=2; do run $w 4 $t; done
done
