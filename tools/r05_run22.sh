#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_rel.py -x -q 2>&1 | tail -2
run() {  # workload batch tune
  MGM_HIP_REL=2 MGM_HIP_TUNE=$3 timeout 300 python bench.py --workload $1 --batch $2 --steps 5 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 x$2 $3', round(d['value'],1), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items() if k in ('k_pass_rel','k_pass2')})"
}
for w in cfg3r cfg3hr; do
  for t in rel_wg=1,rel_ld=2,rel_lead=1,rel_pubq=4 rel_wg=1,rel_ld=2,rel_lead=1,rel_pubq=2 rel_wg=2,rel_ld=2,rel_lead=1,rel_pubq=4 rel_wg=2,rel_ld=2,rel_lead=1,rel_pubq=1; do run $w 1 $t; done
  for t in rel_wg=2,rel_ld=2,rel_lead=1,rel_pubq=4 rel_wg=2,rel_ld=3,rel_lead=2,rel_pubq=4 rel_wg=2,rel_ld=5,rel_lead=8,rel_pubq=4; do run $w 2 $t; done
  for t in rel_wg=3,rel_ld=5,rel_lead=8,rel_pubq=4 rel_wg=3,rel_ld=4,rel_lead=4,rel_pubq=4 rel_wg=3,rel_ld=3,rel_lead=4,rel_pubq=4; do run $w 4 $t; done
done
