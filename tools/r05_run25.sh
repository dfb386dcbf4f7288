#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_rel.py tests/test_gpu_cli.py -x -q 2>&1 | tail -n 3
MGM_FUZZ_N=3000 MGM_FUZZ_BASE=50000 timeout 1000 python -m pytest tests/test_gpu_rel.py -q -k random 2>&1 | tail -n 2
run() {  # workload batch tune
  MGM_HIP_TUNE=$3 timeout 300 python bench.py --workload $1 --batch $2 --steps 8 --warmup 2 --repeats 0 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 x$2 $3', round(d['value'],1), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items() if k in ('k_pass_rel','k_pass2')})"
}
for w in cfg3r cfg3hr; do
  for t in rel_wg=1,rel_ld=2 rel_wg=2,rel_ld=2 rel_wg=2,rel_ld=3 rel_wg=3,rel_ld=2; do run $w 1 $t; done
  for t in rel_wg=2,rel_ld=2 rel_wg=2,rel_ld=3 rel_wg=3,rel_ld=3; do run $w 2 $t; done
  for t in rel_wg=3,rel_ld=2 rel_wg=3,rel_ld=3 rel_wg=3,rel_ld=4 rel_wg=4,rel_ld=3; do run $w 4 $t; done
done
rm -f /tmp/tl.txt
MGM_HIP_TIMELINE=/tmp/tl.txt MGM_BENCH_PLACE_TRIES=0 timeout 300 python bench.py --workload cfg3r --batch 1 --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity > /dev/null 2>&1
python tools/timeline.py /tmp/tl.txt | sed -n 1,22p
