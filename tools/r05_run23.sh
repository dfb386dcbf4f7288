#!/bin/bash
# the rewritten range-proportional kernels: every test that reaches them, their phase clocks, the one-pair timelines
timeout 1500 python -m pytest tests/test_gpu_rel.py tests/test_gpu_cli.py tests/test_gpu_windowed.py tests/test_gpu_placement.py -x -q 2>&1 | tail -3
O=gpurun_out/r05_rel; mkdir -p $O
{
for cfg in "cfg3r 1" "cfg3hr 1" "cfg3r 4" "cfg3hr 4"; do
  set -- $cfg
  rm -f /tmp/tl.txt
  MGM_HIP_LIB=$PWD/mgm_amd/lib/variants/relph/libmgm_hip.so MGM_HIP_REL=2 MGM_HIP_TIMELINE=/tmp/tl.txt MGM_BENCH_PLACE_TRIES=0 timeout 300 python bench.py --workload $1 --batch $2 --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity > /dev/null 2>$O/tl_$1_b$2.err
  echo "== $1 x $2 (development build with phase clocks)"; python tools/timeline.py /tmp/tl.txt | sed -n 1,9p; python tools/rel_phases.py /tmp/tl.txt
done
} > $O/rel_phases.txt 2>&1
for cfg in "cfg3r 1" "cfg3hr 1" "cfg3r 2" "cfg3r 4"; do
  set -- $cfg
  rm -f /tmp/tl.txt
  MGM_HIP_REL=2 MGM_HIP_TIMELINE=/tmp/tl.txt MGM_BENCH_PLACE_TRIES=0 timeout 300 python bench.py --workload $1 --batch $2 --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity > /dev/null 2>$O/tl_$1_b$2.err
  python tools/timeline.py /tmp/tl.txt > $O/timeline_$1_b$2.txt 2>&1
done
cat $O/rel_phases.txt | grep -v "no item\|CU-time\|items per"
for cfg in "cfg3r 1" "cfg3r 2" "cfg3r 4" "cfg3hr 1" "cfg3hr 2" "cfg3hr 4"; do
  set -- $cfg
  timeout 300 python bench.py --workload $1 --batch $2 --steps 10 --warmup 2 --repeats 0 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 x$2', round(d['value'],1), round(d['roofline']['frac'],3), d['roofline'].get('frac_dense_hull_equivalent'), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
done
bash tools/ragged_cli.sh 2>&1 | tail -14
