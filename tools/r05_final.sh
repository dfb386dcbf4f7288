#!/bin/bash
# end-of-round evidence on the final sources: the whole GPU suite, bench matrix + rocprof stats + HBM counters
# (collect_profiles.sh), timelines of the one-pair launches (dense: tl variant; range-proportional: built in), phase clocks, smoke
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 3 > gpurun_out/final_pytest_gpu.log; cat gpurun_out/final_pytest_gpu.log
SKIP_SWEEP=1 bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1
tail -75 gpurun_out/collect.log
O=gpurun_out/r05_final; mkdir -p $O
for wb in "cfg3 1" "cfg3 2" "cfg2 1" "cfg2 2" "cfg3h 1" "cfg1s 2"; do
  set -- $wb
  rm -f /tmp/tl.txt
  MGM_HIP_LIB=$PWD/mgm_amd/lib/variants/tl/libmgm_hip.so MGM_HIP_TIMELINE=/tmp/tl.txt MGM_BENCH_PLACE_TRIES=0 timeout 300 python bench.py --workload $1 --batch $2 --steps 3 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity > /dev/null 2>$O/tl_$1_b$2.err
  python tools/timeline.py /tmp/tl.txt > $O/timeline_$1_b$2.txt 2>&1
done
for wb in "cfg3r 1" "cfg3r 2" "cfg3r 4" "cfg3hr 1"; do
  set -- $wb
  rm -f /tmp/tl.txt
  MGM_HIP_TIMELINE=/tmp/tl.txt MGM_BENCH_PLACE_TRIES=0 timeout 300 python bench.py --workload $1 --batch $2 --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity > /dev/null 2>$O/tl_$1_b$2.err
  python tools/timeline.py /tmp/tl.txt > $O/rel_timeline_$1_b$2.txt 2>&1
done
{
for cfg in "cfg3r 1" "cfg3hr 1" "cfg3r 4" "cfg3hr 4"; do
  set -- $cfg
  rm -f /tmp/tl.txt
  MGM_HIP_LIB=$PWD/mgm_amd/lib/variants/relph/libmgm_hip.so MGM_HIP_TIMELINE=/tmp/tl.txt MGM_BENCH_PLACE_TRIES=0 timeout 300 python bench.py --workload $1 --batch $2 --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity > /dev/null 2>$O/ph_$1_b$2.err
  echo "== $1 x $2 (development build with phase clocks)"; python tools/timeline.py /tmp/tl.txt | sed -n 1,9p; python tools/rel_phases.py /tmp/tl.txt
done
} > $O/rel_phases.txt 2>&1
head -12 $O/timeline_cfg3_b1.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
MGM_FUZZ_N=400 MGM_FUZZ_BASE=9000 timeout 1500 python -m pytest tests/test_gpu_cli.py -q -k "fuzz or random" 2>&1 | tail -n 3 > $O/fuzz_cli.log; cat $O/fuzz_cli.log
