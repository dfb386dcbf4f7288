#!/bin/bash
# end-of-round evidence on the final sources (GPU box): the whole GPU suite, smoke(), bench matrix + rocprof stats + HBM counters
# (collect_profiles.sh), timelines of the range-proportional launches, a random campaign of the range-proportional kernels against the
# dense hull, and LAST (it rebuilds the library in the box's scratch copy) the phase clocks of k_pass_rel.
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 3 > gpurun_out/final_pytest_gpu.log; cat gpurun_out/final_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/final_smoke.log
SKIP_SWEEP=1 bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1
tail -90 gpurun_out/collect.log
O=gpurun_out/r06_final; mkdir -p $O
for wb in "cfg3r 1" "cfg3r 2" "cfg3r 4" "cfg3hr 1" "cfg3r50 1"; do
  set -- $wb
  rm -f /tmp/tl.txt
  MGM_HIP_TIMELINE=/tmp/tl.txt MGM_BENCH_PLACE_TRIES=0 timeout 300 python bench.py --workload $1 --batch $2 --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity > /dev/null 2>$O/tl_$1_b$2.err
  python tools/timeline.py /tmp/tl.txt > $O/rel_timeline_$1_b$2.txt 2>&1
done
MGM_FUZZ_N=1500 MGM_FUZZ_BASE=20000 timeout 1500 python -m pytest tests/test_gpu_rel.py -q -k "random" 2>&1 | tail -n 3 > $O/fuzz_rel.log; cat $O/fuzz_rel.log
# ... and both ragged paths against the ragged ORACLE over everything the widened layout takes (cost forms, widths, update functions, weights)
MGM_FUZZ_N=4000 MGM_FUZZ_BASE=40000 timeout 1500 python -m pytest tests/test_gpu_ragged_oracle.py -q -k "random" 2>&1 | tail -n 3 > $O/fuzz_ragged_oracle.log; cat $O/fuzz_ragged_oracle.log
tools/rel_phases_run.sh rel_diag=1 rel_diag=0 > $O/rel_phases.txt 2>&1
head -40 $O/rel_phases.txt
