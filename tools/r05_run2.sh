#!/bin/bash
# Runs ON THE GPU BOX (round 5): timelines of single launches (tl variant of the library) + the CLI tests and timings
O=gpurun_out/r05b; mkdir -p $O
for wb in "cfg3 1" "cfg3 2" "cfg2 1" "cfg2 2" "cfg3h 1" "cfg1s 2"; do
  set -- $wb
  rm -f /tmp/tl.txt
  MGM_HIP_LIB=$PWD/mgm_amd/lib/variants/tl/libmgm_hip.so MGM_HIP_TIMELINE=/tmp/tl.txt timeout 300 python bench.py --workload $1 --batch $2 --steps 3 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity > $O/tl_$1_b$2.json 2>$O/tl_$1_b$2.err
  python tools/timeline.py /tmp/tl.txt --csv $O/timeline_$1_b$2.csv > $O/timeline_$1_b$2.txt 2>&1
done
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_abi.py -x -q > $O/cli_tests.txt 2>&1
bash tools/cli_fullsize.sh > $O/cli_fullsize.txt 2>&1
cat $O/timeline_cfg3_b1.txt; tail -5 $O/cli_tests.txt; grep -v "mgm stats" $O/cli_fullsize.txt | tail; grep "resident job" $O/cli_fullsize.txt | tail -3
