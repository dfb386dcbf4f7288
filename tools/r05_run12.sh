#!/bin/bash
O=gpurun_out/r05l; mkdir -p $O
{
for w in "cfg3 1" "cfg3h 1" "cfg3ad 1" "cfg4 1" "cfg3 2" "cfg2 2"; do
MGM_HIP_TUNE=show_plan=1 timeout 120 python bench.py --workload ${w% *} --batch ${w#* } --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity 2>&1 >/dev/null | grep "simulated\|mgm plan\] 1\|mgm plan\] 4" | tail -2
REPS=2 bash tools/ab_multi.sh "$w" MGM_HIP_TUNE=diag=0 MGM_HIP_TUNE=diag=1 MGM_HIP_TUNE=diag=1,one_queue=0 MGM_HIP_TUNE=diag=1,one_queue=1
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 1500 python -m pytest tests/test_gpu_xcdq.py tests/test_gpu_fullsize.py tests/test_gpu_atsize.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q 2>&1 | tail -5
