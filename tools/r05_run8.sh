#!/bin/bash
O=gpurun_out/r05h; mkdir -p $O
{
for w in "cfg3 1" "cfg3h 1" "cfg3 2" "cfg2 1" "cfg2 2" "cfg1s 2" "cfg3 4" "cfg4 1" "cfg3 12" "cfg2 16" "cfg5 16" "cfg3w 1" "cfg3ad 1"; do
MGM_HIP_TUNE=show_plan=1 timeout 120 python bench.py --workload ${w% *} --batch ${w#* } --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity 2>&1 >/dev/null | grep "simulated" | tail -1
REPS=2 bash tools/ab_multi.sh "$w" MGM_HIP_TUNE=order=1 MGM_HIP_TUNE=order=2
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 1200 python -m pytest tests/test_gpu_xcdq.py tests/test_gpu_fullsize.py tests/test_gpu_atsize.py tests/test_gpu_batch.py tests/test_gpu_w2.py -x -q 2>&1 | tail -5
