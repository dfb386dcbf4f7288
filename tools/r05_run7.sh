#!/bin/bash
O=gpurun_out/r05g; mkdir -p $O
{
for w in "cfg3 1" "cfg3h 1" "cfg3 2" "cfg2 1" "cfg2 2" "cfg1s 2" "cfg1s 1"; do
REPS=1 bash tools/ab_multi.sh "$w" MGM_HIP_TUNE=order=1 MGM_HIP_TUNE=order=2 MGM_HIP_TUNE=order=2,one_queue=1 MGM_HIP_TUNE=order=1,one_queue=1 MGM_HIP_TUNE=order=2,xcdq_k=2 MGM_HIP_TUNE=order=2,one_queue=1,strips=0 MGM_HIP_TUNE=order=1
done
REPS=1 bash tools/ab_multi.sh "cfg3 4" MGM_HIP_TUNE=order=1 MGM_HIP_TUNE=order=2 MGM_HIP_TUNE=order=2,one_queue=1
REPS=1 bash tools/ab_multi.sh "cfg4 1" MGM_HIP_TUNE=order=1 MGM_HIP_TUNE=order=2 MGM_HIP_TUNE=order=2,one_queue=1
REPS=1 bash tools/ab_multi.sh "cfg3 12" MGM_HIP_TUNE=order=1 MGM_HIP_TUNE=order=2
} > $O/ab.txt 2>&1
cat $O/ab.txt
