#!/bin/bash
O=gpurun_out/r05e; mkdir -p $O
{
for w in "cfg3 1" "cfg3h 1" "cfg3 2" "cfg2 1" "cfg2 2" "cfg1s 2"; do
REPS=1 bash tools/ab_multi.sh "$w" MGM_HIP_TUNE=order=1 MGM_HIP_TUNE=order=1,xcdq=2 MGM_HIP_TUNE=order=0,xcdq=2 MGM_HIP_TUNE=order=1,xcdq=0 MGM_HIP_TUNE=order=0,xcdq=0 MGM_HIP_TUNE=order=1,xcdq=2,strips=0 MGM_HIP_TUNE=order=1
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
