#!/bin/bash
# Runs ON THE GPU BOX (round 5): A/B of the ticket order (relative progress vs latest start time) and of two bands per CU
# for the small launches, alternating in one call
O=gpurun_out/r05c; mkdir -p $O
{
REPS=2 bash tools/ab_multi.sh "cfg3 1" MGM_HIP_TUNE=order=0 MGM_HIP_TUNE=order=1
REPS=2 bash tools/ab_multi.sh "cfg3 2" MGM_HIP_TUNE=order=0 MGM_HIP_TUNE=order=1
REPS=2 bash tools/ab_multi.sh "cfg3h 1" MGM_HIP_TUNE=order=0 MGM_HIP_TUNE=order=1
REPS=2 bash tools/ab_multi.sh "cfg2 1" MGM_HIP_TUNE=order=0 MGM_HIP_TUNE=order=1 MGM_HIP_TUNE=order=1,wg_per_cu=2 MGM_HIP_TUNE=order=0,wg_per_cu=2
REPS=2 bash tools/ab_multi.sh "cfg2 2" MGM_HIP_TUNE=order=0 MGM_HIP_TUNE=order=1 MGM_HIP_TUNE=order=1,wg_per_cu=2
REPS=2 bash tools/ab_multi.sh "cfg1s 2" MGM_HIP_TUNE=order=0 MGM_HIP_TUNE=order=1 MGM_HIP_TUNE=order=1,wg_per_cu=2
REPS=2 bash tools/ab_multi.sh "cfg1s 1" MGM_HIP_TUNE=order=0 MGM_HIP_TUNE=order=1 MGM_HIP_TUNE=order=1,wg_per_cu=2
REPS=1 bash tools/ab_multi.sh "cfg4 1" MGM_HIP_TUNE=order=0 MGM_HIP_TUNE=order=1
REPS=1 bash tools/ab_multi.sh "cfg3 4" MGM_HIP_TUNE=order=0 MGM_HIP_TUNE=order=1
REPS=1 bash tools/ab_multi.sh "cfg3 12" MGM_HIP_TUNE=order=0 MGM_HIP_TUNE=order=1
REPS=1 bash tools/ab_multi.sh "cfg2 16" MGM_HIP_TUNE=order=0 MGM_HIP_TUNE=order=1
} > $O/ab.txt 2>&1
cat $O/ab.txt
