#!/bin/bash
# Runs ON THE GPU BOX (round 5): a pass pinned to an XCD quantises (72 items of 2.3 ms on 32 CUs = 3 rounds): blocks of bands dealt
# over the XCDs, with both ticket orders
O=gpurun_out/r05d; mkdir -p $O
{
for w in "cfg3 1" "cfg3h 1" "cfg3 2" "cfg2 2"; do
REPS=1 bash tools/ab_multi.sh "$w" MGM_HIP_TUNE=order=1 MGM_HIP_TUNE=order=1,xcdq_k=2 MGM_HIP_TUNE=order=1,xcdq_k=4 MGM_HIP_TUNE=order=1,xcdq_k=8 MGM_HIP_TUNE=order=1,xcdq_k=16 MGM_HIP_TUNE=order=0,xcdq_k=2 MGM_HIP_TUNE=order=0,xcdq_k=8 MGM_HIP_TUNE=order=1
done
REPS=1 bash tools/ab_multi.sh "cfg2 1" MGM_HIP_TUNE=order=1 MGM_HIP_TUNE=order=1,xcdq_k=1 MGM_HIP_TUNE=order=1,xcdq_k=4 MGM_HIP_TUNE=order=1,xcdq_k=8 MGM_HIP_TUNE=order=0,xcdq_k=4
} > $O/ab.txt 2>&1
cat $O/ab.txt
