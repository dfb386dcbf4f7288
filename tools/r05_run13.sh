#!/bin/bash
# final timelines of the one-pair launches (tl variant of the library) + a last A/B of two bands per CU under the new schedule
O=gpurun_out/r05m; mkdir -p $O
for wb in "cfg3 1" "cfg3 2" "cfg2 1" "cfg2 2" "cfg3h 1" "cfg1s 2"; do
  set -- $wb
  rm -f /tmp/tl.txt
  MGM_HIP_LIB=$PWD/mgm_amd/lib/variants/tl/libmgm_hip.so MGM_HIP_TIMELINE=/tmp/tl.txt MGM_BENCH_PLACE_TRIES=0 timeout 300 python bench.py --workload $1 --batch $2 --steps 3 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity > /dev/null 2>$O/tl_$1_b$2.err
  python tools/timeline.py /tmp/tl.txt --csv $O/timeline_$1_b$2.csv > $O/timeline_$1_b$2.txt 2>&1
done
{
REPS=2 bash tools/ab_multi.sh "cfg3 1" MGM_HIP_TUNE=order=2 MGM_HIP_TUNE=wg_per_cu=2 MGM_HIP_TUNE=wg_per_cu=2,one_queue=1 MGM_HIP_TUNE=order=0,one_queue=0
REPS=2 bash tools/ab_multi.sh "cfg3h 1" MGM_HIP_TUNE=order=2 MGM_HIP_TUNE=wg_per_cu=2 MGM_HIP_TUNE=order=0,one_queue=0
REPS=2 bash tools/ab_multi.sh "cfg2 2" MGM_HIP_TUNE=order=2 MGM_HIP_TUNE=order=0,one_queue=0
REPS=2 bash tools/ab_multi.sh "cfg3 2" MGM_HIP_TUNE=order=2 MGM_HIP_TUNE=order=0,one_queue=0
} > $O/ab.txt 2>&1
cat $O/ab.txt
head -12 $O/timeline_cfg3_b1.txt; grep -A22 "time slices" $O/timeline_cfg3_b1.txt | awk 'NR>1{printf "%s/%s ", $2,$3} END{print ""}'
