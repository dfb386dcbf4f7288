#!/bin/bash
# Runs ON THE GPU BOX: round-3 exploration of the launch heuristics (deep DMA rings, volumes per wave, bands per CU)
# through the library's development switches.   bash tools/r3_explore.sh [list-file]
set -u
OUT=gpurun_out/explore; mkdir -p $OUT
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); a=d['roofline']['avg_launch_ms']
    print('$1', d['config']['workload'][:6], 'B', d['config']['pairs_per_step'], 'vol/s %.1f'%d['value'], 'K2 %.2f'%a.get('k_cost',0), 'K3 %.2f'%a.get('k_pass2',a.get('k_pass',0)), 'wta %.2f'%a['k_wta'], 'frac %.3f'%d['roofline']['frac'], flush=True)
except Exception as e: print('$1', 'bad', e)
"; }
run() { # tag workload batch [env...]
  tag=$1; w=$2; b=$3; shift 3
  env "$@" timeout 300 python bench.py --workload $w --batch $b --steps ${STEPS:-20} --repeats 0 --no-cpu-baseline --no-parity 2>$OUT/err.txt | tail -1 | line "$tag"
}
for wb in cfg2:1 cfg2:2 cfg2:4 cfg2:8 cfg2:16 cfg3:1 cfg3:2 cfg3:4 cfg3:8 cfg3:12 cfg3h:1 cfg3h:2 cfg3h:12 cfg4:1 cfg4:2 cfg5:16; do
  run auto ${wb%%:*} ${wb##*:} X=1
done
for wb in cfg3:4 cfg3:8 cfg3h:12 cfg2:4 cfg2:8 cfg5:16 cfg4:2; do
  run deep0 ${wb%%:*} ${wb##*:} MGM_HIP_DEEP=0
  run deep1 ${wb%%:*} ${wb##*:} MGM_HIP_DEEP=1
done
for wb in cfg2:2 cfg2:4 cfg2:8 cfg2:16 cfg5:16; do
  run nosubv-wg1 ${wb%%:*} ${wb##*:} MGM_HIP_SUBV=0 MGM_HIP_WG_PER_CU=1
  run nosubv-wg2 ${wb%%:*} ${wb##*:} MGM_HIP_SUBV=0 MGM_HIP_WG_PER_CU=2
  run nosubv-wg2-deep0 ${wb%%:*} ${wb##*:} MGM_HIP_SUBV=0 MGM_HIP_WG_PER_CU=2 MGM_HIP_DEEP=0
done
