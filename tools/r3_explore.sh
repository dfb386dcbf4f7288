#!/bin/bash
# Runs ON THE GPU BOX: round-3 exploration -- single-volume / small-batch launches on pipeline-depth / band-size variants
# built by tools/sweep_build.sh (d<N>: MGM_P2_MAXD=N, n7d<N>: 7 lines per band).
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
V=$PWD/mgm_amd/lib/variants
for wb in cfg2:1 cfg2:2 cfg3:1 cfg3:2 cfg3h:1 cfg4:1 cfg3:12 cfg2:16; do
  w=${wb%%:*}; b=${wb##*:}
  run base $w $b X=1
  for v in d4 d6 d8 n7d4 n7d6; do run $v $w $b MGM_HIP_LIB=$V/$v/libmgm_hip.so; done
done
for wb in cfg2:1 cfg3:1 cfg3:2 cfg3h:1 cfg4:1 cfg3:12 cfg2:16; do
  w=${wb%%:*}; b=${wb##*:}
  for v in n7d4 n7d6; do run $v-wg2 $w $b MGM_HIP_LIB=$V/$v/libmgm_hip.so MGM_HIP_WG_PER_CU=2; done
done
for b in 2 4; do
  run d4-nosubv cfg2 $b MGM_HIP_LIB=$V/d4/libmgm_hip.so MGM_HIP_SUBV=0
  run d4-subv cfg2 $b MGM_HIP_LIB=$V/d4/libmgm_hip.so
done
run base cfg5 16 X=1
run d4 cfg5 16 MGM_HIP_LIB=$V/d4/libmgm_hip.so
run d8 cfg5 16 MGM_HIP_LIB=$V/d8/libmgm_hip.so
