line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); a=d['roofline']['avg_launch_ms']
    print('$1', d['config']['workload'][:6], 'B', d['config']['pairs_per_step'], 'vol/s %.1f'%d['value'], 'K3 %.2f'%a.get('k_pass2',0), 'frac %.3f'%d['roofline']['frac'], flush=True)
except Exception as e: print('$1', 'bad', e)
"; }
run() { tag=$1; w=$2; b=$3; shift 3; env "$@" timeout 300 python bench.py --workload $w --batch $b --steps 20 --repeats 0 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | line "$tag"; }
for wb in cfg3:2 cfg3:1 cfg3:4 cfg3h:2 cfg2:2 cfg2:4 cfg4:1; do
  w=${wb%%:*}; b=${wb##*:}
  run auto $w $b X=1
  run wg2 $w $b MGM_HIP_WG_PER_CU=2
  run wg2-prio $w $b MGM_HIP_WG_PER_CU=2 MGM_HIP_PRIO=-2
  run wg2-prioall $w $b MGM_HIP_WG_PER_CU=2 MGM_HIP_PRIO=255
done
MGM_FUZZ_N=1500 MGM_FUZZ_BASE=50000 timeout 900 python -m pytest tests/test_gpu_cli.py -q -m gpu -k random -n 8 2>&1 | tail -3
