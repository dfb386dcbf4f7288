#!/bin/bash
# Runs ON THE GPU BOX: long random campaigns of the parity tests on the final kernels (the default test run uses short ones).
#   bash tools/fuzz_campaign.sh [out_dir]
O=${1:-gpurun_out/r05_fuzz3}; mkdir -p $O
MGM_FUZZ_N=4000 MGM_FUZZ_BASE=70000 timeout 1500 python -m pytest tests/test_gpu_rel.py -q -k random 2>&1 | tail -n 2 > $O/fuzz_rel.log
MGM_FUZZ_N=3000 MGM_FUZZ_BASE=20000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -n 2 > $O/fuzz_agg.log
MGM_FUZZ_N=300 MGM_FUZZ_BASE=40000 MGM_FUZZ_SCALE=8 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -n 2 > $O/fuzz_agg_scale8.log
MGM_FUZZ_N=600 MGM_FUZZ_BASE=11000 timeout 2400 python -m pytest tests/test_gpu_cli.py -q -k "fuzz or random" 2>&1 | tail -n 2 > $O/fuzz_cli.log
for f in $O/*.log; do echo "== $f"; cat $f; done
