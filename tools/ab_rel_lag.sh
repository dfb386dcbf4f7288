#!/bin/bash
# Development aid (GPU box): the lag per band the simulated schedule assumes (rel_lag: + on the line walks, rel_lagd: + on the anti-diagonal passes)
for tune in "${@:-rel_lagd=0}"; do for cfg in "cfg3r 1" "cfg3r 2" "cfg3r 4" "cfg3hr 1" "cfg3hr 2" "cfg3hr 4" "cfg3r50 1"; do set -- $cfg
MGM_HIP_TUNE=$tune timeout 300 python bench.py --workload $1 --batch $2 --steps 10 --warmup 2 --repeats 0 --no-cpu-baseline --no-parity --extras off 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tune | $1 x$2', round(d['value'],1), round(d['kernel_ms_per_step']['k_pass_rel'],2))"
done; done
