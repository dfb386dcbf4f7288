#!/bin/bash
# Development aid (GPU box): SQ counters of k_pass_rel for cfg3r x 1 / x 4 (counters in a run of their own, kernel trace only)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
O=gpurun_out/r06_final; mkdir -p $O
for cfg in "${@:-cfg3r 1}"; do set -- $cfg
  rm -rf $O/pmc_valu
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_valu -o v -- python bench.py --workload $1 --batch $2 --steps 3 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity --extras off > /dev/null 2> $O/pmc_valu.err
  f=$(find $O/pmc_valu -name "*counter_collection.csv" | head -1)
  echo "== $1 x$2" | tee -a $O/rel_valu_counters.txt
  python tools/rel_valu_counters.py $f | tee -a $O/rel_valu_counters.txt
  python tools/rel_valu_counters.py $f k_wta_rel | tee -a $O/rel_valu_counters.txt
done
rm -rf $O/pmc_valu
