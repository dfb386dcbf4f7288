#!/bin/bash
# Development aid: copies what the end-of-round evidence run (tools/r06_final.sh) left under gpurun_out/ into profiles/<round>_* (the
# tracked copies) and rewrites DESIGN.md's measurement tables.   bash tools/copy_evidence.sh r06
R=${1:-r06}; O=gpurun_out/profiles
for f in $O/*_b[0-9]*_bench_under_rocprof.json $O/*_b[0-9]*_hbm_counters.md $O/*_b[0-9]*_kernel_stats.csv $O/*_b[0-9]*_traffic.json; do
  [ -f "$f" ] && cp $f profiles/${R}_$(basename $f)
done
cp $O/bench_lines.jsonl profiles/${R}_bench_lines.jsonl; cp $O/cli_fullsize.txt profiles/${R}_cli_fullsize.txt
cp $O/bench_default.stderr profiles/${R}_bench_default.stderr 2>/dev/null
cp $O/ragged_cli.txt profiles/${R}_ragged_cli.txt; cp $O/cfg4_pass_blocks.txt profiles/${R}_cfg4_pass_blocks.txt
F=gpurun_out/${R}_final
for f in $F/timeline_*.txt $F/rel_timeline_*.txt; do [ -f "$f" ] && cp $f profiles/${R}_$(basename $f); done
[ -f $F/rel_phases.txt ] && cp $F/rel_phases.txt profiles/${R}_rel_phases.txt
for f in $F/fuzz_*.log; do [ -f "$f" ] && cp $f profiles/${R}_$(basename $f); done
cp gpurun_out/final_pytest_gpu.log profiles/${R}_pytest_gpu.log
[ -f gpurun_out/final_smoke.log ] && cp gpurun_out/final_smoke.log profiles/${R}_smoke.log
python tools/kernel_resources.py > profiles/${R}_kernel_resources.txt 2>/dev/null
python tools/design_tables.py $R
