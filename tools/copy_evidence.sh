#!/bin/bash
# Development aid: copies what tools/r05_final.sh left under gpurun_out/ into profiles/<round>_* (the tracked copies) and rewrites
# DESIGN.md's measurement tables.   bash tools/copy_evidence.sh r05
R=${1:-r05}; O=gpurun_out/profiles
for wb in cfg3:12 cfg3:2 cfg3:1 cfg2:16 cfg2:2 cfg2:1 cfg1s:2 cfg4:1 cfg3w:12 cfg3r:1 cfg3r:4; do
  w=${wb%%:*}; b=${wb##*:}; t=${w}_b${b}
  for suf in bench_under_rocprof.json hbm_counters.md kernel_stats.csv traffic.json; do cp $O/${t}_$suf profiles/${R}_${t}_$suf; done
done
cp $O/bench_lines.jsonl profiles/${R}_bench_lines.jsonl; cp $O/cli_fullsize.txt profiles/${R}_cli_fullsize.txt
cp $O/ragged_cli.txt profiles/${R}_ragged_cli.txt; cp $O/cfg4_pass_blocks.txt profiles/${R}_cfg4_pass_blocks.txt
F=gpurun_out/${R}_final
for f in $F/timeline_*.txt $F/rel_timeline_*.txt; do cp $f profiles/${R}_$(basename $f); done
cp $F/rel_phases.txt profiles/${R}_rel_phases.txt; cp $F/fuzz_cli.log profiles/${R}_fuzz_cli.log
cp gpurun_out/final_pytest_gpu.log profiles/${R}_pytest_gpu.log
python tools/design_tables.py $R
