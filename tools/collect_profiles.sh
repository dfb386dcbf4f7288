#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh'): the round's measurement evidence.
#   1. bench lines: the default command (cfg3, 12 pairs per step; parity gate + CPU baseline), then the workload / batch matrix
#   2. per (workload, batch) in $PROFILED: rocprofv3 --kernel-trace --stats of the bench command  -> <tag>_kernel_stats.csv
#      and rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, counters only)          -> <tag>_hbm_counters.md, <tag>_traffic.json
# Everything lands in gpurun_out/profiles/; copy what is to be judged into profiles/ afterwards (prefixed with the round).
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
PROFILED=${PROFILED:-"cfg3:12 cfg3:2 cfg3:1 cfg2:16 cfg2:2 cfg2:1 cfg1s:2 cfg4:1 cfg3w:12 cfg3r:1 cfg3r:2 cfg3r:4 cfg3hr:1 cfg3r50:1 cfg1sr:2"}
MATRIX=${MATRIX:-"cfg3h:12 cfg2:16 cfg5:16 cfg4:1 cfg4:2 cfg3:8 cfg3:4 cfg3:2 cfg3:1 cfg3h:2 cfg3h:1 cfg2:8 cfg2:4 cfg2:2 cfg2:1 cfg3w:12 cfg3w:1 cfg3hw:12 cfg3hw:1 cfg3ad:12 cfg3ad:1 cfg3ad1:12 cfg3ncc:12 cfg3ncc:1 cfg3bt:12 cfg3c7:12 cfg1s:16 cfg1s:2 cfg1s:1 cfg3L200:8 cfg3L200:1 cfg3L768:4 cfg3L768:1 cfg3r:1 cfg3r:2 cfg3r:4 cfg3hr:1 cfg3hr:2 cfg3hr:4 cfg3r50:1 cfg3r50:4 cfg1sr:2 cfg1sr:4 cfg3i2:1 cfg3w3:1 cfg3w3:4 cfg3hw3:1 cfg3L1536:1 cfg3L2048:1 cfg3neg:1 cfg3nan:1 cfg3rinf:1"}
# a stream of single pairs / small batches through a pipelined context (mgm_ctx_set_pipeline): workload:batch:depth
PIPED=${PIPED:-"cfg3:1:2 cfg3:1:4 cfg3:1:12 cfg3h:1:4 cfg2:1:4 cfg2:1:8 cfg3:2:2"}
: > "$OUT/bench_lines.jsonl"
timeout 900 python bench.py 2> "$OUT/bench_default.stderr" | tail -1 >> "$OUT/bench_lines.jsonl"
for wb in $MATRIX; do
  w=${wb%%:*}; b=${wb##*:}
  st=20; case $w in cfg3nan|cfg3rinf|cfg3L1536|cfg3L2048) st=3;; esac   # (the slow fall-backs: 0.25-0.4 s per step)
  timeout 600 python bench.py --workload $w --batch $b --steps $st --repeats 0 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 >> "$OUT/bench_lines.jsonl"
done
for wbd in $PIPED; do
  w=${wbd%%:*}; rest=${wbd#*:}; b=${rest%%:*}; d=${rest##*:}
  timeout 600 python bench.py --workload $w --batch $b --pipeline $d --repeats 0 --no-cpu-baseline 2>/dev/null | tail -1 >> "$OUT/bench_lines.jsonl"
done
for wb in $PROFILED; do
  w=${wb%%:*}; b=${wb##*:}; tag=${w}_b${b}
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python "$ROOT/bench.py" --workload $w --batch $b --steps 10 --repeats 0 --no-cpu-baseline --no-parity > "$OUT/${tag}_bench_under_rocprof.log" 2> "$OUT/kt.stderr"
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python "$ROOT/bench.py" --workload $w --batch $b --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity > /dev/null 2> "$OUT/fetch.stderr"
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python "$ROOT/bench.py" --workload $w --batch $b --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity > /dev/null 2> "$OUT/write.stderr"
  cd "$ROOT"
  KS=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1)
  FC=$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1)
  WC=$(find "$OUT/write" -name "*counter_collection.csv" | head -1)
  [ -n "$KS" ] && cp "$KS" "$OUT/${tag}_kernel_stats.csv"
  tail -1 "$OUT/${tag}_bench_under_rocprof.log" > "$OUT/${tag}_bench_under_rocprof.json"; rm -f "$OUT/${tag}_bench_under_rocprof.log"
  if [ -n "$FC" ] && [ -n "$WC" ]; then
    python tools/summarize_pmc.py "$FC" "$WC" $w $b "$OUT/${tag}_hbm_counters.md" "$OUT/${tag}_traffic.json" > /dev/null
  fi
  rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write"
done
# the launch plan against every override on shapes it was not tuned on (tools/plan_sweep.py; SKIP_SWEEP=1 leaves it out)
[ -z "${SKIP_SWEEP:-}" ] && timeout 1500 python tools/plan_sweep.py > "$OUT/plan_sweep.txt" 2> "$OUT/plan_sweep.err"
# the single-GPU ingredients of DESIGN.md section 6's direction-sharding model, and the wall time of the whole command line
timeout 600 python tools/time_passes.py cfg4 > "$OUT/cfg4_pass_blocks.txt" 2>&1
for i in 1 2 3; do MGM_HIP_STATS=1 bash tools/cli_fullsize.sh 2>&1 | grep -v "^disp\|^cost"; sleep 2; done > "$OUT/cli_fullsize.txt" 2>&1
bash tools/ragged_cli.sh > "$OUT/ragged_cli.txt" 2>&1
ls -la "$OUT"
for f in "$OUT"/*_kernel_stats.csv; do echo "== $f"; head -8 "$f" | cut -c1-200; done
cat "$OUT"/*_hbm_counters.md 2>/dev/null | grep -E "^#|Aggregation|k_pass|k_wta"
python - <<'PY'
import json
for l in open("gpurun_out/profiles/bench_lines.jsonl"):
    try: d = json.loads(l)
    except Exception: print("bad line", l[:100]); continue
    a = d["roofline"]["avg_launch_ms"]
    print(d["config"]["workload"][:8], "B", d["config"].get("pairs_per_step"), "D", d["config"].get("pipeline_depth"), "value", round(d["value"], 2), "ms/step", round(d["ms_per_step"], 2),
          "K2", round(a.get("k_cost", 0), 3),
          "K3", round(next((a[k] for k in ("k_pass2", "k_pass", "k_pass_rel", "k_pass_exact") if k in a), 0), 2), "wta", round(a["k_wta"], 2), "frac", round(d["roofline"]["frac"], 3),
          "parity", (d.get("parity") or {}).get("status"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
