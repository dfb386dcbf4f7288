#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh'): the round's measurement evidence.
#   1. bench lines of the three workloads (default flags for the headline one: includes the CPU baseline)
#   2. rocprofv3 --kernel-trace --stats of the default bench command        -> kernel_stats.csv
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, counters only) -> hbm_counters.md, traffic.json
# Everything lands in gpurun_out/profiles/; copy what is to be judged into profiles/ afterwards.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
W=${1:-cfg3}
B=${2:-12}
: > "$OUT/bench_lines.jsonl"
timeout 900 python bench.py 2> "$OUT/bench_default.stderr" | tail -1 >> "$OUT/bench_lines.jsonl"
for w in cfg3h cfg2; do timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 >> "$OUT/bench_lines.jsonl"; done
for b in 8 4 1; do timeout 300 python bench.py --workload cfg3 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 >> "$OUT/bench_lines.jsonl"; done

cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python "$ROOT/bench.py" --workload $W --batch $B --steps 10 --no-cpu-baseline > "$OUT/bench_under_rocprof.log" 2> "$OUT/kt.stderr"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python "$ROOT/bench.py" --workload $W --batch $B --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/fetch.stderr"
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python "$ROOT/bench.py" --workload $W --batch $B --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/write.stderr"
cd "$ROOT"
KS=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1)
FC=$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1)
WC=$(find "$OUT/write" -name "*counter_collection.csv" | head -1)
[ -n "$KS" ] && cp "$KS" "$OUT/kernel_stats.csv"
tail -1 "$OUT/bench_under_rocprof.log" > "$OUT/bench_under_rocprof.json"
if [ -n "$FC" ] && [ -n "$WC" ]; then
  python tools/summarize_pmc.py "$FC" "$WC" $W $B "$OUT/hbm_counters.md" "$OUT/traffic.json" > /dev/null
fi
rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write"
ls -la "$OUT"
cat "$OUT/kernel_stats.csv" 2>/dev/null | head -12
cat "$OUT/hbm_counters.md" 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/profiles/bench_lines.jsonl"):
    try: d = json.loads(l)
    except Exception: print("bad line", l[:100]); continue
    print(d["config"]["workload"][:6], "B", d["config"].get("pairs_per_step"), "value", round(d["value"], 2), "ms/step", round(d["ms_per_step"], 2),
          "frac", round(d["roofline"]["frac"], 3), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
