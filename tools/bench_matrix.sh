#!/bin/bash
# Runs ON THE GPU BOX: the bench lines of the workload / batch matrix (no CPU legs), one JSON line each.
#   bash tools/bench_matrix.sh out.jsonl "cfg3:1 cfg3:2 cfg3:4 cfg3:12 cfg3h:1 cfg2:16 ..."
OUT=$1; shift
: > "$OUT"
for wb in $1; do
  w=${wb%%:*}; b=${wb##*:}
  timeout 600 python bench.py --workload $w --batch $b --steps ${STEPS:-20} --repeats 0 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 >> "$OUT"
done
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: print("bad line", l[:100]); continue
    a = d["roofline"]["avg_launch_ms"]
    print(d["config"]["workload"][:6], "B", d["config"]["pairs_per_step"], "vol/s %.1f" % d["value"], "ms/vol %.2f" % (d["ms_per_step"] / d["config"]["pairs_per_step"]),
          "K3 %.2f" % a.get("k_pass2", a.get("k_pass", 0)), "wta %.2f" % a["k_wta"], "frac %.3f" % d["roofline"]["frac"])
PY
