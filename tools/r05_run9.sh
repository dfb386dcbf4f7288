#!/bin/bash
O=gpurun_out/r05i; mkdir -p $O
: > $O/lines.jsonl
for wb in cfg3r:1 cfg3:1 cfg3hr:1 cfg3h:1 cfg3i2:1 cfg3w3:1 cfg3hw3:1 cfg3w:1 cfg3hw:1 cfg3L768:1 cfg3L1536:1 cfg3L2048:1 cfg3neg:1 cfg3r:4 cfg3w3:4; do
  w=${wb%%:*}; b=${wb##*:}
  timeout 600 python bench.py --workload $w --batch $b --steps 5 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity 2>$O/err_$w.txt | tail -1 >> $O/lines.jsonl
done
for w in cfg3nan cfg3rinf; do
  timeout 900 python bench.py --workload $w --batch 1 --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity 2>$O/err_$w.txt | tail -1 >> $O/lines.jsonl
done
python - $O/lines.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: print("bad line", l[:200]); continue
    r = d["roofline"]; a = r["avg_launch_ms"]
    print(d["config"]["workload"][:9], "B", d["config"]["pairs_per_step"], "vol/s %.2f" % d["value"], "ms/step %.2f" % d["ms_per_step"],
          " ".join("%s %.2f" % (k, v) for k, v in sorted(d["kernel_ms_per_step"].items())), "frac %.3f" % r["frac"],
          ("frac_rp %.3f" % r["frac_range_proportional"]) if "frac_range_proportional" in r else "")
PY
tail -3 $O/err_*.txt | grep -v "^$" | head -40
bash tools/ragged_cli.sh > $O/ragged_cli.txt 2>&1; cat $O/ragged_cli.txt
bash tools/cli_fullsize.sh > $O/cli_fullsize.txt 2>&1; cat $O/cli_fullsize.txt | cut -c1-330
