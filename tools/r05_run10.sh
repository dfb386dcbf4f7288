#!/bin/bash
O=gpurun_out/r05j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rel.py tests/test_gpu_cli.py tests/test_gpu_windowed.py tests/test_gpu_post.py -x -q 2>&1 | tail -25
: > $O/lines.jsonl
for wb in cfg3r:1 cfg3hr:1 cfg3r:4; do
  w=${wb%%:*}; b=${wb##*:}
  timeout 600 python bench.py --workload $w --batch $b --steps 5 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity 2>$O/err_$w.txt | tail -1 >> $O/lines.jsonl
done
python - $O/lines.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: print("bad line", l[:300]); continue
    r = d["roofline"]
    print(d["config"]["workload"][:9], "B", d["config"]["pairs_per_step"], "vol/s %.2f" % d["value"], "ms/step %.2f" % d["ms_per_step"],
          " ".join("%s %.2f" % (k, v) for k, v in sorted(d["kernel_ms_per_step"].items())), "frac %.3f" % r["frac"],
          ("frac_rp %.3f" % r["frac_range_proportional"]) if "frac_range_proportional" in r else "")
PY
cat $O/err_*.txt | tail -5
bash tools/ragged_cli.sh > $O/ragged_cli.txt 2>&1; cat $O/ragged_cli.txt
