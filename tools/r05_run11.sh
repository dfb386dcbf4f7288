#!/bin/bash
O=gpurun_out/r05k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_w2.py tests/test_gpu_cli.py -x -q -k "w2 or weight or W2 or aP2" 2>&1 | tail -4
: > $O/lines.jsonl
for wb in cfg3w:1 cfg3w:12 cfg3hw:1 cfg3w:4; do
  w=${wb%%:*}; b=${wb##*:}
  timeout 600 python bench.py --workload $w --batch $b --steps 5 --warmup 1 --repeats 0 --no-cpu-baseline 2>$O/err_$w.txt | tail -1 >> $O/lines.jsonl
done
python - $O/lines.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: print("bad line", l[:300]); continue
    r = d["roofline"]
    print(d["config"]["workload"][:9], "B", d["config"]["pairs_per_step"], "vol/s %.2f" % d["value"], "ms/step %.2f" % d["ms_per_step"],
          " ".join("%s %.2f" % (k, v) for k, v in sorted(d["kernel_ms_per_step"].items())), "frac %.3f" % r["frac"], "parity", (d.get("parity") or {}).get("status"))
PY
