#!/bin/bash
O=gpurun_out/r05n; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_acc.py -x -q 2>&1 | tail -15
: > $O/lines.jsonl
for cfg in "cfg3 12 --accumulate" "cfg3 12" "cfg3 8 --accumulate" "cfg3 4 --accumulate" "cfg3h 12 --accumulate" "cfg2 16 --accumulate" "cfg2 16" "cfg4 2 --accumulate" "cfg4 2" "cfg5 16 --accumulate"; do
  set -- $cfg
  timeout 600 python bench.py --workload $1 --batch $2 $3 --steps 5 --warmup 1 --repeats 0 --no-cpu-baseline --no-parity 2>$O/err.txt | tail -1 >> $O/lines.jsonl
done
python - $O/lines.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: print("bad line", l[:300]); continue
    r = d["roofline"]
    print(d["config"]["workload"][:6], "B", d["config"]["pairs_per_step"], "acc", d["config"].get("accumulate"), "vol/s %.2f" % d["value"], "ms/step %.2f" % d["ms_per_step"],
          " ".join("%s %.2f" % (k, v) for k, v in sorted(d["kernel_ms_per_step"].items())), "frac(step) %.3f" % (12.0*d["config"]["NDIR"]*d["config"]["W"]*d["config"]["H"]*d["config"]["L"]*d["config"]["pairs_per_step"]/(d["ms_per_step"]*1e-3)/8e12))
PY
tail -3 $O/err.txt
