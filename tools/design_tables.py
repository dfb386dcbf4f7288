"""Development aid: rewrites the three measurement tables of DESIGN.md section 5 in place from profiles/<round>_bench_lines.jsonl
and profiles/<round>_*_traffic.json (the BASELINE configurations, the variants, the pipelined streams).
    python tools/design_tables.py r04"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
lines = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", rnd + "_bench_lines.jsonl")) if l.strip().startswith("{")]


def row(d):
    a, c = d["roofline"]["avg_launch_ms"], d["config"]
    return (c["workload"].split(":")[0], c.get("pairs_per_step"), c.get("pipeline_depth", 1), d["value"], a.get("k_cost", 0),
            next((a[k] for k in ("k_pass2", "k_pass", "k_pass_rel", "k_pass_exact") if k in a), 0), a.get("k_wta", 0), d["roofline"]["frac"])


rows = [row(d) for d in lines]
traffic = {}
for f in glob.glob(os.path.join(ROOT, "profiles", rnd + "_*_traffic.json")):
    t = json.load(open(f))
    traffic[(t["workload"], t["pairs_per_step"])] = t
HEAD = ["| workload | pairs/step | volumes/s | K2 ms | K3 ms per launch | k_wta ms | frac (alg.) | frac (PMC counters) |", "|---|---|---|---|---|---|---|---|"]
BASE = ("cfg3", "cfg3h", "cfg2", "cfg5", "cfg4")


def table(sel):
    out = list(HEAD)
    for r in rows:
        if not sel(r):
            continue
        fc, t = "—", traffic.get((r[0], r[1]))
        if t and r[2] == 1:
            agg = r[5] + r[1] * r[6]
            fc = "%.3f (%.1f GB per volume)" % (t["aggregation_hbm_bytes_per_step"] / (agg * 1e-3) / 1e9 / 8000.0, t["aggregation_hbm_bytes_per_volume"] / 1e9)
        out.append("| %s | %s%s | %.1f | %.2f | %.2f | %.2f | **%.3f** | %s |" % (r[0], r[1], (" (pipelined, D = %d)" % r[2]) if r[2] > 1 else "", r[3], r[4], r[5], r[6], r[7], fc))
    return out


tables = [table(lambda r: r[0] in BASE and r[2] == 1), table(lambda r: r[0] not in BASE), table(lambda r: r[2] > 1)]
path = os.path.join(ROOT, "DESIGN.md")
src = open(path).read().split("\n")
out, i, k = [], 0, 0
while i < len(src):
    if src[i] == HEAD[0] and k < 3:
        out += tables[k]
        k += 1
        i += 1
        while i < len(src) and src[i].startswith("|"):
            i += 1
        continue
    out.append(src[i])
    i += 1
assert k == 3, k
open(path, "w").write("\n".join(out))
d0 = lines[0]
print("default line: %.1f volumes/s, frac %.3f, counter %s, parity %s, cpu reference %.3f port %.3f" % (
    d0["value"], d0["roofline"]["frac"], d0["roofline"].get("frac_counter"), d0["parity"]["status"], d0["cpu_baseline"]["reference"]["value"],
    d0["cpu_baseline"]["port"]["value"]))
