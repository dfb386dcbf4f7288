"""Development aid: the markdown rows of DESIGN.md section 5 from profiles/<round>_bench_lines.jsonl and *_traffic.json.
python tools/design_tables.py r03"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = os.path.join(bench.ROOT, "profiles")
rows = {}
for l in open(os.path.join(P, rnd + "_bench_lines.jsonl")):
    d = json.loads(l)
    wn, B = d["config"]["workload"].split(":")[0], d["config"]["pairs_per_step"]
    a = d["roofline"]["avg_launch_ms"]
    rr = bench.roofline_of(bench.WORKLOADS[wn], B, a, wn)
    rows[(wn, B)] = (d["value"], a["k_cost"], a["k_pass2"], a["k_wta"], d["roofline"]["frac"], rr["frac_counter"])
    print("| %s x %d | %.1f | %.2f | %.2f | %.2f | %.3f | %s |" % (wn, B, d["value"], a["k_cost"], a["k_pass2"], a["k_wta"], d["roofline"]["frac"],
                                                                    "%.3f" % rr["frac_counter"] if rr["frac_counter"] else "-"))
print()
for f in sorted(os.listdir(P)):
    if f.startswith(rnd) and f.endswith("_traffic.json"):
        j = json.load(open(os.path.join(P, f)))
        pk = j["per_kernel"]
        kp = [k for k in pk if "k_pass" in k][0]
        kw = [k for k in pk if "k_wta" in k][0]
        wn, B = j["workload"], j["pairs_per_step"]
        w = bench.WORKLOADS[wn]
        alg = 12.0 * w["NDIR"] * w["nx"] * w["ny"] * bench.labels_of(w) / 1e9
        r = rows.get((wn, B))
        print("| %s x %d | %.1f / %.1f GB | %.1f GB | %.2f GB | %.1f GB | %s | %s |" % (
            wn, B, pk[kp]["read_bytes"] / 1e9, pk[kp]["write_bytes"] / 1e9, pk[kw]["read_bytes"] / 1e9, j["aggregation_hbm_bytes_per_volume"] / 1e9, alg,
            "%.3f" % r[4] if r else "-", "%.3f" % r[5] if r and r[5] else "-"))
