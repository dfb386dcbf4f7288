"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py into profiles/<tag>_hbm_counters.md
and profiles/<tag>_traffic.json (which records the hash of the kernel sources it was measured on: bench.py
reports the figure as roofline.traffic only while the kernels are still those).  FETCH_SIZE is doubled (gfx950 counts 128-B requests as 64 B,
MI355X_MICROARCH.md section HBM); WRITE_SIZE is used as reported (it reproduces k_cost's W*H*L*4 exactly)."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def per_kernel(path, counter):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in per.items()}


fetch, write, workload, B, out_md, out_json = sys.argv[1:7]
B = int(B)
F, W = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
rows, total = [], 0.0
for k in sorted(F):
    if not k.startswith(("void mgm", "mgm::")):
        continue
    fb, wb = F[k][0] * 2 * 1024, W.get(k, (0, 0))[0] * 1024
    rows.append((k, F[k][1], F[k][0], fb / 1e9, W.get(k, (0, 0))[0], wb / 1e9))
    if "k_pass" in k:
        total += fb + wb          # one launch per step covers the whole batch
    elif "k_wta" in k:
        total += B * (fb + wb)    # one launch per volume
with open(out_md, "w") as f:
    f.write("# HBM traffic per launch, workload %s (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes)\n\n" % workload)
    f.write("FETCH_SIZE x2 correction per MI355X_MICROARCH.md; averages over the profiled launches.\n\n")
    f.write("| kernel | launches | FETCH_SIZE (KB, raw) | read GB (x2) | WRITE_SIZE (KB) | write GB |\n|---|---|---|---|---|---|\n")
    for r in rows:
        f.write("| `%s` | %d | %.4g | %.3f | %.4g | %.3f |\n" % r)
    f.write("\nAggregation of one batch of %d volumes (one pass-kernel launch + %d k_wta launches): %.2f GB of HBM traffic, %.2f GB per volume\n"
            % (B, B, total / 1e9, total / 1e9 / B))
from bench import kernel_source_hash
json.dump({"workload": workload, "pairs_per_step": B, "kernel_source_sha": kernel_source_hash(), "aggregation_hbm_bytes_per_step": total,
           "aggregation_hbm_bytes_per_volume": total / B,
           "per_kernel": {r[0]: {"read_bytes": r[3] * 1e9, "write_bytes": r[5] * 1e9} for r in rows}},
          open(out_json, "w"), indent=1)
print(open(out_md).read())
