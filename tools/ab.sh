timeout 600 python tools/gpu_check.py --case-timeout 60 2>&1 | grep -E "^AGG" | awk "{print \$2, \$NF}" | sort | uniq -c | head -30
for w in cfg3h cfg3; do timeout 120 python bench.py --workload $w --steps 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"config\"][\"workload\"][:5], round(d[\"ms_per_step\"],2), round(d[\"roofline\"][\"frac\"],3), {k:round(v,2) for k,v in d[\"kernel_ms_per_step\"].items()})"; done
MGM_STATS_REAL=1 timeout 120 python tools/gpu_stats.py cfg3 2>&1 | tail -34 | grep -A3 "pass [04]:"
