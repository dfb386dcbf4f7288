"""Development aid: summary of a rocprofv3 --pmc run (SQ_* + GRBM_GUI_ACTIVE, csv) per kernel -- where a wave's cycles go and how busy the
VALUs are.   python tools/rel_valu_counters.py <counter_collection.csv> [kernel substring]
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over the waves (MI355X_MICROARCH.md, profiling): WAIT_ANY (parked at
s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES; a SIMD issues one VALU instruction at a time, so
SQ_INSTS_VALU x 4 cycles (a wave64 fp32 instruction occupies its SIMD for four; DPP, 64-bit and integer-multiply forms longer) against the
kernel's cycles x 1024 SIMDs is the share of the chip's VALU issue time the kernel used at least.  GRBM_GUI_ACTIVE comes summed over the 8
XCDs (2.3e8 for a 12.2 ms launch at 2.4 GHz)."""
import collections
import csv
import sys

want = sys.argv[2] if len(sys.argv) > 2 else "k_pass_rel"
per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if want in k:
        per[k][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, disp in per.items():
    n = len(disp)
    avg = collections.defaultdict(float)
    for d in disp.values():
        for c, v in d.items():
            avg[c] += v / n
    wc = avg.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    print("%s: %d launches" % (k, n))
    for c in sorted(avg):
        print("  %-22s %.4g" % (c, avg[c]))
    print("  of a wave's cycles: parked (s_waitcnt / barrier) %.1f %%, issue stalls %.1f %%, issuing %.1f %% (VALU %.1f %%)"
          % (100 * avg.get("SQ_WAIT_ANY", 0) / wc, 100 * avg.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * avg.get("SQ_ACTIVE_INST_ANY", 0) / wc,
             100 * avg.get("SQ_ACTIVE_INST_VALU", 0) / wc))
    if avg.get("GRBM_GUI_ACTIVE"):
        cyc = avg["GRBM_GUI_ACTIVE"] / 8.0  # (per XCD)
        print("  kernel cycles %.4g per XCD; VALU instructions per launch %.4g = %.4g per SIMD x 4 cycles = >= %.1f %% of the VALU issue time"
              % (cyc, avg.get("SQ_INSTS_VALU", 0), avg.get("SQ_INSTS_VALU", 0) / 1024, 100 * avg.get("SQ_INSTS_VALU", 0) / 1024 * 4 / cyc))
