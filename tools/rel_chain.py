"""Development aid: one pass of a k_pass_rel timeline dump (MGM_HIP_TIMELINE), band by band: ticket, start, end, waited, start - predecessor's start.
    python tools/rel_chain.py dump.txt <volume*8+pass> [every]"""
import sys

vp = int(sys.argv[2])
every = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rows = []
t0 = None
for line in open(sys.argv[1]):
    f = line.split()
    if f and f[0] == "item":
        t0 = float(f[6]) if t0 is None else min(t0, float(f[6]))
        if int(f[2]) == vp:
            rows.append((int(f[3]), int(f[1]), float(f[6]), float(f[7]), float(f[8]), int(f[12])))
rows.sort()
prev = None
print("band ticket start end waited steps  start-prev.start  end-prev.end")
for b, tk, s, e, w, st in rows:
    if b % every == 0 or b == len(rows) - 1:
        print("%4d %6d %8.1f %8.1f %7.1f %5d  %s" % (b, tk, s, e, w, st, ("%7.1f %7.1f" % (s - prev[0], e - prev[1])) if prev else ""))
    prev = (s, e)
