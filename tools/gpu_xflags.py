"""Development aid: kernel time of one config under MGM_HIP_XFLAGS experiment bits (results are wrong by design)."""
import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3h"
for f in (sys.argv[2:] or ["0", "1", "2", "3", "4", "5", "7"]):
    env = dict(os.environ, MGM_HIP_XFLAGS=f, MGM_HIP_DEBUG_STATS="0")
    out = subprocess.run([sys.executable, os.path.join(here, "gpu_stats.py"), cfg], env=env, capture_output=True, text=True).stdout
    print("XFLAGS=%s" % f, out.strip().splitlines()[-1], flush=True)
