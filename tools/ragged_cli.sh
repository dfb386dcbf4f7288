#!/bin/bash
# Runs ON THE GPU BOX: the whole command line with RANGE IMAGES (-m/-M: per-pixel windows of +-24 labels inside a 256-label
# hull) and with TSGM_ITER=2, ours against the reference CLI (oracle/_ref/mgm_img, OpenMP) on the same files: wall times and a
# comparison of the outputs (what had no number of any kind until round 5).
set -u
T=$(mktemp -d)
python - "$T" <<'PY'
import sys, numpy as np
from PIL import Image
sys.path.insert(0, ".")
from mgm_amd import synth
u, v, gt = synth.stereo_pair(1920, 1080, -191, 0, seed=20150907)
for n, a in (("u", u), ("v", v)):
    Image.fromarray(np.clip(np.round(a[0]), 0, 255).astype(np.uint8)).save("%s/%s.png" % (sys.argv[1], n))
lo = np.clip(gt - 24, -255, 0).astype(np.float32); hi = np.clip(gt + 24, -255, 0).astype(np.float32)
Image.fromarray(lo).save(sys.argv[1] + "/lo.tif"); Image.fromarray(hi).save(sys.argv[1] + "/hi.tif")
PY
ARGS="-r -255 -R 0 -t census -s vfit -O 8 -P1 2 -P2 20000"
export CENSUS_NCC_WIN=5 TSGM=3 USE_TRUNCATED_LINEAR_POTENTIALS=1 MEDIAN=1
run_case() {  # name, extra args, env
  local name=$1; shift
  for rep in 1 2; do
    s=$(date +%s%N); env "$@" MGM_HIP_STATS=1 ./mgm_amd/bin/mgm $ARGS $EXTRA $T/u.png $T/v.png $T/o_$name.tif $T/oc_$name.tif > $T/o_$name.log 2> $T/o_$name.err; e=$(date +%s%N)
    echo "$name ours: $(( (e - s) / 1000000 )) ms wall;  $(grep -o 'upload+enqueue.*' $T/o_$name.err | head -1)"
  done
  if [ -x oracle/_ref/mgm_img ]; then
    s=$(date +%s%N); env "$@" OMP_NUM_THREADS=${OMP_NUM_THREADS:-16} timeout 1200 oracle/_ref/mgm_img $ARGS $EXTRA $T/u.png $T/v.png $T/r_$name.tif $T/rc_$name.tif > $T/r_$name.log; e=$(date +%s%N)
    echo "$name reference (OpenMP, ${OMP_NUM_THREADS:-16} threads): $(( (e - s) / 1000000 )) ms wall"
    for f in o r; do ./mgm_amd/bin/imgconv $T/${f}_$name.tif $T/${f}_$name.npy > /dev/null; ./mgm_amd/bin/imgconv $T/${f}c_$name.tif $T/${f}c_$name.npy > /dev/null; done
    python - "$T" "$name" <<'PY'
import sys, numpy as np
t, n = sys.argv[1], sys.argv[2]
for f in ("", "c"):
    a, b = np.load("%s/o%s_%s.npy" % (t, f, n)), np.load("%s/r%s_%s.npy" % (t, f, n))
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    print(n, "disparity" if f == "" else "cost", a.shape, "differing pixels:", int((~same).sum()))
print(n, "stdout identical:", open("%s/o_%s.log" % (t, n)).read() == open("%s/r_%s.log" % (t, n)).read())
PY
  fi
}
EXTRA=""; run_case dense TSGM_ITER=1
EXTRA="-m $T/lo.tif -M $T/hi.tif"; run_case ragged TSGM_ITER=1
EXTRA=""; run_case iter2 TSGM_ITER=2
rm -rf "$T"
