#!/bin/bash
# Development aid (runs in the build container): build variants of libmgm_hip.so with different
# K3 tuning defines into mgm_amd/lib/variants/<name>/ for A/B timing on the GPU box.
set -e
cd "$(dirname "$0")/.."
mkdir -p mgm_amd/lib/variants
while [ $# -gt 0 ]; do
  name="$1"; defs="$2"; shift 2
  MGM_P2_DEFINES="$defs" MGM_REL_DEFINES="${RELDEFS:-}" python mgm_amd/build.py >/dev/null
  mkdir -p mgm_amd/lib/variants/$name
  cp mgm_amd/lib/libmgm_hip.so mgm_amd/lib/variants/$name/
  echo "built $name: $defs"
done
python mgm_amd/build.py >/dev/null   # (objects whose command line changed are rebuilt: the pass kernels)
