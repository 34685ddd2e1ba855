#!/bin/bash
# bvh_ab.sh -- tools/bvh_counters.py (1,000 disks, 1,000 spheres, 10,000 disks at 3840x2160) with libpt.so and every
# pathtracer_amd/libpt_exp*.so in turn.  Run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${REPS:-2}); do
for lib in $R/pathtracer_amd/libpt.so $R/pathtracer_amd/libpt_exp*.so; do
  [ -f "$lib" ] || continue
  echo "== $(basename $lib .so)"
  PT_LIBPT_PATH=$lib timeout -k 10 300 python3 $R/tools/bvh_counters.py 2>/dev/null || exit 1
done
done
