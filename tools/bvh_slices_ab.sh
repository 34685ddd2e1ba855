#!/bin/bash
# bvh_slices_ab.sh -- the indexed kernel's sample slices: bench.py --scene lines (pipelined ms per frame, serial kernel ms)
# with the second slice 1/16 (shipped), 1/8, and none (PT_BVH_SLICE_DIV=1).  Run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for div in 16 1 8 32; do
  for sc in "disks1000 512 8 3" "spheres1000 512 8 3" "disks10000 256 16 2" "disks10000 512 16 2"; do
    set -- $sc
    PT_BVH_SLICE_DIV=$div timeout -k 10 300 python3 $R/bench.py --scene $1 --height 3840 --width 2160 --spp $2 --bounces $3 --steps $4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print('slice 1/$div', '$1', $2, 'spp: pipelined', j['ms_per_step'], 'ms per frame, serial kernel', j['roofline']['kernel_ms'], 'ms')" || exit 1
  done
done
