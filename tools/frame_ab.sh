#!/bin/bash
# frame_ab.sh <reps> -- the pipelined frame loop: round 2's tree (_base_r02/, if present) against this tree with and
# without the drain gate, at a rank's share of an 8-GPU frame (128 spp; libpt's all-reduce over an RCCL world of 1 in the
# loop) and at 1024 spp (one GPU, no collective).  Run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
REPS=${1:-2}
one() {  # label bench.py spp steps env... -- args
  label=$1; b=$2; spp=$3; steps=$4; e=$5; shift 5
  printf "%-26s %4d spp: " "$label" $spp
  env $e timeout -k 10 200 python3 $b --spp $spp --steps $steps --warmup 6 --no-cpu-baseline "$@" 2>/dev/null \
    | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], 'ms per step,', j['value'], 'Mrays/s, kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'], 'segs', j['config']['segments_per_step'])" || exit 1
}
for rep in $(seq $REPS); do
  for spp in 128 256 1024; do
    steps=200; [ $spp = 1024 ] && steps=40
    env=PT_BENCH_FORCE_DIST=1; [ $spp = 1024 ] && env=PT_X=0
    [ -d $R/_base_r02 ] && { one "round 2" $R/_base_r02/bench.py $spp $steps $env || exit 1; }
    one "this tree, no gate" $R/bench.py $spp $steps $env --no-gate || exit 1
    one "this tree, gate" $R/bench.py $spp $steps $env || exit 1
  done
done
