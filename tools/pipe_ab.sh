#!/bin/bash
# pipe_ab.sh [reps] -- the pipelined frame loop of bench.py (what `value` is timed on) with libpt.so and every
# pathtracer_amd/libpt_exp*.so, alternating, at a rank's 128-spp share of an 8-GPU frame (libpt's all-reduce over an
# RCCL world of 1 in the loop) and at 1024 spp.  Run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
REPS=${1:-3}
one() {  # lib spp steps extra-env
  printf "%-22s %4d spp: " "$(basename $1 .so)" $2
  env PT_LIBPT_PATH=$1 $4 timeout -k 10 200 python3 $R/bench.py --spp $2 --steps $3 --warmup 6 --no-cpu-baseline 2>/dev/null \
    | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], 'ms per step,', j['value'], 'Mrays/s, kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'])" || exit 1
}
for rep in $(seq $REPS); do
  for lib in $R/pathtracer_amd/libpt.so $R/pathtracer_amd/libpt_exp*.so; do
    [ -f "$lib" ] || continue
    one $lib 128 200 PT_BENCH_FORCE_DIST=1 || exit 1
    one $lib 1024 40 PT_X=0 || exit 1
  done
done
