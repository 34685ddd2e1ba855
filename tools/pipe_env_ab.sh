#!/bin/bash
# pipe_env_ab.sh <reps> <ENV=val|-> ... -- the pipelined frame loop of bench.py under each environment setting in
# turn ("-" = none), alternating, at 128 spp (libpt's all-reduce over an RCCL world of 1 in the loop) and 1024 spp.
R=${GRAFT_REPO_ROOT:-$(pwd)}
REPS=$1; shift
one() {  # setting spp steps extra-env
  printf "%-24s %4d spp: " "$1" $2
  s=$1; [ "$s" = "-" ] && s=PT_X=0
  env $s $4 timeout -k 10 200 python3 $R/bench.py --spp $2 --steps $3 --warmup 6 --no-cpu-baseline 2>/dev/null \
    | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], 'ms per step,', j['value'], 'Mrays/s, kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'], 'segs', j['config']['segments_per_step'])" || exit 1
}
for rep in $(seq $REPS); do
  for s in "$@"; do
    one $s 128 200 PT_BENCH_FORCE_DIST=1 || exit 1
    one $s 1024 40 PT_Y=0 || exit 1
  done
done
