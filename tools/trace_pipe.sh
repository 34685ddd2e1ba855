#!/bin/bash
# trace_pipe.sh <out-dir> <spp> [bench args...] -- rocprofv3 kernel trace of bench.py's pipelined frame loop (libpt's
# all-reduce over an RCCL world of 1 in the loop when PT_BENCH_FORCE_DIST=1 is in the environment)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; SPP=$2; shift 2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python3 $R/bench.py --spp $SPP --steps 40 --warmup 4 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
cp $(ls $OUT/raw/*/*_kernel_trace.csv | head -1) $OUT/kernel_trace.csv
tail -1 $OUT/bench.log | cut -c1-200
