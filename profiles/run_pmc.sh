#!/bin/bash
# Collects the rocprofv3 evidence for one build: kernel-trace stats, then PMC
# passes, each pass a run of its own (counters never share a run with a trace).
# usage: profiles/run_pmc.sh <tag>      (run on the GPU box from the repo root)
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --no-pipeline: launches strictly one after the other, so that a kernel's duration in the trace is its own
# (the default frame loop overlaps the tail of one frame's kernel with the head of the next)
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline"
# (the trace pass runs more steps: the first launch of a process is cold -- 2.5 ms longer here -- and should not
# weigh 1/7 in the average that is compared with bench.py's event-timed kernel_ms)
TRACE="python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-pipeline"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $TRACE > $OUT/trace.log 2>&1 || exit 1
[ "${TRACE_ONLY:-0}" = 1 ] && { echo done; exit 0; }
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM" \
         "GRBM_GUI_ACTIVE GRBM_COUNT" \
         "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.log; exit 1; }
done
echo done
