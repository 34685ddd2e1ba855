# kernel timeline of the pipelined frame loop (run on the GPU box from the repo root): which hardware queue each
# trace kernel ran on and how consecutive frames overlap
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_pipe -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/tr_pipe.log 2>&1 || exit 1
echo done
