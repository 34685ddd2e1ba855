# kernel timeline of the pipelined frame loop at a rank's share of an 8-GPU frame (128 spp, RCCL world of 1 in the loop):
# run on the GPU box from the repo root
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
PT_BENCH_FORCE_DIST=1 timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_pipe128 -- python3 $R/bench.py --spp 128 --steps 40 --warmup 4 --no-cpu-baseline > $R/gpurun_out/tr_pipe128.log 2>&1 || exit 1
echo done
