#!/bin/bash
# final_r03.sh -- the round's closing measurements on one MI355X box (run from the repo root on the GPU box):
# bench.py as the driver runs it, the synthetic scenes of BASELINE configs[3]/[4], a rank's 128-spp share with libpt's
# collective over an RCCL world of 1, then the rocprofv3 passes of profiles/run_pmc.sh.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/final; mkdir -p $O
timeout -k 10 300 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
for sc in "disks1000 512 8 3" "spheres1000 512 8 3" "disks10000 256 16 2"; do
  set -- $sc
  timeout -k 10 400 python3 $R/bench.py --scene $1 --height 3840 --width 2160 --spp $2 --bounces $3 --steps $4 --warmup 1 --no-cpu-baseline >> $O/synthetic_scenes.jsonl 2>> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
done
for spp in 128 256 512 1024; do
  printf "collective world 1, %4d spp: " $spp >> $O/rank_share.txt
  PT_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 $R/bench.py --spp $spp --steps 100 --warmup 6 --no-cpu-baseline 2>> $O/bench.err | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], 'ms per step,', j['value'], 'Mrays/s,', j['config']['collective'][:60])" >> $O/rank_share.txt || exit 1
done
PT_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_libpt_collective_world1.json 2>> $O/bench.err || exit 1
$R/profiles/run_pmc.sh r03/final_pmc > $O/run_pmc.log 2>&1 || { tail -5 $O/run_pmc.log; exit 1; }
echo done
