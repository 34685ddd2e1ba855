# frames in flight: 2 against 3, on one GPU (run on the GPU box from the repo root).  With PT_BENCH_FORCE_DIST=1 the
# loop holds libpt's all-reduce over an RCCL world of 1 -- a rank's share of an 8- / 4- / 2-GPU frame is 128 / 256 / 512 spp
set -e
for spp in 128 256 512 1024; do
  for d in 2 3; do
    printf "collective world 1, %4d spp, depth %d: " $spp $d
    PT_BENCH_FORCE_DIST=1 timeout -k 10 120 python3 bench.py --spp $spp --steps 100 --warmup 6 --no-cpu-baseline --pipeline-depth $d 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], 'ms per step,', j['value'], 'Mrays/s')"
  done
done
for d in 2 3; do
  printf "no collective, 1024 spp, depth %d: " $d
  timeout -k 10 120 python3 bench.py --steps 60 --warmup 6 --no-cpu-baseline --pipeline-depth $d 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], 'ms per step,', j['value'], 'Mrays/s')"
done
