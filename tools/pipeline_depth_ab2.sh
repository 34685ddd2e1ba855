# frames in flight, 2 against 3, when the collective takes time: libpt's all-reduce over an RCCL world of 1 plus a
# spin of US microseconds behind it on the same stream (PT_BENCH_COLLECTIVE_EXTRA_US) -- one GPU, a rank's 128-spp share
set -e
for us in 0 100 250 500; do
  for d in 2 3; do
    printf "128 spp, collective + %3d us, depth %d: " $us $d
    PT_BENCH_FORCE_DIST=1 PT_BENCH_COLLECTIVE_EXTRA_US=$us timeout -k 10 120 python3 bench.py --spp 128 --steps 100 --warmup 6 --no-cpu-baseline --pipeline-depth $d 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], 'ms per step')"
  done
done
