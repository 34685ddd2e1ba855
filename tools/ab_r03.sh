#!/bin/bash
# ab_r03.sh -- A/B of libpt.so against pathtracer_amd/libpt_exp*.so with tools/kbench: general and scene kernel,
# 1024 and 128 spp, lone launches (HIP events); each line carries the canvas checksum and the segment count.
cd "$(dirname "$0")/.."
for spec in 0 1; do for spp in 1024 128; do
  echo "== PT_SPECIALISE=$spec spp=$spp"
  for rep in 1 2; do
    PT_SPECIALISE=$spec tools/ab.sh $spp 10 || exit 1
  done
done; done
