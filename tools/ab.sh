#!/bin/bash
# ab.sh -- run tools/kbench against every pathtracer_amd/libpt_exp*.so (and libpt.so)
# usage: tools/ab.sh [kbench args...]
set -e
cd "$(dirname "$0")/.."
for lib in pathtracer_amd/libpt.so pathtracer_amd/libpt_exp*.so; do
    [ -f "$lib" ] || continue
    d=$(mktemp -d)
    cp "$lib" "$d/libpt.so"
    printf "%-34s " "$lib"
    LD_LIBRARY_PATH="$d" timeout -k 10 120 ./tools/kbench "$@"
    rm -rf "$d"
done
