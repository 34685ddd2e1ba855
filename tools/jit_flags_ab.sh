#!/bin/bash
# (PT_HIPRTC needs a library built with `make -C pathtracer_amd/csrc EXP=-DPT_EXPERIMENTS`)
# jit_flags_ab.sh -- the scene kernel of the default scene (PT_SPECIALISE=1) under different compiler options
# (PT_JIT_FLAGS) and compilers (PT_HIPRTC): per-launch time by HIP events and the canvas checksum, which must not change.
# usage (GPU box, repo root): tools/jit_flags_ab.sh [spp]
cd "$(dirname "$0")/.."
SPP=${1:-1024}
run() { printf "%-70s " "[$1] $2"; PT_SPECIALISE=1 PT_HIPRTC="$1" PT_JIT_FLAGS="$2" timeout -k 10 120 ./tools/kbench $SPP 6 || echo "(failed)"; }
TORCH=$(python3 -c "import importlib.util,os;print(os.path.join(list(importlib.util.find_spec('torch').submodule_search_locations)[0],'lib','libhiprtc.so'))")
run /opt/rocm/lib/libhiprtc.so ""
run /opt/rocm/lib/libhiprtc.so "-mllvm -amdgpu-sched-strategy=max-ilp"
run /opt/rocm/lib/libhiprtc.so "-mllvm -amdgpu-sched-strategy=max-memory-clause"
run /opt/rocm/lib/libhiprtc.so "-mllvm -amdgpu-sched-strategy=iterative-ilp"
run /opt/rocm/lib/libhiprtc.so "-mllvm -amdgpu-sched-strategy=iterative-maxocc"
run /opt/rocm/lib/libhiprtc.so "-mllvm -enable-post-misched=0"
run /opt/rocm/lib/libhiprtc.so "-mllvm -enable-misched=0"
run "$TORCH" ""
run /opt/rocm/lib/libhiprtc.so ""
run "$TORCH" ""
printf "%-70s " "general kernel"; timeout -k 10 120 ./tools/kbench $SPP 6
