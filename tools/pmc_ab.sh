#!/bin/bash
# pmc_ab.sh <out-dir> [spp...] -- rocprofv3 counters of the trace kernel for libpt.so and every
# pathtracer_amd/libpt_exp*.so, through tools/kbench (scene kernel, lone launches): VALU instructions, wave cycles
# and kernel cycles per launch are far less noisy than a 1-2 % difference in milliseconds.  Run on the GPU box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; shift
SPPS=${*:-"1024 128"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PT_SPECIALISE=${PT_SPECIALISE:-1}
for lib in $R/pathtracer_amd/libpt.so $R/pathtracer_amd/libpt_exp*.so; do
  [ -f "$lib" ] || continue
  name=$(basename $lib .so)
  d=$(mktemp -d); cp $lib $d/libpt.so
  export LD_LIBRARY_PATH=$d
  for spp in $SPPS; do
    timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS --output-format csv -d $OUT/${name}_${spp}_sq -- $R/tools/kbench $spp 4 > $OUT/${name}_${spp}_sq.log 2>&1 || { echo "sq pass failed: $name $spp"; tail -3 $OUT/${name}_${spp}_sq.log; exit 1; }
    timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/${name}_${spp}_grbm -- $R/tools/kbench $spp 4 > $OUT/${name}_${spp}_grbm.log 2>&1 || { echo "grbm pass failed: $name $spp"; exit 1; }
  done
  rm -rf $d
done
python3 $R/tools/pmc_ab_summary.py $OUT
