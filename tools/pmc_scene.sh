set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/scene_pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for F in 0 2; do
  timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $OUT/f${F}_p$i -- python3 $R/tools/bench_scene.py --height 1920 --width 1080 --spp 32 --n 1000 --flags $F > $OUT/f${F}_p$i.log 2>&1 || { echo fail $i $F; tail -3 $OUT/f${F}_p$i.log; exit 1; }
  done
done
echo done
