# PMC passes over the BVH trace kernel on the 10,000-disk scene (run on the GPU box from the repo root)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/bvh_pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY" \
         "GRBM_GUI_ACTIVE" \
         "TA_BUSY_avr TA_TA_BUSY_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_scene.py --height 1920 --width 1080 --spp 32 --n 10000 > $OUT/p$i.log 2>&1 || { echo fail $i; tail -3 $OUT/p$i.log; }
done
echo done
