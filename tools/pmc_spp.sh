set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for spp in 128 1024; do
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_spp$spp -- python3 $R/bench.py --spp $spp --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline > $R/gpurun_out/pmc_spp$spp.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcg_spp$spp -- python3 $R/bench.py --spp $spp --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline > $R/gpurun_out/pmcg_spp$spp.log 2>&1 || exit 1
done
echo done
