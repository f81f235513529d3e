# SQ/GRBM counters of the multi-query scan under rocprofv3 (run on the GPU box); then python tools/pmc_longest.py gpurun_out/pmc_mq scan_i8_mq
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
N=1e9 NQ=32 STEPS=3 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_mq -- python3 $R/tools/pipeline_probe.py > $R/gpurun_out/pmc_mq.log 2>&1
