# SQ/GRBM counters of scan_query_kernel under rocprofv3 (run on the GPU box); prints the longest dispatch
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_wgq -- python3 $R/tools/wgq_probe.py 0 > $R/gpurun_out/pmc_wgq.log 2>&1
python3 $R/tools/pmc_longest.py $R/gpurun_out/pmc_wgq scan_query
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_wgq2 -- python3 $R/tools/wgq_probe.py 0 > $R/gpurun_out/pmc_wgq2.log 2>&1
python3 $R/tools/pmc_longest.py $R/gpurun_out/pmc_wgq2 scan_query
