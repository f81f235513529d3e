# Round 5: 2-seat and 6-seat forms of the partition-major scan (QADC_MQ_SEATS26; second build: make ab AB_QUERY_FLAGS=
# AB_KERNEL_FLAGS=-DQADC_MQ_SEATS26=0): parity of the grouped paths, then the IVF legs under both libraries (tools/head_ab2.sh).
# -> gpurun_out/mq_seats26_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/mq_seats26_ab.txt
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_config_shapes.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -n 4 -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT
bash tools/head_ab2.sh > /dev/null 2>&1
cat gpurun_out/head_ab2.txt >> $OUT
cat $OUT
