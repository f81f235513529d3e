# 8-GPU pre-flight on ONE GPU (round 6): everything of the driver's N-GPU launch that a one-GPU box can exercise.
#   A  plain `python bench.py` (N = 1)                                  -> value_A
#   B  the driver's launch line with N = 1 (torch.distributed.run)      -> value_B, must agree with A within 2 %
#   C  the multi-rank loop with the one RCCL rank the box allows        -> rccl_ranks 1, native merge
#   D  the driver's launch line with N = 8: eight ranks SHARING the GPU (QADC_BENCH_ONE_GPU=1, gloo for the script's barriers,
#      the library's native merge over its shared-memory transport in place of RCCL)  -> n_gpus 8, rccl_ranks 8, ivf + ivf_c5 legs
#   E  a collective that never completes (QADC_BENCH_IVF_TIMEOUT=0.2): every rank leaves with exit code 3, the launcher returns
#   F  HBM traffic of the headline's mode on a 125 M-code shard (rocprofv3 --pmc child passes): the stand-in for the
#      `roofline.traffic: null` of the multi-rank line
# usage: bash tools/preflight_8gpu.sh   (writes gpurun_out/r06_preflight_8gpu.txt; ~6 GPU-minutes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_preflight_8gpu.txt
mkdir -p $R/gpurun_out; : > $OUT
export MASTER_ADDR=127.0.0.1
LEAN="QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_C2=0 QADC_BENCH_CEILING=0"
P='import sys,json
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
iv=j.get("ivf") or {}; c5=j.get("ivf_c5") or j.get("ivf_c5_one_gpu") or {}
print("n_gpus %s rccl_ranks %s value %.4e codes/s ms_per_step %.3f roofline.frac %.3f traffic %s merge=%s | ivf %s us/query (ranks %s) | ivf_c5 %s us/query" % (j["n_gpus"], j["rccl_ranks"], j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["traffic"], (j.get("multi_gpu_merge") or "-")[:40], iv.get("us_per_query", iv.get("error")), iv.get("rccl_ranks"), c5.get("us_per_query", c5.get("error"))))'
echo "A plain N=1:" >> $OUT
env $LEAN QADC_BENCH_IVF_CODES=0 python3 $R/bench.py --steps 10 --warmup 2 2>/dev/null | python3 -c "$P" >> $OUT
echo "B driver launch line, N=1:" >> $OUT
env $LEAN QADC_BENCH_IVF_CODES=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 $R/bench.py --gpus 1 --steps 10 --warmup 2 2>/dev/null | python3 -c "$P" >> $OUT
echo "C multi-rank loop, one RCCL rank:" >> $OUT
env $LEAN QADC_BENCH_IVF_CODES=0 QADC_BENCH_FORCE_DIST=1 python3 $R/bench.py --steps 10 --warmup 2 2>/dev/null | python3 -c "$P" >> $OUT
echo "D driver launch line, N=8, eight ranks on this one GPU (shared-memory transport in place of RCCL):" >> $OUT
env $LEAN QADC_BENCH_BACKEND=gloo QADC_BENCH_ONE_GPU=1 timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29612 $R/bench.py --gpus 8 --steps 10 --warmup 2 2>$R/gpurun_out/r06_preflight_D.err | python3 -c "$P" >> $OUT
echo "  (launcher exit code ${PIPESTATUS[0]})" >> $OUT
echo "E watchdog: the IVF legs given 0.2 s on 2 ranks:" >> $OUT
env $LEAN QADC_BENCH_BACKEND=gloo QADC_BENCH_ONE_GPU=1 QADC_BENCH_CODES=4e7 QADC_BENCH_IVF_TIMEOUT=0.2 timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 $R/bench.py --gpus 2 --steps 3 --warmup 1 2>/dev/null | python3 -c "$P" >> $OUT
echo "  (launcher exit code ${PIPESTATUS[0]}; 124 would be the outer timeout = a hang)" >> $OUT
echo "F PMC traffic of the headline's mode on a 125 M-code shard (32 queries per step):" >> $OUT
cd /tmp && TMPDIR=/tmp python3 -c "
import sys, json; sys.path.insert(0, '$R'); import bench
print(json.dumps(bench.pmc_traffic_in_run(16, int(125e6), 32)))" >> $OUT 2>/dev/null
cat $OUT
