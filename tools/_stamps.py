import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["QADC_BENCH_CPU_SECONDS"] = "0"
import bench
shape = sys.argv[1]
kw = {} if shape == "c3" else dict(M=32, K=16384, MA=64, dim=96, N=int(1e9), seed0=7000)
bench.ivf_leg(0, **kw)
