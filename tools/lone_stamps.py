"""Phase stamps of the lone query's workgroups 0 and 1 (library built with -DQADC_STAMPS): python tools/lone_stamps.py [codes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np
import pyqadc
pyqadc.LIB_PATH = os.path.join(ROOT, "quick-adc_amd", "libqadc_hip_stamps.so")   # make -C quick-adc_amd stamps
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
M, R = 16, 100
idx = pyqadc.Index(M, 0)
idx.add_partition_synthetic(n, 1)
idx.finalize(bench.KEEP)
rng = np.random.default_rng(0)
cb = rng.normal(size=(M, 16, 8)).astype(np.float32)
tb = bench.make_tables(rng, cb, 1)
a = np.zeros((1, 1), np.int32)
for _ in range(6):
    idx.query_scan(a, tb.copy(), R)
idx.close()
