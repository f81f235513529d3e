"""Phase stamps of the IVF head launch (library built with `make -C quick-adc_amd stamps`): workgroup 1 of every head launch of
bench.py's IVF leg prints its phases in shader cycles; this prints their medians.  python tools/head_stamps.py c3|c5"""
import os, sys, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
    os.environ["QADC_BENCH_CPU_SECONDS"] = "0"
    import pyqadc
    pyqadc.LIB_PATH = os.path.join(ROOT, "quick-adc_amd", "libqadc_hip_stamps.so")
    import bench
    kw = {} if sys.argv[1] == "c3" else dict(M=32, K=16384, MA=64, dim=96, N=int(1e9), seed0=7000)
    bench.ivf_leg(0, **kw)
    sys.exit(0)
out = subprocess.run([sys.executable, os.path.abspath(__file__), sys.argv[1], "child"], capture_output=True, text=True).stdout
rows = [list(map(int, re.findall(r" (\d+)", l.split(":", 1)[1]))) for l in out.splitlines() if l.startswith("HEAD STAMPS")]
names = re.findall(r"([a-z-]+) \d+", [l for l in out.splitlines() if l.startswith("HEAD STAMPS")][0])
import numpy as np
a = np.array(rows)
print("%s head, %d launches, median shader cycles of workgroup 1: " % (sys.argv[1], len(rows)) +
      "  ".join("%s %d" % (n, int(np.median(a[:, i]))) for i, n in enumerate(names)) + "  (sum %d)" % int(np.median(a.sum(1))))
