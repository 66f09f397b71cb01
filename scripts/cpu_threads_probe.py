"""How many host threads does the CPU oracle scale to on this box? (dev aid)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flat_oracle as fo
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "omp", fo.max_threads())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(p):
        print(p, open(p).read().strip())
rng = np.random.default_rng(0)
xb = rng.random((250_000, 512), dtype=np.float32)
xq = rng.random((16, 512), dtype=np.float32)
for nt in (0, 16, 32, 64, 128):
    fo.knn_flat(xb[:1000], xq, 10, 1, nt)
    t = time.perf_counter(); fo.knn_flat(xb, xq, 10, 1, nt); el = time.perf_counter() - t
    print(f"threads={nt:4d}  {el*1e3:8.1f} ms  -> {16/(el*4):8.1f} QPS at 1M rows")
