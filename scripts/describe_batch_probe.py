import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from image_search_engine_amd import descriptors as ds
rng = np.random.default_rng(0)
desc = ds.CNNDescriptor()
desc.warm_up()
imgs = [rng.integers(0, 256, (375, 500, 3), dtype=np.uint8) for _ in range(32)]
for b in (1, 2, 4, 8, 16, 32):
    for _ in range(3): desc.describe_batch(imgs[:b])
    t0 = time.perf_counter()
    for _ in range(20): desc.describe_batch(imgs[:b])
    dt = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        x = desc.preprocessor(imgs[:b]); torch.cuda.synchronize()
    dp = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        f = desc._forward(x); torch.cuda.synchronize()
    df = (time.perf_counter() - t0) / 20
    print(f"batch {b:2d}: describe_batch {dt*1e3:6.2f} ms ({b/dt:6.0f} img/s); preprocess {dp*1e3:6.2f} ms; forward {df*1e3:6.2f} ms")
