"""The reference's own call, as it makes it (backend/engine.py:50-57): index.search(x (1, d) numpy float32, k) on an index
of its own size -- per-call latency through the host entry point, and what the device part of it is.
CASES="n,d,k;..." to choose, METRIC=ip for IndexFlatIP (the reference's "cosine" index); under rocprofv3 --kernel-trace --stats the kernels behind a call show by name."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
cases = os.environ.get("CASES", "1000,2048,20;10000,2048,20;1000,128,9;100000,512,10;1000000,512,10")
for n, d, k in (tuple(int(v) for v in c.split(",")) for c in cases.split(";")):
    xb = np.random.default_rng(1).random((n, d), dtype=np.float32)
    index = (faiss.IndexFlatIP if os.environ.get("METRIC", "l2") == "ip" else faiss.IndexFlatL2)(d); index.add(xb)
    xq = np.random.default_rng(2).random((1, d), dtype=np.float32)
    for _ in range(50): index.search(xq, k)
    t0 = time.perf_counter(); reps = 500
    for _ in range(reps): D, I = index.search(xq, k)
    host_us = (time.perf_counter() - t0) / reps * 1e6
    xq_d = torch.from_numpy(xq).cuda()
    Dd = torch.empty((1, k), dtype=torch.float32, device="cuda"); Id = torch.empty((1, k), dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(50): index.search_into(xq_d, k, Dd, Id, s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): index.search_into(xq_d, k, Dd, Id, s)
    torch.cuda.synchronize(); dev_us = (time.perf_counter() - t0) / reps * 1e6
    t0 = time.perf_counter()
    for _ in range(reps):
        index.search_into(xq_d, k, Dd, Id, s); torch.cuda.synchronize()
    sync_us = (time.perf_counter() - t0) / reps * 1e6
    print(f"{n}x{d} k={k}: host call {host_us:7.1f} us   device entry point back to back {dev_us:7.1f} us per call, "
          f"call + synchronize {sync_us:7.1f} us", flush=True)
