"""Large-batch path (csrc/ise_gemm_scan.hpp) against the streaming passes: QPS and fp32 TFLOP/s at 1M x 512.
ISE_NO_GEMM / ISE_GEMM_SAMPLE_DIV are read once per process, so each arm runs in a child process."""
import os, sys, subprocess, time
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import image_search_engine_amd.faiss_compat as faiss
    n, d, k = int(os.environ.get("N", "1000000")), int(os.environ.get("D", "512")), 10
    g = torch.Generator(device="cuda").manual_seed(1)
    xb = torch.rand((n, d), generator=g, device="cuda")
    storage = os.environ.get("STORAGE", "f32")
    index = faiss.IndexFlat(d, 0 if storage == "bf16" else 1, storage=storage); index.add_torch(xb)
    out = []
    for nq in [int(x) for x in os.environ.get("NQS", "64,128,256,1024,4096").split(",")]:
        xq = torch.rand((nq, d), generator=g, device="cuda")
        for _ in range(3): index.search_torch(xq, k)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 10
        for _ in range(reps): index.search_torch(xq, k)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        out.append((nq, round(dt * 1e3, 3), round(nq / dt / 1e3, 1), round(2.0 * nq * n * d / dt / 1e12, 1)))
    tag = storage + " " + "stream" if os.environ.get("ISE_NO_GEMM") == "1" else "gemm div=" + os.environ.get("ISE_GEMM_SAMPLE_DIV", "32")
    print(tag, "(nq, ms, kQPS, TFLOP/s):", out, index.exact_stats())
else:
    arms = [{"ISE_NO_GEMM": "1"}, {}, {"ISE_GEMM_SAMPLE_DIV": "64"}, {"ISE_GEMM_SAMPLE_DIV": "16"}]
    for a in arms:
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **a))
