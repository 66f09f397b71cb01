"""One-query float32 L2 batches on small indexes: the direct scan against the filtered path (short_scan_kernel +
merge + gate), interleaved in one process after a warm-up (the first second of a process runs ~4 us per call slower)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
from image_search_engine_amd import _native as n
torch.manual_seed(5)
LAT = os.environ.get("LATENCY", "0") == "1"   # 1: synchronise after every call (one caller at a time) instead of back to back
def timed(index, xq, k, reps=400):
    for _ in range(50): index.search_torch(xq, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        index.search_torch(xq, k)
        if LAT: torch.cuda.synchronize()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
warm = faiss.IndexFlatL2(512); warm.add_torch(torch.rand((100_000, 512), device="cuda"))
timed(warm, torch.rand((16, 512), device="cuda"), 10, 20000)
for rows, d in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or ((1000, 512), (2000, 512), (4000, 512), (6000, 512), (8000, 512), (12000, 512), (16000, 512), (1000, 128), (4000, 128), (16000, 128), (1000, 2048), (4000, 2048)):
    index = faiss.IndexFlatL2(d); index.add_torch(torch.rand((rows, d), device="cuda"))
    xq = torch.rand((1, d), device="cuda")
    res = {"": [], "1": []}
    for rep in range(3):
        for nd in ("", "1"):
            if nd: os.environ["ISE_NO_DIRECT"] = "1"
            else: os.environ.pop("ISE_NO_DIRECT", None)
            n.lib.ise_refresh_env_knobs()
            res[nd].append(timed(index, xq, 10))
    os.environ.pop("ISE_NO_DIRECT", None); n.lib.ise_refresh_env_knobs()
    print(f"{rows}x{d}: direct {min(res['']):6.1f} us   filtered {min(res['1']):6.1f} us   (3 rounds each: {[round(v,1) for v in res['']]} {[round(v,1) for v in res['1']]})", flush=True)
