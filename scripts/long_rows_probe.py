"""Long rows (the reference's real descriptor: 2048 floats): short-index kernel against the streaming kernel."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
from image_search_engine_amd import _native as n
k = 20
for rows, d in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or ((1_000, 2048), (10_000, 2048), (100_000, 2048), (30_000, 1024)):
    xb = torch.rand((rows, d), device="cuda")
    index = faiss.IndexFlatL2(d); index.add_torch(xb)
    for nq in (1, 16):
        xq = torch.rand((nq, d), device="cuda")
        res = {}
        for ns in ("", "1"):
            if ns: os.environ["ISE_NO_SHORT"] = "1"
            else: os.environ.pop("ISE_NO_SHORT", None)
            n.lib.ise_refresh_env_knobs()
            for _ in range(20): index.search_torch(xq, k)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reps = 300
            for _ in range(reps): out = index.search_torch(xq, k)
            torch.cuda.synchronize(); res[ns] = ((time.perf_counter() - t0) / reps * 1e6, out)
        same = torch.equal(res[""][1][1], res["1"][1][1]) and torch.equal(res[""][1][0], res["1"][1][0])
        print(f"{rows}x{d} nq={nq}: short {res[''][0]:7.1f} us  streaming {res['1'][0]:7.1f} us per batch (one stream)  same bits {same}  short_batches {index.short_stats()['short_batches']}", flush=True)
os.environ.pop("ISE_NO_SHORT", None); n.lib.ise_refresh_env_knobs()
