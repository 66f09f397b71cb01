"""The reference's request path under concurrency (backend/engine.py:68-107: decode -> describe -> search):
threads each run describe(image) + index.search(features, 20) in a loop against a 100k x 2048 index (config 2's
row count at the reference's 2048-d), with the shared forward pass / shared index pass and without."""
import json, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
from image_search_engine_amd import descriptors as ds

def main():
    rng = np.random.default_rng(0)
    desc = ds.CNNDescriptor()
    desc.warm_up()
    n, d, k = 100_000, 2048, 20
    index = faiss.IndexFlatL2(d)
    g = torch.Generator(device="cuda").manual_seed(0)
    index.add_torch(torch.rand((n, d), generator=g, device="cuda") * 3.0)
    imgs = [rng.integers(0, 256, (375, 500, 3), dtype=np.uint8) for _ in range(64)]
    res = {}
    for name, cmax in (("combined", 32), ("one forward pass per call", 0)):
        ds.config.DESCRIBE_COMBINE_MAX = cmax
        rows = []
        for threads in (1, 4, 16, 64):
            per = max(8, 256 // threads)
            lat = []
            start = threading.Barrier(threads + 1)
            def work(i):
                f = desc.describe(imgs[i]); index.search(f.reshape(1, -1).numpy(), k)
                start.wait()
                for _ in range(per):
                    t0 = time.perf_counter()
                    f = desc.describe(imgs[i])
                    index.search(f.reshape(1, -1).numpy(), k)
                    lat.append(time.perf_counter() - t0)
            th = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
            [t.start() for t in th]
            start.wait(); t0 = time.perf_counter()
            [t.join() for t in th]
            el = time.perf_counter() - t0
            lat.sort()
            rows.append({"threads": threads, "requests_per_s": round(threads * per / el, 1),
                         "latency_ms_median": round(lat[len(lat) // 2] * 1e3, 2), "latency_ms_p99": round(lat[int(len(lat) * 0.99)] * 1e3, 2)})
            print(name, rows[-1], flush=True)
        res[name] = rows
    res["describe_combined"] = {"batches": desc.combined_batches, "calls": desc.combined_calls}
    res["index_combined"] = index.host_stats()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "serve_route.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k.endswith("combined")}))

if __name__ == "__main__":
    main()
