"""Concurrent one-query callers (the reference's serving pattern: one index.search per HTTP request on a
threaded Flask, backend/engine.py:55,137) against one 1M x 512 index: queries/s over the number of caller
threads, with the host API combining concurrent calls into shared passes and without (subprocess per setting:
the knob is read once)."""
import json, os, subprocess, sys, threading, time

def run():
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import image_search_engine_amd.faiss_compat as faiss
    n, d, k = int(os.environ.get("N", "1000000")), 512, 10
    rng = np.random.default_rng(1234)
    index = faiss.IndexFlatL2(d)
    for i0 in range(0, n, 250_000):
        index.add(rng.random((min(250_000, n - i0), d), dtype=np.float32))
    out = []
    for threads in (1, 2, 4, 8, 16, 32, 64):
        qs = [rng.random((1, d), dtype=np.float32) for _ in range(threads)]
        per = max(20, 600 // threads)
        lat = [[] for _ in range(threads)]
        start = threading.Barrier(threads + 1)
        def work(i):
            index.search(qs[i], k)
            start.wait()
            for _ in range(per):
                t0 = time.perf_counter()
                index.search(qs[i], k)
                lat[i].append(time.perf_counter() - t0)
        th = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        [t.start() for t in th]
        start.wait(); t0 = time.perf_counter()
        [t.join() for t in th]
        el = time.perf_counter() - t0
        allv = sorted(v for l in lat for v in l)
        st = index.host_stats()
        out.append({"threads": threads, "qps": round(threads * per / el, 1), "latency_ms_median": round(allv[len(allv) // 2] * 1e3, 3),
                    "latency_ms_p99": round(allv[int(len(allv) * 0.99)] * 1e3, 3), "calls_per_batch_so_far": round(st["combined_calls"] / max(1, st["combined_batches"]), 2)})
        print(out[-1], flush=True)
    return out

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        print("RESULT " + json.dumps(run()))
    else:
        res = {}
        for name, v in (("combined (default, up to 64 queries per pass)", "64"), ("one scan per call (ISE_HOST_COMBINE_MAX=0)", "0")):
            env = dict(os.environ, ISE_HOST_COMBINE_MAX=v)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
            print(name); print("\n".join(l for l in p.stdout.splitlines() if not l.startswith("RESULT")))
            if p.returncode:
                print(p.stderr[-2000:])
            r = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            res[name] = json.loads(r[0][7:]) if r else None
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "serve_threads.json")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        json.dump(res, open(out, "w"), indent=1)
