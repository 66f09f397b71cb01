"""10M x 512 on ONE MI355X (BASELINE config 5's row count; 20.5 GB float32 / 10.2 GB bf16): ids against a float64
torch reference computed chunk by chunk on the device, and the time per nq=16 batch.  Rows come from per-chunk seeds,
so the reference regenerates them instead of keeping a second copy."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss

N, d, nq, k, CH = int(os.environ.get("N", "10000000")), 512, 16, 10, 1_000_000
dev = torch.device("cuda")

def chunk(i, unit):
    g = torch.Generator(device=dev).manual_seed(1000 + i)
    x = torch.rand((min(CH, N - i * CH), d), generator=g, device=dev)
    if unit:
        x = x - 0.5
        x = x / x.norm(dim=1, keepdim=True)
    return x

res = {}
for name, metric, storage in (("float32 L2", faiss.METRIC_L2, "f32"), ("bf16 inner product (unit rows)", faiss.METRIC_INNER_PRODUCT, "bf16")):
    unit = storage == "bf16"
    index = faiss.IndexFlat(d, metric, storage=storage)
    t0 = time.time()
    for i in range((N + CH - 1) // CH):
        index.add_torch(chunk(i, unit))
    torch.cuda.synchronize(); t_add = time.time() - t0
    assert index.ntotal == N
    g = torch.Generator(device=dev).manual_seed(7)
    xq = torch.rand((nq, d), generator=g, device=dev)
    if unit:
        xq = xq - 0.5; xq = xq / xq.norm(dim=1, keepdim=True)
    D, I = index.search_torch(xq, k)
    for _ in range(20): index.search_torch(xq, k)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(50): index.search_torch(xq, k)
    torch.cuda.synchronize(); per = (time.time() - t0) / 50
    # float64 reference, chunk by chunk (bf16 storage: on the rounded values, as the index holds them)
    best_d = torch.full((nq, k), float("inf"), dtype=torch.float64, device=dev)
    best_i = torch.full((nq, k), -1, dtype=torch.int64, device=dev)
    q64 = (xq.to(torch.bfloat16).double() if unit else xq.double())
    for i in range((N + CH - 1) // CH):
        x = chunk(i, unit)
        x64 = x.to(torch.bfloat16).double() if unit else x.double()
        if metric == faiss.METRIC_L2:
            s = (q64 * q64).sum(1, keepdim=True) + (x64 * x64).sum(1)[None, :] - 2.0 * q64 @ x64.T
        else:
            s = -(q64 @ x64.T)
        ids = torch.arange(i * CH, i * CH + x.shape[0], device=dev)[None, :].expand(nq, -1)
        cd = torch.cat([best_d, s], 1); ci = torch.cat([best_i, ids], 1)
        o = torch.argsort(cd, dim=1, stable=True)[:, :k]
        best_d, best_i = torch.gather(cd, 1, o), torch.gather(ci, 1, o)
        del x, x64, s
    same = bool(torch.equal(best_i, I))
    recall = float((I[:, :, None] == best_i[:, None, :]).any(2).float().mean())
    got = D.double() if metric == faiss.METRIC_L2 else -D.double()
    err = float((got - best_d).abs().max())
    gb = N * d * (2 if unit else 4) / 1e9
    res[name] = {"n": N, "index_GB": round(gb, 2), "add_s": round(t_add, 2), "ids_identical_to_float64_reference": same,
                 "recall_at_10": recall, "max_abs_score_err": err, "us_per_batch_of_16": round(per * 1e6, 1),
                 "effective_GB_per_s": round(gb / per, 1)}
    print(name, res[name], flush=True)
    del index
    torch.cuda.empty_cache()
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "big_index_probe.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
