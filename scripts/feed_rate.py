"""how fast can decoded images be fed to the batched CNN?  Writes N synthetic JPEG
(and PNG) files, runs the product's describe_dataset end to end (PIL decode on a thread pool ->
batched resize/normalise + ResNet-50 on the GPU) for several decode pool sizes, and prints images/s
beside the device-only rate of the extractor (resident uint8 tensors).
Reference path: backend/descriptors.py:64-68 (cv2.imread), :104-139 (describe_dataset), batch 1."""
import json, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from image_search_engine_amd import descriptors as ds


def main():
    N = int(os.environ.get("N", "3000"))
    H, W = 375, 500   # a typical photo-collection size; the CNN sees 224 x 224
    rng = np.random.default_rng(0)
    tmp = tempfile.mkdtemp(prefix="ise_feed_")
    ds.config.BOVW_CORNER_DESCRIPTIONS_PATH = type(ds.config.BOVW_CORNER_DESCRIPTIONS_PATH)(os.path.join(tmp, "absent.joblib"))
    yy, xx = np.mgrid[0:H, 0:W]
    paths = []
    t0 = time.time()
    for i in range(N):
        # smooth structure + noise, so that JPEG decode does real entropy decoding
        base = 127 + 90 * np.sin(xx / (7.0 + i % 13) + i) * np.cos(yy / (5.0 + i % 7))
        img = np.clip(base[..., None] + rng.normal(0, 25, (H, W, 3)), 0, 255).astype(np.uint8)
        ext = "png" if i % 10 == 0 else "jpg"
        p = os.path.join(tmp, f"im_{i:05d}.{ext}")
        Image.fromarray(img).save(p, quality=90)
        paths.append(p)
    gen_s = time.time() - t0
    arr = np.array(paths).reshape(-1, 1)
    res = {"n_images": N, "image_size": [H, W], "formats": "90% JPEG q90, 10% PNG", "host_cpus": len(os.sched_getaffinity(0)),
           "generate_s": round(gen_s, 1), "runs": []}
    modes = os.environ.get("MODES", "threads,procs,decode").split(",")
    for dtype in (torch.float32, torch.bfloat16) if "threads" in modes else ():
        desc = ds.CNNDescriptor(dtype=dtype)
        # device-only rate: resident uint8 batches
        x = torch.randint(0, 256, (256, 224, 224, 3), dtype=torch.uint8, device="cuda")
        for _ in range(3): desc.extract_features_tensor(x)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10): desc.extract_features_tensor(x)
        torch.cuda.synchronize(); dev_rate = 2560 / (time.time() - t0)
        for workers in (1, 2, 4, 8, 16, 32):
            ds.config.DECODE_WORKERS = workers
            describer = ds.Describer({"conv_features": desc}, batch_size=128)
            ds.describe_dataset(describer, arr[:256])            # warm-up
            t0 = time.time()
            out = ds.describe_dataset(describer, arr)
            dt = time.time() - t0
            assert len(out) == N and len(describer.described_paths) == N
            res["runs"].append({"dtype": str(dtype).split(".")[-1], "decode_workers": workers,
                                "images_per_s": round(N / dt, 1), "device_only_images_per_s": round(dev_rate, 1)})
            print(res["runs"][-1], flush=True)
    # the same with a pool of spawned decode processes (config.DECODE_PROCESSES)
    for dtype in (torch.float32, torch.bfloat16) if "procs" in modes else ():
        desc = ds.CNNDescriptor(dtype=dtype)
        for procs in (4, 8, 16):
            ds.config.DECODE_PROCESSES = procs
            describer = ds.Describer({"conv_features": desc}, batch_size=128)
            ds.describe_dataset(describer, arr[:512])            # starts the workers
            for rnd in range(2):                                  # the host CPUs are shared: alternate, twice
                for asyn, slot in ((True, 3 << 20), (False, 3 << 20), (True, 0)):
                    ds.config.DESCRIBE_ASYNC, ds.config.DECODE_SLOT_BYTES = asyn, slot
                    t0 = time.time(); c0 = time.process_time()
                    out = ds.describe_dataset(describer, arr)
                    dt = time.time() - t0
                    assert len(out) == N and len(describer.described_paths) == N
                    res["runs"].append({"dtype": str(dtype).split(".")[-1], "decode_processes": procs, "async": asyn,
                                        "slot_bytes": slot, "images_per_s": round(N / dt, 1),
                                        "parent_cpu_us_per_image": round(1e6 * (time.process_time() - c0) / N)})
                    print(res["runs"][-1], flush=True)
            ds.config.DESCRIBE_ASYNC, ds.config.DECODE_SLOT_BYTES = True, 3 << 20
            describer.close()
    ds.config.DECODE_PROCESSES = 0
    # decode alone (no GPU): what the thread pool delivers
    from concurrent.futures import ThreadPoolExecutor
    d = ds.Describer({"x": object()})
    for workers in (1, 8, 16, 32) if "decode" in modes else ():
        t0 = time.time()
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(d.read_image, paths[:1500]))
        res.setdefault("decode_only", []).append({"workers": workers, "images_per_s": round(1500 / (time.time() - t0), 1)})
    print(json.dumps(res))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "feed_rate.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":  # the decode processes are spawned: they import this file
    main()
