#!/usr/bin/env python3
"""Headline benchmark: queries/sec of exact brute-force L2 kNN, 1M x 512 fp32, k=10.

  python bench.py --gpus 1 --steps K --warmup W            (default)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step = one search batch of --nq queries (default 16, the HBM-bound headline
batch of SURVEY.md 8d) against the whole index, inputs resident in HBM.  With N
GPUs the 1M rows are row-sharded (strong scaling) and a step also contains the
one all-gather + merge.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3  # dense fp32 MFMA peak, same guide


def make_inputs(n, d, nq, lo, hi):
    """BASELINE.md section 4 inputs; only rows [lo, hi) are materialised."""
    rng = np.random.default_rng(1234)
    out = np.empty((hi - lo, d), dtype=np.float32)
    step = 1 << 16
    for s in range(0, n, step):
        e = min(n, s + step)
        blk = rng.random((e - s, d), dtype=np.float32)
        a, b = max(s, lo), min(e, hi)
        if a < b:
            out[a - lo: b - lo] = blk[a - s: b - s]
        if e >= hi:
            break
    xq = np.random.default_rng(4321).random((nq, d), dtype=np.float32)
    return out, xq


def host_cores():
    """Threads this process may really run: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


PROFILE_ROUND = "r03"
CLOCK_WARMUP_STEPS = 400  # untimed, before the W warm-up steps: see main()


def kernel_source_hash():
    """sha256 over the kernel sources the library is built from (csrc/*.hip, *.hpp, the C header):
    what a committed counter capture is tied to."""
    import glob
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "image-search-engine_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")))
    files.append(os.path.join(ROOT, "include", "ise_knn.h"))
    for p in files:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(n, d, nq, k, world):
    """HBM bytes per scan launch from the committed rocprofv3 PMC passes of this same command
    (profiles/<round>/bench_n<n>_nq<nq>_hbm_pmc.json; counters cannot be read from inside the timed
    process).  The capture names the kernel sources it was taken on: if they have changed since,
    the figure is stale and None is reported instead."""
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND, f"bench_n{n}_nq{nq}_hbm_pmc.json")
    if world != 1 or (d, k) != (512, 10) or not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return None
    return rec["scan_kernel"]["traffic_bytes_per_launch"]


def committed_overlap(n, d, nq, k, world):
    """How the scan kernels of consecutive steps overlap on the GPU in THIS command's timed configuration
    (16 issue streams): reduced from a `rocprofv3 --kernel-trace` of the same command by
    scripts/overlap_from_trace.py and committed under profiles/<round>/ (a process cannot trace itself).
    Tied to the kernel sources like the counter capture: stale captures are not reported."""
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND, f"bench_n{n}_nq{nq}_streams16_overlap.json")
    if world != 1 or d != 512 or k != 10 or not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return None
    keep = ("kernel", "launches", "mean_duration_us", "std_duration_us", "mean_start_to_start_us",
            "fraction_of_time_with_n_of_this_kernel_resident", "mean_kernels_resident", "hardware_queues")
    return {key: rec[key] for key in keep if key in rec}


class BoardPower:
    """Board power of the card this rank runs on, read from the amdgpu hwmon files while the steps run
    (a thread, one read per 10 ms; no GPU API involved).  The one-tile scan runs at the board's power cap
    (DESIGN.md 5): this is the evidence for the run the line comes from.  Everything here is best effort:
    unreadable files give ``None``."""

    def __init__(self, device_index):
        import glob
        import threading

        self.samples, self.cap, self._stop, self._thread, self._file = [], None, False, None, None
        try:
            bus = int(torch.cuda.get_device_properties(device_index).pci_bus_id)
            for c in glob.glob("/sys/class/drm/card*/device"):
                addr = os.path.basename(os.path.realpath(c))
                if addr.count(":") == 2 and int(addr.split(":")[1], 16) == bus:
                    hw = sorted(glob.glob(os.path.join(c, "hwmon", "hwmon*")))
                    if hw and os.path.exists(os.path.join(hw[0], "power1_input")):
                        self._file = os.path.join(hw[0], "power1_input")
                        capf = os.path.join(hw[0], "power1_cap")
                        if os.path.exists(capf):
                            with open(capf) as f:
                                self.cap = float(f.read()) / 1e6
            if self._file:
                self._thread = threading.Thread(target=self._run, daemon=True)
                self._thread.start()
        except Exception:  # noqa: BLE001
            self._file = None

    def _run(self):
        while not self._stop:
            try:
                with open(self._file) as f:
                    self.samples.append((time.perf_counter(), float(f.read()) / 1e6))
            except Exception:  # noqa: BLE001
                pass
            time.sleep(0.01)

    def stop(self, t_from):
        """mean / max of the samples taken since ``t_from`` (perf_counter), or None."""
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=1.0)
        v = [w for t, w in self.samples if t >= t_from]
        if not v:
            return None
        return {"mean_w": round(sum(v) / len(v), 1), "max_w": round(max(v), 1), "cap_w": self.cap, "samples": len(v),
                "what": "hwmon power1_input of this card, one read per 10 ms over the last 0.8 s of 2 s of the same "
                        "steps issued back to back after the timed region (the reading is a moving average)"}


def cpu_baseline(xb, xq, k, budget_s=12.0):
    """Faiss-equivalent CPU restatement (oracle/flat_oracle.c, kind 'port') timed on this
    host's cores on a bounded sample: the same query batch against the first `rows` index
    rows, repeated until ~budget_s of CPU work; QPS is scaled to the full index."""
    from oracle import flat_oracle as fo

    fo.build()
    cores = min(fo.max_threads(), host_cores())
    n = xb.shape[0]
    rows = min(n, 250_000)
    sample = np.ascontiguousarray(xb[:rows])
    fo.knn_flat(sample[:1000], xq, k, 1, cores)  # warm the thread pool
    t0 = time.perf_counter()
    reps = 0
    while True:
        D, I, nt = fo.knn_flat(sample, xq, k, 1, cores)
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 50:
            break
    per_batch = el / reps * (n / rows)
    return {
        "value": xq.shape[0] / per_batch,
        "unit": "queries/s",
        "cores": int(nt),
        "kind": "port",
        "sample": f"nq={xq.shape[0]} batch vs first {rows} of {n} rows x{reps} reps, scaled by {n / rows:.1f} "
                  f"(oracle/flat_oracle.c: Faiss small-batch scan restated, {nt} OpenMP threads)",
    }


def main():
    # stdout carries exactly one line, the JSON result: whatever libraries print there on the way
    # (RCCL writes a version banner to stdout when a communicator is created) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=512)
    ap.add_argument("--nq", type=int, default=16)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if not 1 <= args.k <= 32:
        raise SystemExit("bench.py times one scan pass per batch: 1 <= k <= 32 (larger k is covered by the tests)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # rehearsal knobs (one-GPU box): ISE_BENCH_SHARE_GPU=1 puts every rank on cuda:0,
    # ISE_BENCH_BACKEND=gloo swaps RCCL for gloo; the driver's runs use neither
    if os.environ.get("ISE_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import image_search_engine_amd.faiss_compat as faiss

    n, d, nq, k = args.n, args.d, args.nq, args.k
    lo, hi = n * rank // world, n * (rank + 1) // world
    xb_host, xq_host = make_inputs(n, d, nq, lo, hi)
    xq = torch.from_numpy(xq_host).to(dev)

    # Independent batches are issued round-robin onto a few HIP streams so that the
    # latency phases of one batch (query staging, threshold boot, final selection, merge,
    # and for N > 1 the all-gather) overlap the streaming phase of the next one.
    # 16 streams over the library's six workspace slots: at most six scans in flight, chained slot to
    # slot on the GPU, the next ones already queued (303 us per step against 328 with 4 streams)
    n_streams = max(1, int(os.environ.get("ISE_BENCH_STREAMS", "16")))
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]

    # rehearsal knob: ISE_BENCH_FORCE_SHARDED=1 runs the N > 1 code path (shard scan, all-gather,
    # merge) in a world of one, to time its host side on a one-GPU box
    sharded = world > 1 or os.environ.get("ISE_BENCH_FORCE_SHARDED") == "1"
    if sharded:
        import torch.distributed as dist
        from image_search_engine_amd.sharded import SearchPipeline, ShardedIndexFlat

        backend = os.environ.get("ISE_BENCH_BACKEND", "nccl")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        index = ShardedIndexFlat(d, faiss.METRIC_L2)
        index.add_local(torch.from_numpy(xb_host).to(dev))
        local = index.backend.index

        # batches in flight: the shard scans of `depth` consecutive steps share one all-gather + merge
        # (bucketed collective), a bucket per stream, four buckets in flight; results come back one
        # bucket late, everything is drained inside the timed region
        depth = max(1, int(os.environ.get("ISE_BENCH_BUCKET", "4")))
        n_buckets = max(2, int(os.environ.get("ISE_BENCH_BUCKETS", "4")))
        pipe = SearchPipeline(index, nq, k, depth=depth, buckets=n_buckets)

        def run(steps):
            out = None
            for _ in range(steps):
                done = pipe.submit(xq, xq_ready=True)  # xq is resident before the timed region
                if done:
                    out = done[-1]
            done = pipe.flush()
            return done[-1] if done else out

        def barrier():
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
    else:
        index = faiss.IndexFlatL2(d)
        index.add_torch(torch.from_numpy(xb_host).to(dev))
        local = index

        outs = [(torch.empty((nq, k), dtype=torch.float32, device=dev),
                 torch.empty((nq, k), dtype=torch.int64, device=dev)) for _ in range(n_streams)]
        handles = [s.cuda_stream for s in streams]

        def run(steps):
            # outputs are a ring of caller-owned buffers, the stream is passed by handle: the
            # timed loop is one C call per step
            for i in range(steps):
                j = i % n_streams
                index.search_into(xq, k, outs[j][0], outs[j][1], handles[j])
            return outs[(steps - 1) % n_streams]

        def barrier():
            pass

    # every workspace slot of the library is sized before the first step (a serving process does the
    # same): no allocation, fill or device-wide synchronisation can fall into the timed region however
    # short the warm-up is
    local.reserve(nq, k)
    torch.cuda.synchronize()
    # The GPU leaves its idle power state over the first tens of milliseconds of work (measured,
    # scripts/short_run_probe.py: the same 20 steps take 400 us each right after the index build, 315
    # after 30 ms of searches, 380 again after half a second of idling).  A server is never idle, so
    # the clocks are brought up with untimed steps of the same kind before the W warm-up steps; the
    # timed region itself is untouched (exactly K steps, nothing skipped).
    run(CLOCK_WARMUP_STEPS)
    torch.cuda.synchronize()
    run(max(1, args.warmup))
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    D, I = run(args.steps)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # roofline of the dominant kernel (the shard-local scan), timed with HIP events on
    # the stream it runs on, inside the library
    n_local = hi - lo
    _, _, scan_ms, merge_ms = local.search_timed_torch(xq, k, 50 if nq <= 64 else 10)
    alg_bytes = 4.0 * n_local * d + 4.0 * nq * d + 12.0 * nq * k
    achieved = alg_bytes / (scan_ms * 1e-3) / 1e9
    # batches of >= 256 queries take the GEMM-shaped pass (csrc/ise_gemm_scan.hpp): MFMA-bound, priced in flops
    gemm_batch = local.exact_stats().get("gemm_chunks", 0) > 0
    alg_flops = 2.0 * nq * n_local * d
    # which kernel scanned the rows (the dominant kernel of the step)
    if gemm_batch:
        kernel_name = "gemm_scan_kernel (threshold sample + main pass)"
    elif local.host_stats()["direct_queries"] > 0:
        kernel_name = "exact_scan_kernel<1, nt> (one-query batches: the direct-difference scan is the whole search)"
    elif local.short_stats()["short_batches"] > 0:
        kernel_name = "short_scan_kernel (short index: scores dumped to LDS, one selection per block)"
    else:
        kernel_name = "scan_kernel"

    # per-batch latency (median + p10/p90): one batch at a time on one stream, an
    # event pair around scan + merge.  Single-GPU runs only; the timed region above is the metric.
    latency = None
    if not sharded:
        s0 = streams[0]
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
        for e0, e1 in evs:
            e0.record(s0)
            index.search_into(xq, k, outs[0][0], outs[0][1], s0.cuda_stream)
            e1.record(s0)
        torch.cuda.synchronize()
        us = np.sort(np.array([e0.elapsed_time(e1) * 1e3 for e0, e1 in evs]))
        latency = {"p10": float(us[10]), "median": float(us[50]), "p90": float(us[90]),
                   "what": "one batch at a time on one stream: scan + merge, HIP event pair per batch, 100 batches"}
    # board power under the same steps, sustained (untimed, after everything that is timed): the hwmon figure is
    # a moving average that needs about a second of load to settle, the K timed steps are over in milliseconds
    board_power = None
    if not sharded and nq <= 64:
        power = BoardPower(dev.index if dev.index is not None else 0)
        t_seg = time.perf_counter()
        while time.perf_counter() - t_seg < 2.0:
            run(200)
            torch.cuda.synchronize()
        board_power = power.stop(t_seg + 1.2)

    # N > 1: what a SCALE record needs to describe itself -- every rank's own shard roofline, the collective in
    # use and the ranks it reached, and the check that all ranks issued the same sequence of all-gathers.
    # Then the communicator and the process group go away BEFORE rank 0 regenerates the index and runs the
    # CPU oracle pass (15-20 s): the other ranks are not parked in a barrier meanwhile.
    multi = None
    D_host, I_host = D.cpu(), I.cpu()
    if sharded:
        mine = {"rank": rank, "rows": n_local, "kernel_ms": scan_ms, "achieved_gbs": achieved,
                "frac": achieved / HBM_PEAK_GBS}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        order = index.check_collective_order()   # raises on every rank if the sequences differ
        multi = {"per_rank_roofline": per_rank, "collective": index.collective, "ranks_seen": index.ranks_seen,
                 "collective_order": order}
        if index.comm is not None:  # the library's communicator goes before the process group that bootstrapped it
            torch.cuda.synchronize()
            barrier()
            index.comm.close()
        dist.destroy_process_group()

    if rank == 0:
        res = {
            "metric": "queries/sec, exact brute-force L2 kNN (recall@10 vs exact CPU), 1Mx512 fp32 index, k=10",
            "value": nq * args.steps / el,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{n}x{d} fp32 uniform[0,1) index (default_rng 1234), L2, k={k}, "
                                   f"nq={nq} queries per step, index resident in HBM, "
                                   + (f"row-sharded over {world} GPUs, one all-gather ({index.collective}) + merge per {depth} "
                                      f"steps, {n_buckets} buckets (streams) in flight" if sharded else
                                      f"steps issued round-robin on {n_streams} HIP streams")
                                   + f"; {CLOCK_WARMUP_STEPS} untimed steps ahead of the warm-up bring the GPU out of "
                                     f"its idle power state",
                       "n": n, "d": d, "k": k, "nq": nq},
            "roofline": {"bound": "mfma" if gemm_batch else "hbm",
                         "achieved": alg_flops / (scan_ms * 1e-3) / 1e12 if gemm_batch else achieved,
                         "peak": MFMA_F32_PEAK_TF if gemm_batch else HBM_PEAK_GBS,
                         "unit": "TFLOP/s" if gemm_batch else "GB/s",
                         "frac": (alg_flops / (scan_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF) if gemm_batch
                         else achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(n, d, nq, k, world),
                         "kernel": kernel_name,
                         "kernel_overlap": committed_overlap(n, d, nq, k, world),
                         "kernel_ms": scan_ms, "merge_kernel_ms": merge_ms, "board_power": board_power,
                         "algorithmic_bytes": alg_bytes, "algorithmic_flops": alg_flops,
                         # the same bytes over the measured step time (batches overlapped on the GPU):
                         # what the whole step sustains, beside the isolated kernel's figure above
                         "step_effective": {"achieved": alg_bytes / (el / args.steps) / 1e9,
                                            "frac": alg_bytes / (el / args.steps) / 1e9 / HBM_PEAK_GBS},
                         "kernel_ms_source": "50 back-to-back launches on one stream, HIP events around the kernel "
                                             "(ise_index_search_timed_device); merge_kernel_ms = everything behind "
                                             "the scan (merge + exact re-rank + the gated exact-scan launches); "
                                             f"rocprofv3 agreement: profiles/{PROFILE_ROUND}/ (kernel_stats CSVs, README.md)"},
        }
        if latency is not None:
            res["batch_latency_us"] = latency
        if multi is not None:
            res["multi_gpu"] = multi
        if not args.no_cpu_baseline:
            from oracle import flat_oracle as fo

            fo.build()
            cores = min(fo.max_threads(), host_cores())
            if world == 1:
                cb = cpu_baseline(xb_host, xq_host, k)
                res["cpu_baseline"] = cb
                xb_full = xb_host
            else:  # rank 0 holds one shard: the gate needs the whole index once
                xb_full, _ = make_inputs(n, d, nq, 0, n)
            # parity gate on the benchmark data itself, at every N: full-index CPU pass (oracle) for the
            # batch the last timed step answered
            Dc, Ic, _ = fo.knn_flat(xb_full, xq_host, k, 1, cores)
            In, Dn = I_host.numpy(), D_host.numpy()
            res["recall_at_k"] = float(np.mean([len(set(In[q]) & set(Ic[q])) / k for q in range(nq)]))
            res["ids_identical"] = bool(np.array_equal(In, Ic))
            res["max_abs_dist_err"] = float(np.abs(Dn - Dc).max())
            st = local.exact_stats()
            res["exact_path"] = {"queries_reranked": st["reranked"], "queries_sent_to_exact_scan": st["exact_scan"]}
            # the gate (north_star: identical ids, distances within 1e-4 fp32): the tests' checker --
            # ids bit-exact except where two rows' float64 distances are closer than float32 resolves
            # (either order is then a correct float32 answer; the C oracle and the GPU sum in different
            # orders), distances within the ABSOLUTE 1e-4
            from tests.knn_checks import assert_knn_matches

            try:
                res["id_mismatches_at_float32_ties"] = assert_knn_matches(Dn, In, Dc, Ic, xb_full, xq_host, 1, atol=1e-4)
            except AssertionError as e:
                sys.stderr.write(f"bench.py: PARITY GATE FAILED ({e}): " + json.dumps(res) + "\n")
                raise SystemExit(3)
        if sharded and world == 1:
            res["config"]["workload"] += " [rehearsal: sharded code path in a world of one]"
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())


if __name__ == "__main__":
    main()
