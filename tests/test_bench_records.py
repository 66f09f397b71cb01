"""CPU tests of the measurement plumbing around bench.py: the reduction of a rocprofv3 kernel trace to the overlap
record (scripts/overlap_from_trace.py) on a synthetic trace with known answers, and the rule that committed
counter / overlap records are reported only for the kernel sources they were captured on."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _write_trace(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id"])
        w.writerows(rows)


def test_overlap_reduction_on_a_synthetic_trace(tmp_path):
    """100 launches of a 900-ns kernel starting 300 ns apart on four queues in turn (three are resident at any
    time in the steady state) with a 100-ns kernel behind each: the reducer must report exactly that."""
    rows = []
    for i in range(100):
        s = 1000 + 300 * i
        rows.append(["void scan_kernel<4>(P)", s, s + 900, str(1 + i % 4)])
        rows.append(["void merge_kernel<true>(M)", s + 900, s + 1000, str(1 + i % 4)])
    trace = tmp_path / "trace.csv"
    _write_trace(trace, rows)
    out = tmp_path / "ov.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "overlap_from_trace.py"), str(trace),
                        "--kernel", "void scan_kernel", "--json", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rec = json.loads(out.read_text())
    assert rec["kernel"] == "void scan_kernel<4>(P)" and rec["launches"] == 100
    assert abs(rec["mean_duration_us"] - 0.9) < 1e-9 and abs(rec["mean_start_to_start_us"] - 0.3) < 1e-9
    assert rec["hardware_queues"] == ["1", "2", "3", "4"]
    frac = rec["fraction_of_time_with_n_of_this_kernel_resident"]
    assert max(frac, key=frac.get) == "3" and frac["3"] > 0.9      # ramp-up and drain are the rest
    assert rec["other_kernels_in_window"]["void merge_kernel<true>(M)"]["calls"] >= 99
    import bench

    assert rec["kernel_source_hash"] == bench.kernel_source_hash()
    # the default choice of the dominant kernel is the one with the largest total time
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "overlap_from_trace.py"), str(trace)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and json.loads(r.stdout)["kernel"] == "void scan_kernel<4>(P)"


def test_committed_records_are_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """A record is reported only while the sources hash to what it was captured on, only for the configuration
    it describes (one GPU, d = 512, k = 10), and a missing record reads as None, not as an error."""
    import bench

    prof = tmp_path / "profiles" / bench.PROFILE_ROUND
    prof.mkdir(parents=True)
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    (prof / "bench_n1000_nq16_hbm_pmc.json").write_text(json.dumps(
        {"kernel_source_hash": h, "scan_kernel": {"traffic_bytes_per_launch": 12345.0}}))
    (prof / "bench_n1000_nq16_streams16_overlap.json").write_text(json.dumps(
        {"kernel_source_hash": h, "kernel": "k", "launches": 3, "mean_duration_us": 1.0, "mean_start_to_start_us": 0.5,
         "mean_kernels_resident": 2.0, "hardware_queues": ["1"], "window_us": 9.0}))
    (prof / "bench_n2000_nq16_hbm_pmc.json").write_text(json.dumps(
        {"kernel_source_hash": "0" * 16, "scan_kernel": {"traffic_bytes_per_launch": 1.0}}))
    (prof / "bench_n2000_nq16_streams16_overlap.json").write_text(json.dumps({"kernel_source_hash": "0" * 16, "kernel": "k"}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: h)   # (its own ROOT-relative glob is not under test here)
    assert bench.pmc_traffic(1000, 512, 16, 10, 1) == 12345.0
    ov = bench.committed_overlap(1000, 512, 16, 10, 1)
    assert ov["mean_start_to_start_us"] == 0.5 and "window_us" not in ov      # only the documented keys travel
    assert bench.pmc_traffic(2000, 512, 16, 10, 1) is None and bench.committed_overlap(2000, 512, 16, 10, 1) is None  # stale
    assert bench.pmc_traffic(3000, 512, 16, 10, 1) is None and bench.committed_overlap(3000, 512, 16, 10, 1) is None  # absent
    assert bench.pmc_traffic(1000, 512, 16, 10, 2) is None and bench.committed_overlap(1000, 512, 16, 10, 2) is None  # N > 1
    assert bench.pmc_traffic(1000, 256, 16, 10, 1) is None and bench.committed_overlap(1000, 512, 16, 5, 1) is None   # other config


def test_kernel_source_hash_follows_the_sources(tmp_path, monkeypatch):
    import bench

    csrc = tmp_path / "image-search-engine_amd" / "csrc"
    csrc.mkdir(parents=True)
    (tmp_path / "include").mkdir()
    (csrc / "a.hip").write_text("x")
    (csrc / "b.hpp").write_text("y")
    (csrc / "notes.txt").write_text("not a source")
    (tmp_path / "include" / "ise_knn.h").write_text("z")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    h0 = bench.kernel_source_hash()
    (csrc / "notes.txt").write_text("changed")
    assert bench.kernel_source_hash() == h0
    (csrc / "b.hpp").write_text("y2")
    h1 = bench.kernel_source_hash()
    assert h1 != h0
    (tmp_path / "include" / "ise_knn.h").write_text("z2")
    assert bench.kernel_source_hash() not in (h0, h1)


def test_pmc_reduction_calibrates_reads_on_the_known_copy(tmp_path):
    """scripts/pmc_to_json.py on synthetic counter files: FETCH_SIZE is reported in KB and, on gfx950, at half
    the bytes of a wide read -- the device-to-device copy of the index in the same trace (known byte count) gives
    the ratio the scan's reads are divided by; WRITE_SIZE is taken as it is."""
    n, nq, d, k = 1000, 16, 512, 10
    known = 4.0 * n * d
    alg = known + 4.0 * nq * d + 12.0 * nq * k
    kname = "void short_scan_kernel<4, 2, 16, 4, 1, false, true>(ScanParams, ShortParams)"

    def counter_file(sub, counter, rows):
        p = tmp_path / "prof" / sub / "runc"
        p.mkdir(parents=True)
        with open(p / "1_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
            for name, v in rows:
                w.writerow([name, counter, v])

    half = 0.5
    counter_file("pmc_fetch", "FETCH_SIZE",
                 [("__amd_rocclr_copyBuffer", 3.0), ("__amd_rocclr_copyBuffer", known * half / 1024.0)] +
                 [(kname, 1.01 * alg * half / 1024.0)] * 5 + [("void merge_kernel<true>(M, E)", 7.0)] * 5)
    counter_file("pmc_write", "WRITE_SIZE", [(kname, 2.0)] * 5 + [("__amd_rocclr_copyBuffer", known / 1024.0)])
    out = tmp_path / "rec" / "bench_n1000_nq16_hbm_pmc.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_to_json.py"), str(tmp_path / "prof"), str(out),
                        str(n), str(nq), "void short_scan_kernel"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rec = json.loads(out.read_text())
    assert abs(rec["calibration"]["ratio"] - half) < 1e-12
    sk = rec["scan_kernel"]
    assert abs(sk["fetch_bytes_per_launch_corrected"] - 1.01 * alg) < 1e-3 and sk["write_bytes_per_launch"] == 2048.0
    assert abs(sk["algorithmic_bytes_per_launch"] - alg) < 1e-9
    assert abs(sk["traffic_over_algorithmic"] - (1.01 * alg + 2048.0) / alg) < 1e-9
    import bench

    assert rec["kernel"] == kname and rec["kernel_source_hash"] == bench.kernel_source_hash()
