#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (scripts/profile_bench.sh) -> the JSON
bench.py reads `roofline.traffic` from.  usage: scripts/pmc_to_json.py gpurun_out/prof_<tag> profiles/rNN/bench_n<n>_nq<nq>_hbm_pmc.json [n [nq [kernel-name-prefix]]]

Counter handling as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: separate passes;
values are KB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read -- calibrated here on
the 2.048 GB device-to-device copy that index.add_torch makes in the same trace (known byte count) -- so
reads are doubled; WRITE_SIZE is taken as is.  The JSON records the hash of the kernel sources it was
captured on (bench.kernel_source_hash): bench.py reports traffic = null when they have changed since."""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402

prof, out = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 16
kprefix = sys.argv[5] if len(sys.argv) > 5 else "void scan_kernel"
d, k = 512, 10
raw = defaultdict(dict)
for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    files = glob.glob(os.path.join(prof, sub, "*", "*counter_collection.csv"))
    assert files, f"no counter_collection.csv under {prof}/{sub}"
    vals = defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter:
            vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for name, v in vals.items():
        raw[name][counter] = {"launches": len(v), "mean_KB": sum(v) / len(v), "min_KB": min(v), "max_KB": max(v)}
scan = [kname for kname in raw if kname.startswith(kprefix)]
assert len(scan) == 1, scan
copy = raw["__amd_rocclr_copyBuffer"]["FETCH_SIZE"]
known = 4.0 * n * d
ratio = copy["max_KB"] * 1024.0 / known
fetch = raw[scan[0]]["FETCH_SIZE"]["mean_KB"] * 1024.0 / ratio
write = raw[scan[0]]["WRITE_SIZE"]["mean_KB"] * 1024.0
alg = 4.0 * n * d + 4.0 * nq * d + 12.0 * nq * k
rec = {
    "command": "ISE_BENCH_STREAMS=1 scripts/profile_bench.sh <tag> [bench args]: rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE "
               "(separate passes) -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline [bench args]",
    "workload": f"{n}x{d} fp32, L2, k={k}, nq={nq}, 1 GPU",
    "kernel": scan[0],
    "kernel_source_hash": kernel_source_hash(),
    "calibration": {"known_copy_bytes": known, "FETCH_SIZE_bytes_reported": copy["max_KB"] * 1024.0, "ratio": ratio,
                    "note": "hipMemcpy D2D of the index (known byte count) reports 1/2 -> reads are divided by this ratio "
                            "(MI355X_MICROARCH.md, HBM section)"},
    "scan_kernel": {"fetch_bytes_per_launch_corrected": fetch, "write_bytes_per_launch": write,
                    "traffic_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": alg,
                    "traffic_over_algorithmic": (fetch + write) / alg},
    "raw": raw,
}
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec["scan_kernel"]), "ratio", ratio)
