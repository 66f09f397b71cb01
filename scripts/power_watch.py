"""Samples board power of every card for SECONDS (argv[1]) and prints, per card, mean/max over the busiest half
(dev aid: run beside bench.py to see what the timed loop draws).  Touches no GPU API."""
import glob, os, sys, time
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
files = {}
for c in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")):
    files[c.split("/")[4]] = c
acc = {c: [] for c in files}
t0 = time.time()
while time.time() - t0 < secs:
    for c, p in files.items():
        try:
            with open(p) as f:
                acc[c].append(float(f.read()) / 1e6)
        except Exception:
            pass
    time.sleep(0.02)
for c, v in acc.items():
    if v:
        top = sorted(v)[len(v) // 2:]
        print(f"{c}: mean of upper half {sum(top)/len(top):6.0f} W, max {max(v):6.0f} W, min {min(v):6.0f} W, samples {len(v)}")
