"""Board power and clocks while the one-tile scan kernel runs back to back (dev aid; ablate build for the
stream-only / compute-only variants).  Samples rocm-smi's sysfs files from a thread; prints per-variant means."""
import glob, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss

def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except Exception:
        return None

bus = torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else None
print("our device: pci bus id", bus, "| domain", getattr(torch.cuda.get_device_properties(0), "pci_domain_id", None),
      "| device id", getattr(torch.cuda.get_device_properties(0), "pci_device_id", None))
cards = {}
for c in sorted(glob.glob("/sys/class/drm/card*/device")):
    if "-" in c.split("/")[4]:
        continue
    cards[c.split("/")[4]] = os.path.basename(os.path.realpath(c))
print("cards:", {c: a for c, a in cards.items() if a.count(":") == 2})
mine = [c for c, addr in cards.items() if bus is not None and addr.count(":") == 2 and int(addr.split(":")[1], 16) == int(bus)]
print("ours:", mine)
hw = sorted(glob.glob(f"/sys/class/drm/{mine[0]}/device/hwmon/hwmon*")) if mine else []

def sample():
    out = {}
    for h in hw:
        for name in ("power1_input", "freq1_input", "freq2_input", "temp2_input", "power1_cap"):
            v = read(os.path.join(h, name))
            if v is not None:
                out[name] = float(v)
    return out

n, d, nq, k = 1_000_000, 512, 16, 10
xb = torch.rand((n, d), device="cuda"); xq = torch.rand((nq, d), device="cuda")
index = faiss.IndexFlatL2(d); index.add_torch(xb)
print("idle:", sample())
variants = {"full": 0, "stream only": 1 | 8 | 4 | 64, "mfma+lds only": 128 | 1 | 8 | 4, "mfma only (no lds reads)": 128 | 1 | 8 | 4 | 512,
            "loop skeleton": 128 | 64 | 1 | 8 | 4, "full again": 0}
if "ablate" not in os.environ.get("ISE_KNN_LIB", ""):
    variants = {"full": 0}
for nm, abl in variants.items():
    os.environ["ISE_ABLATE"] = str(abl)
    acc, stop = [], False
    def run():
        while not stop:
            acc.append(sample()); time.sleep(0.05)
    th = threading.Thread(target=run); th.start()
    t0 = time.perf_counter(); steps = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(200): index.search_torch(xq, k)
        torch.cuda.synchronize(); steps += 200
    el = time.perf_counter() - t0
    stop = True; th.join()
    late = acc[len(acc) // 2:]
    keys = sorted({kk for a in late for kk in a})
    means = {kk: sum(a.get(kk, 0) for a in late) / len(late) for kk in keys}
    active = {kk: v for kk, v in means.items() if v}
    print(f"{nm:28s}: {el / steps * 1e6:7.1f} us/step   " + "  ".join(f"{kk}={v:.3g}" for kk, v in active.items()))
