"""Is the one-tile scan power-bound?  The same kernel, same bytes, on index contents that toggle fewer wires
(all zeros / one repeated row) against uniform random rows; isolated kernel time, board power, in-kernel clock."""
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

bus = torch.cuda.get_device_properties(0).pci_bus_id
cards = {c.split("/")[4]: os.path.basename(os.path.realpath(c)) for c in glob.glob("/sys/class/drm/card*/device")}
mine = [c for c, a in cards.items() if a.count(":") == 2 and int(a.split(":")[1], 16) == int(bus)]
hw = sorted(glob.glob(f"/sys/class/drm/{mine[0]}/device/hwmon/hwmon*")) if mine else []
def power():
    v = read(os.path.join(hw[0], "power1_input")) if hw else None
    return float(v) / 1e6 if v else float("nan")

n, d, nq, k = 1_000_000, 512, 16, 10
g = torch.Generator(device="cuda").manual_seed(1)
xq = torch.rand((nq, d), generator=g, device="cuda")
contents = {
    "uniform random": lambda: torch.rand((n, d), generator=g, device="cuda"),
    "all zeros": lambda: torch.zeros((n, d), device="cuda"),
    "one row repeated": lambda: torch.rand((1, d), generator=g, device="cuda").expand(n, d).contiguous(),
    "random, 8 mantissa bits": lambda: (torch.rand((n, d), generator=g, device="cuda").to(torch.bfloat16).to(torch.float32)),
    "uniform random again": lambda: torch.rand((n, d), generator=g, device="cuda"),
}
for nm, make in contents.items():
    xb = make()
    index = faiss.IndexFlatL2(d); index.add_torch(xb)
    for _ in range(300): index.search_torch(xq, k)
    torch.cuda.synchronize()
    acc, stop = [], False
    def run():
        while not stop:
            acc.append(power()); time.sleep(0.02)
    th = threading.Thread(target=run); th.start()
    t0 = time.perf_counter(); steps = 0
    while time.perf_counter() - t0 < 3.0:
        for _ in range(200): index.search_torch(xq, k)
        torch.cuda.synchronize(); steps += 200
    el = time.perf_counter() - t0
    stop = True; th.join()
    late = acc[len(acc) // 2:]
    _, _, a, b = index.search_timed_torch(xq, k, 100)
    print(f"{nm:26s}: isolated scan {a*1e3:6.1f} us (+{b*1e3:4.1f})  back-to-back {el/steps*1e6:6.1f} us/step  "
          f"board power {sum(late)/len(late):6.0f} W", flush=True)
    del index, xb
