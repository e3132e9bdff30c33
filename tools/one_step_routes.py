"""Diagnostic: `filter()` one step per call (the reference's own calling pattern) at configs[1] on each route of the library --
back to back and with a synchronisation behind every call.  Round 4: one-launch 101 / 129 us per call, three streams 155 / 192, one stream 155 / 187."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vjf_amd
torch.manual_seed(0)
B, dz, dy, n = 4096, 10, 50, 200
g = torch.Generator(device="cuda").manual_seed(1)
T = 300
y = torch.randn(T, B, dy, device="cuda", generator=g); eps = torch.randn(T, 2, B, dz, device="cuda", generator=g)
for ov in (1, 3, 0):
    m = vjf_amd.VJF.make_model(dy, dz, 0, n, [128], likelihood="gaussian", noise="device")
    if ov != 1: m.set_overlap(ov)
    q = None
    for t in range(40): q, _ = m.filter(y[t], None, q, eps=(eps[t, 0], eps[t, 1]))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(40, 240): q, _ = m.filter(y[t], None, q, eps=(eps[t, 0], eps[t, 1]))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    ts = []
    for t in range(240, 280):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        q, _ = m.filter(y[t], None, q, eps=(eps[t, 0], eps[t, 1]))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"overlap {ov} route {m.route()}: filter() back to back {dt*1e6:.1f} us/call; + synchronize {ts[len(ts)//2]*1e6:.1f} us/call; status {m.status()}")
