"""Diagnostic: host time of one vjf_filter_seq call (enqueue only) and its end-to-end latency for a short sequence, config B.
python tools/launch_overhead.py [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vjf_amd
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B, dz, dy, n = 4096, 10, 50, 200
torch.manual_seed(0)
m = vjf_amd.VJF.make_model(dy, dz, 0, n, [128], likelihood="gaussian", noise="device")
y = torch.randn(T, B, dy, device="cuda"); eps = torch.randn(T, 2, B, dz, device="cuda")
for _ in range(3):
    m.filter_sequence(y, eps=eps)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(30):
    t0 = time.perf_counter()
    m.filter_sequence(y, eps=eps)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
enq.sort(); tot.sort()
print(f"T={T} route={m.route()} enqueue median {enq[15]*1e6:.1f} us  end-to-end median {tot[15]*1e6:.1f} us  (min {tot[0]*1e6:.1f})")
