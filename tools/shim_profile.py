"""Diagnostic: where the host time of a synchronised `filter_sequence` call goes: Python in front of the C call, the C call
(vjf_filter_seq: argument checks, launch), Python behind it, and the wait for the device.    python tools/shim_profile.py [calls] [T]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, vjf_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.manual_seed(0)
m = vjf_amd.VJF.make_model(50, 10, 0, 200, [128], likelihood="gaussian", noise="device")
y = torch.randn(T, 4096, 50, device="cuda"); eps = torch.randn(T, 2, 4096, 10, device="cuda")
mu, lv, _ = m.filter_sequence(y, eps=eps)
q = vjf_amd.Gaussian(mu[-1], lv[-1])
L = m._backend()
real = L.vjf_filter_seq
tc = [0.0, 0.0]
def timed(*a):
    tc[0] = time.perf_counter()
    r = real(*a)
    tc[1] = time.perf_counter()
    return r
class Shim:                                   # (the binding object is a ctypes CDLL: wrap the one attribute)
    def __init__(self, lib): self.__dict__["_l"] = lib
    def __getattr__(self, k): return timed if k == "vjf_filter_seq" else getattr(self._l, k)
m._lib = Shim(L)
rows = []
for i in range(n + 20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mu, lv, _ = m.filter_sequence(y, qs=q, eps=eps)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if i >= 20: rows.append((tc[0] - t0, tc[1] - tc[0], t1 - tc[1], t2 - t1, t2 - t0))
r = np.median(np.array(rows), axis=0) * 1e6
print(f"T={T}: python before the C call {r[0]:.1f} us | vjf_filter_seq {r[1]:.1f} | python behind it {r[2]:.1f} | wait for the device {r[3]:.1f} | whole call {r[4]:.1f}")
print("status", m.check_status())
