"""Diagnostic: when each workgroup of the trial kernel's backward half ends, and where it ran (last even step of a sequence)."""
import ctypes, sys, os, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vjf_amd
from vjf_amd import _native as N
torch.manual_seed(0)
B, dz, dy, n = 4096, 10, 50, 200
m = vjf_amd.VJF.make_model(dy, dz, 0, n, [128], likelihood="gaussian", noise="device")
y = torch.randn(17, B, dy, device="cuda")
m.filter_sequence(y[:8])
N.check(m._backend().vjf_debug_stamps(m._ctx, 2, None))
m.filter_sequence(y[8:])          # 9 steps: the last one is even
rows = []
for k in range(32):
    out = (ctypes.c_uint64 * 32)()
    N.check(m._backend().vjf_debug_stamps(m._ctx, 64 + k, out))
    f = struct.unpack("64f", bytes(out))
    for b in range(8):
        rows.append((f[b * 8 + 7] / 100.0, int(f[b * 8 + 6]), k * 8 + b, f[b * 8 + 5] / 100.0))
rows.sort()
from collections import Counter
place = Counter(r[1] for r in rows)
print("workgroups:", len(rows), " distinct (xcc,se,cu):", len(place), " CUs with 2:", sum(1 for v in place.values() if v > 1))
for t, pl, b, t0 in rows[:3] + rows[-24:]:
    print(f"block {b:4d}  start {t0:6.2f}  end {t:7.2f} us  xcc {pl >> 8} se {(pl >> 4) & 7} cu {pl & 15}  shared_cu={place[pl]}")
