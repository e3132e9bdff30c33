"""Diagnostic: phase stamps of the serial kernel at the bench configuration (not a benchmark)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vjf_amd
from vjf_amd import _native as N
torch.manual_seed(0)
B, dz, dy, n = 4096, 10, 50, 200
m = vjf_amd.VJF.make_model(dy, dz, 0, n, [128], likelihood="gaussian", noise="device")
y = torch.randn(16, B, dy, device="cuda"); 
m.filter_sequence(y[:8])
MODE = 2 if len(sys.argv) > 1 and sys.argv[1] == 'overlap' else 1
N.check(m._backend().vjf_debug_stamps(m._ctx, MODE, None))
m.filter_sequence(y[8:])
out = (ctypes.c_uint64 * 32)()
N.check(m._backend().vjf_debug_stamps(m._ctx, 0, out))
t = list(out)[:9]
names = ["load", "chol", "writeL", "dinv", "inverse", "diagcopy+wchol", "solveW", "residual"]
for i, nm in enumerate(names):
    print(f"{nm:16s} {(t[i+1]-t[i]):8d} cycles(100MHz ticks?)")
print("total", t[8]-t[0])
T=list(out)
prev = T[1]
for k in range(7):
    print(f'chol column {k}: {T[9 + k] - prev:7d}')
    prev = T[9 + k]
print('column 0: wave0 panel product', T[8]-T[1], ' panel', T[3]-T[1], ' wave0 trail', T[4]-T[3], ' wave0 chain', T[5]-T[4], ' column end', T[9]-T[3])
for i, nm in enumerate(["prefetch+stage", "forward", "backward", "W store", "sigma tail"]):
    print(f'post y/W workgroup {nm:16s} {T[17 + i] - T[16 + i]:7d}')
names1 = ["stage0 inputs", "stage1 rbf", "stage2 var+mean", "stage3 recognition", "stage4 xt+decoder", "stage5 losses", "stage6 backward", "stage7 rows out"]
for i, nm in enumerate(names1):
    print(f"K1 {nm:20s} {T[23+i]-T[22+i]:8d}")
print("K1 total", T[30]-T[22])
