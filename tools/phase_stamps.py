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
# one clock for all kernels (s_memrealtime, 10 ns ticks): cross-kernel order of the last step, relative to the Cholesky loop's step start
if MODE == 2:
    t0 = T[0]
    ev = [("chol: step start (statistics wait)", T[0]), ("chol: operands + sigma there, chain starts", T[1]), ("chol: factor done", T[2]),
          ("post y/W: start of step", T[16]), ("post y/W: first column staged", T[17]), ("post y/W: forward done, factor good, K1 done", T[18]),
          ("post y/W: backward done", T[19]), ("post y/W: W stored", T[20]),
          ("K1 bwd blk0: start", T[22]), ("K1 bwd blk0: RLS(t-1) there (stage 2 starts)", T[24]), ("K1 bwd blk0: stage 4 done", T[26]),
          ("post y/W: sigma stored", T[21]), ("K1 bwd blk0: end", T[30])]
    for nm, t in sorted(ev, key=lambda e: e[1]):
        print(f"{(t - t0) / 100.0:9.2f} us  {nm}")
# the same for every step of the sequence (ring of 8 stamp sets, indexed by the step's epoch % 8)
if MODE == 2:
    allev = []
    for k in range(8):
        o = (ctypes.c_uint64 * 32)()
        N.check(m._backend().vjf_debug_stamps(m._ctx, 16 + k, o))
        R = list(o)
        for nm, i in (("chol start/stat wait", 0), ("chol chain starts", 1), ("chol factor done", 2), ("post first column", 17),
                      ("post fwd done+K1 done", 18), ("post W stored", 20), ("post sigma stored", 21),
                      ("K1 bwd start", 22), ("K1 bwd: RLS(t-1) there, stage 2 starts", 24), ("K1 bwd blk0 end", 30), ("K1 bwd LAST workgroup ends", 29)):
            if R[i]: allev.append((R[i], k, nm))
    allev.sort()
    tmin = allev[0][0]
    print("---- all steps (ring index = epoch % 8)")
    for t, k, nm in allev:
        print(f"{(t - tmin) / 100.0:9.2f} us  [{k}] {nm}")
    o = (ctypes.c_uint64 * 32)()
    N.check(m._backend().vjf_debug_stamps(m._ctx, 16 + 8, o))
    F = list(o)
    print("---- forward half, block 0 (us):", " ".join(f"{nm}={(F[23 + i] - F[22 + i]) / 100.0:.2f}" for i, nm in enumerate(
        ["inputs", "rbf", "-", "recognition", "xt+post", "sdx2", "-", "rows out"])), f" total={(F[30] - F[22]) / 100.0:.2f}  last workgroup +{(F[29] - F[30]) / 100.0:.2f}")
