"""Diagnostic (not a benchmark): s_memrealtime stamps (one 100 MHz clock for the whole device) of workgroup 0 of every role of the
one-launch route, for the last steps of a sequence at the bench configuration.  python tools/mega_stamps.py [B] [T]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vjf_amd
from vjf_amd import _native as N
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfgC = os.environ.get("CFG", "B") == "C"                   # CFG=C: BASELINE configs[2] (Poisson, d_y = 200)
cfgA = os.environ.get("CFG", "B") == "A"                   # CFG=A: BASELINE configs[0] (one trial, d_z = 3, RBF(100), hidden [20]): pass B = 1
dz, dy, n = (10, 200, 200) if cfgC else (3, 10, 100) if cfgA else (10, 50, 200)
m = vjf_amd.VJF.make_model(dy, dz, 0, n, [20] if cfgA else [128], likelihood="poisson" if cfgC else "gaussian", noise="device")
y = torch.poisson(torch.rand(T + 8, B, dy, device="cuda")) if cfgC else torch.randn(T + 8, B, dy, device="cuda")
_FL_PLACEHOLDER = None
FL = {"train": {}, "warmup": dict(warm_up=True), "infer": dict(sgd=False, update=False), "sgd-only": dict(update=False)}[os.environ.get("FLAGS", "train")]   # FLAGS=warmup|infer|sgd-only: the launches without an RLS update
m.filter_sequence(y[:8], **FL)
N.check(m._backend().vjf_debug_stamps(m._ctx, 2, None))
m.filter_sequence(y[8:], **FL)
torch.cuda.synchronize()
ev = []
TR = ["step start", "theta staged", "features done", "recognition done", "early slab out", "RLS(t-1) there", "var+mean done", "seeds+dxt done",
      "backward+grads done", "late slab out", "moments saved"] + (["gram: may start", "gram: partials out", "gram: reduced",
      "operand: inputs there", "operand: done"] if not os.environ.get("FLAGS") else ["moments role: posterior(t-1) there", "moments role: xs done", "moments role: features done",
      "moments role: variance + mean done", "moments role: moments out, tags posted"]) + [ "sgd: late slabs there", "sgd: done", "trial: step end", "trial: deltas done", "trial: inputs in LDS", "trial: xs done",
      "trial: wave0 variance tiles done", "trial: rec layers done", "trial: heads partials done", "trial: xt/post/decoder done", "trial: early slab stored"]
for t in range(max(0, T - 6), T):
    o = (ctypes.c_uint64 * 32)()
    N.check(m._backend().vjf_debug_stamps(m._ctx, 128 + (t & 31), o))
    R = list(o)
    for i, nm in ((31, "trial: loss terms + seeds done"), (27, "trial: LAST workgroup has theta"), (28, "trial: LAST early slab out"), (29, "trial: LAST late slab out")):
        if R[i]:
            ev.append((R[i], t, nm))
    if R[30] and t >= 32:        # (kept complemented; a slot holds steps t and t - 32: only the later one's minimum is meaningful)
        ev.append(((~R[30]) & 0xffffffffffffffff, t, "trial: FIRST late slab out"))
    for i, nm in enumerate(TR):
        if R[i] and nm:
            ev.append((R[i], t, ("trial: " if i < 11 else "") + nm))
    o = (ctypes.c_uint64 * 32)()
    N.check(m._backend().vjf_debug_stamps(m._ctx, 16 + ((t + 1) & 7), o))      # RLS loops: ring entry = epoch % 8, epoch = step + 1
    R = list(o) if not FL else [0] * 32                                        # (no RLS roles in those launches)
    for nm, i in (("chol: step start (stat wait)", 0), ("chol: operands + sigma there, chain starts", 1), ("chol: factor done", 2), ("chol: column 0 panel done (before the barrier)", 8), ("chol: column 0 panel barrier passed", 3),
                  ("chol: column 0, block (1,1) updated", 4), ("chol: column 0, chain of block (1,1) done", 5),
                  ("y/W: first column staged", 17), ("y/W: forward done, factor good, trial readers done", 18), ("y/W: backward done", 19),
                  ("y/W: W stored", 20), ("y/W: sigma stored", 21), ("inverse loops: LAST one done", 22)):
        if R[i]:
            ev.append((R[i], t, nm))
    for k in range(7):
        if R[9 + k]:
            ev.append((R[9 + k], t, f"chol: column {k} done"))
ev.sort()
t0 = ev[0][0]
for tt, t, nm in ev:
    print(f"{(tt - t0) / 100.0:9.2f} us  [{t}] {nm}")
# the last step, every trial workgroup: times relative to the earliest "theta staged"
rows = []
for j in range(32):
    o = (ctypes.c_uint64 * 32)()
    N.check(m._backend().vjf_debug_stamps(m._ctx, 256 + j, o))
    R = list(o)
    for k in range(4):
        w = R[8 * k:8 * k + 8]
        if w[0]:
            rows.append((4 * j + k, w))
if rows:
    t00 = min(w[0] for _, w in rows)
    print("last step, per trial workgroup (us after the first one had theta): wg xcc rls-at-gate | theta staged, early slab out, RLS there, var+mean done, grads done, late slab out")
    for wgi, w in rows:
        hw = w[6] >> 8                                  # HW_ID: cu_id [11:8], sh_id [12], se_id [15:13] (gfx9 layout)
        print(f"  {wgi:3d} {w[6] & 15:2d} {w[7]} | " + " ".join(f"{(x - t00) / 100.0:7.2f}" for x in w[:6]) + f" | se {(hw >> 13) & 7} sh {(hw >> 12) & 1} cu {(hw >> 8) & 15}")
print("status", m.status())
