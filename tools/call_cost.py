"""Fixed cost of a call: wall time of single-step `filter()` calls and of 20-step `filter_sequence` calls at the bench
configuration, against the per-step time of a 200-step call.   python tools/call_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vjf_amd
torch.manual_seed(0)
B, dz, dy, n = 4096, 10, 50, 200
m = vjf_amd.VJF.make_model(dy, dz, 0, n, [128], likelihood="gaussian", noise="device")
T = 400
y = torch.randn(T, B, dy, device="cuda"); eps = torch.randn(T, 2, B, dz, device="cuda")
mu, lv, _ = m.filter_sequence(y[:20], eps=eps[:20]); q = vjf_amd.Gaussian(mu[-1], lv[-1])
def wall(f, reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
t = [20]
def one_step():
    global q
    q, _ = m.filter(y[t[0]], None, q, eps=(eps[t[0], 0], eps[t[0], 1])); t[0] += 1
def one_step_sync():
    one_step(); torch.cuda.synchronize()
print("filter(), back to back (asynchronous): %.1f us per call" % wall(one_step, 100))
print("filter() + synchronize:                %.1f us per call" % wall(one_step_sync, 100))
def seq(k):
    def f():
        global q
        lo = t[0]; mu, lv, _ = m.filter_sequence(y[lo:lo + k], qs=q, eps=eps[lo:lo + k]); q = vjf_amd.Gaussian(mu[-1], lv[-1]); t[0] = 220 if lo + 2 * k > T else lo + k
    return f
t[0] = 220
w20 = wall(seq(20), 8)
t[0] = 220
def seq_sync(k):
    g = seq(k)
    def f(): g(); torch.cuda.synchronize()
    return f
w20s = wall(seq_sync(20), 8)
t[0] = 200
w200 = wall(seq_sync(200), 1)
print("filter_sequence(20 steps) back to back: %.1f us per step; + synchronize: %.1f us per step;  200 steps: %.1f us per step" % (w20 / 20, w20s / 20, w200 / 200))
print("fixed cost of a 20-step call (synchronised) beside 20 x the 200-step rate: %.1f us" % (w20s - 20 * w200 / 200))
print("status", m.status())
