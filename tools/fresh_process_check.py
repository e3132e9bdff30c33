"""Diagnostic: the first one-launch sequence of a FRESH process (cold instruction caches, low clocks, uninitialised workspace) against
the known result of the same tiny model -- run N times in child processes, one after the other.  python tools/fresh_process_check.py [N]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys; sys.path.insert(0, %r)
import numpy as np, torch
from tests.test_dist_gloo import _make, _gpu_data
m = _make(); y, u, eps = _gpu_data()
mu, lv, ls = m.filter_sequence(y.cuda(), u.cuda(), None, eps=eps.cuda())
st = m.status()
m2 = _make(); m2.set_overlap(False)
mu2, lv2, ls2 = m2.filter_sequence(y.cuda(), u.cuda(), None, eps=eps.cuda())
d = float((ls[:, 0] - ls2[:, 0]).abs().max())
if d > 1e-4:
    print("DETAIL one-launch losses", ls.cpu().numpy().tolist(), "per-step", ls2.cpu().numpy().tolist(), flush=True)
    a, b = m.get_state(), m2.get_state()
    def walk(x, y, path):
        if isinstance(x, dict):
            for k in x: walk(x[k], y[k], path + "/" + str(k))
        elif isinstance(x, (list, tuple)):
            for i, (p, q) in enumerate(zip(x, y)): walk(p, q, path + "/" + str(i))
        elif isinstance(x, torch.Tensor):
            print("DETAIL", path, float((x.double() - y.double()).abs().max()), flush=True)
        else:
            print("DETAIL", path, x, y, flush=True)
    walk(a, b, "")
print("RESULT", st, d, ls[:, 0].cpu().numpy().tolist(), "per-step:", ls2[:, 0].cpu().numpy().tolist(), "max |mu - mu2|", float((mu - mu2).abs().max()), flush=True)
''' % ROOT
if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    bad = 0
    for i in range(n):
        out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=120).stdout
        line = [l for l in out.splitlines() if l.startswith("RESULT")]
        print(i, line[0] if line else "no result", flush=True)
        for l in out.splitlines():
            if l.startswith("DETAIL"):
                print("   ", l, flush=True)
        if line:
            parts = line[0].split()
            if int(parts[1]) != 0 or float(parts[2]) > 1e-4:
                bad += 1
    print("deviating runs:", bad, "of", n)
