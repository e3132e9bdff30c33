#!/usr/bin/env python3
"""BASELINE.json configs[0]: one trial of a Lorenz system, d_z = 3, d_y = 10, Gaussian likelihood -- the counterpart of the
reference's `script/example.py:12-47` (which fits a 2-D limit cycle the same way): make_model -> fit -> forecast.

    python examples/lorenz_fit.py [--epochs 20] [--T 1000] [--plot out.png]

Needs an MI355X (the filtering step runs as HIP kernels); `tests/test_host_cpu.py::test_lorenz_example_plumbing` runs the same
script against the oracle-backed stand-in on the CPU."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--n-rbf", type=int, default=100)
    ap.add_argument("--forecast", type=int, default=200)
    ap.add_argument("--plot", default=None)
    a = ap.parse_args(argv)
    import vjf_amd
    from vjf_amd.data import lorenz, observe

    torch.manual_seed(0)
    xdim, ydim, udim = 3, 10, 0
    g = torch.Generator().manual_seed(1)
    x = lorenz(a.T, generator=g, noise=0.05).to(torch.get_default_dtype())          # (time, 3)
    y, C, d = observe(x, ydim, "gaussian", generator=g)                              # (time, 10)

    model = vjf_amd.VJF.make_model(ydim, xdim, udim=udim, n_rbf=a.n_rbf, hidden_sizes=[20], likelihood="gaussian")
    t0 = time.perf_counter()
    m, logvar, loss = model.fit(y, max_iter=a.epochs)                                # posterior means / log-variances, last epoch's loss
    dt = time.perf_counter() - t0
    m = m.detach().cpu().squeeze(1)
    print(f"fit: {a.epochs} epochs x {a.T} steps in {dt:.2f} s  ({a.epochs * a.T / dt:.0f} trial-timesteps/s), final epoch loss {float(loss):.4f}")
    xf, yf = model.forecast(x0=m[9:10], n_step=a.forecast, noise=False)              # (vjf/model.py:321-324)
    print("forecast:", tuple(xf.shape), tuple(yf.shape), "finite:", bool(torch.isfinite(xf).all() and torch.isfinite(yf).all()))
    if a.plot:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        fig, ax = plt.subplots(1, 3, figsize=(12, 3))
        ax[0].plot(x.numpy()); ax[0].set_title("True state")
        ax[1].plot(m.numpy()); ax[1].set_title("Posterior mean")
        ax[2].plot(xf.detach().cpu().squeeze(1).numpy()); ax[2].set_title("Forecast")
        fig.tight_layout(); fig.savefig(a.plot)
    return m, loss


if __name__ == "__main__":
    main()
