"""Synthetic data of BASELINE.json's configurations.  The reference ships no generator: `script/example.py:17-33`
draws a noisy 2-D limit cycle inline; BASELINE configs[0] names a Lorenz system, configs[1..4] "RBF data".  Plain torch on
the CPU generator, deterministic under a seed; nothing here touches the device."""
import math

import torch


def lorenz(T: int, *, dt: float = 0.01, burn_in: int = 500, sigma: float = 10.0, rho: float = 28.0, beta: float = 8.0 / 3.0,
           x0=(1.0, 1.0, 1.0), generator: torch.Generator = None, noise: float = 0.0) -> torch.Tensor:
    """Lorenz-63 by classical RK4, `burn_in` steps discarded, each coordinate z-scored: (T, 3) float64.
    SURVEY.md 8d: sigma = 10, rho = 28, beta = 8/3, dt = 0.01, 500 burn-in steps."""
    def f(s):
        x, y, z = s
        return torch.stack((sigma * (y - x), x * (rho - z) - y, x * y - beta * z))
    s = torch.tensor(x0, dtype=torch.float64)
    out = torch.empty(T, 3, dtype=torch.float64)
    for t in range(burn_in + T):
        k1 = f(s); k2 = f(s + 0.5 * dt * k1); k3 = f(s + 0.5 * dt * k2); k4 = f(s + dt * k3)
        s = s + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
        if t >= burn_in:
            out[t - burn_in] = s
    out = (out - out.mean(0)) / out.std(0)
    if noise > 0:
        out = out + noise * torch.randn(out.shape, dtype=torch.float64, generator=generator)
    return out


def observe(x: torch.Tensor, ydim: int, likelihood: str = "gaussian", *, generator: torch.Generator = None, noise: float = 0.1):
    """Linear read-out of latent states as `script/example.py:23-33` draws it: C ~ N(0,1)^{dz x dy}, d ~ N(0,1)^{dy};
    Gaussian: y = x C + d + noise N(0,1); Poisson: y ~ Poisson(exp(0.5 (x C) / sqrt(dz) + d - 1))."""
    dz = x.shape[-1]
    C = torch.randn(dz, ydim, dtype=x.dtype, generator=generator)
    d = torch.randn(ydim, dtype=x.dtype, generator=generator)
    if likelihood == "gaussian":
        y = x @ C + d
        return y + noise * torch.randn(y.shape, dtype=x.dtype, generator=generator), C, d
    rate = torch.exp(0.5 * (x @ C) / math.sqrt(dz) + d - 1.0)
    return torch.poisson(rate, generator=generator), C, d


def rbf_system(T: int, B: int, dz: int, *, n_true: int = 50, generator: torch.Generator = None) -> torch.Tensor:
    """'RBF data' (SURVEY.md 8d): x[t+1] = x[t] + Phi_true(x[t]) W_true + 0.1 xi, 50 centres U(-2,2)^dz of width sqrt(dz),
    W_true ~ 0.1 N(0,1), x[0] ~ N(0,1), independent per trial: (T, B, dz) float32."""
    g = generator
    cen = torch.rand(n_true, dz, generator=g) * 4 - 2
    Wt = 0.1 * torch.randn(n_true, dz, generator=g)
    x = torch.randn(B, dz, generator=g)
    out = torch.empty(T, B, dz)
    for t in range(T):
        d2 = torch.cdist(x, cen) ** 2
        x = x + torch.exp(-0.5 * d2 / float(dz)) @ Wt + 0.1 * torch.randn(B, dz, generator=g)
        out[t] = x
    return out
