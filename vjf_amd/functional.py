"""vjf.functional on the GPU: every function forwards to a HIP operator through the C ABI."""
from typing import Union

import torch
from torch import Tensor

from . import _native as N
from .distribution import Gaussian
from .util import at_least2d, dev32, stream_ptr


def rbf(x: Tensor, c: Tensor, w: Tensor) -> Tensor:
    """Gaussian radial basis functions exp(-1/2 (|x - c| / w)^2)   (vjf/functional.py:11-22).
    :param x: (batch, dim)  :param c: (basis, dim)  :param w: width, (basis)"""
    x, c = dev32(x), dev32(c)
    logw = torch.log(dev32(w, ndim2=False).reshape(-1)).contiguous()
    B, d = x.shape
    n = c.shape[0]
    assert c.shape[1] == d and logw.numel() == n
    out = torch.empty(B, n, device=x.device, dtype=torch.float32)
    N.check(N.lib().vjf_rbf_forward(N.ptr(x), N.ptr(c), N.ptr(logw), N.ptr(out), B, n, d, stream_ptr()), "vjf_rbf_forward")
    return out


def gaussian_entropy(q: Gaussian) -> Tensor:
    """1/2 mean_b sum_j logvar   (vjf/functional.py:25-29)"""
    _, logvar = q
    assert logvar.ndim >= 2
    lv = dev32(logvar).reshape(-1, logvar.shape[-1])
    out = torch.empty((), device=lv.device, dtype=torch.float32)
    N.check(N.lib().vjf_gaussian_entropy(N.ptr(lv), N.ptr(out), lv.shape[0], lv.shape[1], stream_ptr()), "vjf_gaussian_entropy")
    return out


def gaussian_loss(a: Union[Tensor, Gaussian], b: Union[Tensor, Gaussian], logvar: Tensor) -> Tensor:
    """Expected negative Gaussian log-likelihood incl. the reference's trace term
    (vjf/functional.py:32-75)."""
    a = at_least2d(a)
    b = at_least2d(b)
    m1, lv1 = (a, None) if isinstance(a, Tensor) else a
    m2, lv2 = (b, None) if isinstance(b, Tensor) else b
    m1, m2 = dev32(m1), dev32(m2)
    lv1 = None if lv1 is None else dev32(lv1)
    lv2 = None if lv2 is None else dev32(lv2)
    assert m1.shape == m2.shape and m1.ndim >= 2
    d = m1.shape[-1]
    B = m1.numel() // d
    lvs = dev32(logvar, ndim2=False).reshape(1)
    out = torch.empty((), device=m1.device, dtype=torch.float32)
    N.check(N.lib().vjf_gaussian_loss(N.ptr(m1), N.ptr(lv1), N.ptr(m2), N.ptr(lv2), N.ptr(lvs), N.ptr(out), B, d,
                                     stream_ptr()), "vjf_gaussian_loss")
    return out


def poisson_loss(eta: Tensor, target: Tensor) -> Tensor:
    """Poisson NLL, log link, eta clamped at 10   (vjf/likelihood.py:51-62)."""
    eta, target = dev32(eta), dev32(target)
    assert eta.shape == target.shape and eta.ndim == 2
    out = torch.empty((), device=eta.device, dtype=torch.float32)
    N.check(N.lib().vjf_poisson_loss(N.ptr(eta), N.ptr(target), N.ptr(out), eta.shape[0], eta.shape[1], stream_ptr()),
            "vjf_poisson_loss")
    return out
