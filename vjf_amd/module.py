"""vjf.module on the GPU: RBF features and the Bayesian linear regression on them.

Same classes, constructor arguments, attributes and return types as the reference
(vjf/module.py); the arithmetic runs in HIP kernels behind the C ABI.  Parameters are created
with torch's CPU generator in the reference's order (so a seed reproduces the reference's
initial values bit for bit) and then live on the GPU in fp32.
"""
import math
import warnings
from typing import Union

import torch
from torch import Tensor, nn
from torch.nn import Module, Parameter

from . import _native as N
from .distribution import Gaussian
from .util import dev32, stream_ptr


def _to_dev(t: Tensor) -> Tensor:
    return dev32(t.detach(), ndim2=False)


def rebind(module: Module, name: str, view: Tensor):
    """Move a parameter / plain tensor attribute into `view` (a slice of a VJF state blob)."""
    old = getattr(module, name)
    view.copy_(old.detach().to(view.device, view.dtype).reshape(view.shape))
    if name in module._parameters:
        module._parameters[name] = Parameter(view, requires_grad=False)
    else:
        setattr(module, name, view)


class RBF(Module):
    """Radial basis functions   (vjf/module.py:14-34)"""
    def __init__(self, n_dim: int, n_basis: int, intercept: bool = False, requires_grad: bool = False):
        super().__init__()
        self.n_basis = n_basis
        self.intercept = intercept
        # the HIP path has no autograd: the reference's live path keeps these frozen too (module.py:16)
        self.register_parameter('centroid', Parameter(_to_dev(torch.rand(n_basis, n_dim) * 4 - 2.), requires_grad=False))
        self.register_parameter('logwidth', Parameter(_to_dev(torch.zeros(n_basis)), requires_grad=False))

    @property
    def n_feature(self):
        return self.n_basis + 1 if self.intercept else self.n_basis

    def forward(self, x: Tensor) -> Tensor:
        x = dev32(x)
        B, d = x.shape
        out = torch.empty(B, self.n_basis, device=x.device, dtype=torch.float32)
        N.check(N.lib().vjf_rbf_forward(N.ptr(x), N.ptr(self.centroid), N.ptr(self.logwidth), N.ptr(out), B, self.n_basis, d,
                                        stream_ptr()), "vjf_rbf_forward")
        if self.intercept:
            out = torch.column_stack((torch.ones(B, device=out.device), out))
        return out


class LinearRegression(Module):
    """Bayesian linear regression   (vjf/module.py:37-150)"""
    def __init__(self, feature: Module, n_output: int, bayes=True):
        super().__init__()
        if not bayes:
            raise NotImplementedError("bayes=False needs autograd on w_mean; the HIP path implements the Bayesian form "
                                      "that VJF uses (module.py:49-54)")
        if getattr(feature, "intercept", False):
            raise NotImplementedError("intercept features are not on VJF's path (module.py:16 default False)")
        self.bayes = bayes
        self.add_module('feature', feature)
        self.n_output = n_output
        n = self.feature.n_feature
        self.w_mean = _to_dev(torch.zeros(n, n_output))
        self.w_chol = _to_dev(torch.eye(n))
        self.w_precision = _to_dev(torch.eye(n))
        self.w_pchol = _to_dev(torch.eye(n))
        self._rls_scratch = None
        self._w_colmajor = False

    def _dims(self, x):
        n = self.feature.n_basis
        return x.shape[0], n, x.shape[1], self.n_output

    def forward(self, x: Tensor, sampling=True, noise: Tensor = None) -> Union[Tensor, Gaussian]:
        """Predictive distribution (sampling=False) or a sample (vjf/module.py:56-77).
        `noise` (n_feature, n_output) replaces the randn_like(w) draw of the sampling branch."""
        x = dev32(x)
        B, n, d, dout = self._dims(x)
        c, lw = self.feature.centroid, self.feature.logwidth
        L = N.lib()
        if sampling:
            if noise is None:
                noise = self._draw_weight_noise()
            noise = dev32(noise)
            out = torch.empty(B, dout, device=x.device, dtype=torch.float32)
            scratch = torch.empty(n, dout, device=x.device, dtype=torch.float32)
            N.check(L.vjf_blr_sample(N.ptr(x), N.ptr(c), N.ptr(lw), N.ptr(self.w_mean), N.ptr(self.w_chol), N.ptr(noise),
                                     N.ptr(out), N.ptr(scratch), B, n, d, dout, stream_ptr()), "vjf_blr_sample")
            return out
        mean = torch.empty(B, dout, device=x.device, dtype=torch.float32)
        logvar = torch.empty(B, dout, device=x.device, dtype=torch.float32)
        N.check(L.vjf_blr_predict(N.ptr(x), N.ptr(c), N.ptr(lw), N.ptr(self.w_mean), N.ptr(self.w_chol), N.ptr(mean),
                                  N.ptr(logvar), B, n, d, dout, stream_ptr()), "vjf_blr_predict")
        return Gaussian(mean, logvar)

    def _draw_weight_noise(self) -> Tensor:
        """randn_like(w_mean) from torch's CPU generator.  In the reference w_mean is column-major
        once rls has run (cholesky_solve returns a Fortran-ordered tensor, module.py:101) and
        randn_like then takes torch's non-contiguous (serial, memory-order) path: the draw is
        made on a tensor with the same strides so the same values come out."""
        n, dout = self.w_mean.shape
        if self._w_colmajor:
            # same strides as the reference's w_mean => same (serial, memory-order) normal_ path
            return torch.randn_like(torch.empty(dout, n, dtype=torch.get_default_dtype()).t())
        return torch.randn(n, dout, dtype=torch.get_default_dtype())

    @torch.no_grad()
    def rls(self, x: Tensor, target: Tensor, v: Union[Tensor, float], shrink: float = 1.):
        """RLS weight update   (vjf/module.py:79-112)"""
        x, target = dev32(x), dev32(target)
        B, n, d, dout = self._dims(x)
        assert target.shape == (B, dout)
        L = N.lib()
        import ctypes
        nbytes = ctypes.c_int64()
        N.check(L.vjf_rls_scratch_size(B, n, dout, ctypes.byref(nbytes)), "vjf_rls_scratch_size")
        if self._rls_scratch is None or self._rls_scratch.numel() < nbytes.value:
            self._rls_scratch = torch.empty(nbytes.value, dtype=torch.uint8, device=x.device)
        vt = dev32(v, ndim2=False).reshape(1)
        status = torch.zeros(1, dtype=torch.int32, device=x.device)
        N.check(L.vjf_blr_rls(N.ptr(x), N.ptr(target), N.ptr(vt), float(shrink), N.ptr(self.feature.centroid),
                              N.ptr(self.feature.logwidth), N.ptr(self.w_mean), N.ptr(self.w_chol), N.ptr(self.w_precision),
                              N.ptr(self.w_pchol), N.ptr(self._rls_scratch), N.ptr(status), B, n, d, dout, stream_ptr()),
                "vjf_blr_rls")
        self._w_colmajor = True
        if int(status.item()) != 0:
            warnings.warn('RLS failed.')          # module.py:112; state left unchanged

    @torch.no_grad()
    def kalman(self, x: Tensor, target: Tensor, v: Union[Tensor, float], diffusion: float = 0.):
        """Update weight using Kalman   (vjf/module.py:114-142): w[t] = w[t-1] + Q, target[t] = f(x[t])'w[t] + v, Q = diffusion I.
        The reference's update (vjf/kalman.py:102-145, where S^-1 enters the gain twice) reproduced in n x n algebra on the Gram
        statistics H'H, H'y: the (samples x samples) innovation covariance never exists.  w_chol leaves as the lower Cholesky
        factor of the covariance."""
        assert diffusion >= 0., 'diffusion needs to be non-negative'
        x, target = dev32(x), dev32(target)
        B, n, d, dout = self._dims(x)
        assert target.shape == (B, dout)
        L = N.lib()
        import ctypes
        nbytes = ctypes.c_int64()
        N.check(L.vjf_kalman_scratch_size(B, n, dout, ctypes.byref(nbytes)), "vjf_kalman_scratch_size")
        if self._rls_scratch is None or self._rls_scratch.numel() < nbytes.value:
            self._rls_scratch = torch.empty(nbytes.value, dtype=torch.uint8, device=x.device)
        vt = dev32(v, ndim2=False).reshape(1)
        status = torch.zeros(1, dtype=torch.int32, device=x.device)
        N.check(L.vjf_blr_kalman(N.ptr(x), N.ptr(target), N.ptr(vt), float(diffusion), N.ptr(self.feature.centroid),
                                 N.ptr(self.feature.logwidth), N.ptr(self.w_mean), N.ptr(self.w_chol), N.ptr(self._rls_scratch),
                                 N.ptr(status), B, n, d, dout, stream_ptr()), "vjf_blr_kalman")
        self._w_colmajor = False
        if int(status.item()) != 0:
            warnings.warn('Kalman update failed.')    # covariance / precision not positive definite: state left unchanged

    @torch.no_grad()
    def initialize(self, x: Tensor, target: Tensor, v):
        """vjf/module.py:144-150: centroids ~ U(-r, r), width r, then one RLS."""
        x = dev32(x)
        r = x.norm(dim=1).max().item()
        c = torch.empty(self.feature.centroid.shape, dtype=torch.get_default_dtype())
        nn.init.uniform_(c, a=-r, b=r)            # CPU generator, as the reference draws it
        self.feature.centroid.copy_(c.to(self.feature.centroid.device, torch.float32))
        self.feature.logwidth.fill_(math.log(r))
        self.rls(x, target, v)
