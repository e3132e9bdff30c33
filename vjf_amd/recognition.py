"""vjf.recognition on the GPU   (mirror of vjf/recognition.py)."""
import ctypes
from typing import Sequence, Union

import torch
from torch import Tensor
from torch.nn import Linear, Module, Sequential, Tanh

from . import _native as N
from .distribution import Gaussian
from .util import dev32, stream_ptr

__all__ = ['Recognition']


class Recognition(Module):
    """tanh MLP on [y, u, mean, logvar] -> Gaussian(mean head without bias, logvar head with bias)
    (vjf/recognition.py:16-42).  Layers are torch Linear modules so that a seed gives the
    reference's initial weights and state_dict keys; their forward is never used -- the forward
    pass is the HIP operator."""
    def __init__(self, ydim: int, xdim: int, udim: int, hidden_sizes: Sequence[int], activation=Tanh):
        super().__init__()
        if activation is not Tanh:
            raise NotImplementedError("the HIP recognition kernels implement the reference's default Tanh")
        self.ydim, self.xdim, self.udim = ydim, xdim, udim
        self.hidden_sizes = [int(h) for h in hidden_sizes]
        layers = [Linear(ydim + udim + 2 * xdim, hidden_sizes[0]), activation()]
        for k in range(len(hidden_sizes) - 1):
            layers.append(Linear(hidden_sizes[k], hidden_sizes[k + 1]))
            layers.append(activation())
        self.add_module('mlp', Sequential(*layers))
        self.add_module('mean', Linear(hidden_sizes[-1], xdim, bias=False))
        self.add_module('logvar', Linear(hidden_sizes[-1], xdim, bias=True))
        for p in self.parameters():
            p.requires_grad_(False)
            p.data = dev32(p.data, ndim2=False)

    def linears(self):
        return [m for m in self.mlp if isinstance(m, Linear)]

    def forward(self, y: Tensor, xs: Union[Tensor, Gaussian], u: Tensor = None) -> Gaussian:
        if isinstance(xs, Tensor):
            raise NotImplementedError("Recognition on a point state is not used by VJF (model.py:116 passes a Gaussian)")
        elif not isinstance(xs, Gaussian):
            raise TypeError
        y = dev32(y)
        mu_s, lv_s = dev32(xs.mean), dev32(xs.logvar)
        u = None if (u is None or self.udim == 0) else dev32(u)
        B = y.shape[0]
        lins = self.linears()
        L = len(lins)
        Wp = (ctypes.c_void_p * L)(*[l.weight.data_ptr() for l in lins])
        bp = (ctypes.c_void_p * L)(*[l.bias.data_ptr() for l in lins])
        hid = (ctypes.c_int32 * L)(*self.hidden_sizes)
        mu_t = torch.empty(B, self.xdim, device=y.device, dtype=torch.float32)
        lv_t = torch.empty(B, self.xdim, device=y.device, dtype=torch.float32)
        N.check(N.lib().vjf_recognition_forward(N.ptr(y), N.ptr(u), N.ptr(mu_s), N.ptr(lv_s), Wp, bp, N.ptr(self.mean.weight),
                                                N.ptr(self.logvar.weight), N.ptr(self.logvar.bias), N.ptr(mu_t), N.ptr(lv_t),
                                                B, self.ydim, self.udim, self.xdim, L, hid, stream_ptr()),
                "vjf_recognition_forward")
        return Gaussian(mu_t, lv_t)
