"""vjf.likelihood on the GPU   (mirror of vjf/likelihood.py)."""
import torch
from torch import Tensor
from torch.nn import Module, Parameter

from .functional import gaussian_loss, poisson_loss
from .util import dev32, running_var


class GaussianLikelihood(Module):
    """Gaussian likelihood with a scalar learnable log-variance   (vjf/likelihood.py:9-40)"""
    def __init__(self):
        super().__init__()
        self.register_parameter('logvar', Parameter(dev32(torch.tensor(.1).log(), ndim2=False), requires_grad=False))
        self._n_sample = 0
        object.__setattr__(self, '_owner', None)   # weakref to the owning VJF: counters then live in its device blob

    @property
    def n_sample(self):
        o = self._owner() if self._owner is not None else None
        return o._get_counter("lik") if o is not None else self._n_sample

    @n_sample.setter
    def n_sample(self, v):
        o = self._owner() if self._owner is not None else None
        if o is not None:
            o._set_counter("lik", v)
        else:
            self._n_sample = v

    def loss(self, eta: Tensor, target: Tensor) -> Tensor:
        return gaussian_loss(target, eta, self.logvar)              # likelihood.py:26

    @torch.no_grad()
    def update(self, eta: Tensor, target: Tensor):
        """Running-variance update of the noise   (vjf/likelihood.py:28-40)"""
        eta, target = dev32(eta), dev32(target)
        mse = (target - eta).pow(2).mean()
        var, n_sample = running_var(self.logvar.exp(), self.n_sample, mse, eta.shape[0])
        self.logvar.copy_(var.log())
        self.n_sample = n_sample


class PoissonLikelihood(Module):
    """Poisson likelihood, log link   (vjf/likelihood.py:43-66)"""
    def __init__(self):
        super().__init__()

    @staticmethod
    def loss(eta: Tensor, target: Tensor) -> Tensor:
        if not isinstance(eta, Tensor):
            raise NotImplementedError
        return poisson_loss(eta, target)

    @torch.no_grad()
    def update(self, eta: Tensor, target: Tensor):
        pass
