"""vjf.model on the MI355X: VJF, RBFDS, LinearDecoder with the reference's signatures
(vjf/model.py), the filtering step running as HIP kernels behind the C ABI (include/vjf_hip.h).

Drop-in notes:
  * `filter()` is the reference's API; `feed()` is an alias (the reference has no `feed`).
  * the optimiser is the reference's clipped plain SGD with an ExponentialLR stepped per epoch in
    `fit` (vjf/model.py:69-78, 210-211, 303) -- executed on the device; `optimizer.param_groups`
    and `scheduler.step()` are kept as light host objects.
  * all parameters, the RLS state and the counters live in ONE device blob owned by the VJF
    object; module attributes (`likelihood.logvar`, `transition.velocity.w_mean`, ...) are views
    into it, updated in place by the kernels.
  * device arithmetic is fp32 whatever torch's default dtype is.
  * noise: by default the two reparametrisation draws per step come from torch's CPU generator
    in the reference's order (xs then xt), so `torch.manual_seed` reproduces the reference's
    trajectory; `noise="device"` draws on the GPU instead (faster, different stream).
"""
import atexit
import os
import ctypes
import logging
import sys
import weakref
from itertools import zip_longest
from typing import Sequence, Tuple, Union

import torch
from torch import Tensor, nn
from torch.nn import Linear, Module, Parameter

from . import _native as N
from .distribution import Gaussian
from .functional import gaussian_entropy as entropy
from .functional import gaussian_loss
from .likelihood import GaussianLikelihood, PoissonLikelihood
from .module import RBF, LinearRegression, rebind
from .recognition import Recognition
from .util import dev32, nonecat, reparametrize, running_var, storage_device, stream_ptr

try:                                   # progress bar as in the reference (vjf/model.py:245)
    from tqdm import trange
except Exception:                      # pragma: no cover
    def trange(n):
        class _R:
            def __enter__(self_): return self_
            def __exit__(self_, *a): return False
            def __iter__(self_): return iter(range(n))
            def set_postfix(self_, *a, **k): pass
        return _R()


# Native contexts are destroyed while the HIP runtime is alive: explicitly (`VJF.close()`), when a model is collected in a running
# interpreter, or -- for whatever is left -- from this hook, which is registered when the module is imported (after torch, so it runs
# BEFORE torch's and the HIP runtime's own exit handlers).  Nothing calls into the library from `__del__` during interpreter
# finalisation: by then streams and communicators may be gone.
_LIVE = weakref.WeakSet()


def _close_all():
    for m in list(_LIVE):
        try:
            m.close()
        except Exception:
            pass


atexit.register(_close_all)


class LinearDecoder(Module):
    """vjf/model.py:21-42"""
    def __init__(self, xdim: int, ydim: int):
        super().__init__()
        self.add_module('decode', Linear(xdim, ydim))
        for p in self.decode.parameters():
            p.requires_grad_(False)
            p.data = dev32(p.data, ndim2=False)
        self.n_sample = 0

    def forward(self, x: Union[Tensor, Gaussian]) -> Union[Tensor, Gaussian]:
        if isinstance(x, Tensor):
            x = dev32(x)
            lead = x.shape[:-1]
            x2 = x.reshape(-1, x.shape[-1])
            W, b = self.decode.weight, self.decode.bias
            out = torch.empty(x2.shape[0], W.shape[0], device=x.device, dtype=torch.float32)
            N.check(N.lib().vjf_linear_forward(N.ptr(x2), N.ptr(W), N.ptr(b), N.ptr(out), x2.shape[0], W.shape[1], W.shape[0],
                                               stream_ptr()), "vjf_linear_forward")
            return out.reshape(*lead, W.shape[0])
        elif isinstance(x, Gaussian):
            raise NotImplementedError("LinearDecoder on a Gaussian (vjf/model.py:31-40) is never taken by filter()")
        else:
            raise NotImplementedError


def detach(q: Gaussian) -> Gaussian:
    """vjf/model.py:45-47 (no autograd graph exists on the HIP path; kept for API parity)"""
    mean, logvar = q
    return Gaussian(mean.detach(), logvar.detach())


class _ParamGroups:
    """Stand-in for torch.optim.SGD's `param_groups` (vjf/model.py:69-77): four groups
    [likelihood, decoder, transition, recognition], each with its own 'lr'."""
    def __init__(self, lr):
        self.param_groups = [{'lr': lr, 'name': n} for n in ('likelihood', 'decoder', 'transition', 'recognition')]

    def zero_grad(self):
        pass


class _ExponentialLR:
    """ExponentialLR(gamma) stepped once per epoch (vjf/model.py:78, 303): lr <- lr * gamma, recursively."""
    def __init__(self, optimizer, gamma):
        self.optimizer, self.gamma = optimizer, gamma

    def step(self):
        for g in self.optimizer.param_groups:
            g['lr'] = g['lr'] * self.gamma


class RBFDS(Module):
    """RBF dynamical system  x[t] = x[t-1] + Phi([x[t-1], u[t]]) W    (vjf/model.py:327-391)"""
    def __init__(self, n_rbf: int, xdim: int, udim: int):
        super().__init__()
        self.add_module('velocity', LinearRegression(RBF(xdim + udim, n_rbf), xdim))
        self.register_parameter('logvar', Parameter(dev32(torch.tensor(0.), ndim2=False), requires_grad=False))
        self._n_sample = 0
        object.__setattr__(self, '_owner', None)

    @property
    def n_sample(self):
        o = self._owner() if self._owner is not None else None
        return o._get_counter("tr") if o is not None else self._n_sample

    @n_sample.setter
    def n_sample(self, v):
        o = self._owner() if self._owner is not None else None
        if o is not None:
            o._set_counter("tr", v)
        else:
            self._n_sample = v

    def forward(self, x: Tensor, u: Tensor = None, sampling: bool = True, leak: float = 0., noise: Tensor = None):
        x = dev32(x)
        xu = nonecat(x, None if u is None else dev32(u))
        dx = self.velocity(xu, sampling=sampling, noise=noise)
        if isinstance(dx, Gaussian):
            return Gaussian((1 - leak) * x + dx.mean, dx.logvar)
        else:
            return (1 - leak) * x + dx

    def forecast(self, x0: Tensor, u: Tensor = None, n_step: int = 1, *, noise: bool = False) -> Tensor:
        """vjf/model.py:342-361: sequential sampled roll-out (a fresh weight sample every step)."""
        x0 = dev32(x0)
        x = torch.empty(n_step + 1, *x0.shape, device=x0.device, dtype=torch.float32)
        x[0] = x0
        s = torch.exp(.5 * self.logvar)
        if u is None:
            u = [None] * n_step
        else:
            u = dev32(u)
            assert u.shape[0] == n_step, 'u must have length of n_step if present'
        for t in range(n_step):
            x[t + 1] = self.forward(x[t], u[t], sampling=True)
            if noise:
                e = torch.randn(x[t + 1].shape, dtype=torch.get_default_dtype()).to(x.device, torch.float32)
                x[t + 1] = x[t + 1] + e * s
        return x

    @torch.no_grad()
    def update(self, xt: Tensor, xs: Tensor, ut: Tensor = None, *, warm_up=False):
        """Train regression, operator by operator (vjf/model.py:363-377).  VJF.filter does the same
        inside its fused step; this method serves direct callers."""
        xs, xt = dev32(xs), dev32(xt)
        xu = nonecat(xs, None if ut is None else dev32(ut))
        dx = xt - xs
        if not warm_up:
            self.velocity.rls(xu, dx, self.logvar.exp(), shrink=1.)
        residual = dx - self.velocity(xu, sampling=False).mean
        mse = residual.pow(2).mean()
        var, n_sample = running_var(self.logvar.exp(), self.n_sample, mse, xs.shape[0], size_cap=500)
        self.logvar.copy_(var.log())
        self.n_sample = n_sample

    @torch.no_grad()
    def initialize(self, xt: Tensor, xs: Tensor, ut: Tensor = None):
        """vjf/model.py:379-388"""
        xs, xt = dev32(xs), dev32(xt)
        xu = nonecat(xs, None if ut is None else dev32(ut))
        mse = (xt - xs).pow(2).mean()
        self.velocity.initialize(xu, xt - xs, mse)
        d, V = self.velocity(xu, sampling=False)
        mse = (xt - xs - d).pow(2).mean()
        self.logvar.copy_(mse.log())

    def loss(self, pt, qt) -> Tensor:
        return gaussian_loss(pt, qt, self.logvar)


class VJF(Module):
    def __init__(self, ydim: int, xdim: int, likelihood: Module, transition: Module, recognition: Module,
                 *, lr: float = 1e-4, lr_decay: float = .9, noise: str = "reference"):
        """
        Use VJF.make_model   (vjf/model.py:50-78)
        :param likelihood: GLM likelihood, Gaussian or Poisson
        :param transition: f(x[t-1], u[t]) -> x[t]
        :param recognition: y[t], f(x[t-1], u[t]) -> x[t]
        :param lr_decay: multiplicative factor of learning rate decay
        :param noise: "reference" (CPU generator, reference draw order) or "device"
        """
        super().__init__()
        self.add_module('likelihood', likelihood)
        self.add_module('transition', transition)
        self.add_module('recognition', recognition)
        self.add_module('decoder', LinearDecoder(xdim, ydim))
        self.register_parameter('mean', Parameter(dev32(torch.zeros(xdim), ndim2=False), requires_grad=False))
        self.register_parameter('logvar', Parameter(dev32(torch.zeros(xdim), ndim2=False), requires_grad=False))
        self.optimizer = _ParamGroups(lr)
        self.scheduler = _ExponentialLR(self.optimizer, gamma=lr_decay)

        if noise not in ("reference", "device"):
            raise ValueError("noise must be 'reference' or 'device'")
        self.noise = noise
        self.ydim, self.xdim = ydim, xdim
        feat = transition.velocity.feature
        self.udim = feat.centroid.shape[1] - xdim
        self.n_rbf = feat.n_basis
        self.hidden_sizes = list(recognition.hidden_sizes)
        if isinstance(likelihood, GaussianLikelihood):
            self._lik = N.LIK_GAUSSIAN
        elif isinstance(likelihood, PoissonLikelihood):
            self._lik = N.LIK_POISSON
        else:
            raise TypeError("likelihood must be GaussianLikelihood or PoissonLikelihood")
        self._lib = None            # tests may inject a stand-in for the C ABI
        self._ctx = None
        self._ctx_batch = 0
        self._workspace = None
        self._pushed_lr = None
        self._adopt_state()

    # ------------------------------------------------------------------ device state
    def _backend(self):
        if self._lib is None:
            self._lib = N.lib()
        return self._lib

    def _config(self, max_batch):
        dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
        return N.make_config(self.ydim, self.xdim, self.udim, self.n_rbf, self.hidden_sizes, self._lik, max_batch, dev)

    def _adopt_state(self):
        """Allocate the state blob and turn every module tensor into a view of it (include/vjf_hip.h, enum vjf_slot)."""
        n, off, size = N.state_layout(self._config(1))
        self._blob = torch.zeros(n, dtype=torch.float32, device=storage_device())

        def view(slot, shape):
            return self._blob[off[slot]:off[slot] + size[slot]].view(shape)

        vel, feat, rec = self.transition.velocity, self.transition.velocity.feature, self.recognition
        nr, dz, du, dy = self.n_rbf, self.xdim, self.udim, self.ydim
        rebind(self, 'mean', view(N.SLOT_PRIOR_MEAN, (dz,)))
        rebind(self, 'logvar', view(N.SLOT_PRIOR_LOGVAR, (dz,)))
        if self._lik == N.LIK_GAUSSIAN:
            rebind(self.likelihood, 'logvar', view(N.SLOT_LIK_LOGVAR, ()))
            object.__setattr__(self.likelihood, '_owner', weakref.ref(self))
        rebind(self.transition, 'logvar', view(N.SLOT_TR_LOGVAR, ()))
        object.__setattr__(self.transition, '_owner', weakref.ref(self))
        rebind(feat, 'centroid', view(N.SLOT_CENTROID, (nr, dz + du)))
        rebind(feat, 'logwidth', view(N.SLOT_LOGWIDTH, (nr,)))
        prev = dy + du + 2 * dz
        for k, lin in enumerate(rec.linears()):
            rebind(lin, 'weight', view(N.SLOT_REC_W0 + 2 * k, (self.hidden_sizes[k], prev)))
            rebind(lin, 'bias', view(N.SLOT_REC_B0 + 2 * k, (self.hidden_sizes[k],)))
            prev = self.hidden_sizes[k]
        rebind(rec.mean, 'weight', view(N.SLOT_MEAN_W, (dz, prev)))
        rebind(rec.logvar, 'weight', view(N.SLOT_LV_W, (dz, prev)))
        rebind(rec.logvar, 'bias', view(N.SLOT_LV_B, (dz,)))
        rebind(self.decoder.decode, 'weight', view(N.SLOT_DEC_W, (dy, dz)))
        rebind(self.decoder.decode, 'bias', view(N.SLOT_DEC_B, (dy,)))
        rebind(vel, 'w_mean', view(N.SLOT_W_MEAN, (nr, dz)))
        rebind(vel, 'w_chol', view(N.SLOT_W_CHOL, (nr, nr)))
        rebind(vel, 'w_precision', view(N.SLOT_W_PREC, (nr, nr)))
        rebind(vel, 'w_pchol', view(N.SLOT_W_PCHOL, (nr, nr)))
        self._scalars = self._blob[off[N.SLOT_SCALARS]:off[N.SLOT_SCALARS] + N.N_SCALARS]
        self._scalars[N.SC_N_LIK] = float(getattr(self.likelihood, "_n_sample", 0))
        self._scalars[N.SC_N_TR] = float(self.transition._n_sample)
        self._push_lr(force=True)

    def _get_counter(self, which):
        return int(self._scalars[N.SC_N_LIK if which == "lik" else N.SC_N_TR].item())

    def _set_counter(self, which, v):
        self._scalars[N.SC_N_LIK if which == "lik" else N.SC_N_TR] = float(v)

    def _push_lr(self, force=False):
        lrs = [float(g['lr']) for g in self.optimizer.param_groups]
        if force or lrs != self._pushed_lr:
            self._scalars[N.SC_LR_LIK:N.SC_LR_LIK + 4] = torch.tensor(lrs, dtype=torch.float32)
            self._pushed_lr = lrs

    def freeze_decoder(self, frozen: bool = True):
        """decoder.requires_grad_(False) of the reference (vjf/model.py:283)."""
        self._scalars[N.SC_FREEZE_DEC] = 1.0 if frozen else 0.0

    def set_overlap(self, enable: bool = True):
        """filter_sequence runs a step's RLS chain on a second stream beside the trial / SGD chain (default);
        False forces the one-stream order.  Results are bit-identical either way."""
        self._overlap = int(enable) if enable in (0, 1, 2, 3, True, False) else 1     # 3: the per-step three-stream route on one rank
        if self._ctx is not None:
            rc = self._backend().vjf_set_overlap(self._ctx, int(self._overlap))       # returns the resulting setting
            if rc < 0:
                N.check(rc, "vjf_set_overlap")

    def route(self, sgd: bool = True, update: bool = True, warm_up: bool = False) -> str:
        """The schedule `filter_sequence` would use now for this batch size: 'one-launch', 'streams' (three streams: the RCCL
        path), 'two-stream' (plans whose RLS update is a sequence of launches: that update beside the trial chain) or 'per-step'."""
        if self._ctx is None:
            return "unsized"
        rc = self._backend().vjf_route(self._ctx, (N.FLAG_SGD if sgd else 0) | (N.FLAG_UPDATE if update else 0)
                                       | (N.FLAG_WARM_UP if warm_up else 0))
        if rc < 0:
            N.check(rc, "vjf_route")
        return {1: "one-launch", 2: "two-stream", 3: "streams", 4: "packed"}.get(rc, "per-step")

    # ------------------------------------------------------------------ state I/O (the reference has no format)
    _RLS_KEYS = ("w_mean", "w_chol", "w_precision", "w_pchol")

    def get_state(self) -> dict:
        """Everything a run needs to resume, as numpy arrays: the 13 `state_dict` tensors, the RLS tensors of
        `transition.velocity` (plain attributes in the reference: vjf/module.py:45-54), the two sample counters, the
        four group learning rates, the decoder-freeze flag and the shape of the model."""
        import numpy as np
        self.check_status()                  # (raises if a device-side wait of an earlier call timed out: that state is not one to keep)
        st = {k: v.detach().cpu().numpy().copy() for k, v in self.state_dict().items()}
        lr = self.transition.velocity
        for k in self._RLS_KEYS:
            st["transition.velocity." + k] = getattr(lr, k).detach().cpu().numpy().copy()
        st["likelihood.n_sample"] = np.int64(self._get_counter("lik"))
        st["transition.n_sample"] = np.int64(self._get_counter("tr"))
        st["optimizer.lr"] = np.array([float(g["lr"]) for g in self.optimizer.param_groups], np.float64)
        st["decoder.frozen"] = np.int64(int(self._scalars[N.SC_FREEZE_DEC].item() != 0))
        st["config"] = np.array([self.ydim, self.xdim, self.udim, self.n_rbf, self._lik] + list(self.hidden_sizes), np.int64)
        return st

    def set_state(self, st: dict):
        """Inverse of `get_state` (same model shape required).  Clears the sticky status bits and marks `w_chol` /
        `w_pchol` as possibly dense, so the first RLS update re-establishes their triangles."""
        import numpy as np
        cfg = [int(v) for v in np.asarray(st["config"]).tolist()]
        mine = [self.ydim, self.xdim, self.udim, self.n_rbf, self._lik] + list(self.hidden_sizes)
        if cfg != mine:
            raise ValueError(f"state is for a model of shape {cfg}, this one is {mine}")
        own = self.state_dict()
        with torch.no_grad():
            for k, v in own.items():
                v.copy_(torch.as_tensor(np.asarray(st[k]), dtype=torch.float32).reshape(v.shape))
            lr = self.transition.velocity
            for k in self._RLS_KEYS:
                t = getattr(lr, k)
                t.copy_(torch.as_tensor(np.asarray(st["transition.velocity." + k]), dtype=torch.float32).reshape(t.shape))
        self._set_counter("lik", int(st["likelihood.n_sample"]))
        self._set_counter("tr", int(st["transition.n_sample"]))
        for g, v in zip(self.optimizer.param_groups, np.asarray(st["optimizer.lr"]).tolist()):
            g["lr"] = float(v)
        self._push_lr(force=True)
        self.freeze_decoder(bool(int(st["decoder.frozen"])))
        self._scalars[N.SC_TRI_CLEAN] = 0.0
        self._scalars[N.SC_STATUS] = 0.0

    def save_state(self, path):
        import numpy as np
        np.savez(path, **self.get_state())

    def load_state(self, path):
        import numpy as np
        with np.load(path) as z:
            self.set_state({k: z[k] for k in z.files})

    def status(self) -> int:
        """Sticky VJF_STATUS_* bits raised by the device since the last call (clears them)."""
        if self._ctx is None:
            return 0
        s = ctypes.c_uint32()
        N.check(self._backend().vjf_get_status(self._ctx, ctypes.byref(s)), "vjf_get_status")
        return s.value

    def _ensure_ctx(self, B):
        if self._ctx is not None and B <= self._ctx_batch:
            return
        L = self._backend()
        if self._ctx is not None:
            torch.cuda.synchronize() if torch.cuda.is_available() else None
            L.vjf_ctx_destroy(self._ctx)
            self._ctx = None
        cfg = self._config(B)
        nbytes = ctypes.c_int64()
        N.check(L.vjf_workspace_size(ctypes.byref(cfg), ctypes.byref(nbytes)), "vjf_workspace_size")
        self._workspace = torch.empty(nbytes.value, dtype=torch.uint8, device=self._blob.device)
        poison = os.environ.get("VJF_DEBUG_POISON_WS")         # (tests: a workspace of NaNs / of a byte pattern shows any read of a
        if poison:                                             #  region the library has not written)
            self._workspace.fill_(int(poison, 0) & 0xFF)
        ctx = ctypes.c_void_p()
        N.check(L.vjf_ctx_create(ctypes.byref(cfg), N.ptr(self._blob), N.ptr(self._workspace), nbytes.value, stream_ptr(),
                                 ctypes.byref(ctx)), "vjf_ctx_create")
        self._ctx, self._ctx_batch = ctx, B
        _LIVE.add(self)
        if getattr(self, "_collectives", 2) != 2:
            N.check(L.vjf_set_collectives(ctx, int(self._collectives)), "vjf_set_collectives")
        if getattr(self, "_overlap", 1) != 1:
            rc = L.vjf_set_overlap(ctx, int(self._overlap))
            if rc < 0:
                N.check(rc, "vjf_set_overlap")
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        N.check(L.vjf_reduce_buffer(ctx, ctypes.byref(p), ctypes.byref(n)), "vjf_reduce_buffer")
        o = p.value - self._workspace.data_ptr()
        self._reduce = self._workspace[o:o + 4 * n.value].view(torch.float32)

    def close(self):
        """Destroy the native context (its streams are synchronised first; RCCL communicators and the extra streams of the sharded
        route are released).  The model stays usable: the next call makes a new context.  Called for every live model at
        interpreter exit, before the HIP runtime's own exit handlers."""
        ctx, self._ctx = self._ctx, None
        self._ctx_batch = 0
        if ctx is not None and self._lib is not None:
            self._lib.vjf_ctx_destroy(ctx)

    def __del__(self):
        if sys.is_finalizing():               # (no native call at interpreter shutdown: `_close_all` has run)
            return
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ noise
    def _draw(self, shape):
        """One reparametrisation draw.  "reference": torch.randn on the CPU generator in the default
        dtype == randn_like(mean) of vjf/util.py:13 for the contiguous means filter produces."""
        if self.noise == "reference":
            return torch.randn(shape, dtype=torch.get_default_dtype()).to(self._blob.device, torch.float32)
        return torch.randn(shape, device=self._blob.device, dtype=torch.float32)

    @staticmethod
    def _world():
        import os
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            w = dist.get_world_size()
            # VJF_FORCE_DIST=1: take the sharded path (local half -> all-reduce -> global half) even with one rank, so that it
            # can be exercised on a single GPU
            return (w, True) if (w > 1 or os.environ.get("VJF_FORCE_DIST") == "1") else (1, False)
        return 1, False

    @staticmethod
    def _all_reduce_sum(t: Tensor):
        """Sum over ranks, in place.  A gloo group (several ranks sharing one GPU in tests, or a CPU control plane) gets the
        buffer through the host; RCCL takes the device tensor as it is."""
        import torch.distributed as dist
        if dist.get_backend() == "gloo" and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def _native_comm(self, world) -> bool:
        """RCCL communicators inside the context (collective over the ranks; once per context).  False -> the caller keeps
        the all-reduce on its side (vjf_filter_local / torch.distributed.all_reduce / vjf_filter_global)."""
        import os
        import torch.distributed as dist
        if getattr(self, "_comm_ctx", None) is self._ctx:
            return self._comm_ok
        self._comm_ctx, self._comm_ok = self._ctx, False
        L = self._backend()
        ids = (ctypes.c_char * 256)()
        ok = (os.environ.get("VJF_NATIVE_RCCL", "1") != "0" and self._overlap_flag() and L.vjf_set_overlap(self._ctx, int(getattr(self, "_overlap", 1))) >= 1
              and L.vjf_comm_unique_id(ids) == 0)
        # (control plane only: with a gloo group the two exchanges go through host tensors and no collective stream of
        #  torch's is created beside the four of vjf_filter_seq)
        cdev = "cpu" if dist.get_backend() == "gloo" else self._blob.device
        flag = torch.tensor([1 if ok else 0], device=cdev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)                 # every rank takes the same route
        if int(flag.item()) == 0:
            return False
        t = torch.frombuffer(bytearray(bytes(ids)), dtype=torch.uint8).to(cdev)
        dist.broadcast(t, src=0)
        buf = (ctypes.c_char * 256).from_buffer_copy(t.cpu().numpy().tobytes())
        N.check(L.vjf_comm_init(self._ctx, buf, dist.get_rank(), world), "vjf_comm_init")
        self._comm_ok = True
        return True

    def set_collectives(self, per_step: int = 2):
        """Sums over ranks per step on the in-library RCCL route: 2 (two all-reduces on two overlapping chains, default) or 1 (the
        whole reduce buffer in one).  Takes effect from the next call."""
        if per_step not in (1, 2):
            raise ValueError("per_step must be 1 or 2")
        self._collectives = per_step
        if self._ctx is not None:
            N.check(self._backend().vjf_set_collectives(self._ctx, per_step), "vjf_set_collectives")

    def comm_ranks(self):
        """[ranks of the gradient chain's communicator, of the statistics chain's] as RCCL reports them (ncclCommCount);
        [0, 0] when the context has none (single rank, or the sums over ranks on the caller's side)."""
        if self._ctx is None:
            return [0, 0]
        r = (ctypes.c_int32 * 2)()
        N.check(self._backend().vjf_comm_ranks(self._ctx, r), "vjf_comm_ranks")
        return [int(r[0]), int(r[1])]

    def _overlap_flag(self) -> bool:
        return bool(getattr(self, "_overlap", 1))

    # Trials sharded over ranks: True replays a step whose loss has a non-finite component exactly as the reference defines it
    # (vjf/model.py:138-149) at the price of a second sum over ranks in every step; False (default) skips such a step's SGD update
    # and raises the status bit.  One rank always replays (include/vjf_hip.h, VJF_FLAG_EXACT_NONFINITE).
    exact_nonfinite = False

    def _flags(self, sgd, update, warm_up):
        return ((N.FLAG_SGD if sgd else 0) | (N.FLAG_UPDATE if update else 0) | (N.FLAG_WARM_UP if warm_up else 0)
                | (N.FLAG_EXACT_NONFINITE if self.exact_nonfinite else 0))

    # ------------------------------------------------------------------ operator-level API (reference structure)
    def prior(self, y: Tensor) -> Gaussian:
        """vjf/model.py:80-95"""
        assert y.ndim == 2
        n_batch = y.shape[0]
        one = torch.ones(n_batch, self.xdim, device=self._blob.device)
        return Gaussian(one * torch.atleast_2d(self.mean), one * torch.atleast_2d(self.logvar))

    def forward(self, y: Tensor, qs: Gaussian, u: Tensor = None, *, eps=None) -> Tuple:
        """encode / predict / decode, one operator at a time (vjf/model.py:97-122).
        :return: xs, pt, qt, xt, py"""
        y = dev32(y)
        qs = self.prior(y) if qs is None else detach(Gaussian(dev32(qs.mean), dev32(qs.logvar)))
        e1, e2 = (None, None) if eps is None else (dev32(eps[0]), dev32(eps[1]))
        xs = reparametrize(qs, e1 if e1 is not None else self._draw(qs.mean.shape))
        pt = self.transition(xs, u, sampling=False)
        qt = self.recognition(y, qs, u)
        xt = reparametrize(qt, e2 if e2 is not None else self._draw(qt.mean.shape))
        py = self.decoder(xt)
        return xs, pt, qt, xt, py

    def loss(self, y: Tensor, xs: Tensor, pt, qt: Gaussian, xt: Tensor, py: Tensor,
             components: bool = False, warm_up: bool = False):
        """vjf/model.py:124-154"""
        l_recon = self.likelihood.loss(py, dev32(y))
        l_dynamics = self.transition.loss(pt, qt)
        h = entropy(qt)
        zero = torch.zeros((), device=l_recon.device)
        if not torch.isfinite(l_recon):
            l_recon = zero
        if not torch.isfinite(l_dynamics):
            l_dynamics = zero
        if not torch.isfinite(h):
            h = zero
        loss = l_recon - h
        if not warm_up:
            loss = loss + l_dynamics
        if components:
            return loss, -l_recon, -l_dynamics, h
        return loss

    @torch.no_grad()
    def update(self, y: Tensor, xs: Tensor, u: Tensor, pt, qt: Gaussian, xt: Tensor, py: Tensor, *,
               likelhood=True, decoder=True, transition=True, recognition=True, warm_up=False):
        """Learning without gradient (vjf/model.py:156-177)"""
        if likelhood:
            self.likelihood.update(py, y)
        if transition:
            self.transition.update(xt, xs, u, warm_up=warm_up)

    # ------------------------------------------------------------------ the hot path
    def filter(self, y: Tensor, u: Tensor = None, qs: Gaussian = None, *,
               sgd: bool = True, update: bool = True, verbose: bool = False, warm_up: bool = False, eps=None):
        """
        Filter a step   (vjf/model.py:179-221) -- one fused device step for the whole batch of trials.
        :param y: observation (batch, dim); a missing batch axis is prepended.
        :param u: control
        :param qs: previous posterior. use prior if None.
        :param sgd: flag to enable gradient step
        :param update: flag to update DS
        :param verbose: verbose output
        :param warm_up: do not learn dynamics if True, default=False
        :param eps: optional (eps_s, eps_t) noise, each (batch, xdim); drawn if None
        (Asynchronous, like `filter_sequence`: the device's status word -- 'RLS failed.', a timed-out hand-off -- is read by
        `check_status()`, which a caller runs before relying on the outputs or the state; `fit`, `get_state`, `save_state` do.)
        :return:
            qt: posterior
            loss: negative elbo   [, -l_recon, -l_dynamics, entropy if verbose]
        """
        y = dev32(y)
        if y.shape[1] != self.ydim:
            raise AssertionError(f"y has {y.shape[1]} columns, expected ydim={self.ydim}")
        B = y.shape[0]
        if self.udim > 0:
            if u is None:
                raise TypeError("u is required when udim > 0")
            u = dev32(u)
            assert u.shape == (B, self.udim)
        else:
            u = None
        if qs is None:
            mu_s = lv_s = None
        else:
            mu_s, lv_s = dev32(qs.mean), dev32(qs.logvar)
            assert mu_s.shape == (B, self.xdim) and lv_s.shape == (B, self.xdim)
        if eps is None:
            eps_s = self._draw((B, self.xdim))
            eps_t = self._draw((B, self.xdim))
        else:
            eps_s, eps_t = dev32(eps[0]), dev32(eps[1])
            assert eps_s.shape == (B, self.xdim) and eps_t.shape == (B, self.xdim)
        self._ensure_ctx(B)
        self._push_lr()
        L = self._backend()
        dev = self._blob.device
        mu_t = torch.empty(B, self.xdim, device=dev, dtype=torch.float32)
        lv_t = torch.empty(B, self.xdim, device=dev, dtype=torch.float32)
        loss4 = torch.empty(4, device=dev, dtype=torch.float32)
        flags = self._flags(sgd, update, warm_up)
        L.vjf_set_stream(self._ctx, stream_ptr())
        world, sharded = self._world()
        if not sharded:
            N.check(L.vjf_filter_step(self._ctx, B, N.ptr(y), N.ptr(u), N.ptr(mu_s), N.ptr(lv_s), N.ptr(eps_s), N.ptr(eps_t),
                                      N.ptr(mu_t), N.ptr(lv_t), N.ptr(loss4), flags), "vjf_filter_step")
        else:
            import torch.distributed as dist
            N.check(L.vjf_filter_local(self._ctx, B, N.ptr(y), N.ptr(u), N.ptr(mu_s), N.ptr(lv_s), N.ptr(eps_s), N.ptr(eps_t),
                                       N.ptr(mu_t), N.ptr(lv_t), flags), "vjf_filter_local")
            self._all_reduce_sum(self._reduce)                        # the ONE collective of a step
            N.check(L.vjf_filter_global(self._ctx, B * world, N.ptr(loss4), flags), "vjf_filter_global")
        if update and not warm_up:
            if not self.transition.velocity._w_colmajor:     # (nn.Module.__setattr__ costs 2.5 us: only when it changes)
                self.transition.velocity._w_colmajor = True  # see LinearRegression._draw_weight_noise
        qt = Gaussian(mu_t, lv_t)
        if verbose:
            return qt, loss4[0], loss4[1], loss4[2], loss4[3]
        return qt, loss4[0]

    feed = filter      # the north-star's name for one step; the reference itself has no `feed`

    def filter_sequence(self, y: Tensor, u: Tensor = None, qs: Gaussian = None, *, sgd: bool = True, update: bool = True,
                        warm_up: bool = False, eps: Tensor = None):
        """T successive `filter` steps in one C-ABI call (the inner loop of fit, vjf/model.py:252-261).
        y (T,B,ydim); u (T,B,udim) or None; eps (T,2,B,xdim) or None (drawn in the reference's order).
        :return: mu (T,B,xdim), logvar (T,B,xdim), loss (T,4) = [loss, -l_recon, -l_dynamics, entropy]

        The call is ASYNCHRONOUS and does not read the device's status word.  **Call `check_status()` before you rely on the
        outputs or on the model's state**: it warns 'RLS failed.' as the reference does, and RAISES when a hand-off inside the
        launch timed out (compute units held by another process for seconds: the outputs and the SGD / RLS state of that call are
        then not valid -- restore from `get_state()` of an earlier point and run again).  `fit` does this after every sequence;
        `get_state` / `save_state` do it before they copy anything."""
        y = dev32(y, ndim2=False)
        assert y.ndim == 3 and y.shape[2] == self.ydim
        T, B = y.shape[:2]
        u = dev32(u, ndim2=False) if (u is not None and self.udim > 0) else None
        if self.udim > 0 and u is None:
            raise TypeError("u is required when udim > 0")
        if eps is None:
            if self.noise == "reference":      # per step: xs draw then xt draw, each its own randn call
                eps = torch.stack([torch.stack([torch.randn(B, self.xdim, dtype=torch.get_default_dtype()) for _ in range(2)])
                                   for _ in range(T)])
            else:
                eps = torch.randn(T, 2, B, self.xdim, device=self._blob.device)
        eps = dev32(eps, ndim2=False)
        assert eps.shape == (T, 2, B, self.xdim)
        mu0 = lv0 = None
        if qs is not None:
            mu0, lv0 = dev32(qs.mean), dev32(qs.logvar)
        self._ensure_ctx(B)
        self._push_lr()
        L = self._backend()
        dev = self._blob.device
        # (one allocation for the three outputs: the call's host time is what a short sequence pays per step)
        nz = T * B * self.xdim
        out = torch.empty(2 * nz + 4 * T, device=dev, dtype=torch.float32)
        st3 = (B * self.xdim, self.xdim, 1)                     # (as_strided: a third of the host time of slice + view)
        mu, lv, loss = out.as_strided((T, B, self.xdim), st3, 0), out.as_strided((T, B, self.xdim), st3, nz), out.as_strided((T, 4), (4, 1), 2 * nz)
        flags = self._flags(sgd, update, warm_up)
        L.vjf_set_stream(self._ctx, stream_ptr())
        world, sharded = self._world()
        if not sharded or ((getattr(self, "_collectives", 2) == 1 or (update and not warm_up and T > 1)) and self._native_comm(world)):
            # one C-ABI call for the whole sequence; with ranks, the library sums statistics and gradients over them itself
            N.check(L.vjf_filter_seq(self._ctx, T, B, N.ptr(y), N.ptr(u), N.ptr(eps), N.ptr(mu0), N.ptr(lv0), N.ptr(mu), N.ptr(lv),
                                     N.ptr(loss), flags), "vjf_filter_seq")
        else:
            import torch.distributed as dist
            ms, ls = mu0, lv0
            for t in range(T):
                N.check(L.vjf_filter_local(self._ctx, B, N.ptr(y[t]), N.ptr(None if u is None else u[t]), N.ptr(ms), N.ptr(ls),
                                           N.ptr(eps[t, 0]), N.ptr(eps[t, 1]), N.ptr(mu[t]), N.ptr(lv[t]), flags), "vjf_filter_local")
                self._all_reduce_sum(self._reduce)
                N.check(L.vjf_filter_global(self._ctx, B * world, N.ptr(loss[t]), flags), "vjf_filter_global")
                ms, ls = mu[t], lv[t]
        if update and not warm_up and not self.transition.velocity._w_colmajor:
            self.transition.velocity._w_colmajor = True
        return mu, lv, loss

    # sticky status bits that mean "the results of the call are not to be used" (include/vjf_hip.h: VJF_STATUS_WAIT_*)
    _WAIT_BITS = 0x3ff00

    def check_status(self) -> int:
        """Read (and clear) the device's sticky status word -- one host synchronisation.  'RLS failed.' is warned about as the
        reference does from LinearRegression.rls (vjf/module.py:112); a hand-off time-out inside a launch raises."""
        import warnings
        st = self.status()
        if st & self._WAIT_BITS:
            raise RuntimeError(f"vjf: a device-side wait timed out (status 0x{st:x}); the outputs of the call are not valid")
        if st & N.STATUS_RLS_FAILED:
            warnings.warn('RLS failed.')
        return st

    # ------------------------------------------------------------------ harness
    def fit(self, y: Tensor, u: Tensor = None, *,
            max_iter: int = 200, beta: float = 0.1, verbose: bool = False, rtol: float = 1e-4):
        """
        vjf/model.py:223-307: epochs over the time axis with warm-up, decoder freeze, RBF
        re-initialisation, convergence test and per-epoch learning-rate decay.  Each epoch is one
        `filter_sequence` call.
        :param y: observation, (time, ..., dim)
        :param u: control input, None if autonomous
        :return: mu (time, batch, xdim), logvar (time, batch, xdim), epoch_loss
        """
        y = dev32(y)
        if y.ndim == 2:
            y = y[:, None, :]                    # (time, dim): every step is a batch of one trial
        if u is not None and self.udim > 0:
            u_ = dev32(u)
            if u_.ndim == 2:
                u_ = u_[:, None, :]
        else:
            u_ = None
        T = y.shape[0]

        warm_up = True
        epoch_loss = torch.tensor(float('nan'))
        mu = lv = None
        with trange(max_iter) as progress:
            running_loss = torch.tensor(float('nan'))
            for i in progress:
                mu, lv, losses = self.filter_sequence(y, u_, None, sgd=True, update=True, warm_up=warm_up)
                losses = losses.detach().cpu().to(torch.get_default_dtype())
                self.check_status()                  # (the copy above has synchronised the stream already)
                epoch_loss = losses[:, 0].sum() / T
                if verbose:
                    progress.set_postfix({'Loss': running_loss.item(), 'Recon': losses[-1, 1].item(),
                                          'Dynamics': losses[-1, 2].item(), 'Entropy': losses[-1, 3].item()})
                if warm_up:
                    if epoch_loss.isclose(running_loss, rtol=rtol):
                        warm_up = False
                        running_loss = epoch_loss
                        print('\nWarm up stopped.\n')
                        self.freeze_decoder(True)                    # freeze decoder after warm up
                        m = mu
                        if u_ is not None:
                            u_init = u_[1:].reshape(-1, u_.shape[-1])
                        else:
                            u_init = None
                        self.transition.initialize(m[1:].reshape(-1, m.shape[-1]),
                                                   m[:-1].reshape(-1, m.shape[-1]),
                                                   u_init)
                else:
                    if epoch_loss.isclose(running_loss, rtol=rtol):
                        print('\nConverged.\n')
                        break
                running_loss = beta * running_loss + (1 - beta) * epoch_loss if i > 0 else epoch_loss
                progress.set_postfix({'Loss': running_loss.item()})
                self.scheduler.step()
        return mu, lv, epoch_loss

    @classmethod
    def make_model(cls, ydim: int, xdim: int, udim: int, n_rbf: int, hidden_sizes: Sequence[int],
                   likelihood: str = 'poisson', *args, **kwargs):
        """vjf/model.py:309-319 -- same construction (and RNG consumption) order as the reference."""
        if likelihood.lower() == 'poisson':
            likelihood = PoissonLikelihood()
        elif likelihood.lower() == 'gaussian':
            likelihood = GaussianLikelihood()
        model = VJF(ydim, xdim, likelihood, RBFDS(n_rbf, xdim, udim), Recognition(ydim, xdim, udim, hidden_sizes),
                    *args, **kwargs)
        return model

    def forecast(self, x0: Tensor, u: Tensor = None, n_step: int = 1, *, noise: bool = False) -> Tuple[Tensor, Tensor]:
        """vjf/model.py:321-324"""
        x = self.transition.forecast(x0, u, n_step, noise=noise)
        y = self.decoder(x)
        return x, y
