"""Host-side helpers mirroring vjf/util.py, plus device plumbing shared by the operator shims."""
from typing import Tuple, Union

import torch
from torch import Tensor

from .distribution import Gaussian


def device() -> torch.device:
    """The MI355X this process drives.  There is no CPU path: raises when no GPU is visible."""
    if not torch.cuda.is_available():
        raise RuntimeError("vjf_amd needs an AMD GPU (gfx950); torch.cuda.is_available() is False and "
                           "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def storage_device() -> torch.device:
    """Where parameters are kept: the GPU when there is one.  Without a GPU objects can still be
    constructed and inspected (state_dict, shapes, seeds) on the CPU, but every compute call goes
    to the HIP library, which fails without a device -- there is no CPU compute path."""
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def dev32(a, ndim2: bool = True) -> Tensor:
    """as_tensor -> fp32 -> device -> contiguous (-> at least 2-D), as VJF.filter coerces its inputs
    (vjf/model.py:194-198).  The device path computes in fp32 whatever torch's default dtype is."""
    if (isinstance(a, Tensor) and a.is_cuda and a.dtype == torch.float32 and a.is_contiguous() and (a.ndim >= 2 or not ndim2)
            and a.device.index == torch.cuda.current_device()):
        return a                                   # (already what the library takes: the usual case inside a filtering loop)
    t = torch.as_tensor(a)
    t = t.to(device=storage_device(), dtype=torch.float32)
    if ndim2:
        t = torch.atleast_2d(t)
    return t.contiguous()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """The HIP stream torch is on (the library launches on the caller's stream): the raw handle, without a Stream object."""
    import ctypes
    if not torch.cuda.is_available():
        return ctypes.c_void_p(0)
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def reparametrize(q: Tuple[Tensor, Tensor], eps: Tensor = None) -> Tensor:
    """mean + eps * exp(logvar / 2)   (vjf/util.py:11-13).  `eps` defaults to a CPU draw in the
    reference's generator (torch.randn, same stream as randn_like on a contiguous CPU tensor)."""
    mean, logvar = q
    if eps is None:
        eps = torch.randn(mean.shape, dtype=torch.get_default_dtype()).to(mean.device, mean.dtype)
    return mean + eps * torch.exp(.5 * logvar)


def symmetric(a: Tensor) -> bool:
    return torch.allclose(a, a.transpose(-1, -2))


def running_var(acc_var, acc_size, new_var, new_size, *, size_cap=1000):
    """Capped running variance   (vjf/util.py:20-35)."""
    acc_size = min(acc_size, size_cap)
    tot_size = acc_size + new_size
    f1 = acc_size / tot_size
    f2 = new_size / tot_size
    return f1 * acc_var + f2 * new_var, tot_size


def nonecat(a: Tensor, u: Tensor):
    """Concatenation allowing None input   (vjf/util.py:38-49)."""
    au = torch.atleast_2d(a)
    if u is not None:
        if u.shape[-1] > 0:
            au = torch.cat((au, torch.atleast_2d(u)), -1)
    return au


def at_least2d(a: Union[Tensor, Gaussian]) -> Union[Tensor, Gaussian]:
    """vjf/util.py:52-63"""
    if isinstance(a, Tensor):
        return torch.atleast_2d(a)
    elif isinstance(a, Gaussian):
        return Gaussian(torch.atleast_2d(a.mean), torch.atleast_2d(a.logvar))
    else:
        raise TypeError(a.__class__)
