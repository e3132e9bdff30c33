"""Which library faults in the exit handlers of a process profiled by rocprofv3?  

    rocprofv3 --kernel-trace -d <dir> -- python tools/exit_probe.py <mode> <maps-file>

mode  torch : torch only -- one matmul on the GPU; libvjf_hip.so is never loaded
      load  : + libvjf_hip.so loaded through ctypes, no call into it beyond the ABI version
      step  : + a model, three filter steps on the per-step kernels (plain launches)
      coop  : + a three-step sequence on the one-launch route (cooperative launch)
      close : as `coop`, and the context destroyed explicitly before the interpreter shuts down
The process's load map goes to <maps-file> from an atexit hook registered before anything else (it runs last, with every library
still mapped), so that the frame addresses rocprofv3's signal handler prints can be attributed offline."""
import atexit
import os
import sys

mode, maps_out = sys.argv[1], sys.argv[2]


def _dump():
    with open("/proc/self/maps") as f, open(maps_out, "w") as o:
        o.write(f.read())


atexit.register(_dump)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

a = torch.randn(256, 256, device="cuda")
print("matmul", float((a @ a).sum()))
if mode != "torch":
    from vjf_amd import _native
    print("abi", _native.lib().vjf_abi_version())
if mode in ("step", "coop", "close"):
    import vjf_amd
    torch.manual_seed(0)
    m = vjf_amd.VJF.make_model(12, 4, 0, 40, [16], likelihood="gaussian")
    if mode == "step":
        m.set_overlap(False)
    g = torch.Generator().manual_seed(1)
    y = torch.randn(3, 48, 12, generator=g)
    eps = torch.randn(3, 2, 48, 4, generator=g)
    mu, lv, loss = m.filter_sequence(y, eps=eps)
    torch.cuda.synchronize()
    print("route", m.route(), "loss", float(loss[-1, 0]), "status", m.status())
    if mode == "close":
        m.close()
print("probe done", mode, flush=True)
