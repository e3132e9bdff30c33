"""Build the HIP library in-tree:  vjf_amd/libvjf_hip.so  (gfx950 only).

    python -m vjf_amd._build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the source tree.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvjf_hip.so")
CHAOS_LIB = os.path.join(HERE, "libvjf_hip_chaos.so")
SOURCES = ["vjf_abi.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build for gfx950)")
    return exe


def _inputs():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    files.append(os.path.join(os.path.dirname(HERE), "include", "vjf_hip.h"))
    return files


def is_stale(lib=None):
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(f) > t for f in _inputs())


def build(force=False, verbose=False, chaos=False):
    """Compile csrc/*.hip into libvjf_hip.so if missing or older than its sources.  chaos=True: the diagnostic build
    libvjf_hip_chaos.so (-DVJF_CHAOS, see vjf_plan.h) instead."""
    lib = CHAOS_LIB if chaos else LIB
    if not force and not is_stale(lib):
        return lib
    cmd = [_hipcc()] + FLAGS + (["-DVJF_CHAOS", "-Wno-pass-failed"] if chaos else []) + ["-o", lib + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(lib + ".tmp", lib)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, chaos="--chaos" in sys.argv))
