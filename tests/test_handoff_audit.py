"""Build-time audit of the hand-off form of the resident kernels (tools/audit_plain_loads.py): their waits do not acquire, so every
load of a byte another workgroup of the launch stored must be an sc1 load.  The ISA of vjf_mega_kernel, vjf_mega_lite_kernel and
vjf_rlsc_loop_kernel is listed; a vector-memory load without sc1 whose source line has not been certified (launch constant, the
workgroup's own bytes, behind an explicit acquire, or code of the per-step routes) fails the test.  No GPU needed: hipcc
cross-compiles (about a minute)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_every_plain_load_of_the_resident_kernels_is_certified(tmp_path):
    spec = importlib.util.spec_from_file_location("audit_plain_loads", os.path.join(ROOT, "tools", "audit_plain_loads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    asm = str(tmp_path / "vjf_abi.s")
    mod.build_asm(asm)
    counts, report, bad = mod.audit(asm)
    assert not bad, "\n".join(bad)
    for k in mod.KERNELS:
        assert counts[k][0] > 100, (k, counts[k])          # (the kernels are there and their hand-off loads are sc1)
    # the audit does find what it is for: an uncertified line is reported
    key = ("vjf_mega_kernel.h", "grp = A.sl_grp[quad];")
    why = mod.ALLOW.pop(key)
    try:
        _, _, bad2 = mod.audit(asm)
        assert any("sl_grp" in b for b in bad2)
    finally:
        mod.ALLOW[key] = why
