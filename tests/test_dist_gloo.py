"""world_size = 2 (gloo, CPU) test of the multi-GPU protocol of VJF.filter: each rank runs the
trial-parallel half on its shard, ONE all-reduce (sum) of the reduce buffer, every rank runs the
serial half.  The C ABI is served by the oracle-backed stand-in (tests/fake_backend.py); what is
under test is the host-side protocol: sharding, the single collective, B_total, replicated state."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(seed=5):
    import vjf_amd
    torch.manual_seed(seed)
    return vjf_amd.VJF.make_model(8, 3, 1, 12, [6], likelihood="gaussian", lr=1e-2)


def _data():
    g = torch.Generator().manual_seed(9)
    T, B = 3, 10
    return torch.randn(T, B, 8, generator=g), torch.randn(T, B, 1, generator=g), torch.randn(T, 2, B, 3, generator=g)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from tests import fake_backend
    fake_backend.install()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _make()
        y, u, eps = _data()
        B = y.shape[1]
        lo, hi = rank * B // world, (rank + 1) * B // world
        q, losses = None, []
        for t in range(y.shape[0]):
            q, loss = m.filter(y[t, lo:hi], u[t, lo:hi], q, eps=(eps[t, 0, lo:hi], eps[t, 1, lo:hi]))
            losses.append(float(loss))
        np.savez(out_path + f".{rank}.npz", blob=m._blob.numpy(), mu=q.mean.numpy(), losses=np.asarray(losses), lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.is_available(), reason="the stand-in backend works on CPU tensors")
def test_two_rank_filter_equals_single_process(tmp_path):
    from tests import fake_backend
    undo = fake_backend.install()
    try:
        m = _make()
        y, u, eps = _data()
        q, ref_losses = None, []
        for t in range(y.shape[0]):
            q, loss = m.filter(y[t], u[t], q, eps=(eps[t, 0], eps[t, 1]))
            ref_losses.append(float(loss))
        ref_blob, ref_mu = m._blob.numpy().copy(), q.mean.numpy().copy()
    finally:
        undo()
    out = str(tmp_path / "rank")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = [np.load(out + f".{k}.npz") for k in range(2)]
    for k in range(2):
        np.testing.assert_allclose(r[k]["losses"], ref_losses, rtol=1e-5)             # global loss on every rank
        np.testing.assert_allclose(r[k]["blob"], ref_blob, rtol=2e-4, atol=1e-6)      # replicated state == single-process state
        np.testing.assert_allclose(r[k]["mu"], ref_mu[int(r[k]["lo"]):int(r[k]["hi"])], rtol=1e-4, atol=1e-6)
    np.testing.assert_array_equal(r[0]["blob"], r[1]["blob"])                         # ranks stay bit-identical
