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


# ---------------------------------------------------------------------------------------------------------------- on the GPU
def _gpu_data():
    g = torch.Generator().manual_seed(19)
    T, B = 3, 80                                    # 40 trials per rank: a ragged second tile on each
    return torch.randn(T, B, 8, generator=g), torch.randn(T, B, 1, generator=g), torch.randn(T, 2, B, 3, generator=g)


def _gpu_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VJF_NATIVE_RCCL="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)                        # both ranks on the one GPU of the box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _make()
        y, u, eps = _gpu_data()
        B = y.shape[1]
        lo, hi = rank * B // world, (rank + 1) * B // world
        # steps 0, 1 through filter(), the rest through filter_sequence(): both take the caller-side route
        q, losses = None, []
        for t in range(2):
            q, loss = m.filter(y[t, lo:hi].cuda(), u[t, lo:hi].cuda(), q, eps=(eps[t, 0, lo:hi].cuda(), eps[t, 1, lo:hi].cuda()))
            losses.append(float(loss))
        mu, lv, ls = m.filter_sequence(y[2:, lo:hi].cuda(), u[2:, lo:hi].cuda(), q, eps=eps[2:, :, lo:hi].cuda())
        losses += [float(v) for v in ls[:, 0]]
        assert m.status() == 0
        np.savez(out_path + f".{rank}.npz", blob=m._blob.cpu().numpy(), mu=mu[-1].cpu().numpy(), losses=np.asarray(losses), lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_caller_side_route(tmp_path):
    """The sharded protocol with REAL kernels and a real second rank: two processes on the one GPU, a gloo group, the all-reduce on
    the caller's side (vjf_filter_local -> sum over ranks -> vjf_filter_global with B_total = 80).  Against the single-process run
    of all 80 trials: the same losses on every rank, replicated states that agree with it and with each other bit for bit."""
    m = _make()
    y, u, eps = _gpu_data()
    mu, lv, ls = m.filter_sequence(y.cuda(), u.cuda(), None, eps=eps.cuda())
    assert m.check_status() == 0                    # (raises on a timed-out hand-off inside the launch)
    ref_losses, ref_blob, ref_mu = ls[:, 0].cpu().numpy(), m._blob.cpu().numpy().copy(), mu[-1].cpu().numpy()
    # the same on this process's per-step kernels: tells a deviation of the single-process reference from one of the ranks
    m1 = _make()
    m1.set_overlap(False)
    _, _, ls1 = m1.filter_sequence(y.cuda(), u.cuda(), None, eps=eps.cuda())
    np.testing.assert_allclose(ref_losses, ls1[:, 0].cpu().numpy(), rtol=2e-5, err_msg="one-launch route vs per-step kernels, single process")
    out = str(tmp_path / "rank")
    mp.spawn(_gpu_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = [np.load(out + f".{k}.npz") for k in range(2)]
    for k in range(2):
        np.testing.assert_allclose(r[k]["losses"], ref_losses, rtol=2e-5)
        np.testing.assert_allclose(r[k]["blob"], ref_blob, rtol=5e-4, atol=2e-5)
        np.testing.assert_allclose(r[k]["mu"], ref_mu[int(r[k]["lo"]):int(r[k]["hi"])], rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(r[0]["blob"], r[1]["blob"])
