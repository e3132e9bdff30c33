#!/usr/bin/env python3
"""Headline benchmark: VJF online filtering step throughput (trial-timesteps/sec) + ELBO.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md 8d "config B"): 4096 trials per GPU, d_z=10, d_y=50, 200 RBF centres,
recognition hidden=[128], Gaussian likelihood, fp32, synthetic "RBF data", explicit pre-drawn noise, sgd=True / update=True /
warm_up=False.  One "step" = one VJF.filter call on the whole batch (everything of the step, incl. the serial RLS update).
W untimed warm-up steps, then EXACTLY K timed steps bracketed by barrier + synchronize -> `value`.  After that region the
sequence continues for --repeats - 1 further regions of K steps (reported as `ms_per_step_repeats` with their median: the
spread of the measurement; they do not enter `value`).  The ELBO of the first timed steps is compared with the
fp64 oracle run from the model's own state on the same inputs (after the timed region).  With N > 1 GPUs the trials are
sharded (4096 per GPU, weak scaling); the sums over trials go through RCCL inside the step.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own ranks: a fresh child
(`python -m torch.distributed.run --nproc-per-node N ... bench.py <the same arguments>`), decided before anything of this
process has touched the GPU; the child's rank 0 prints the line, this process relays it and exits with the child's code.

Prints ONE JSON line on rank 0.
"""
import argparse
import copy
import json
import os
import socket
import subprocess
import sys
import time


def _launch_own_ranks(argv):
    """N > 1 without a launcher's environment: run the ranks as children of a fresh torch.distributed.run and relay rank 0's
    line.  This process never initialises the GPU (no re-exec of a process that has: it only waits for the child)."""
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1 or "WORLD_SIZE" in os.environ:
        return None
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    def attempt(extra):
        with socket.socket() as s:                               # a free port for the rendezvous
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + list(argv) + extra
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=float(os.environ.get("VJF_BENCH_ATTEMPT_TIMEOUT", "900")))
            rc, out = r.returncode, r.stdout
        except subprocess.TimeoutExpired as e:
            rc, out = 124, e.stdout or b""
        return rc, [l for l in out.decode("utf-8", "replace").splitlines() if l.startswith("{") and '"metric"' in l]
    rc, lines = attempt([])
    # The in-library RCCL route has never run with >= 2 ranks on the builder's one-GPU boxes.  The ranks themselves fall back to the
    # caller-side sums when a wait of its warm-up steps gives up; a run that dies or hangs instead is started over once, on the
    # caller-side route from the beginning -- the line then says so ("sums_over_ranks": "caller").
    if (rc != 0 or not lines) and "--dist-route" not in " ".join(argv) and os.environ.get("VJF_BENCH_ROUTE", "native") == "native":
        print(f"bench.py: the ranks ended with code {rc}{'' if lines else ' and no result line'}; once more with --dist-route caller", file=sys.stderr)
        rc, lines = attempt(["--dist-route", "caller"])
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    if rc == 0 and not lines:
        print("bench.py: the ranks exited 0 without a result line", file=sys.stderr)
        return 4
    return rc


if __name__ == "__main__":
    _rc = _launch_own_ranks(sys.argv[1:])
    if _rc is not None:
        sys.exit(_rc)

if os.environ.get("VJF_BENCH_MAPS"):      # diagnostic: the process's load map, written by the LAST exit hook of the interpreter
    import atexit

    def _dump_maps(dst=os.environ["VJF_BENCH_MAPS"]):
        with open("/proc/self/maps") as f, open(dst, "w") as o:
            o.write(f.read())
    atexit.register(_dump_maps)

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFGS = {"A": dict(B=1, dz=3, dy=10, du=0, n=100, hidden=[20], lik="gaussian"),                     # configs[0]: one Lorenz trial (plumbing)
        "B": dict(B=4096, dz=10, dy=50, du=0, n=200, hidden=[128], lik="gaussian"),                # configs[1]: the headline workload
        "C": dict(B=4096, dz=10, dy=200, du=0, n=200, hidden=[128], lik="poisson"),                # configs[2]
        "E": dict(B=4096, dz=64, dy=512, du=0, n=1000, hidden=[512, 512], lik="gaussian"),         # configs[4]
        # configs[3] UNSHARDED: its 32768 trials on ONE GPU (the comparator of the 8-way shard, which is `--gpus 8` of config B)
        "D1": dict(B=32768, dz=10, dy=50, du=0, n=200, hidden=[128], lik="gaussian")}
CFG_INDEX = dict(A=0, B=1, C=2, E=4, D1=3)
# torch.distributed is the control plane only (rendezvous, barrier, max of the wall times, the RCCL ids)
CTRL_BACKEND = os.environ.get("VJF_BENCH_BACKEND", "nccl")
PEAK_FP32_TFLOPS = 157.3     # MI355X_MICROARCH.md: f32 vector == f32 MFMA peak (v_mfma_f32_*_f32 is exact fp32 at that rate)
PEAK_HBM_GBS = 8000.0
ELBO_RTOL = 5e-5             # tolerance of the parity tests (fp32 device path against the fp64 oracle)
# measured by the survey on the UNMODIFIED reference, 8 Xeon cores, torch CPU fp32: trial-timesteps/s
REFERENCE_CPU = {"A": {"as_is": 533}, "B": {"as_is": 96700, "with_O(B)_variance": 256600},
                 "D1": {"as_is": 14500, "with_O(B)_variance": 831500},
                 "C": {"as_is": 109900, "with_O(B)_variance": 336500}, "E": {"as_is": 393, "with_usable_rbf_init": 25100}}


def algorithmic_work(c, flags="train"):
    """FLOPs and bytes per trial-timestep, SURVEY.md 8d (dense count of the O(B) restatement).  `flags`: what the steps run with --
    train is SURVEY's figure; the others drop the terms their steps do not have (warmup: no Phi^T Phi, Phi^T dx, Cholesky / solves;
    sgd-only: no closed-form updates at all; infer: the forward pass, the predictive moments and the loss terms only)."""
    dz, dy, du, n, h = c["dz"], c["dy"], c["du"], c["n"], c["hidden"]
    din = dy + du + 2 * dz
    chain = sum(a * b for a, b in zip(h[:-1], h[1:]))
    rec_fwd = 2 * (din * h[0] + chain + 2 * h[-1] * dz)
    rec_bwd = 2 * din * h[0] + 4 * (chain + 2 * h[-1] * dz)
    flops = 2 * n * (dz + du) + 6 * n * dz + 4 * n * n + rec_fwd + rec_bwd + 6 * dz * dy
    if flags == "warmup":
        flops -= 2 * n * n + 2 * n * dz
    elif flags == "sgd-only":
        flops -= 2 * n * n + 4 * n * dz
    elif flags == "infer":
        flops = 2 * n * (dz + du) + 2 * n * dz + 2 * n * n + rec_fwd + 2 * dz * dy
    ntheta = din * h[0] + h[0] + sum(a * b + b for a, b in zip(h[:-1], h[1:])) + 2 * h[-1] * dz + dz + dy * dz + dy + 1
    bytes_trial = 4 * (dy + du + 6 * dz)
    bytes_shared = 4 * (2 * ntheta + n * (dz + du + 1 + 2 * dz) + 4 * n * n)
    serial_flops = (2 * n ** 3) // 3 + 4 * n * n * dz if flags == "train" else 0
    return flops, bytes_trial, bytes_shared, serial_flops


def synth_data(c, T, seed, device, config):
    """config A: one z-scored Lorenz trial; else the 'RBF data' of SURVEY.md 8d.  y = x C + d + 0.1 N(0,1) (script/example.py:23-33)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    B, dz, dy = c["B"], c["dz"], c["dy"]
    cen = (torch.rand(50, dz, generator=g) * 4 - 2).to(device)
    Wt = (0.1 * torch.randn(50, dz, generator=g)).to(device)
    C = torch.randn(dz, dy, generator=g).to(device)
    d = torch.randn(dy, generator=g).to(device)
    gd = torch.Generator(device=device).manual_seed(seed + 1)
    if config == "A":
        from vjf_amd.data import lorenz
        x = lorenz(T).float().to(device)[:, None, :]
        return x @ C + d + 0.1 * torch.randn(T, B, dy, device=device, generator=gd)
    x = torch.randn(B, dz, device=device, generator=gd)
    y = torch.empty(T, B, dy, device=device)
    w2 = float(dz)
    for t in range(T):
        d2 = torch.cdist(x, cen) ** 2
        x = x + torch.exp(-0.5 * d2 / w2) @ Wt + 0.1 * torch.randn(B, dz, device=device, generator=gd)
        y[t] = x @ C + d + 0.1 * torch.randn(B, dy, device=device, generator=gd)
    if c["lik"] == "poisson":                                   # spike counts with log-rate = the linear read-out (clamped)
        y = torch.poisson(torch.exp(torch.clamp(0.5 * y - 1.0, max=3.0)), generator=gd)
    return y


def cpu_baseline(c, config, s32, q0, y_cpu, eps_cpu, budget_s=12.0):
    """The oracle -- a numpy/BLAS port of the reference's step, started from the SAME state and inputs as the timed GPU run, in
    fp32 as the reference runs -- timed on this box's host cores on a bounded sample: the O(B) variance form (row sums of squares,
    squared distances by one GEMM as torch.cdist does), then two steps of the reference's own cost form (the (B, B) product whose
    diagonal vjf/module.py:76 takes)."""
    from oracle import vjf_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    s = copy.deepcopy(s32)
    mu, lv = (None, None) if q0 is None else (q0[0].astype(np.float32), q0[1].astype(np.float32))
    y_cpu, eps_cpu = y_cpu.astype(np.float32), eps_cpu.astype(np.float32)
    orc.RBF_GEMM = True
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    step = [0]

    def run(nsteps, **kw):
        nonlocal mu, lv
        t0 = time.perf_counter()
        for _ in range(nsteps):
            t = step[0]; step[0] += 1
            o = orc.filter_step(s, y_cpu[t], None, mu, lv, eps_cpu[t, 0], eps_cpu[t, 1], **kw)
            mu, lv = o.mu_t, o.lv_t
        return (time.perf_counter() - t0) / nsteps
    try:
        run(2)                                                   # discard two steps (allocator / thread-pool warm-up)
        # BLAS threads: these matrices are small, every core of a large host is not the fastest setting -- take the best of a few
        best = (run(3), threads)
        if threadpool_limits is not None:
            for nt in (8, 16, 32, 64):
                if nt < threads:
                    with threadpool_limits(limits=nt):
                        tt = run(3)
                    if tt < best[0]:
                        best = (tt, nt)
        t1, threads = best
        nstep = int(min(max(budget_s / max(t1, 1e-5), 5), y_cpu.shape[0] - step[0] - 3, 400))
        nf = 2
        if threadpool_limits is not None:
            with threadpool_limits(limits=threads):
                dt = run(nstep) * nstep
                dtf = run(nf, faithful_cost=True) * nf
        else:
            dt = run(nstep) * nstep
            dtf = run(nf, faithful_cost=True) * nf
        val = c["B"] * nstep / dt
    finally:
        orc.RBF_GEMM = False
    ref = REFERENCE_CPU.get(config, {})
    return {"value": val, "unit": "trial-timesteps/s", "cores": int(os.cpu_count() or 1), "blas_threads": int(threads), "kind": "port",
            "sample": f"(host cores: {os.cpu_count()}; BLAS threads: the fastest of 8/16/32/64/all over 3 steps each = {int(threads)}) {nstep} steps of the bench workload from the timed run's own state (B={c['B']}, fp32 numpy/BLAS oracle, O(B) variance "
                      f"form, GEMM distances), {dt:.1f} s; the reference's cost form ((B,B) product for its diagonal): "
                      f"{c['B'] * nf / dtf:.0f} trial-timesteps/s over {nf} steps; the UNMODIFIED reference (torch CPU) on the survey's 8 Xeon "
                      f"cores, BASELINE.md section 2: {json.dumps(ref)}",
            "reference_cost_form_value": c["B"] * nf / dtf, "reference_measured_8_xeon_cores": ref}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="regions of --steps steps each (the sequence continues): the first is `value`, "
                                                           "all are listed with their median")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-elbo-check", action="store_true")
    ap.add_argument("--elbo-steps", type=int, default=10)
    ap.add_argument("--no-overlap", action="store_true", help="A/B: the per-step kernels in the one-stream order")
    ap.add_argument("--no-call-cost", action="store_true", help="skip the fixed-cost-of-a-call measurement behind the timed regions (profiler runs)")
    ap.add_argument("--streams-route", action="store_true", help="A/B: the per-step three-stream route instead of the one-launch route")
    ap.add_argument("--config", default="B", choices=sorted(CFGS),
                    help="B: the headline workload (BASELINE configs[1]); A / C / E: configs[0] / [2] / [4]; D1: configs[3]'s 32768 trials "
                         "on ONE GPU -- extra lines, not the headline")
    ap.add_argument("--flags", default="train", choices=["train", "warmup", "infer", "sgd-only"],
                    help="the flags of VJF.filter the steps run with: train = sgd + update (the headline), warmup = sgd + update + "
                         "warm_up (the first epochs of fit, vjf/model.py:243-259), infer = sgd=False, update=False (a deployed filter), "
                         "sgd-only = sgd, no update")
    ap.add_argument("--collectives", type=int, default=2, choices=[1, 2],
                    help="N > 1, in-library route: sums over ranks per step -- 2: [grad | loss sums] and [G | Phi^T dx | sums] on the two "
                         "chains of the three-stream schedule; 1: the whole reduce buffer in ONE all-reduce (SURVEY 8e), one stream")
    ap.add_argument("--dist-route", default=os.environ.get("VJF_BENCH_ROUTE", "native"), choices=["native", "caller"],
                    help="N > 1: the sums over ranks inside the library (RCCL communicators in the context) or on the caller's side "
                         "(vjf_filter_local / torch.distributed.all_reduce / vjf_filter_global)")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1 only: run the sharded route (RCCL communicators inside the context) with a one-rank group")
    a = ap.parse_args()

    # stdout carries the ONE JSON line and nothing else: libraries that write to file descriptor 1 (RCCL prints a version banner
    # there when a communicator is made) go to stderr for the length of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world} (a launcher's environment with another rank count)"
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    ndev = max(1, torch.cuda.device_count())
    # more ranks than GPUs: a rehearsal of the N > 1 path on a box with fewer cards (tests/test_gpu_bench_ranks.py: two ranks on the
    # one GPU) -- ranks share devices, RCCL cannot span two ranks of one device, so the sums over ranks stay on the caller's side
    # and torch.distributed runs on gloo
    shared_gpu = world > ndev
    ctrl = "gloo" if shared_gpu else CTRL_BACKEND
    if shared_gpu and os.environ.get("VJF_BENCH_TEST_SAFE") != "1":
        a.dist_route = "caller"
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.dist_route == "caller":
            os.environ["VJF_NATIVE_RCCL"] = "0"
        dist.init_process_group(ctrl, **({"device_id": dev} if ctrl == "nccl" else {}))
    elif a.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ["VJF_FORCE_DIST"] = "1"
        dist.init_process_group(ctrl, rank=0, world_size=1, **({"device_id": dev} if ctrl == "nccl" else {}))

    import vjf_amd
    c = dict(CFGS[a.config])
    K, W, R = a.steps, a.warmup, max(1, a.repeats)
    if a.config == "E" and a.steps == 200:
        K, W, R = 40, 5, 3                                      # ms-scale steps: keep the default run short
        a.elbo_steps = min(a.elbo_steps, 2)                     # (an fp64 oracle step takes seconds at this size)
    if a.config == "D1":
        a.elbo_steps = min(a.elbo_steps, 3)
        if a.steps == 200:
            K, W, R = 100, 10, 3
    if world > 1:
        a.elbo_steps = min(a.elbo_steps, 2)                     # (the fp64 oracle on the batch of ALL ranks)
    if a.config == "A" and a.steps == 200:
        K = 2000                                                # T_meas = 2000 for the single-trial configuration
    T = W + K * R
    fkw = {"train": dict(sgd=True, update=True, warm_up=False), "warmup": dict(sgd=True, update=True, warm_up=True),
           "infer": dict(sgd=False, update=False, warm_up=False), "sgd-only": dict(sgd=True, update=False, warm_up=False)}[a.flags]

    def barrier_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def dist_share(d):
        """rank 0's dict on every rank (the watchdog of any rank may have to end the run; only rank 0 writes)"""
        if world == 1:
            return d
        box = [d]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def run_once(route_choice, stash):
        """the whole measurement on one route of the sums over ranks -> (the JSON line's dict on rank 0, exit code, the model)"""
        a.dist_route = route_choice
        if world > 1:
            if route_choice == "caller":
                os.environ["VJF_NATIVE_RCCL"] = "0"
            else:
                os.environ.pop("VJF_NATIVE_RCCL", None)
        def make_model():
            torch.manual_seed(0)                                    # identical parameters on every rank
            m = vjf_amd.VJF.make_model(c["dy"], c["dz"], c["du"], c["n"], c["hidden"], likelihood=c["lik"], noise="device")
            if c["dz"] >= 32:  # the default RBF init underflows every feature at d_z = 64: SURVEY 8d's init
                r = float(np.sqrt(c["dz"]))
                m.transition.velocity.feature.centroid.uniform_(-r, r)
                m.transition.velocity.feature.logwidth.fill_(float(np.log(r)))
            if a.no_overlap:
                m.set_overlap(False)
            if a.streams_route:
                m.set_overlap(3)
            if a.collectives == 1:
                m.set_collectives(1)
            return m
        model = make_model()
        Td = max(T, 100) if a.config == "B" else T                  # (the call-cost measurement behind the timed regions takes 100 steps)
        y = synth_data(c, Td, 1234 + rank, dev, a.config)           # each rank filters its own trials
        eps = torch.randn(Td, 2, c["B"], c["dz"], device=dev, generator=torch.Generator(device=dev).manual_seed(4321 + rank))
        cdev = dev if ctrl == "nccl" else "cpu"                      # where the control plane's tensors live

        def barrier():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        def agree_max(v):
            """max over ranks of a small non-negative integer (control plane)"""
            if world == 1:
                return int(v)
            t = torch.tensor([int(v)], device=cdev, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return int(t.item())

        def gather_batch(t, dim):
            """rank 0: the tensors of all ranks joined along the trial axis `dim` (rank r holds trials [r B, (r + 1) B)); else None"""
            if world == 1:
                return t
            src = t.contiguous().to(cdev)
            parts = [torch.empty_like(src) for _ in range(world)]
            dist.all_gather(parts, src)
            return torch.cat(parts, dim=dim) if rank == 0 else None

        # warm-up (untimed): also sizes the context and, for N > 1, the RCCL communicators
        def warm_up_steps(m):
            q_ = None
            if W > 0:                                               # (two calls when W > 1: the second takes the posterior of the first, as the timed calls do)
                w1 = W - 1 if W > 1 else W
                mu_, lv_, _ = m.filter_sequence(y[:w1], eps=eps[:w1], **fkw)
                q_ = vjf_amd.Gaussian(mu_[-1], lv_[-1])
                if w1 < W:
                    mu_, lv_, _ = m.filter_sequence(y[w1:W], qs=q_, eps=eps[w1:W], **fkw)
                    q_ = vjf_amd.Gaussian(mu_[-1], lv_[-1])
            torch.cuda.synchronize()
            return q_
        native_error = None
        watchdog = None
        if world > 1 and a.dist_route == "native":
            # (a collective that never completes -- a rank that died, a route that deadlocks -- would hang the run for good: the warm-up
            #  steps get two minutes, then the rank ends itself and its launcher starts the run over on the caller-side route)
            import threading

            def _give_up():
                print(f"bench.py[rank {rank}]: the in-library RCCL route did not finish in time", file=sys.stderr, flush=True)
                if stash is not None:                                # (the caller-side route's line of this same run, measured first)
                    if rank == 0:
                        stash["dist"]["native_route_error"] = "the in-library RCCL route did not finish in time: this is the caller-side route's line"
                        os.write(json_fd, (json.dumps(stash) + "\n").encode())
                    os._exit(0)
                os._exit(86)
            watchdog = threading.Timer(float(os.environ.get("VJF_BENCH_WARMUP_TIMEOUT", "120")), _give_up)
            watchdog.daemon = True
            watchdog.start()
        try:
            q = warm_up_steps(model)
            wst = model.status()
            if wst & model._WAIT_BITS:
                native_error = f"a device-side wait timed out during the warm-up steps (status 0x{wst:x})"
        except Exception as e:                                       # (an error return of the library: the process is intact)
            if world == 1:
                raise
            native_error = f"{type(e).__name__}: {e}"
        if os.environ.get("VJF_BENCH_FAKE_NATIVE_HANG") == "1" and world > 1 and a.dist_route == "native" and stash is not None:
            time.sleep(3600)                                         # (test hook: the in-library route never comes back)
        if os.environ.get("VJF_BENCH_FAKE_NATIVE_FAILURE") == "1" and world > 1 and not getattr(main, "_faked", False):
            main._faked = True                                       # (test hook: rank 0 reports a failed warm-up once, as the in-library route would)
            if rank == 0:
                native_error = "injected (VJF_BENCH_FAKE_NATIVE_FAILURE)"
            a.dist_route = "native"
        if world > 1 and agree_max(1 if native_error else 0):
            # The in-library RCCL route failed on some rank (it cannot be rehearsed with >= 2 ranks on the builder's one-GPU boxes): every
            # rank starts over on the caller-side route -- a fresh model from the same seed, the sums over ranks through torch.distributed
            if a.dist_route == "caller":
                raise RuntimeError(f"bench.py: the warm-up steps failed on the caller-side route: {native_error}")
            print(f"bench.py[rank {rank}]: leaving the in-library RCCL route ({native_error or 'another rank failed'}); "
                  "caller-side all-reduce from here", file=sys.stderr)
            if watchdog is not None:
                watchdog.cancel()
                watchdog = None
            os.environ["VJF_NATIVE_RCCL"] = "0"
            a.dist_route = "caller (the in-library RCCL route failed in the warm-up steps)"
            try:
                model.close()
            except Exception:
                pass
            model = make_model()
            q = warm_up_steps(model)
        # (the allocator's cache holds blocks of the timed regions' output sizes before the first of them runs, as it does in any loop
        #  that has been running for a while: two sets, a region's outputs are alive while the next one's are allocated)
        # (filter_sequence makes ONE allocation for its three outputs: blocks of exactly that size)
        prime = [torch.empty(2 * K * c["B"] * c["dz"] + 4 * K, device=dev) for _ in range(2)]
        del prime
        # the state the timed region starts from, kept on the device (the oracle's copies -- the checker of the ELBO and the CPU
        # baseline; nothing of the timed path goes through them -- are made from it AFTER the timed regions: no host work, and no idle
        # device, between the warm-up steps and the timed ones).  N > 1: every rank holds the same state (the sums over ranks are
        # bit-identical on all of them); rank 0's copy serves, the posteriors of all ranks are gathered behind the timed regions.
        want_check = not (a.no_elbo_check and (a.no_cpu_baseline or world > 1))
        checker = rank == 0 and want_check
        snap = model._blob.clone() if checker else None
        q0d = None if (q is None or not want_check) else (q.mean.clone(), q.logvar.clone())
        barrier()
        walls, devs, enqs, elbos, first_losses = [], [], [], [], None
        stream = torch.cuda.current_stream()                         # the stream vjf_filter_seq launches on (vjf_set_stream)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(R)]
        for e0, e1 in evs:                                           # (the events exist before the first timed region)
            e0.record(stream); e1.record(stream)
        for r in range(R):
            lo = W + r * K
            ev0, ev1 = evs[r]
            barrier()
            ev0.record(stream)                                       # (the stream is empty: the event is the device-side start of the region)
            t0 = time.perf_counter()
            mu, lv, loss = model.filter_sequence(y[lo:lo + K], qs=q, eps=eps[lo:lo + K], **fkw)
            enq = time.perf_counter() - t0                           # host time to enqueue the K steps (asynchronous)
            ev1.record(stream)
            barrier()
            wall = time.perf_counter() - t0
            tt = torch.tensor([wall], device=cdev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            walls.append(float(tt.item())); devs.append(ev0.elapsed_time(ev1) * 1e-3); enqs.append(enq)
            elbos.append(float(-loss[:, 0].mean().item()))
            if r == 0:
                first_losses = loss[:min(a.elbo_steps, K)].cpu().numpy().astype(np.float64)
            q = vjf_amd.Gaussian(mu[-1], lv[-1])
        if watchdog is not None:
            watchdog.cancel()
        status = agree_max(model.status())                           # (bits are small integers: the max over ranks shows any wait bit)
        route = model.route(sgd=fkw["sgd"], update=fkw["update"], warm_up=fkw["warm_up"])
        rccl_ranks = model.comm_ranks() if world > 1 or a.force_dist else None
        if world > 1 and not getattr(model, "_comm_ok", False):
            route = f"caller-side all-reduce (vjf_filter_local / torch.distributed[{ctrl}] / vjf_filter_global per step)"
        wall_max, dev_s, enq = walls[0], devs[0], enqs[0]           # `value`: the first region = exactly K steps after W warm-up steps
        ne = first_losses.shape[0]
        # what the checker needs of the other ranks: their trials' inputs of the first `ne` timed steps and the posterior they started from
        y_chk = eps_chk = q0 = None
        if want_check and not a.no_elbo_check:
            y_chk = gather_batch(y[W:W + ne], 1)
            eps_chk = gather_batch(eps[W:W + ne], 2)
            if q0d is not None:
                qm, ql = gather_batch(q0d[0], 0), gather_batch(q0d[1], 0)
                if rank == 0:
                    q0 = (qm.cpu().numpy().astype(np.float64), ql.cpu().numpy().astype(np.float64))
        elif want_check and q0d is not None and world == 1:
            q0 = (q0d[0].cpu().numpy().astype(np.float64), q0d[1].cpu().numpy().astype(np.float64))
        s64 = s32 = None
        if checker:
            from tests.helpers import load_oracle_state
            torch.manual_seed(0)
            twin = vjf_amd.VJF.make_model(c["dy"], c["dz"], c["du"], c["n"], c["hidden"], likelihood=c["lik"], noise="device")
            for g_t, g_m in zip(twin.optimizer.param_groups, model.optimizer.param_groups):
                g_t["lr"] = g_m["lr"]
            twin._blob.copy_(snap)
            s64, s32 = load_oracle_state(twin, np.float64), load_oracle_state(twin, np.float32)
            del twin

        # ELBO of the first timed steps against the oracle on identical state / inputs / noise (N > 1: on the trials of ALL ranks)
        elbo_check = None
        if s64 is not None and not a.no_elbo_check:
            from oracle import vjf_oracle as orc
            s = copy.deepcopy(s64)
            om, ol = (None, None) if q0 is None else q0
            ref = np.empty(ne)
            yc, ec = y_chk.cpu().numpy().astype(np.float64), eps_chk.cpu().numpy().astype(np.float64)
            for t in range(ne):
                o = orc.filter_step(s, yc[t], None, om, ol, ec[t, 0], ec[t, 1], **fkw)
                om, ol = o.mu_t, o.lv_t
                ref[t] = o.loss
            rel = float(np.max(np.abs(first_losses[:, 0] - ref) / np.maximum(np.abs(ref), 1e-12)))
            elbo_check = {"steps": int(ne), "trials": int(yc.shape[1]), "max_rel_err": rel, "rtol": ELBO_RTOL, "elbo_gpu": float(-first_losses[:, 0].mean()),
                          "elbo_oracle_fp64": float(-ref.mean()), "ok": bool(rel < ELBO_RTOL)}

        exit_code = 0
        result = {"out": None}
        if rank == 0:
            flops, b_trial, b_shared, serial_flops = algorithmic_work(c, a.flags)
            units = c["B"] * world * K
            value = units / wall_max
            step_s = dev_s / K
            ach_tf = (flops * c["B"] + serial_flops) / step_s / 1e12
            ach_gbs = (b_trial * c["B"] + b_shared) / step_s / 1e9
            traffic = traffic_note = None
            try:                                                    # HBM-side bytes per step from the committed PMC passes of this build
                if a.config != "B" or world != 1 or a.no_overlap or a.streams_route or a.force_dist or a.flags != "train":
                    raise LookupError("the committed PMC passes are of the headline configuration on the one-launch route")
                tj = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")))
                traffic, traffic_note = float(tj["bytes_per_step_corrected"]), tj.get("note")
            except Exception:
                pass
            kstats = None
            try:                                                    # the committed rocprofv3 --kernel-trace --stats summary of this build
                import csv
                if a.config == "E":                                 # (tools/profile_configE.sh: every kernel of the per-step route)
                    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r04_configE_kernel_stats.csv"))))
                    nst = 23
                    kstats = {"file": "profiles/r04_configE_kernel_stats.csv",
                              "command": "rocprofv3 --kernel-trace --stats -- python bench.py --config E --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-elbo-check",
                              "steps_in_profile": nst,
                              "kernels": [{"name": r_["Name"].split("(")[0].replace("void ", ""), "calls": int(r_["Calls"]), "avg_us": float(r_["AverageNs"]) / 1e3,
                                           "us_per_step": float(r_["TotalDurationNs"]) / 1e3 / nst} for r_ in rows if "vjf_" in r_["Name"]]}
                elif a.config == "B":
                    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r04_kernel_stats.csv"))))
                    meta = json.load(open(os.path.join(ROOT, "profiles", "r04_kernel_stats_meta.json")))
                    nst = meta.get("steps_in_all_launches")
                    kstats = {"file": "profiles/r04_kernel_stats.csv", "command": meta.get("command"), "steps_per_launch": meta.get("steps_per_launch"),
                              "note": meta.get("note"), "mega_launches_us": meta.get("mega_launches_us"),
                              "kernels": [{"name": r_["Name"].split("(")[0].replace("void ", ""), "calls": int(r_["Calls"]), "avg_us": float(r_["AverageNs"]) / 1e3,
                                           "max_us": float(r_["MaxNs"]) / 1e3,
                                           "us_per_step": (float(r_["TotalDurationNs"]) / 1e3 / nst) if (nst and "mega" in r_["Name"]) else None}
                                          for r_ in rows if "vjf_" in r_["Name"]]}
            except Exception:
                pass
            one_launch = world == 1 and not a.no_overlap and not a.streams_route and not a.force_dist and route == "one-launch"
            spl = K if one_launch else 1
            # the fixed cost of a call (one-launch route: host shim + launch + filling and draining the roles' pipeline + the wake-up of
            # the synchronising host), measured BEHIND the timed regions: synchronised calls of 20 and of 100 steps -- the slope is the
            # steady step, the intercept the fixed cost -- and single `filter` calls, each followed by a synchronisation
            call_cost = None
            if one_launch and a.config == "B" and a.flags == "train" and y.shape[0] >= 100 and not a.no_call_cost:
                def timed_calls(nst, reps):
                    ts = []
                    qq = q
                    for _ in range(reps):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        m_, l_, _ = model.filter_sequence(y[:nst], qs=qq, eps=eps[:nst])
                        torch.cuda.synchronize()
                        ts.append(time.perf_counter() - t0)
                        qq = vjf_amd.Gaussian(m_[-1], l_[-1])
                    return float(np.median(ts))
                timed_calls(20, 2)
                t20, t100 = timed_calls(20, 7), timed_calls(100, 5)
                steady = (t100 - t20) / 80.0
                ts, qq = [], q
                for i in range(24):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    qq, _ = model.filter(y[i], None, qq, eps=(eps[i, 0], eps[i, 1]))
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                call_cost = {"fixed_us_per_call": (t20 - 20.0 * steady) * 1e6, "steady_us_per_step": steady * 1e6,
                             "call_20_steps_us": t20 * 1e6, "call_100_steps_us": t100 * 1e6,
                             "single_filter_call_us": float(np.median(ts[4:])) * 1e6,
                             "note": "wall clock, host synchronised before and after every call; medians of 7 / 5 / 20 calls behind the timed regions"}
            out = {
                "metric": "trial-timesteps/sec", "value": value, "unit": "trial-timesteps/s", "n_gpus": world, "steps": K,
                "warmup": W, "ms_per_step": wall_max / K * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": (f"BASELINE configs[{CFG_INDEX[a.config]}]: VJF.filter, {c['B']} trials/GPU, d_z={c['dz']}, d_y={c['dy']}, "
                                        f"RBF({c['n']}), hidden={c['hidden']}, {c['lik'].capitalize()} likelihood, "
                                        + {"train": "sgd+update", "warmup": "sgd+update+warm_up", "infer": "sgd=False, update=False",
                                           "sgd-only": "sgd, update=False"}[a.flags] + ", explicit noise"
                                        + (", one Lorenz trial" if a.config == "A" else "")
                                        + (" (configs[3] unsharded: all 32768 trials on ONE GPU)" if a.config == "D1" else "")),
                           "global_batch": c["B"] * world, "trials_per_gpu": c["B"], "parallelism": f"trial-shard x{world}",
                           "flags": a.flags},
                "route": route,
                "dist": None if world == 1 and not a.force_dist else {
                    "sums_over_ranks": a.dist_route, "collectives_per_step": a.collectives, "control_plane": ctrl, "rccl_comm_ranks": rccl_ranks,
                    "ranks_share_a_gpu": bool(shared_gpu), "native_route_error": native_error},
                "repeats": R, "ms_per_step_repeats": [w / K * 1e3 for w in walls], "ms_per_step_median": float(np.median(walls)) / K * 1e3,
                "elbo": elbos[0], "elbo_check": elbo_check, "status_bits": status, "call_cost": call_cost,
                "roofline": {"bound": "mfma", "achieved": ach_tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                             "frac": ach_tf / PEAK_FP32_TFLOPS, "traffic": None if traffic is None else traffic * spl, "traffic_note": traffic_note,
                             "kernel": ("vjf_mega_kernel: ONE launch carries the K timed steps (the trial, Gram, operand, SGD and RLS "
                                        "roles are workgroups of one grid that is resident as a whole); a launch processes K x trials trial-timesteps; its duration "
                                        "is measured with HIP events on its stream around the timed region" if one_launch else
                                        "one filter step = the per-step kernels of the route in use (HIP events around the timed region / steps)"),
                             "launch_us": dev_s * 1e6 if one_launch else step_s * 1e6, "steps_per_launch": spl,
                             "algorithmic_flops_per_launch": (flops * c["B"] + serial_flops) * spl,
                             "flops_per_trial_step": flops, "serial_flops_per_step": serial_flops,
                             "rocprof_kernel_averages": kstats,
                             "step_us": step_s * 1e6, "host_enqueue_us_per_step": enq / K * 1e6,
                             "hbm_achieved_GBs": ach_gbs, "hbm_frac": ach_gbs / PEAK_HBM_GBS,
                             "bytes_per_trial_step": b_trial + b_shared / c["B"],
                             "peak_note": ("the fp32 matrix peak; on this part a SIMD runs MFMA or VALU instructions, never both at once "
                                           "(profiles/r04_mfma_valu_overlap.txt), so a kernel with scalar work beside its MFMAs cannot reach it: "
                                           "per step of the headline launch MFMA + VALU issue time is ~35-40 % of a trial workgroup's SIMD cycles "
                                           "(profiles/r04_pmc_sq.json; DESIGN.md section 3), the rest is LDS / L2 / fabric latency")},
            }
            if not a.no_cpu_baseline and s32 is not None and world == 1 and a.flags == "train":
                out["cpu_baseline"] = cpu_baseline(c, a.config, s32, q0, y[W:].cpu().numpy(), eps[W:].cpu().numpy())
            result["out"] = out
            if elbo_check is not None and not elbo_check["ok"]:
                print(f"bench.py: the ELBO of the timed steps differs from the oracle: {elbo_check}", file=sys.stderr)
                exit_code = 3
            if status & model._WAIT_BITS:
                print(f"bench.py: a device-side wait timed out in the timed regions (status 0x{status:x})", file=sys.stderr)
                exit_code = 5
        return result["out"], exit_code, model

    # N > 1 under somebody else's launcher (the round driver's torch.distributed.run): the in-library RCCL route has never run with two
    # real ranks on the builder's one-GPU boxes, and a rank that hangs in it cannot be recovered in-process.  So the caller-side
    # route (torch.distributed's all-reduce between vjf_filter_local and vjf_filter_global: nothing untried in it) is measured
    # FIRST and its line kept; then the in-library route runs under a watchdog -- if it comes back its line is the run's (with the
    # other route's value beside it), if it does not, the kept line goes out and the ranks exit 0.
    out = None
    if world > 1 and a.dist_route == "native" and (not shared_gpu or os.environ.get("VJF_BENCH_TEST_SAFE") == "1"):
        stash, exit_code, model = run_once("caller", None)
        stash = dist_share(stash)
        model.close()
        del model
        barrier_all()
        try:
            out, exit_code, model = run_once("native", stash)
            if rank == 0 and out is not None and out["dist"] is not None:
                out["dist"]["caller_side_route_value"] = stash["value"]
        except Exception as e:
            print(f"bench.py[rank {rank}]: the in-library RCCL route failed ({type(e).__name__}: {e}); the caller-side route's line stands", file=sys.stderr)
            out, model = stash, None
            if rank == 0:
                out["dist"]["native_route_error"] = f"{type(e).__name__}: {e}"
    else:
        out, exit_code, model = run_once(a.dist_route, None)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
    if (world > 1 or a.force_dist) and model is not None:
        model.close()                                           # (RCCL communicators of the context, before the process group goes)
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
