#!/usr/bin/env python3
"""Headline benchmark: VJF online filtering step throughput (trial-timesteps/sec).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md 8d "config B"): 4096 trials per GPU, d_z=10,
d_y=50, 200 RBF centres, recognition hidden=[128], Gaussian likelihood, fp32, synthetic "RBF
data", explicit pre-drawn noise, sgd=True / update=True / warm_up=False.  One "step" = one
VJF.filter call on the whole batch (all kernels of the step, incl. the serial RLS update).
With N > 1 GPUs the trials are sharded (4096 per GPU, weak scaling) with one RCCL all-reduce of
the gradient / RLS-statistics buffer per step.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG = dict(B=4096, dz=10, dy=50, du=0, n=200, hidden=[128], lik="gaussian")     # BASELINE.json configs[1]: the headline workload
OTHER_CFGS = {"C": dict(B=4096, dz=10, dy=200, du=0, n=200, hidden=[128], lik="poisson"),          # configs[2]
              "E": dict(B=4096, dz=64, dy=512, du=0, n=1000, hidden=[512, 512], lik="gaussian")}   # configs[4]
# torch.distributed is the control plane only (rendezvous, barrier, max of the wall times, the RCCL ids): the data path's two
# all-reduces per step are RCCL calls inside vjf_filter_seq on its own streams
CTRL_BACKEND = os.environ.get("VJF_BENCH_BACKEND", "nccl")
PEAK_FP32_TFLOPS = 157.3     # MI355X_MICROARCH.md: f32 vector == f32 MFMA peak
PEAK_HBM_GBS = 8000.0
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic_current.json")   # tools/pmc_traffic.sh on this build


def algorithmic_work(c):
    """FLOPs and bytes per trial-timestep, SURVEY.md 8d (dense count of the O(B) restatement)."""
    dz, dy, du, n, h = c["dz"], c["dy"], c["du"], c["n"], c["hidden"]
    din = dy + du + 2 * dz
    chain = sum(a * b for a, b in zip(h[:-1], h[1:]))
    rec_fwd = 2 * (din * h[0] + chain + 2 * h[-1] * dz)
    rec_bwd = 2 * din * h[0] + 4 * (chain + 2 * h[-1] * dz)
    flops = 2 * n * (dz + du) + 6 * n * dz + 4 * n * n + rec_fwd + rec_bwd + 6 * dz * dy
    ntheta = din * h[0] + h[0] + sum(a * b + b for a, b in zip(h[:-1], h[1:])) + 2 * h[-1] * dz + dz + dy * dz + dy + 1
    bytes_trial = 4 * (dy + du + 6 * dz)
    bytes_shared = 4 * (2 * ntheta + n * (dz + du + 1 + 2 * dz) + 4 * n * n)
    serial_flops = (2 * n ** 3) // 3 + 4 * n * n * dz
    return flops, bytes_trial, bytes_shared, serial_flops


def synth_data(c, T, seed, device):
    """'RBF data' of SURVEY.md 8d: x[t+1] = x[t] + Phi_true(x[t]) W_true + 0.1 xi, y = x C + d + 0.1 N(0,1)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    B, dz, dy = c["B"], c["dz"], c["dy"]
    cen = (torch.rand(50, dz, generator=g) * 4 - 2).to(device)
    Wt = (0.1 * torch.randn(50, dz, generator=g)).to(device)
    C = torch.randn(dz, dy, generator=g).to(device)
    d = torch.randn(dy, generator=g).to(device)
    gd = torch.Generator(device=device).manual_seed(seed + 1)
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


def cpu_baseline(c, y_cpu, eps_cpu, budget_s=15.0):
    """The numpy oracle (a port of the reference's step, O(B) variance form) timed on the host cores
    on a bounded sample of the same workload; plus a short faithful-cost (B x B) sample."""
    from oracle import vjf_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    rng = np.random.default_rng(0)
    s = orc.init_state(c["dy"], c["dz"], c["du"], c["n"], c["hidden"], c["lik"], rng, dtype=np.float32)
    mu = lv = None
    t0 = time.perf_counter()
    o = orc.filter_step(s, y_cpu[0], None, mu, lv, eps_cpu[0, 0], eps_cpu[0, 1])      # discard step 0 (allocator warm-up)
    mu, lv = o.mu_t, o.lv_t
    o = orc.filter_step(s, y_cpu[1], None, mu, lv, eps_cpu[1, 0], eps_cpu[1, 1])
    mu, lv = o.mu_t, o.lv_t
    t1 = (time.perf_counter() - t0) / 2
    nstep = int(min(max(budget_s / max(t1, 1e-4), 5), y_cpu.shape[0] - 4))
    t0 = time.perf_counter()
    for t in range(2, 2 + nstep):
        o = orc.filter_step(s, y_cpu[t], None, mu, lv, eps_cpu[t, 0], eps_cpu[t, 1])
        mu, lv = o.mu_t, o.lv_t
    dt = time.perf_counter() - t0
    val = c["B"] * nstep / dt
    t0 = time.perf_counter()
    nf = 2
    for t in range(2 + nstep, 2 + nstep + nf):
        o = orc.filter_step(s, y_cpu[t], None, mu, lv, eps_cpu[t, 0], eps_cpu[t, 1], faithful_cost=True)
        mu, lv = o.mu_t, o.lv_t
    dtf = time.perf_counter() - t0
    return {"value": val, "unit": "trial-timesteps/s", "cores": int(threads), "kind": "port",
            "sample": f"{nstep} steps of the bench workload (B={c['B']}, fp32 numpy/OpenBLAS oracle, O(B) variance form), "
                      f"{dt:.1f} s; faithful-cost (B x B product, as the reference computes it): "
                      f"{c['B'] * nf / dtf:.0f} trial-timesteps/s over {nf} steps"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown-steps", type=int, default=20)
    ap.add_argument("--no-overlap", action="store_true", help="one-stream order inside filter_sequence (A/B of the schedule)")
    ap.add_argument("--config", default="B", choices=sorted(OTHER_CFGS) + ["B"],
                    help="B: the headline workload (BASELINE configs[1]); C / E: configs[2] / [4], extra lines, not the headline")
    ap.add_argument("--serial-schedule", action="store_true", help="(obsolete: the one-launch route is a single kernel)")
    ap.add_argument("--streams-route", action="store_true", help="A/B: the per-step three-stream route instead of the one-launch route")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1 only: run the sharded path (local half, RCCL all-reduce, global half) with a one-rank group")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1"
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(CTRL_BACKEND, **({"device_id": dev} if CTRL_BACKEND == "nccl" else {}))
    elif a.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ["VJF_FORCE_DIST"] = "1"
        dist.init_process_group(CTRL_BACKEND, rank=0, world_size=1, **({"device_id": dev} if CTRL_BACKEND == "nccl" else {}))

    import vjf_amd
    from vjf_amd import _native as N
    c = dict(CFG if a.config == "B" else OTHER_CFGS[a.config])
    K, W = a.steps, a.warmup
    if a.config == "E" and a.steps == 500:
        K, W = 40, 5                                            # ms-scale steps: keep the default run short
    T = W + K
    torch.manual_seed(0)                                       # identical parameters on every rank
    model = vjf_amd.VJF.make_model(c["dy"], c["dz"], c["du"], c["n"], c["hidden"], likelihood=c["lik"], noise="device")
    if a.no_overlap:
        model.set_overlap(False)
    if a.streams_route:
        model.set_overlap(3)
    y = synth_data(c, T, 1234 + rank, dev)                      # each rank filters its own trials
    eps = torch.randn(T, 2, c["B"], c["dz"], device=dev, generator=torch.Generator(device=dev).manual_seed(4321 + rank))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up (untimed): also sizes the context and, for N > 1, the RCCL communicator
    q = None
    if W > 0:
        mu, lv, _ = model.filter_sequence(y[:W], eps=eps[:W])
        q = vjf_amd.Gaussian(mu[-1], lv[-1])
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    mu, lv, loss = model.filter_sequence(y[W:], qs=q, eps=eps[W:])
    enq = time.perf_counter() - t0                             # host time to enqueue the K steps (asynchronous launches)
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_s = ev0.elapsed_time(ev1) * 1e-3
    tt = torch.tensor([wall], device=dev if CTRL_BACKEND == "nccl" else "cpu", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    wall_max = float(tt.item())
    status = model.status()
    elbo = float(-loss[:, 0].mean().item())

    # per-half breakdown on the same stream, outside the timed region
    nb = a.breakdown_steps
    loc = glob = 0.0
    if nb > 0 and world == 1:
        model._ensure_ctx(c["B"])
        L, ctx = model._backend(), model._ctx
        flags = N.FLAG_SGD | N.FLAG_UPDATE
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        loss4 = torch.empty(4, device=dev)
        ms, ls = mu[-1].clone(), lv[-1].clone()
        mo, lo = torch.empty_like(ms), torch.empty_like(ls)
        for t in range(nb):
            tt_ = W + (t % K)
            e[0].record()
            N.check(L.vjf_filter_local(ctx, c["B"], N.ptr(y[tt_]), None, N.ptr(ms), N.ptr(ls), N.ptr(eps[tt_, 0]), N.ptr(eps[tt_, 1]),
                                       N.ptr(mo), N.ptr(lo), flags))
            e[1].record()
            N.check(L.vjf_filter_global(ctx, c["B"], N.ptr(loss4), flags))
            e[2].record()
            torch.cuda.synchronize()
            loc += e[0].elapsed_time(e[1])
            glob += e[1].elapsed_time(e[2])
            ms, ls = mo.clone(), lo.clone()
        loc, glob = loc / nb * 1e3, glob / nb * 1e3      # us

    if rank == 0:
        flops, b_trial, b_shared, serial_flops = algorithmic_work(c)
        units = c["B"] * world * K
        value = units / wall_max
        step_s = dev_s / K
        ach_tf = (flops * c["B"] + serial_flops) / step_s / 1e12
        ach_gbs = (b_trial * c["B"] + b_shared) / step_s / 1e9
        traffic = None
        try:                                                    # HBM bytes per step from the committed PMC run of this build
            traffic = float(json.load(open(TRAFFIC_FILE))["bytes_per_step_corrected"])
        except Exception:
            pass
        kstats = None
        try:                                                    # per-kernel averages of the committed rocprofv3 --stats run of this build
            import csv
            import re
            ks = sorted(__import__("glob").glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_kernel_stats.csv")),
                        key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])     # r01_v10 after r01_v9
            rows = list(csv.DictReader(open(ks[-1])))
            # (one operand kernel / Cholesky / serial kernel per step; the persistent RLS kernels are one launch per sequence)
            steps_in_profile = max(int(r["Calls"]) for r in rows if "prepg" in r["Name"] or "chol" in r["Name"] or "serial" in r["Name"])
            kstats = {"file": "profiles/" + os.path.basename(ks[-1]),
                      "kernels": [{"name": r["Name"].split("(")[0].replace("void ", ""), "avg_us": float(r["AverageNs"]) / 1e3,
                                   "launches_per_step": round(int(r["Calls"]) / steps_in_profile, 2)}
                                  for r in rows if int(r["Calls"]) >= steps_in_profile]}
        except Exception:
            pass
        out = {
            "metric": "trial-timesteps/sec", "value": value, "unit": "trial-timesteps/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": wall_max / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[{dict(B=1, C=2, E=4)[a.config]}]: VJF.filter, {c['B']} trials/GPU, d_z={c['dz']}, d_y={c['dy']}, "
                                    f"RBF({c['n']}), hidden={c['hidden']}, {c['lik'].capitalize()} likelihood, sgd+update, explicit noise"),
                       "global_batch": c["B"] * world, "trials_per_gpu": c["B"], "parallelism": f"trial-shard x{world}"},
            "elbo": elbo, "status_bits": status,
            "roofline": {"bound": "mfma", "achieved": ach_tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tf / PEAK_FP32_TFLOPS, "traffic": traffic,
                         "traffic_note": "HBM bytes per step: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes, "
                                         "summed over the kernels of a step (profiles/pmc_traffic_current.json); algorithmic bytes "
                                         "per step = bytes_per_trial_step x trials",
                         "kernel": "one filter step = vjf_trial_mfma_kernel (one launch: forward part, row signals, in-kernel wait for the "
                                   "RLS update, backward part), vjf_gram_kernel x3 (Phi^T Phi a step ahead, Phi^T dx, gradients), "
                                   "vjf_gram_reduce_kernel x2, vjf_sgd_kernel, vjf_prepg_kernel, gate kernels, and one step's share "
                                   "of the persistent vjf_rls_pair_kernel (Cholesky, y/W and inverse workgroups) that "
                                   "serve the whole sequence; three streams (HIP events around the timed region / steps)",
                         "flops_per_trial_step": flops, "serial_flops_per_step": serial_flops,
                         "rocprof_kernel_averages": kstats,
                         "step_us": step_s * 1e6, "host_enqueue_us_per_step": enq / K * 1e6, "trial_half_us": loc, "serial_half_us": glob,
                         "hbm_achieved_GBs": ach_gbs, "hbm_frac": ach_gbs / PEAK_HBM_GBS,
                         "bytes_per_trial_step": b_trial + b_shared / c["B"]},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(c, y.cpu().numpy(), eps.cpu().numpy())
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
