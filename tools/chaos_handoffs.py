"""Diagnostic and regression check of the hand-offs between roles: sequences on the routes whose workgroups or kernels run BESIDE each
other (the one-launch route; the three-stream per-step route) under the diagnostic build, in which one workgroup in eight is held for
up to 200 us in front of a wait or a signal (-DVJF_CHAOS, vjf_plan.h), against the one-stream per-step kernels, which have no hand-offs.
An access that is ordered only by the usual timing of the roles -- not by a count -- shows as a wrong result within a few sequences
(round 2: the Gram role's double buffer, which a late RLS role read one step too late; 21 to 40 of 40 sequences deviated at
configs[1]'s shape with the Cholesky or the operand workgroups held, one fresh process in fifty without any hold).

    python -m vjf_amd._build --chaos && VJF_LIB=chaos python tools/chaos_handoffs.py [reps_small] [reps_configB]
    CHAOS_FLAGS=warmup|infer|sgd-only ...   the same for the launches without an RLS update

Holds: every workgroup; workgroups 0..11 one at a time; eight slices of 32 workgroups.  Exit code 1 if any sequence deviates."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VJF_LIB", "chaos")
import torch
import vjf_amd as vjf
from vjf_amd import _native

assert _native.LIB_PATH.endswith("libvjf_hip_chaos.so"), "run with VJF_LIB=chaos"
# CHAOS_FLAGS=warmup|infer|sgd-only: the launches without an RLS update (trial + SGD + moments roles: tags, the ring of loss sums)
KW = {"train": {}, "warmup": dict(warm_up=True), "infer": dict(sgd=False, update=False), "sgd-only": dict(update=False)}[os.environ.get("CHAOS_FLAGS", "train")]


def run(make, y, u, eps, reps, tag, overlap, expect_status=0):
    ref_model = make(); ref_model.set_overlap(False)
    os.environ["VJF_CHAOS_LO"], os.environ["VJF_CHAOS_HI"] = "0", "0"            # (nothing held while the comparison values are formed)
    ref = ref_model.filter_sequence(y, u, None, eps=eps, **KW)
    holds = [(0, 1 << 30)] + [(i, i + 1) for i in range(12)] + [(32 * i, 32 * i + 32) for i in range(8)]
    bad = 0
    for lo, hi in holds:
        os.environ["VJF_CHAOS_LO"], os.environ["VJF_CHAOS_HI"] = str(lo), str(hi)
        for r in range(reps):
            m = make()
            if overlap != 1:
                m.set_overlap(overlap)
            out = m.filter_sequence(y, u, None, eps=eps, **KW)
            st = m.status()
            dl = float(((out[2] - ref[2]).abs().amax(1) / ref[2].abs().amax(1)).max())
            dm = float((out[0] - ref[0]).abs().max())
            if overlap == 3 and r == 0 and lo == 0 and hi > 256: print(tag, "route:", m.route(), flush=True)
            if st != expect_status or not (dl < 2e-5 and dm < 2e-5):
                bad += 1
                print(f"{tag}: workgroups [{lo}, {hi}) held, sequence {r}: route {m.route()} status {st:#x} loss rel. diff {dl:.3e} mean diff {dm:.3e}", flush=True)
    print(f"{tag}: {bad} deviating of {len(holds) * reps} sequences", flush=True)
    return bad


def main():
    reps_small = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    reps_b = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda", 0)
    bad = 0

    def small():
        torch.manual_seed(5)
        return vjf.VJF.make_model(8, 3, 1, 12, [6], likelihood="gaussian", lr=1e-2)
    g = torch.Generator().manual_seed(9)
    T, B = 4, 80
    y, u, eps = torch.randn(T, B, 8, generator=g).to(dev), torch.randn(T, B, 1, generator=g).to(dev), torch.randn(T, 2, B, 3, generator=g).to(dev)
    bad += run(small, y, u, eps, reps_small, "n=12 B=80, one launch", 1)
    bad += run(small, y, u, eps, max(1, reps_small // 4), "n=12 B=80, three streams", 3)

    def config_b():
        torch.manual_seed(3)
        return vjf.VJF.make_model(50, 10, 0, 200, [128], likelihood="gaussian", lr=1e-3)
    g = torch.Generator().manual_seed(4)
    T, B = 12, 4096
    y, eps = torch.randn(T, B, 50, generator=g).to(dev), torch.randn(T, 2, B, 10, generator=g).to(dev)
    bad += run(config_b, y, None, eps, reps_b, "configs[1] shape, one launch", 1)
    bad += run(config_b, y, None, eps, max(1, reps_b // 2), "configs[1] shape, three streams", 3)

    def poisson():
        torch.manual_seed(6)
        return vjf.VJF.make_model(200, 10, 2, 160, [128, 64], likelihood="poisson", lr=1e-3)
    g = torch.Generator().manual_seed(7)
    T, B = 8, 1000
    y = torch.poisson(torch.exp(0.5 * torch.randn(T, B, 200, generator=g) - 0.5), generator=g).to(dev)
    u, eps = torch.randn(T, B, 2, generator=g).to(dev), torch.randn(T, 2, B, 10, generator=g).to(dev)
    bad += run(poisson, y, u, eps, reps_b, "Poisson d_y=200 n=160 B=1000 (ragged tiles), one launch", 1)

    def two_cols():
        torch.manual_seed(8)
        return vjf.VJF.make_model(20, 16, 0, 64, [32, 32], likelihood="gaussian", lr=1e-3)
    g = torch.Generator().manual_seed(10)
    T, B = 6, 300
    y, eps = torch.randn(T, B, 20, generator=g).to(dev), torch.randn(T, 2, B, 16, generator=g).to(dev)
    bad += run(two_cols, y, None, eps, reps_b, "d_z=16 n=64 two layers B=300, one launch", 1)
    g = torch.Generator().manual_seed(11)
    T, B = 5, 12000                                              # (three tiles per trial workgroup)
    y, eps = torch.randn(T, B, 50, generator=g).to(dev), torch.randn(T, 2, B, 10, generator=g).to(dev)
    bad += run(config_b, y, None, eps, max(1, reps_b // 2), "configs[1] model, 12000 trials, one launch", 1)

    # every step with a non-finite reconstruction term (one decoder bias at -3e38 under counts of 2: vjf/model.py:138-149): the
    # backward half, the gradient sums and the SGD pass of every step run twice inside the launch (the REDO counts)
    def replayed():
        m = poisson()
        with torch.no_grad():
            m.decoder.decode.bias[0] = -3e38
        return m
    g = torch.Generator().manual_seed(12)
    T, B = 6, 1000
    y = torch.poisson(torch.exp(0.5 * torch.randn(T, B, 200, generator=g) - 0.5), generator=g)
    y[:, :, 0] = 2.0
    y = y.to(dev)
    u, eps = torch.randn(T, B, 2, generator=g).to(dev), torch.randn(T, 2, B, 10, generator=g).to(dev)
    bad += run(replayed, y, u, eps, reps_b, "Poisson, every step replayed without its reconstruction term, one launch", 1, expect_status=1)
    print("deviating sequences in all:", bad, flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
