"""Build-time audit of the one-launch route's hand-off form (DESIGN.md "Hand-offs").

The waits of vjf_mega_kernel / vjf_mega_lite_kernel / vjf_rlsc_loop_kernel do not acquire: they are valid only while EVERY load of a
byte that another workgroup of the same launch stored is an sc1 load (MI355X guide, "sc1 loads in place of the acquire").  A plain
load added later would read stale data only when the line happens to sit in that compute unit's L1 -- no timing test finds that.
This tool compiles the library to ISA with line tables, lists every vector-memory load WITHOUT sc1 in those kernels with the
source line it comes from, and fails on any whose source line is not in the allow-list below (the line's text, so an edited line
has to be certified again).  Every entry says why a plain load is right there:
    const   launch constants: inputs, centroids, tables, learning rates, the state as the launch found it (a kernel boundary lies
            between their last store and this launch)
    own     bytes this same workgroup stored (same compute unit, same L1), or an address written once per launch
    acq     behind an explicit agent-scope acquire
    other   code of the per-step routes that shares a template with the resident kernels and is dead in them at run time

    python tools/audit_plain_loads.py [--asm FILE]      exit code 1 on an uncertified plain load
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("_Z15vjf_mega_kernel", "_Z20vjf_mega_lite_kernel", "_Z20vjf_rlsc_loop_kernel")
# module globals that are set by the host between launches (plain loads of their address-holding words)
CONST_SYMBOLS = ("vjf_host_mirror", "vjf_chaos_range", "vjf_chaos_base")

ALLOW = {
    # ---- vjf_plan.h
    ("vjf_plan.h", "unsigned old = *u, assumed;"): "own: the first guess of a compare-and-swap loop (a stale guess costs one more round)",
    # ---- vjf_mega_kernel.h: trial / Gram roles
    ("vjf_mega_kernel.h", "for (int e = tid; e < npad * dxu; e += NT) { const int c = e / npad, k = e - c * npad; s_cen[e] = k < n ? cen[k * dxu + c] : 0.f; }"): "const: centroids",
    ("vjf_mega_kernel.h", "for (int e = tid; e < npad; e += NT) { float v = 0.f; if (e < n) { const float w = expf(lw[e]); v = -0.5f / (w * w); } s_iw[e] = v; }"): "const: widths",
    ("vjf_mega_kernel.h", "v = (src != nullptr && b < nb) ? src[(size_t)b0 * d + e] : 0.f;"):
        "const / own: y, u, eps are inputs; mu_s, lv_s of step t are rows THIS workgroup stored at step t - 1, at addresses written once per launch",
    ("vjf_mega_kernel.h", "vs[0] = S[P.off[VJF_SLOT_PRIOR_MEAN] + j]; vs[1] = S[P.off[VJF_SLOT_PRIOR_LOGVAR] + j];"): "const: the prior",
    ("vjf_mega_kernel.h", "const float v = Wc[e];"): "const: w_chol of a launch without an RLS update",
    ("vjf_mega_kernel.h", "if (tp) touch = *tp;"): "const: the observations and the noise of the NEXT step (a cache touch in launches without updates, value unused)",
    ("vjf_mega_kernel.h", "else if (j < dz) s_pm[j * LD + b] = *sv;"): "own: the predictive moments this workgroup saved for a replay of its step",
    ("vjf_mega_kernel.h", "const float bf = tl ? ((mg_lds_cf*)bias_l)[f] : ((mg_glb_cf*)bias_g)[f];"): "acq: LDS when the parameters are staged; else the state, behind the gate's acquire (!tl)",
    ("vjf_mega_kernel.h", "if (f < dz) s_mu[f * LD + b] = v; else s_lv[(f - dz) * LD + b] = v + (tl ? ((mg_lds_cf*)bl_l)[f - dz] : ((mg_glb_cf*)bl_g)[f - dz]);"): "acq: as the layer biases",
    ("vjf_mega_kernel.h", "if (f < dy) { const float df = tl ? ((mg_lds_cf*)d_l)[f] : ((mg_glb_cf*)d_g)[f]; s_py[f * LD + (lane & 15)] = acc0[r] + df; s_py[f * LD + 16 + (lane & 15)] = acc1[r] + df; }"): "acq: as the layer biases",
    ("vjf_mega_kernel.h", "const float m = mu_s ? mg_ld(mu_s + (size_t)b * dz + c2) : S[P.off[VJF_SLOT_PRIOR_MEAN] + c2];   // (sc1: no acquire"): "const: the prior (the posterior is an sc1 load)",
    ("vjf_mega_kernel.h", "const float l = mu_s ? mg_ld(lv_s + (size_t)b * dz + c2) : S[P.off[VJF_SLOT_PRIOR_LOGVAR] + c2]; //  behind the waits)"): "const: the prior",
    ("vjf_mega_kernel.h", "v = fmaf(eps_s[(size_t)b * dz + c2], expf(0.5f * l), m);"): "const: the noise input",
    ("vjf_mega_kernel.h", "v = u_e[(size_t)b * du + c2 - dz];"): "const: the control input",
    # ---- moments role, image builders (launches without an RLS update)
    ("vjf_mega_kernel.h", "for (int e = tid0; e < npad * dxu; e += NT) { const int c = e / npad, k = e - c * npad; s_cen[e] = k < n ? cen[k * dxu + c] : 0.f; }"): "const: centroids",
    ("vjf_mega_kernel.h", "for (int e = tid0; e < npad; e += NT) { float v = 0.f; if (e < n) { const float w = expf(lw[e]); v = -0.5f / (w * w); } s_iw[e] = v; }"): "const: widths",
    ("vjf_mega_kernel.h", "const mg_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, float_index * 4, 0, 0);"):
        "own: mg_ld4_plain -- used for L^-1 of a launch WITHOUT an RLS update only (mg_varN): written once at the start of the launch, "
        "read for the first time behind the MG_C_XT count (no earlier copy in this CU's L1: a launch starts with it invalidated)",
    ("vjf_mega_kernel.h", "else { m = S[P.off[VJF_SLOT_PRIOR_MEAN] + c]; l = S[P.off[VJF_SLOT_PRIOR_LOGVAR] + c]; }"): "const: the prior (the posterior is an sc1 load)",
    ("vjf_mega_kernel.h", "if (b < nb) ep = eps_s[(size_t)(b0 + b) * dz + c];"): "const: the noise input",
    ("vjf_mega_kernel.h", "} else if (b < nb) v = u_t[(size_t)(b0 + b) * du + c - dz];"): "const: the control input",
    ("vjf_mega_kernel.h", "const int4 pi = *reinterpret_cast<const int4*>(A.sl_pidx + (size_t)quad * 4);"): "const: slab tables",
    ("vjf_mega_kernel.h", "const int4 ci = *reinterpret_cast<const int4*>(A.sl_cidx + (size_t)quad * 4);"): "const: slab tables",
    ("vjf_mega_kernel.h", "if (pi.x >= 0 && ci.x >= 0) mg_st(img + ci.x, th[pi.x]);"): "const: the parameters of a launch that does not update them",
    ("vjf_mega_kernel.h", "if (pi.y >= 0 && ci.y >= 0) mg_st(img + ci.y, th[pi.y]);"): "const: as above",
    ("vjf_mega_kernel.h", "if (pi.z >= 0 && ci.z >= 0) mg_st(img + ci.z, th[pi.z]);"): "const: as above",
    ("vjf_mega_kernel.h", "if (pi.w >= 0 && ci.w >= 0) mg_st(img + ci.w, th[pi.w]);"): "const: as above",
    # ---- SGD role
    ("vjf_mega_kernel.h", "const float lr_dec = SC[VJF_SC_LR_DEC], lr_rec = SC[VJF_SC_LR_REC];"): "const: set by the host between launches",
    ("vjf_mega_kernel.h", "pi = *reinterpret_cast<const int4*>(A.sl_pidx + (size_t)quad * 4);"): "const: slab tables",
    ("vjf_mega_kernel.h", "ci = *reinterpret_cast<const int4*>(A.sl_cidx + (size_t)quad * 4);"): "const: slab tables",
    ("vjf_mega_kernel.h", "grp = A.sl_grp[quad];"): "const: slab tables",
    ("vjf_mega_kernel.h", "rho -= SC[VJF_SC_LR_LIK] * g;"): "const: learning rate",
    ("vjf_mega_kernel.h", "if (sw == 0 && tid == 0 && mode_rls && SC[VJF_SC_TRI_CLEAN] == 0.f) {"): "own: written by this lane alone, at the end of a launch",
    # ---- vjf_post_kernel.h
    ("vjf_post_kernel.h", "s_x[r * LX + c] = (r < n && c < dz) ? Wold[(size_t)r * dz + c] : 0.f;"): "acq: the failed-factorisation path acquires before it reads the state",
    ("vjf_post_kernel.h", "for (int e = tid; e < n * n; e += VJF_POST_THREADS) vjf_store_wt(Pm + e, fmaf(-G[e], inv_v, Pm[e]));"): "acq: same path",
    ("vjf_post_kernel.h", "vjf_store_wt(A.xt + (size_t)j * n + k, Wc[e]);"): "const: w_chol as the launch found it (the launch's first act)",
    # ---- vjf_chol_kernel.h (vjf_chol_body is shared with the per-step kernels)
    ("vjf_chol_kernel.h", "float sig = S[P.off[VJF_SLOT_TR_LOGVAR]];"): "const: the value the launch found; from the second step on it is replaced by the hand-off word's",
    ("vjf_chol_kernel.h", "return *reinterpret_cast<const float4*>(G + (size_t)gi * n + gj);"): "other: !A.stat_count (per-step kernels: a kernel boundary lies before)",
    ("vjf_chol_kernel.h", "if (sp && !it_src_state) v[q] = *reinterpret_cast<const float4*>(A.pscr + (size_t)idx * 4);"): "own: this workgroup's copy of P of the step before",
    ("vjf_chol_kernel.h", "else v[q] = (gi < n && gj < n) ? *reinterpret_cast<const float4*>(Pm + (size_t)gi * n + gj) : pad4(gi, gj);"):
        "const: the state's P at the first step of a launch, before the operand role may overwrite it (the 'operands loaded' word follows these loads)",
    ("vjf_chol_kernel.h", "s_g[e] = (r < n && j < dz) ? A.gbuf[(size_t)r * dz + j] : 0.f;"): "other: !A.post",
    ("vjf_chol_kernel.h", "if (!sp) for (int e = tid; e < n * n; e += VJF_CHOL_THREADS) Pm[e] = fmaf(-G[e], inv_v, Pm[e]);"): "other: !self_prep",
    ("vjf_chol_kernel.h", "float4 pv = *reinterpret_cast<const float4*>(A.pscr + (size_t)idx * 4);"): "own: this workgroup's copy of P",
    ("vjf_chol_kernel.h", "if (!A.no_triclean && SC[VJF_SC_TRI_CLEAN] == 0.f) {"): "other: the resident loop sets no_triclean",
    ("vjf_chol_kernel.h", "if (SC[VJF_SC_TRI_CLEAN] == 0.f) {"): "other: !A.post",
    ("vjf_chol_kernel.h", "s_g[e] = (r < n && j < dz) ? Wm[(size_t)r * dz + j] : 0.f;"): "other: !A.post (the post mode returns above)",
    ("vjf_chol_kernel.h", "for (int q = 0; q < 8; ++q) gv[q] = (cb + 4 * q < c1) ? *reinterpret_cast<const float4*>(grow + cb + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);"): "other: !A.post",
    ("vjf_chol_kernel.h", "for (int j = 0; j < dz; ++j) f = fmaf(s_g[i * DZP + j], FDX[(size_t)i * dz + j], f);"): "other: !A.post",
    ("vjf_chol_kernel.h", "double t = (double)it_red[P.red_SC + RS_SDX2];"): "other: !A.post",
    ("vjf_chol_kernel.h", "const float acc = fminf(SC[VJF_SC_N_TR], 500.f), tot = acc + Bf;   // running_var, size_cap=500 (model.py:375)"): "other: !A.post",
}


def build_asm(out, extra=()):
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-w", "--cuda-device-only", "-gline-tables-only", "-S", "-o", out,
           os.path.join(ROOT, "vjf_amd", "csrc", "vjf_abi.hip")] + list(extra)
    subprocess.run(cmd, check=True)


def audit(asm_path):
    src = open(asm_path).read().split("\n")
    files = {}
    for l in src:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    text = {}

    def line_text(fn, ln):
        if fn not in text:
            p = os.path.join(ROOT, "vjf_amd", "csrc", fn)
            text[fn] = open(p).read().split("\n") if os.path.exists(p) else []
        t = text[fn]
        return t[ln - 1].strip() if 0 < ln <= len(t) else ""
    report, bad, counts = [], [], {}
    for kern in KERNELS:
        try:
            start = next(i for i, l in enumerate(src) if l.startswith(kern))
        except StopIteration:
            bad.append(f"{kern}: not found in the ISA")
            continue
        cur, n_sc1, n_plain = None, 0, 0
        for i in range(start + 1, len(src)):
            l = src[i]
            m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
            if m:
                cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
                continue
            t = l.strip()
            if t.startswith("s_endpgm") or t.startswith(".Lfunc_end"):
                break
            if not re.match(r"(global_load|buffer_load|flat_load)", t):
                continue
            flat = t.startswith("flat_load")              # (a generic pointer: LDS or memory -- never the sc1 form of the guide)
            if re.search(r"\bsc1\b", t) and not flat:
                n_sc1 += 1
                continue
            n_plain += 1
            fn, ln = cur if cur else ("?", 0)
            if ln == 0:                                   # compiler-generated location: the address words of host-set module globals,
                ctx = " ".join(x.strip() for x in src[max(start, i - 6):i])   # or an instruction the scheduler moved: its statement follows
                if any(sym + "@rel32" in ctx for sym in CONST_SYMBOLS):
                    continue
                for x in src[i + 1:i + 5]:
                    m2 = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", x)
                    if m2 and int(m2.group(2)) > 0:
                        fn, ln = files.get(int(m2.group(1)), "?"), int(m2.group(2))
                        break
                if ln == 0:
                    bad.append(f"{kern}: plain load without a source line: {t}")
                    continue
            tx = line_text(fn, ln)
            why = ALLOW.get((fn, tx))
            if why is None and flat and re.search(r"sc0 sc1", t) and re.search(r"\b(s_ctl|v_ok)\b", tx):
                why = "lds: a hand-off word between the wavefronts of one workgroup (LDS through a generic pointer, an atomic load)"
            if why is None:
                bad.append(f"{kern}: uncertified plain load at {fn}:{ln}: `{tx}`   [{t.split()[0]}]")
            else:
                report.append((kern, fn, ln, why))
        counts[kern] = (n_sc1, n_plain)
    return counts, report, bad


def main(argv):
    asm = None
    if "--asm" in argv:
        asm = argv[argv.index("--asm") + 1]
    tmp = None
    if asm is None:
        tmp = tempfile.mkdtemp(prefix="vjf_audit_")
        asm = os.path.join(tmp, "vjf_abi.s")
        build_asm(asm)
    counts, report, bad = audit(asm)
    for k, (a, b) in counts.items():
        print(f"{k}: {a} sc1 loads, {b} plain loads")
    seen = set()
    for kern, fn, ln, why in report:
        if (fn, ln) not in seen:
            seen.add((fn, ln))
            print(f"   certified {fn}:{ln}  {why}")
    for b in bad:
        print("UNCERTIFIED:", b)
    print("audit:", "FAILED" if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
