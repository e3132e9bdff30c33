// vjf_abi.hip -- extern "C" entry points declared in include/vjf_hip.h.  gfx950 only.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <atomic>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/vjf_hip.h"
#include "vjf_chol_kernel.h"
#include "vjf_gram_kernel.h"
#include "vjf_mega_kernel.h"
#include "vjf_ops_kernels.h"
#include "vjf_plan.h"
#include "vjf_post_kernel.h"
#include "vjf_rlsb_kernels.h"
#include "vjf_serial_kernel.h"
#include "vjf_trial_kernel.h"
#include "vjf_trial_mfma_kernel.h"
#include "vjf_trial_wide.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define VJF_HIP(call)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return fail(-100, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Launch; with `stop` non-null the event rides on the kernel's own completion signal (no marker packet behind it)
#define VJF_LAUNCH(kernel, grid, block, lds, st, stop, ...)                                             \
    do {                                                                                               \
        if (stop) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)(lds), st, nullptr, stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                            \
    } while (0)

// ---- RCCL, resolved at run time from the copy already in the process (torch's) or librccl.so: the library has no link-time
//      dependency on it, and a single-GPU user never touches it
struct VjfNcclId { char internal[128]; };
typedef int (*nccl_get_unique_id_t)(VjfNcclId*);
typedef int (*nccl_comm_init_rank_t)(void**, int, VjfNcclId, int);
typedef int (*nccl_comm_destroy_t)(void*);
typedef int (*nccl_all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_group_t)();
typedef int (*nccl_comm_count_t)(void*, int*);
typedef const char* (*nccl_err_t)(int);
struct VjfNccl {
    nccl_get_unique_id_t get_unique_id; nccl_comm_init_rank_t comm_init_rank; nccl_comm_destroy_t comm_destroy;
    nccl_all_reduce_t all_reduce; nccl_group_t group_start, group_end; nccl_err_t err;
    nccl_comm_count_t comm_count;
    bool ok;
};
constexpr int kNcclFloat = 7, kNcclSum = 0;        // ncclFloat32, ncclSum (rccl.h)
const VjfNccl& nccl() {
    static VjfNccl n = [] {
        VjfNccl v{};
        void* h = RTLD_DEFAULT;
        if (!dlsym(h, "ncclAllReduce")) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return v;
        v.get_unique_id = (nccl_get_unique_id_t)dlsym(h, "ncclGetUniqueId");
        v.comm_init_rank = (nccl_comm_init_rank_t)dlsym(h, "ncclCommInitRank");
        v.comm_destroy = (nccl_comm_destroy_t)dlsym(h, "ncclCommDestroy");
        v.all_reduce = (nccl_all_reduce_t)dlsym(h, "ncclAllReduce");
        v.group_start = (nccl_group_t)dlsym(h, "ncclGroupStart");
        v.group_end = (nccl_group_t)dlsym(h, "ncclGroupEnd");
        v.err = (nccl_err_t)dlsym(h, "ncclGetErrorString");
        v.comm_count = (nccl_comm_count_t)dlsym(h, "ncclCommCount");
        v.ok = v.get_unique_id && v.comm_init_rank && v.comm_destroy && v.all_reduce && v.group_start && v.group_end;
        return v;
    }();
    return n;
}
#define VJF_NCCL(call)                                                                           \
    do {                                                                                         \
        int e_ = (call);                                                                         \
        if (e_ != 0) return fail(-110, "%s failed: %s", #call, nccl().err ? nccl().err(e_) : "rccl error"); \
    } while (0)

constexpr size_t kMaxLds = 160 * 1024;

// ---- what the contexts of a process share, per device
//  * the chain of one-launch grids: such a grid must be resident as a whole (every workgroup wants a whole compute unit's LDS), so
//    two of them -- two models on two streams -- must never be dispatched side by side: each launch waits for the completion event
//    of the previous one, whatever context and stream that came from.  While a device has a single context with the route the
//    stream's own order does this and no event is used; the second context's creation synchronises the device once and switches
//    the chain on for good.
//  * a page of pinned host memory through which a launch that has given up a wait tells the host (vjf_plan.h, vjf_status_or).
struct DevShared {
    std::mutex mu;
    int mega_ctxs = 0;          // live contexts whose plan the one-launch route serves
    bool chained = false;
    hipEvent_t last = nullptr;  // completion of the most recent one-launch grid on this device
    hipStream_t last_stream = nullptr;
    bool last_valid = false;
    unsigned* mirror_h = nullptr;   // the page of pinned host memory (VJF_MIRROR_WORDS words) ...
    unsigned* mirror_d = nullptr;   // ... as the device addresses it
};
constexpr int kMaxDevices = 64;
DevShared g_dev[kMaxDevices];
DevShared* dev_shared(int device) { return device >= 0 && device < kMaxDevices ? &g_dev[device] : nullptr; }

int split_for(int B) {
    int s = B / 256;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

void build_jobs(const VjfPlan& P, std::vector<VjfJob>& jobs) {
    jobs.clear();
    // kind 0: lower tiles of E^T E that touch Phi columns
    const int nt = P.ldE / VJF_TILE;
    for (int ti = 0; ti < nt; ++ti)
        for (int tj = 0; tj <= ti; ++tj) {
            if (tj * VJF_TILE >= P.n) continue;                 // dx x dx tiles are not needed
            if (ti * VJF_TILE >= P.n + P.dz) continue;          // pure padding rows
            VjfJob j{};
            j.kind = 0; j.xc = ti * VJF_TILE; j.yc = tj * VJF_TILE; j.xn = VJF_TILE; j.yn = VJF_TILE;
            j.ti = ti; j.tj = tj; j.dst = 0; j.ld = 0; j.ncol_w = 0; j.dst_b = -1;
            jobs.push_back(j);
        }
    // kind 1: DEL[:, xcol..+M]^T ACT[:, ycol..+K+1]  ->  weight (M,K) + bias (M)
    auto tensor_of = [&](int slot) {
        for (int t = 0; t < P.n_train; ++t) if (P.tr_off[t] == P.off[slot]) return t;
        return -1;
    };
    auto grad = [&](int xcol, int M, int ycol, int K, int slotW, int slotB) {
        const int offW = P.off[slotW] - P.train_off;
        const int offB = slotB >= 0 ? P.off[slotB] - P.train_off : -1;
        const int tW = tensor_of(slotW), tB = slotB >= 0 ? tensor_of(slotB) : -1;
        const int ncols = K + 1;
        for (int ri = 0; ri * VJF_TILE < M; ++ri)
            for (int ci = 0; ci * VJF_TILE < ncols; ++ci) {
                VjfJob j{};
                j.kind = 1;
                j.xc = xcol + ri * VJF_TILE; j.xn = M - ri * VJF_TILE < VJF_TILE ? M - ri * VJF_TILE : VJF_TILE;
                j.yc = ycol + ci * VJF_TILE; j.yn = ncols - ci * VJF_TILE < VJF_TILE ? ncols - ci * VJF_TILE : VJF_TILE;
                j.dst = offW + ri * VJF_TILE * K + ci * VJF_TILE;
                j.ld = K;
                int nw = K - ci * VJF_TILE;
                j.ncol_w = nw < 0 ? 0 : (nw > VJF_TILE ? VJF_TILE : nw);
                j.dst_b = offB >= 0 ? offB + ri * VJF_TILE : -1;
                j.tw = tW; j.tb = tB;
                jobs.push_back(j);
            }
    };
    int prev = P.din;
    for (int l = 0; l < P.L; ++l) {
        grad(P.colD_da[l], P.h[l], P.colA_act[l], prev, VJF_SLOT_REC_W0 + 2 * l, VJF_SLOT_REC_B0 + 2 * l);
        prev = P.h[l];
    }
    grad(P.colD_dmu, P.dz, P.colA_act[P.L], prev, VJF_SLOT_MEAN_W, -1);
    grad(P.colD_dlv, P.dz, P.colA_act[P.L], prev, VJF_SLOT_LV_W, VJF_SLOT_LV_B);
    grad(P.colD_dpy, P.dy, P.colA_xt, P.dz, VJF_SLOT_DEC_W, VJF_SLOT_DEC_B);
}

// ---- the one-launch route (vjf_mega_kernel.h): which plans it serves and how the grid's workgroups are dealt to its roles
constexpr size_t kMegaLds = kMaxLds - 512;             // dynamic LDS of every workgroup of the launch (one workgroup per CU)
constexpr int kMegaMaxTrialWg = 256, kMegaMaxGramWg = 64;
struct MegaShape { int n_rls, n_trial, n_gram, n_prep, n_sgd, ntiles, gram_rows, n_mom; };
constexpr int kMegaRefused = 1 << 20;                  // filter_seq_mega: the grid cannot be resident as a whole (not an error code of the ABI)

bool mega_plan_ok(const VjfPlan& P) {
    const int nbl = (P.n + 31) / 32;
    if (!vjf_chol_lds_ok(P) || P.dz > 16 || nbl > VJF_CHOL_MAXBLK) return false;          // LDS Cholesky loop + y / W and inverse loops
    if ((size_t)(nbl * (nbl + 1) / 2 + nbl) * 1024 * 4 + (size_t)nbl * 32 * 16 * 4 + 768 > kMegaLds) return false;   // vjf_chol_loop<16>
    if (vjf_post_lds_bytes(P) > kMegaLds) return false;
    if ((size_t)vjf_mega_trial_lds(P).total * 4 > kMegaLds) return false;                 // 32 trials' working set
    if (vjf_mega_gram_lds_floats(P) * 4 > kMegaLds || vjf_mega_prep_lds_floats(P) * 4 > kMegaLds) return false;
    if (vjf_mega_mom_lds_floats(P) * 4 > kMegaLds) return false;
    if (P.du > 16) return false;                                                         // (one element of a 32 x du tile per thread)
    if (nbl * (nbl + 1) / 2 > VJF_MG_WAVES * VJF_MG_MAXQ) return false;
    return true;
}

bool mega_shape(const VjfPlan& P, int B, int ncu, uint32_t flags, MegaShape* m) {
    // one workgroup per compute unit: the RLS loops and the operand role have fixed sizes; the trial role gets 128 / 227 of the
    // rest (one 32-trial tile per workgroup at 256 CUs and 4096 trials), then the SGD role (below), the Gram role whatever remains
    const int nbl = (P.n + 31) / 32;
    const bool rls = (flags & (VJF_FLAG_UPDATE | VJF_FLAG_WARM_UP)) == VJF_FLAG_UPDATE;   // (else: no RLS, Gram, operand roles)
    m->n_rls = rls ? 2 + 2 * nbl : 0;
    m->n_prep = rls ? (P.n + 15) / 16 : 0;
    m->ntiles = (B + VJF_MG_TR - 1) / VJF_MG_TR;
    m->n_mom = 0;
    const int rest = ncu - m->n_rls - m->n_prep;
    if (rest < 3) return false;
    if (!rls) {
        // trial + SGD roles only: the SGD role as many workgroups as its fewest rounds of slab loads need (they also build the
        // parameter image at the start of the launch), the trial role the rest
        const int quads = vjf_mega_slab_layout(P).len / 4, gpw = VJF_MG_THREADS / 8;
        int want = (quads + gpw - 1) / gpw;
        if (!(flags & VJF_FLAG_SGD) && want > 16) want = 16;          // (no gradient steps: these only build the parameter image at the start)
        if (want > rest / 4) want = rest / 4;
        if (want < 1) want = 1;
        m->n_sgd = want;
        int cap = rest - m->n_sgd;
        if (cap > kMegaMaxTrialWg) cap = kMegaMaxTrialWg;
        m->n_trial = m->ntiles < cap ? m->ntiles : cap;
        m->n_gram = 0; m->gram_rows = 0;
        // the moments role (vjf_mega_moments): the compute units that are left, when they can keep up -- a tile takes such a workgroup
        // about as long as the rest of the step takes the trial role, so at most two tiles each; else the trial role forms its
        // moments itself
        {
            static const bool off = getenv("VJF_NO_MOMENTS_ROLE") != nullptr;    // (A/B)
            int nm = rest - m->n_sgd - m->n_trial;
            if (nm > m->ntiles) nm = m->ntiles;
            if (!(flags & (VJF_FLAG_SGD | VJF_FLAG_UPDATE))) nm += m->n_sgd;        // (the image builders go on as moments workgroups)
            if (nm > m->ntiles) nm = m->ntiles;
            if (!off && nm >= 1 && 2 * nm >= m->ntiles && m->ntiles <= VJF_MG_TAG_TILES && m->n_trial == m->ntiles) m->n_mom = nm;
            if (m->n_mom > 0 && m->n_sgd > m->n_mom && !(flags & (VJF_FLAG_SGD | VJF_FLAG_UPDATE))) m->n_sgd = m->n_mom;
        }
        return true;
    }
    int cap_t = rest * 128 / 227;
    if (cap_t < 1) cap_t = 1;
    if (cap_t > kMegaMaxTrialWg) cap_t = kMegaMaxTrialWg;
    m->n_trial = m->ntiles < cap_t ? m->ntiles : cap_t;
    if (m->ntiles > cap_t) {
        // More tiles than the trial role's usual share of the chip (B > 4096 at 256 compute units: BASELINE configs[3] on ONE GPU has
        // 1024): the step is then the trial role's tiles in sequence (~41 us each) plus the SGD role's loop, and the other roles have
        // slack -- the SGD role may take two or three rounds of slab loads (+7 us each, once per step), the Gram role several
        // passes of rows (22 us each, a step ahead).  The fewest tiles per trial workgroup that leave those two enough workgroups:
        const int quads = vjf_mega_slab_layout(P).len / 4, gpw = VJF_MG_THREADS / 8, want = (quads + gpw - 1) / gpw;
        int big_sgd = 0, big_gram = 0;
        for (int ntl = (m->ntiles + cap_t - 1) / cap_t; ntl >= 1; --ntl) {
            int nt = (m->ntiles + ntl - 1) / ntl;
            if (nt > kMegaMaxTrialWg || nt > rest - 2) break;
            bool found = false;
            for (int rounds = 1; rounds <= 3 && !found; ++rounds) {
                const int ns = (want + rounds - 1) / rounds;
                int ng = rest - nt - ns;
                if (ng > kMegaMaxGramWg) ng = kMegaMaxGramWg;
                if (ng < nbl * (nbl + 1) / 2) continue;                        // (one Gram workgroup per lower tile at least: the slab sum's shares)
                const int passes = ((B + ng - 1) / ng + VJF_MG_GROWS - 1) / VJF_MG_GROWS;
                const double cycle = 41.0 * ntl + 12.0 + 7.0 * (rounds - 1), gram = 22.0 * passes + 17.0;
                if (gram <= 0.95 * cycle) { m->n_trial = nt; big_sgd = ns; big_gram = ng; found = true; }
            }
            if (!found) break;
        }
        if (big_sgd > 0) {
            m->n_sgd = big_sgd; m->n_gram = big_gram;
            m->gram_rows = ((B + m->n_gram - 1) / m->n_gram + 1) & ~1;
            return true;
        }
    }
    const int left = rest - m->n_trial;                                               // >= 2
    // SGD role: one 8-lane group per quad of the late slab and ROUND of slab loads; its time is the bytes of the slabs over the
    // compute units it has (a unit takes in ~33 GB/s of slabs written on other XCDs), so the fewest rounds win.  The Gram role
    // runs a step ahead with slack: if a second pass of rows per Gram workgroup (fewer of them) saves the SGD role a round, take
    // it; the SGD role then gets just the workgroups that round count needs, the Gram role the rest.
    const int quads = vjf_mega_slab_layout(P).len / 4, gpw = VJF_MG_THREADS / 8;          // lane groups per workgroup
    auto rounds = [&](int nwg) { return (quads + gpw * nwg - 1) / (gpw * nwg); };
    auto clampg = [&](int g) { if (g > left / 2) g = left / 2; return g < 1 ? 1 : g; };
    const int want = (quads + gpw - 1) / gpw;
    const int g1 = clampg((B + VJF_MG_GROWS - 1) / VJF_MG_GROWS), g2 = clampg((B + 2 * VJF_MG_GROWS - 1) / (2 * VJF_MG_GROWS));
    const int n1 = want < left - g1 ? want : left - g1, n2 = want < left - g2 ? want : left - g2;
    const int r = rounds(n2 < 1 ? 1 : n2) < rounds(n1 < 1 ? 1 : n1) ? rounds(n2 < 1 ? 1 : n2) : rounds(n1 < 1 ? 1 : n1);
    m->n_sgd = (quads + gpw * r - 1) / (gpw * r);                                         // the fewest workgroups with that many rounds
    if (m->n_sgd > left - g2) m->n_sgd = left - g2;
    if (m->n_sgd < 1) m->n_sgd = 1;
    m->n_gram = (B + 63) / 64;
    // (at least one Gram workgroup per lower tile of Phi^T Phi, rows or not: the slab sum deals its quads over the role's
    //  workgroups, two per thread and ROUND TRIP -- one workgroup alone took ten of them for the ten tiles of RBF(100), and at
    //  one trial that loop, 35 us, was the step; compute units are idle at such batch sizes)
    { const int ntri = nbl * (nbl + 1) / 2; if (m->n_gram < ntri) m->n_gram = ntri; }
    if (m->n_gram > left - m->n_sgd) m->n_gram = left - m->n_sgd;
    if (m->n_gram > kMegaMaxGramWg) m->n_gram = kMegaMaxGramWg;
    if (m->n_gram < 1) m->n_gram = 1;
    m->gram_rows = ((B + m->n_gram - 1) / m->n_gram + 1) & ~1;
    return true;
}

struct Carve {
    size_t pscr; size_t mg_mom, mg_xt, mg_early, mg_late, mg_gslab, mg_cnt, mg_stamps, mg_pidx, mg_cidx, mg_grp, mg_img, mg_pmsave; size_t E, E2, ACT, DEL, partial, partial2, slabs, red, red2, red3, tbig, wide, work, jobs, aux, post, lscr, flags, resid, total;
};

Carve carve_ws(const VjfPlan& P, int max_batch, int njobs) {
    Carve c{};
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = (o + bytes + 255) / 256 * 256; return at; };
    c.E = take((size_t)max_batch * P.ldE * 4);
    c.E2 = take((size_t)max_batch * P.ldE * 4);            // odd steps' E rows in the multi-stream sequence
    c.ACT = take((size_t)max_batch * P.ldA * 4);
    c.DEL = take((size_t)max_batch * P.ldD * 4);
    c.partial = take(((size_t)max_batch / 4 + 2) * RS_N * 4);
    c.partial2 = take(((size_t)max_batch / 4 + 2) * RS_N * 4);
    c.slabs = take((size_t)njobs * split_for(max_batch) * 1024 * 4);
    c.red = take((size_t)P.red_len * 4);
    c.red2 = take((size_t)P.red_len * 4);                  // RLS statistics of even / odd steps in the multi-stream sequence
    c.red3 = take((size_t)P.red_len * 4);
    c.tbig = take(P.n > 32 * VJF_CHOL_MAXBLK ? (size_t)((P.n + 31) / 32) * 1024 * 4 * 3 : 16);   // multi-launch RLS: the diagonal blocks of L, two sets of column sums
    // GEMM-per-layer trial path (working set beyond LDS): [xs|u], pt.mean, pt.logvar, decoder output, Phi w_chol per trial
    c.wide = take(vjf_trial_mfma_lds_floats(P) * 4 > kMaxLds - 1024 ? (size_t)max_batch * (P.dxu + P.dz + 1 + P.dy + P.n) * 4 + 1024 : 16);
    c.work = take(vjf_serial_work_floats(P) * 4 + 10 * 256);  // + 10 x 32 u64 diagnostic stamps (a ring over the steps of a sequence)
    c.post = take((size_t)((P.n + 31) / 32) * 1024 * 4 + VJF_RESID_BLOCKS * 8 + 64);   // Dinv blocks | resid partials | ok flag
    c.flags = take(256);                                   // column flags of the Cholesky -> post hand-off (a block of their own)
    c.lscr = take((size_t)P.n * P.n * 4);                  // L, column by column, from the Cholesky kernel to the post kernel
    c.pscr = take((size_t)(VJF_CHOL_MAXBLK * (VJF_CHOL_MAXBLK + 1) / 2) * 1024 * 4);   // lower blocks of P, from one Cholesky kernel to the next
    if (mega_plan_ok(P)) {                                 // slabs of the one-launch route, sized for the largest role counts
        const int nbl = (P.n + 31) / 32;
        const size_t slab_len = (size_t)vjf_mega_slab_layout(P).len;
        c.mg_early = take((size_t)2 * kMegaMaxTrialWg * ((size_t)((P.n + 3) & ~3) * 16 + 8) * 4);   // (two sets: even / odd steps)
        c.mg_late = take((size_t)kMegaMaxTrialWg * (slab_len + 8 * VJF_MG_RING) * 4);
        c.mg_gslab = take((size_t)kMegaMaxGramWg * (nbl * (nbl + 1) / 2) * 1024 * 4);
        c.mg_cnt = take((size_t)2 * MG_C_WORDS * 4);           // two counter blocks: a launch runs on one and zeroes the other for the next
        c.mg_stamps = take((32 * 32 + kMegaMaxTrialWg * 8) * 8);   // ring of role stamps | 8 words per trial workgroup (last step)
        c.mg_pidx = take(slab_len * 4); c.mg_cidx = take(slab_len * 4); c.mg_grp = take(slab_len);
        c.mg_img = take((size_t)vjf_mega_trial_lds(P, (int)(kMegaLds / 4) - 8).th_len * 4 + 64);   // the parameters in the trial role's LDS layout
        c.mg_pmsave = take((size_t)max_batch * (P.dz + 1) * 4);
        c.mg_xt = take((size_t)P.n * P.n * 4);               // row-major L^-1 (= w_chol^T) for the trial role's 16-byte operand loads
        {   // moments role -> trial role: [tile][step parity][(2 dz + 1) x 32]
            int nt = (max_batch + VJF_MG_TR - 1) / VJF_MG_TR;
            if (nt > VJF_MG_TAG_TILES) nt = VJF_MG_TAG_TILES;
            c.mg_mom = take((size_t)nt * 2 * (2 * P.dz + 1) * VJF_MG_TR * 4);
        }
    }
    c.jobs = take((size_t)njobs * sizeof(VjfJob));
    c.aux = take((size_t)P.aux_len * 4);
    // multi-launch RLS on a stream of its own (filter_seq_two): Phi W of the state-noise update, beside the trial chain's DEL rows
    c.resid = take(P.n > 32 * VJF_CHOL_MAXBLK ? (size_t)max_batch * P.dz * 4 : 16);
    c.total = o;
    return c;
}

template <class K>
void allow_lds(K kernel, size_t bytes) {
    // raise the dynamic-LDS cap (kernels here use up to ~150 KiB of the CU's 160 KiB); a refusal is
    // not fatal by itself -- the launch reports it -- so clear the sticky error state
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    (void)hipGetLastError();
}

}  // namespace

struct vjf_ctx {
    vjf_config cfg;
    VjfPlan plan;
    float* state;
    char* ws;
    int64_t ws_bytes;
    hipStream_t stream;
    Carve cv;
    int njobs;
    size_t lds_k2;
    bool post_kernels;     // RLS tail (inverse, solve, residual) on many CUs beside / after the Cholesky kernel
    size_t lds_post;
    bool mfma_trial;       // 16 trials' working set fits LDS: matrix-core trial kernel of the per-step routes
    size_t lds_k1m;
    bool stamps;           // diagnostic: s_memrealtime phase stamps
    bool stamps_keep_overlap;
    bool fast_chol;        // n_rbf <= 224: prep kernel + LDS-resident MFMA Cholesky; else the generic serial kernel / multi-launch RLS
    size_t lds_chol;
    int n_ejobs;           // jobs [0, n_ejobs) are the E^T E tiles, the rest gradient tiles
    bool overlap;          // 1: single rank -> the one-launch route, ranks -> the three-stream per-step route; 0: one-stream order
    bool handoff_acquire;  // VJF_HANDOFF_ACQUIRE=1: the one-launch route's waits acquire at agent scope beside the sc1 loads (default: sc1 loads alone)
    bool force_streams;    // vjf_set_overlap(ctx, 3): the three-stream per-step route on a single rank too (A/B measurements)
    bool mega_ok;          // the plan fits the one-launch route (vjf_mega_kernel.h)
    int ncu;               // compute units of the device: the one-launch grid has one workgroup per CU
    int mega_wg_per_cu;    // workgroups of vjf_mega_kernel a compute unit can hold (occupancy query): the residency check of the route
    int lite_wg_per_cu;    // the same for vjf_mega_lite_kernel (the launches without an RLS update)
    unsigned mega_launches; // launches of vjf_mega_kernel so far: launch k counts in counter block k & 1
    hipStream_t stream2, stream3;
    hipEvent_t ev_s, ev_c;
    hipEvent_t ev_f[2], ev_r[2], ev_b[2], ev_g[2];   // (ev_g: the RLS statistics) two-stream route of the multi-launch RLS plans: forward half / RLS update / backward half of even, odd steps
    unsigned epoch;        // launches of the Cholesky / post pair so far (the hand-off flags carry it)
    unsigned k1_count;     // workgroups of the matrix-core trial kernel (whole step or backward half) launched so far
    unsigned post_count;   // workgroups of the post kernel launched so far
    unsigned fwd_count;    // workgroups of forward halves launched with a completion count
    unsigned stats_count;  // steps whose RLS statistics the three-stream route has launched (host mirror of flag word kStatsWord)
    unsigned start_count;  // host mirror of the post kernel's "workgroups started" count
    bool mega_counted;     // counted in its device's DevShared::mega_ctxs
    bool on_mega;          // the context's last sequence ran on the one-launch route (a timed-out wait then makes it leave the route)
    void* comm_a; void* comm_b;   // RCCL communicators of the two chains of the three-stream route (null: single rank)
    int world;
    int collectives;       // sums over ranks per step on the in-library route: 2 (default) [grad | loss sums] and [G | Phi^T dx | sums], one on each
                           // chain of the three-stream schedule; 1 ONE all-reduce of the whole reduce buffer (SURVEY 8e's layout) between the
                           // trial-parallel and the serial half of a step, on one stream (vjf_set_collectives)
    int fake_world;        // test hook (VJF_DEBUG_FAKE_WORLD=k at vjf_comm_init, one-rank communicators): behave as rank 0 of k ranks that
                           // all hold the same trials -- every all-reduced buffer is multiplied by k and B_total = k B
};

extern "C" {

int vjf_abi_version(void) { return VJF_ABI_VERSION; }
const char* vjf_last_error(void) { return g_err.c_str(); }

int vjf_state_size(const vjf_config* cfg, int64_t* n_floats) {
    VjfPlan P;
    int rc = vjf_make_plan(cfg, &P);
    if (rc) return fail(rc, "vjf_state_size: invalid config (%d)", rc);
    if (!n_floats) return fail(-1, "vjf_state_size: null output");
    *n_floats = P.n_state;
    return 0;
}

int vjf_state_layout(const vjf_config* cfg, int64_t* offsets, int64_t* sizes) {
    VjfPlan P;
    int rc = vjf_make_plan(cfg, &P);
    if (rc) return fail(rc, "vjf_state_layout: invalid config (%d)", rc);
    if (!offsets || !sizes) return fail(-1, "vjf_state_layout: null output");
    for (int s = 0; s < VJF_N_SLOTS; ++s) { offsets[s] = P.off[s]; sizes[s] = P.size[s]; }
    return 0;
}

int vjf_workspace_size(const vjf_config* cfg, int64_t* bytes) {
    VjfPlan P;
    int rc = vjf_make_plan(cfg, &P);
    if (rc) return fail(rc, "vjf_workspace_size: invalid config (%d)", rc);
    if (!bytes) return fail(-1, "vjf_workspace_size: null output");
    if (cfg->max_batch < 1) return fail(-7, "vjf_workspace_size: max_batch must be >= 1");
    std::vector<VjfJob> jobs;
    build_jobs(P, jobs);
    *bytes = (int64_t)carve_ws(P, cfg->max_batch, (int)jobs.size()).total;
    return 0;
}

int vjf_ctx_create(const vjf_config* cfg, float* state, void* workspace, int64_t workspace_bytes, void* stream,
                   vjf_ctx** out) {
    if (!cfg || !state || !workspace || !out) return fail(-1, "vjf_ctx_create: null argument");
    VjfPlan P;
    int rc = vjf_make_plan(cfg, &P);
    if (rc) return fail(rc, "vjf_ctx_create: invalid config (%d)", rc);
    if (cfg->max_batch < 1) return fail(-7, "vjf_ctx_create: max_batch must be >= 1");
    std::vector<VjfJob> jobs;
    build_jobs(P, jobs);
    if ((int)jobs.size() > VJF_MAX_JOBS) return fail(-8, "vjf_ctx_create: %zu Gram tiles exceed the limit", jobs.size());
    Carve cv = carve_ws(P, cfg->max_batch, (int)jobs.size());
    if ((int64_t)cv.total > workspace_bytes)
        return fail(-9, "vjf_ctx_create: workspace too small (%lld < %zu bytes)", (long long)workspace_bytes, cv.total);
    const size_t lds_k2 = vjf_serial_lds_floats(P) * 4;
    const bool fast_chol = vjf_chol_lds_ok(P);
    if (!fast_chol && lds_k2 > kMaxLds - 1024 && P.n <= 32 * VJF_CHOL_MAXBLK)
        return fail(-11, "vjf_ctx_create: n_rbf=%d too large for the single-workgroup RLS kernel", P.n);
    VJF_HIP(hipSetDevice(cfg->device));
    vjf_ctx* c = new (std::nothrow) vjf_ctx();
    if (!c) return fail(-12, "vjf_ctx_create: out of host memory");
    c->cfg = *cfg; c->plan = P; c->state = state; c->ws = (char*)workspace; c->ws_bytes = workspace_bytes;
    c->stream = (hipStream_t)stream; c->cv = cv; c->njobs = (int)jobs.size();
    c->lds_k2 = lds_k2;
    c->fast_chol = fast_chol; c->lds_chol = vjf_chol_lds_bytes(P); c->stamps = false; c->stamps_keep_overlap = false;
    c->lds_post = vjf_post_lds_bytes(P);
    c->post_kernels = fast_chol && P.dz <= 16 && c->lds_post <= kMaxLds - 1024;
    // the single-workgroup chain kernels ask for the whole LDS of their compute unit: nothing else (every other kernel of
    // a step uses some LDS) is then placed beside them to share their SIMDs' issue slots and matrix cores
    if (fast_chol) c->lds_chol = kMaxLds - 256;
    if (c->post_kernels) c->lds_post = kMaxLds - 256;
    c->lds_k1m = vjf_trial_mfma_lds_floats(P) * 4;
    c->mfma_trial = c->lds_k1m <= kMaxLds - 1024;
    c->n_ejobs = 0;
    for (const VjfJob& j : jobs) c->n_ejobs += j.kind == 0;
    c->overlap = true; c->force_streams = false;
    // the one-launch route's hand-offs: the producer stores write-through (sc1), every storing wavefront drains vmcnt, the workgroup
    // barrier, ONE lane's agent-scope add; the consumer polls that count with one lane (an sc1 load), the workgroup barrier, and then
    // EVERY load of a handed-off byte is an sc1 load (4- or 16-byte, global_ / buffer_, never flat_), one workgroup per compute unit,
    // hipMalloc memory: the first row of the MI355X guide's table "hand-offs measured with sc1 loads in place of the acquire", in every
    // cell.  That form is the default.  VJF_HANDOFF_ACQUIRE=1 adds an agent-scope acquire (L1 invalidate, ~1.7 us the polling
    // wavefront waits for) behind every wait -- the form with an architectural guarantee, ~2.5 % slower at config B; the perturbed-
    // timing test (tests/test_gpu_handoffs.py) runs in both.
    { const char* ha = getenv("VJF_HANDOFF_ACQUIRE"); c->handoff_acquire = ha && atoi(ha) != 0; }
    c->mega_ok = mega_plan_ok(P);
    {
        int v = 0;
        VJF_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, cfg->device));
        c->ncu = v;
        c->mega_wg_per_cu = 0; c->lite_wg_per_cu = 0;
    }
    c->start_count = 0; c->mega_launches = 0; c->on_mega = false; c->mega_counted = false;
    c->stream2 = c->stream3 = nullptr; c->ev_s = c->ev_c = nullptr;
    for (int i = 0; i < 2; ++i) c->ev_f[i] = c->ev_r[i] = c->ev_b[i] = c->ev_g[i] = nullptr;
    c->epoch = 0; c->k1_count = 0; c->post_count = 0; c->fwd_count = 0; c->stats_count = 0;
    c->comm_a = c->comm_b = nullptr; c->world = 1; c->fake_world = 1; c->collectives = 2;
    if (const char* ce = getenv("VJF_COLLECTIVES")) { if (atoi(ce) == 1) c->collectives = 1; }
    hipError_t e = hipMemcpyAsync(c->ws + cv.jobs, jobs.data(), jobs.size() * sizeof(VjfJob), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->ws + cv.red, 0, (size_t)P.red_len * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->ws + cv.red2, 0, (size_t)P.red_len * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->ws + cv.red3, 0, (size_t)P.red_len * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->ws + cv.flags, 0, 256, c->stream);
    std::vector<int> pidx, cidx, grpv;
    if (c->mega_ok) {
        // the late slab's tables (vjf_mega_slab_layout): per float the parameter it is the gradient of and that parameter's copy
        // for the trial role -- its place in the image of the LDS region (vjf_mega_trial_lds) when the parameters fit there, else
        // in the transposed aux copies; per quad the optimizer group
        const VjfMegaTrialLds Lo = vjf_mega_trial_lds(P, (int)(kMegaLds / 4) - 8);
        const VjfMegaSlab SL = vjf_mega_slab_layout(P);
        pidx.assign((size_t)SL.len, -1); cidx.assign((size_t)SL.len, -1); grpv.assign((size_t)SL.len / 4, 0);
        std::vector<int> img_of((size_t)P.train_len, -1), aux_of((size_t)P.train_len, -1), dec_of((size_t)P.train_len, 0);
        auto place = [&](int slot, int rows, int cols, int at, int ld) {
            const int o = P.off[slot] - P.train_off;
            for (int r = 0; r < rows; ++r)
                for (int cc = 0; cc < cols; ++cc) img_of[(size_t)o + (size_t)r * cols + cc] = at - Lo.th0 + r * ld + cc;
        };
        int prev = P.din;
        for (int l = 0; l < P.L; ++l) {
            place(VJF_SLOT_REC_W0 + 2 * l, P.h[l], prev, Lo.th_w[l], Lo.th_ldw[l]);
            place(VJF_SLOT_REC_B0 + 2 * l, 1, P.h[l], Lo.th_b[l], P.h[l]);
            prev = P.h[l];
        }
        place(VJF_SLOT_MEAN_W, P.dz, prev, Lo.th_head, Lo.th_ldh);
        place(VJF_SLOT_LV_W, P.dz, prev, Lo.th_head + P.dz * Lo.th_ldh, Lo.th_ldh);
        place(VJF_SLOT_LV_B, 1, P.dz, Lo.th_bl, P.dz);
        place(VJF_SLOT_DEC_W, P.dy, P.dz, Lo.th_dec, Lo.th_ldd);
        place(VJF_SLOT_DEC_B, 1, P.dy, Lo.th_bd, P.dy);
        for (int t = 0; t < P.n_train; ++t) {
            const int o = P.tr_off[t] - P.train_off, rows = P.tr_rows[t], cols = P.tr_cols[t];
            for (int el = 0; el < rows * cols; ++el) {
                const int r = el / cols, cc = el - r * cols;
                dec_of[(size_t)o + el] = P.tr_dec[t] ? 1 : 0;
                aux_of[(size_t)o + el] = P.tr_aux[t] >= 0 ? P.tr_aux[t] + cc * P.tr_auxld[t] + P.tr_auxcol[t] + r : -1;
            }
        }
        // block b of the slab: weight (M, Kin) [+ bias (M)] stored as rows j = 0 .. Kin - 1 [, Kin] of ldm columns m
        auto block = [&](int b, int slotW, int slotB, int M, int Kin) {
            const int ow = P.off[slotW] - P.train_off, ob = slotB >= 0 ? P.off[slotB] - P.train_off : -1;
            for (int j = 0; j < SL.rows[b]; ++j)
                for (int m2 = 0; m2 < SL.ldm[b]; ++m2) {
                    const size_t at = (size_t)SL.off[b] + (size_t)j * SL.ldm[b] + m2;
                    int pe = -1;
                    if (m2 < M) pe = j < Kin ? ow + m2 * Kin + j : (ob >= 0 ? ob + m2 : -1);
                    pidx[at] = pe;
                    if (pe >= 0) {
                        cidx[at] = Lo.theta ? img_of[(size_t)pe] : aux_of[(size_t)pe];
                        grpv[at / 4] = dec_of[(size_t)pe];
                    }
                }
        };
        const int hL = P.h[P.L - 1];
        block(0, VJF_SLOT_DEC_W, VJF_SLOT_DEC_B, P.dy, P.dz);
        block(1, VJF_SLOT_MEAN_W, -1, P.dz, hL);
        block(2, VJF_SLOT_LV_W, VJF_SLOT_LV_B, P.dz, hL);
        for (int l = P.L - 1; l >= 0; --l) block(3 + (P.L - 1 - l), VJF_SLOT_REC_W0 + 2 * l, VJF_SLOT_REC_B0 + 2 * l, P.h[l], l > 0 ? P.h[l - 1] : P.din);
        if (e == hipSuccess) e = hipMemcpyAsync(c->ws + cv.mg_pidx, pidx.data(), pidx.size() * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(c->ws + cv.mg_cidx, cidx.data(), cidx.size() * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(c->ws + cv.mg_grp, grpv.data(), grpv.size() * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(c->ws + cv.mg_img, 0, (size_t)Lo.th_len * 4, c->stream);   // (its padding stays 0)
        if (e == hipSuccess) e = hipMemsetAsync(c->ws + cv.mg_cnt, 0, (size_t)2 * MG_C_WORDS * 4, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // `jobs`, `meta` (host) must outlive the copies
    if (e != hipSuccess) { delete c; return fail(-100, "vjf_ctx_create: %s", hipGetErrorString(e)); }
    allow_lds(vjf_serial_kernel, c->lds_k2);
    allow_lds(vjf_rls_post_kernel, c->lds_post);
    allow_lds(vjf_prepg_kernel, vjf_prepg_lds_bytes(P));
    if (c->mfma_trial) allow_lds(vjf_trial_mfma_kernel, c->lds_k1m);
    allow_lds(vjf_chol_lds_kernel<4>, c->lds_chol); allow_lds(vjf_chol_lds_kernel<8>, c->lds_chol);
    allow_lds(vjf_chol_lds_kernel<12>, c->lds_chol); allow_lds(vjf_chol_lds_kernel<16>, c->lds_chol);
    allow_lds(vjf_chol_lds_kernel<32>, c->lds_chol);
    allow_lds(vjf_rls_pair_kernel<4>, c->lds_chol); allow_lds(vjf_rls_pair_kernel<8>, c->lds_chol);
    allow_lds(vjf_rls_pair_kernel<12>, c->lds_chol); allow_lds(vjf_rls_pair_kernel<16>, c->lds_chol);
    if (c->mega_ok) {
        allow_lds(vjf_mega_kernel, kMegaLds);
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, vjf_mega_kernel, VJF_MG_THREADS, kMegaLds) != hipSuccess) { (void)hipGetLastError(); nb = 0; }
        c->mega_wg_per_cu = nb;
        if (nb < 1) c->mega_ok = false;
        allow_lds(vjf_mega_lite_kernel, kMegaLds);
        nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, vjf_mega_lite_kernel, VJF_MG_THREADS, kMegaLds) != hipSuccess) { (void)hipGetLastError(); nb = 0; }
        c->lite_wg_per_cu = nb;
    }
    if (DevShared* d = dev_shared(cfg->device)) {
        std::lock_guard<std::mutex> lk(d->mu);
        if (!d->mirror_h) {
            unsigned* h = nullptr; unsigned* dp = nullptr;
            if (hipHostMalloc((void**)&h, VJF_MIRROR_WORDS * sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
                hipHostGetDevicePointer((void**)&dp, h, 0) == hipSuccess) {
                memset(h, 0, VJF_MIRROR_WORDS * sizeof(unsigned));
                d->mirror_h = h; d->mirror_d = dp;
            }
            (void)hipGetLastError();                                // (without the page: no early notice, the status word still tells)
        }
        if (c->mega_ok) c->mega_counted = true;
        if (c->mega_ok && ++d->mega_ctxs == 2 && !d->chained) {
            (void)hipDeviceSynchronize();                           // (the first context's launches so far carry no event)
            if (!d->last) (void)hipEventCreateWithFlags(&d->last, hipEventDisableTiming);
            d->chained = d->last != nullptr;
        }
    }
    *out = c;
    return 0;
}

int vjf_ctx_destroy(vjf_ctx* ctx) {
    if (ctx)
        if (DevShared* d = dev_shared(ctx->cfg.device)) {
            std::lock_guard<std::mutex> lk(d->mu);
            if (ctx->mega_counted && d->mega_ctxs > 0) --d->mega_ctxs;
        }
    if (ctx && ctx->comm_a) {
        if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamSynchronize(ctx->stream3); }
        (void)hipStreamSynchronize(ctx->stream);
        (void)nccl().comm_destroy(ctx->comm_a); (void)nccl().comm_destroy(ctx->comm_b);
        ctx->comm_a = ctx->comm_b = nullptr;
    }
    if (ctx && ctx->stream2) {
        (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamSynchronize(ctx->stream3);
        (void)hipEventDestroy(ctx->ev_s); (void)hipEventDestroy(ctx->ev_c);
        for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(ctx->ev_f[i]); (void)hipEventDestroy(ctx->ev_r[i]); (void)hipEventDestroy(ctx->ev_b[i]); (void)hipEventDestroy(ctx->ev_g[i]); }
        (void)hipStreamDestroy(ctx->stream2); (void)hipStreamDestroy(ctx->stream3);
    }
    delete ctx;
    return 0;
}

int vjf_set_overlap(vjf_ctx* ctx, int enable) {
    if (!ctx) return fail(-1, "vjf_set_overlap: null context");
    ctx->overlap = enable != 0;
    ctx->force_streams = enable == 3;
    return ctx->overlap ? (ctx->force_streams ? 3 : 1) : 0;
}

int vjf_comm_unique_id(void* ids256) {
    if (!ids256) return fail(-1, "vjf_comm_unique_id: null output");
    if (!nccl().ok) return fail(-111, "vjf_comm_unique_id: RCCL is not available in this process");
    VJF_NCCL(nccl().get_unique_id((VjfNcclId*)ids256));
    VJF_NCCL(nccl().get_unique_id((VjfNcclId*)ids256 + 1));
    return 0;
}

int vjf_comm_init(vjf_ctx* ctx, const void* ids256, int32_t rank, int32_t world) {
    if (!ctx || !ids256) return fail(-1, "vjf_comm_init: null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(-20, "vjf_comm_init: rank %d of %d", rank, world);
    if (!nccl().ok) return fail(-111, "vjf_comm_init: RCCL is not available in this process");
    if (ctx->comm_a) return fail(-112, "vjf_comm_init: the context already has communicators");
    if (ctx->collectives != 1 && !(ctx->fast_chol && ctx->post_kernels && ctx->mfma_trial))
        return fail(-113, "vjf_comm_init: this plan has no multi-stream route; keep the all-reduce on the caller's side (vjf_filter_local / vjf_filter_global)");
    VJF_HIP(hipSetDevice(ctx->cfg.device));
    VjfNcclId ids[2];
    memcpy(ids, ids256, sizeof ids);
    void* ca = nullptr; void* cb = nullptr;
    VJF_NCCL(nccl().comm_init_rank(&ca, world, ids[0], rank));
    int e = nccl().comm_init_rank(&cb, world, ids[1], rank);
    if (e != 0) { (void)nccl().comm_destroy(ca); return fail(-110, "ncclCommInitRank failed: %s", nccl().err ? nccl().err(e) : "rccl error"); }
    ctx->comm_a = ca; ctx->comm_b = cb; ctx->world = world;
    ctx->fake_world = 1;
    if (const char* fw = getenv("VJF_DEBUG_FAKE_WORLD")) { const int k = atoi(fw); if (world == 1 && k > 1 && k <= 64) ctx->fake_world = k; }
    return 0;
}

int vjf_comm_ranks(vjf_ctx* ctx, int32_t* ranks2) {
    if (!ctx || !ranks2) return fail(-1, "vjf_comm_ranks: null argument");
    ranks2[0] = ranks2[1] = 0;
    if (!ctx->comm_a) return 0;
    if (!nccl().comm_count) return fail(-111, "vjf_comm_ranks: ncclCommCount is not available in this process");
    int na = 0, nb = 0;
    VJF_NCCL(nccl().comm_count(ctx->comm_a, &na));
    VJF_NCCL(nccl().comm_count(ctx->comm_b, &nb));
    ranks2[0] = na; ranks2[1] = nb;
    return 0;
}

int vjf_set_collectives(vjf_ctx* ctx, int32_t per_step) {
    if (!ctx) return fail(-1, "vjf_set_collectives: null context");
    if (per_step != 1 && per_step != 2) return fail(-20, "vjf_set_collectives: %d (1 or 2)", per_step);
    ctx->collectives = per_step;
    return 0;
}

int vjf_set_stream(vjf_ctx* ctx, void* stream) {
    if (!ctx) return fail(-1, "vjf_set_stream: null context");
    ctx->stream = (hipStream_t)stream;
    return 0;
}

namespace {
// A wait of an earlier call of this context gave up (the device said so through the host page, vjf_plan.h): this call must not
// build on its results.  One plain load of host memory in the usual case.
int refuse_if_poisoned(vjf_ctx* c, const char* who) {
    DevShared* d = dev_shared(c->cfg.device);
    float* p = c->state + c->plan.off[VJF_SLOT_SCALARS] + VJF_SC_STATUS;
    if (!d || !d->mirror_h || __atomic_load_n(d->mirror_h + VJF_MIRROR_SLOT(p), __ATOMIC_RELAXED) == 0u) return 0;
    float v = 0.f;                                                 // (this context's own word: the slot may be another's)
    VJF_HIP(hipMemcpyAsync(&v, p, 4, hipMemcpyDeviceToHost, c->stream));
    VJF_HIP(hipStreamSynchronize(c->stream));
    const unsigned st = (unsigned)v;
    if (!(st & VJF_STATUS_WAIT_MASK)) return 0;
    if (c->on_mega) c->mega_ok = false;                            // (the one-launch route did not hold on this device: per-step kernels from here)
    return fail(-30, "%s: a device-side wait of an earlier call of this context timed out (status 0x%x%s); its outputs and what it left "
                     "of the state are not to be used -- read the status (vjf_get_status), restore the state, run again", who, st,
                (st & VJF_STATUS_NOT_RESIDENT) ? ": the grid was not resident as a whole, the state is untouched" : "");
}
}  // namespace

int vjf_get_status(vjf_ctx* ctx, uint32_t* status) {
    if (!ctx || !status) return fail(-1, "vjf_get_status: null argument");
    float* p = ctx->state + ctx->plan.off[VJF_SLOT_SCALARS] + VJF_SC_STATUS;
    float v = 0.f;
    VJF_HIP(hipMemcpyAsync(&v, p, 4, hipMemcpyDeviceToHost, ctx->stream));
    VJF_HIP(hipMemsetAsync(p, 0, 4, ctx->stream));
    VJF_HIP(hipStreamSynchronize(ctx->stream));
    *status = (uint32_t)v;
    if (*status & VJF_STATUS_WAIT_MASK) {
        if (ctx->on_mega) ctx->mega_ok = false;                    // (see refuse_if_poisoned)
        if (DevShared* d = dev_shared(ctx->cfg.device))
            if (d->mirror_h) __atomic_store_n(d->mirror_h + VJF_MIRROR_SLOT(p), 0u, __ATOMIC_RELAXED);   // acknowledged
    }
    return 0;
}

int vjf_debug_stamps(vjf_ctx* ctx, int enable, uint64_t* out32) {
    if (!ctx) return fail(-1, "vjf_debug_stamps: null context");
    int ring = 0;
    if (enable >= 128) {                                     // 128 + k: the one-launch route's role stamps of step k % 32 (32 words);
        if (out32 && ctx->mega_ok) {                         // 256 + j: the last step's 8 words of trial workgroups 4 j .. 4 j + 3
            const size_t at = enable >= 256 ? (size_t)32 * 256 + (size_t)((enable - 256) % (kMegaMaxTrialWg / 4)) * 256 : (size_t)((enable - 128) & 31) * 256;
            VJF_HIP(hipMemcpyAsync(out32, ctx->ws + ctx->cv.mg_stamps + at, 256, hipMemcpyDeviceToHost, ctx->stream));
            VJF_HIP(hipStreamSynchronize(ctx->stream));
        }
        return 0;
    }
    if (enable >= 64) {                                      // 64 + k: 256-byte chunk k of the trial kernel's per-workgroup partials (even steps)
        if (out32) {
            VJF_HIP(hipMemcpyAsync(out32, ctx->ws + ctx->cv.partial + (size_t)(enable - 64) * 256, 256, hipMemcpyDeviceToHost, ctx->stream));
            VJF_HIP(hipStreamSynchronize(ctx->stream));
        }
        return 0;
    }
    if (enable >= 16) ring = (enable - 16) % 10;             // 16 + k: ring entry k (steps with epoch % 8 == k), mode unchanged
    else {
        ctx->stamps = enable != 0;
        ctx->stamps_keep_overlap = enable == 2;
    }
    if (out32) {
        VJF_HIP(hipMemcpyAsync(out32, ctx->ws + ctx->cv.work + vjf_serial_work_floats(ctx->plan) * 4 + ring * 256, 256, hipMemcpyDeviceToHost, ctx->stream));
        VJF_HIP(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

int vjf_reduce_buffer(vjf_ctx* ctx, float** ptr, int64_t* n_floats) {
    if (!ctx || !ptr || !n_floats) return fail(-1, "vjf_reduce_buffer: null argument");
    *ptr = (float*)(ctx->ws + ctx->cv.red);
    *n_floats = ctx->plan.red_len;
    return 0;
}

namespace {
int refresh_aux(vjf_ctx* c, hipStream_t st = nullptr) {
    if (!c->mfma_trial && !c->mega_ok) return 0;
    hipLaunchKernelGGL(vjf_aux_kernel, dim3(32), dim3(256), 0, st ? st : c->stream, c->plan, (const float*)c->state, (float*)(c->ws + c->cv.aux));
    VJF_HIP(hipGetLastError());
    return 0;
}

// the context's device for the duration of an entry point; the caller's current device is put back on the way out
struct DeviceGuard {
    int prev = -1, dev;
    explicit DeviceGuard(int d) : dev(d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); }
    ~DeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};

constexpr int kReplayMaskWord = 56, kReplayRhoWord = 57;   // words of the flag block no hand-off uses
constexpr int kStatsWord = 58;                            // three-stream route: RLS statistics (summed over ranks) of how many steps are in memory
constexpr unsigned kScAll = (1u << RS_N) - 1u;
constexpr unsigned kScRls = 1u << RS_SDX2;                 // the one loss sum the RLS chain reads

int check_step_args(vjf_ctx* c, int32_t B, const float* y, const float* u, const float* mu_s, const float* lv_s,
                    const float* eps_s, const float* eps_t, float* mu_t, float* lv_t) {
    if (B < 1 || B > c->cfg.max_batch) return fail(-20, "vjf_filter: B=%d outside [1, max_batch=%d]", B, c->cfg.max_batch);
    if (!y || !eps_s || !eps_t || !mu_t || !lv_t) return fail(-1, "vjf_filter: null tensor");
    if (c->plan.du > 0 && !u) return fail(-21, "vjf_filter: u is required when udim > 0");
    if ((mu_s == nullptr) != (lv_s == nullptr)) return fail(-22, "vjf_filter: mu_s and lv_s must both be given or both be null");
    return 0;
}

VjfTrialArgs trial_args(vjf_ctx* c, int32_t B, const float* y, const float* u, const float* mu_s, const float* lv_s,
                        const float* eps_s, const float* eps_t, float* mu_t, float* lv_t, uint32_t flags, int gen = 0) {
    VjfTrialArgs a{};
    a.y = y; a.u = u; a.mu_s = mu_s; a.lv_s = lv_s; a.eps_s = eps_s; a.eps_t = eps_t; a.mu_t = mu_t; a.lv_t = lv_t;
    a.state = c->state;
    a.E = (float*)(c->ws + (gen ? c->cv.E2 : c->cv.E)); a.ACT = (float*)(c->ws + c->cv.ACT); a.DEL = (float*)(c->ws + c->cv.DEL);
    a.partial = (float*)(c->ws + (gen ? c->cv.partial2 : c->cv.partial));
    a.B = B; a.flags = flags;
    return a;
}

int trial_blocks(const vjf_ctx* c, int B) { return c->mfma_trial ? (B + 15) / 16 : (B + 3) / 4; }   // (wide path: 4 trials per loss workgroup)

// one GEMM of the wide routes: a narrow output (N <= 128) splits K over the wavefronts of 32 x 32-tile workgroups; else 128 x 128 or
// 128 x 64 tiles when the shape fills the chip with them, else the 64 x 64 kernel
void launch_wide_gemm(const VjfWideGemm& g0, hipStream_t st) {
    VjfWideGemm g = g0;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15u) == 0; };
    // 16-byte loads: aligned rows whose extent along the contiguous direction is a multiple of 4
    g.va = (g.lda % 4 == 0) && al16(g.A) && ((g.ta ? g.M : g.K) % 4 == 0);
    g.vb = (g.ldb % 4 == 0) && al16(g.Bm) && ((g.nt ? g.K : g.N) % 4 == 0);
    const int tm = (g.M + 127) / 128;
    if (g.N <= 128)
        hipLaunchKernelGGL(vjf_skinny_gemm_kernel<8>, dim3((g.M + 31) / 32, (g.N + 31) / 32), dim3(512), 0, st, g);
    else if (!g.ta && g.va && g.vb && g.M >= 256 && g.N >= 256 && tm * ((g.N + 127) / 128) >= 192) {
        const dim3 grid((g.N + 127) / 128, tm);
        if (g.nt) hipLaunchKernelGGL((vjf_wide_gemm3_kernel<128, 16, 4, false, true>), grid, dim3(512), 0, st, g);
        else hipLaunchKernelGGL((vjf_wide_gemm3_kernel<128, 16, 4, false, false>), grid, dim3(512), 0, st, g);
    } else if (!g.ta && g.va && g.vb && g.M >= 256 && g.N >= 64 && tm * ((g.N + 63) / 64) >= 32) {
        const dim3 grid((g.N + 63) / 64, tm);
        if (g.nt) hipLaunchKernelGGL((vjf_wide_gemm3_kernel<64, 32, 4, false, true>), grid, dim3(512), 0, st, g);
        else hipLaunchKernelGGL((vjf_wide_gemm3_kernel<64, 32, 4, false, false>), grid, dim3(512), 0, st, g);
    }
    else
        hipLaunchKernelGGL(vjf_wide_gemm_kernel, dim3((g.N + 63) / 64, (g.M + 63) / 64), dim3(256), 0, st, g);
}

// K1 of the per-step routes.  part: 0 whole step, 1 forward half, 2 backward half (matrix-core kernel only)
int launch_trial(vjf_ctx* c, const VjfTrialArgs& a, int part, hipStream_t st, bool count_fwd = false,
                 const unsigned* rls_done = nullptr, unsigned rls_target = 0) {
    const VjfPlan& P = c->plan;
    const int nblk = trial_blocks(c, a.B);
    if (c->mfma_trial) {
        VjfTrialMfmaArgs m{};
        m.t = a; m.aux = (const float*)(c->ws + c->cv.aux); m.part = part;
        m.rls_done = rls_done; m.rls_target = rls_target;
        if (a.replay) {                                            // the backward half again; counted only where an RLS update on another
            m.part = 2;                                            // stream must not overwrite W, w_chol, sigma under it (count_fwd)
            if (count_fwd) m.done = (unsigned*)(c->ws + c->cv.flags) + 16;
            hipLaunchKernelGGL(vjf_trial_mfma_kernel, dim3(nblk), dim3(VJF_K1M_THREADS), c->lds_k1m, st, P, m);
            VJF_HIP(hipGetLastError());
            return 0;
        }
        m.done = (unsigned*)(c->ws + c->cv.flags) + 16;
        if (part != 1) c->k1_count += (unsigned)nblk;
        if (part == 1 && count_fwd) { m.fwd_done = (unsigned*)(c->ws + c->cv.flags) + 48; c->fwd_count += (unsigned)nblk; }
        m.stamps = c->stamps ? (unsigned long long*)(c->ws + c->cv.work + vjf_serial_work_floats(P) * 4) : nullptr;
        hipLaunchKernelGGL(vjf_trial_mfma_kernel, dim3(nblk), dim3(VJF_K1M_THREADS), c->lds_k1m, st, P, m);
    } else {
        // working set beyond LDS: one GEMM over all trials per layer (vjf_trial_wide.h)
        VjfWideArgs w{};
        w.t = a;
        float* wb = (float*)(c->ws + c->cv.wide);
        const size_t Bz = (size_t)a.B;
        w.XU = wb; w.PM = w.XU + Bz * P.dxu; w.PY = w.PM + Bz * P.dz + ((Bz + 3) / 4) * 4; w.Z = w.PY + Bz * P.dy;
        const float* S = c->state;
        auto gemm = [&](const float* A_, int lda, const float* Bm, int ldb, float* C_, int ldc, int N, int K, int nt, int epi,
                        const float* bias = nullptr, const float* src = nullptr, int lds = 0) {
            VjfWideGemm g{};
            g.A = A_; g.lda = lda; g.Bm = Bm; g.ldb = ldb; g.C = C_; g.ldc = ldc; g.M = a.B; g.N = N; g.K = K; g.nt = nt; g.epi = epi;
            g.bias = bias; g.src = src; g.lds = lds; g.src_scale = 1.f; g.eps_t = a.eps_t; g.lv_t = a.lv_t;
            g.ok = a.replay ? (const int*)a.replay_mask : nullptr;     // (a replay's launches do nothing when the word is 0)
            launch_wide_gemm(g, st);
        };
        const int gx = 1024;
        // part 1: everything up to the decoder (no use of W, w_chol, sigma); part 2: predictive moments, losses, backward; 0: both
        if (!a.replay && part != 2) {                              // (a replay: what the backward half reads stays in place)
        hipLaunchKernelGGL(vjf_wide_in_kernel, dim3(a.B < 2048 ? a.B : 2048), dim3(256), 0, st, P, w);
        hipLaunchKernelGGL(vjf_wide_rbf_kernel, dim3((P.n + 255) / 256, (a.B + 15) / 16), dim3(256), 0, st, P, w);
        int kin = P.din;
        for (int l = 0; l < P.L; ++l) {                            // h_l = tanh(h_{l-1} W_l^T + b_l)   (recognition.py:31-36)
            gemm(a.ACT + P.colA_act[l], P.ldA, S + P.off[VJF_SLOT_REC_W0 + 2 * l], kin, a.ACT + P.colA_act[l + 1], P.ldA, P.h[l], kin, 1,
                 WEPI_TANH_BIAS, S + P.off[VJF_SLOT_REC_B0 + 2 * l]);
            kin = P.h[l];
        }
        if (P.off[VJF_SLOT_LV_W] == P.off[VJF_SLOT_MEAN_W] + P.dz * kin) {   // the heads' weights lie one behind the other: ONE product, N = 2 dz
            VjfWideGemm g{};
            g.A = a.ACT + P.colA_act[P.L]; g.lda = P.ldA; g.Bm = S + P.off[VJF_SLOT_MEAN_W]; g.ldb = kin; g.C = a.mu_t; g.C2 = a.lv_t; g.ldc = P.dz;
            g.M = a.B; g.N = 2 * P.dz; g.K = kin; g.nt = 1; g.epi = WEPI_HEADS; g.bias = S + P.off[VJF_SLOT_LV_B];
            launch_wide_gemm(g, st);
        } else {
        gemm(a.ACT + P.colA_act[P.L], P.ldA, S + P.off[VJF_SLOT_MEAN_W], kin, a.mu_t, P.dz, P.dz, kin, 1, WEPI_NONE);
        gemm(a.ACT + P.colA_act[P.L], P.ldA, S + P.off[VJF_SLOT_LV_W], kin, a.lv_t, P.dz, P.dz, kin, 1, WEPI_BIAS, S + P.off[VJF_SLOT_LV_B]);
        }
        hipLaunchKernelGGL(vjf_wide_mid_kernel, dim3(gx), dim3(256), 0, st, P, w);
        gemm(a.ACT + P.colA_xt, P.ldA, S + P.off[VJF_SLOT_DEC_W], P.dz, w.PY, P.dy, P.dy, P.dz, 1, WEPI_BIAS, S + P.off[VJF_SLOT_DEC_B]);
        }
        if (part == 1) { VJF_HIP(hipGetLastError()); return 0; }
        if (!a.replay) {
        gemm(a.E, P.ldE, S + P.off[VJF_SLOT_W_MEAN], P.dz, w.PM, P.dz, P.dz, P.n, 0, WEPI_ADD_SRC, nullptr, w.XU, P.dxu);
        gemm(a.E, P.ldE, S + P.off[VJF_SLOT_W_CHOL], P.n, w.Z, P.n, P.n, P.n, 0, WEPI_NONE);
        }
        hipLaunchKernelGGL(vjf_wide_loss_kernel, dim3((a.B + 3) / 4), dim3(256), 0, st, P, w);
        // backward: dxt = dpy C into dmu / dlv; dh_L = dmu Wm + dlv Wl; da_l = (da_{l+1} W_{l+1}) (1 - h_l^2)
        gemm(a.DEL + P.colD_dpy, P.ldD, S + P.off[VJF_SLOT_DEC_W], P.dz, a.DEL + P.colD_dmu, P.ldD, P.dz, P.dy, 0, WEPI_SEED);
        const int hL = P.h[P.L - 1];
        if (P.off[VJF_SLOT_LV_W] == P.off[VJF_SLOT_MEAN_W] + P.dz * hL && P.colD_dlv == P.colD_dmu + P.dz)
            // the two heads' weights lie one behind the other in the state, their seeds side by side in DEL: ONE product with K = 2 dz
            gemm(a.DEL + P.colD_dmu, P.ldD, S + P.off[VJF_SLOT_MEAN_W], hL, a.DEL + P.colD_da[P.L - 1], P.ldD, hL, 2 * P.dz, 0, WEPI_DTANH, nullptr,
                 a.ACT + P.colA_act[P.L], P.ldA);
        else {
        gemm(a.DEL + P.colD_dmu, P.ldD, S + P.off[VJF_SLOT_MEAN_W], hL, a.DEL + P.colD_da[P.L - 1], P.ldD, hL, P.dz, 0, WEPI_NONE);
        gemm(a.DEL + P.colD_dlv, P.ldD, S + P.off[VJF_SLOT_LV_W], hL, a.DEL + P.colD_da[P.L - 1], P.ldD, hL, P.dz, 0, WEPI_ADDC_DTANH, nullptr,
             a.ACT + P.colA_act[P.L], P.ldA);
        }
        for (int l = P.L - 1; l >= 1; --l)
            gemm(a.DEL + P.colD_da[l], P.ldD, S + P.off[VJF_SLOT_REC_W0 + 2 * l], P.h[l - 1], a.DEL + P.colD_da[l - 1], P.ldD, P.h[l - 1], P.h[l], 0,
                 WEPI_DTANH, nullptr, a.ACT + P.colA_act[l], P.ldA);
    }
    VJF_HIP(hipGetLastError());
    return 0;
}

// Gram tiles of jobs [job0, job0 + njobs) and their slab reduction into `red`
int launch_gram(vjf_ctx* c, int B, int job0, int njobs, unsigned sc_mask, float* red, hipStream_t st, int gen = 0, const unsigned* run_if = nullptr,
                unsigned* done_count = nullptr) {   // done_count: see VjfReduceArgs (njobs + (sc_mask ? 1 : 0) arrivals)
    const VjfPlan& P = c->plan;
    const int nsplit = split_for(B);
    VjfGramArgs g{};
    g.jobs = (const VjfJob*)(c->ws + c->cv.jobs);
    g.E = (const float*)(c->ws + (gen ? c->cv.E2 : c->cv.E)); g.ACT = (const float*)(c->ws + c->cv.ACT); g.DEL = (const float*)(c->ws + c->cv.DEL);
    g.slabs = (float*)(c->ws + c->cv.slabs);
    g.B = B; g.nsplit = nsplit; g.job0 = job0;
    g.rows_per_split = ((B + nsplit - 1) / nsplit + 7) / 8 * 8;
    g.run_if = run_if;
    hipLaunchKernelGGL(vjf_gram_kernel, dim3(njobs * nsplit), dim3(VJF_GRAM_THREADS), 0, st, P, g);
    VJF_HIP(hipGetLastError());
    VjfReduceArgs r{};
    r.jobs = g.jobs; r.slabs = g.slabs; r.partial = (const float*)(c->ws + (gen ? c->cv.partial2 : c->cv.partial)); r.red = red;
    r.njobs = njobs; r.nsplit = nsplit; r.nblocks_k1 = trial_blocks(c, B); r.job0 = job0; r.sc_mask = sc_mask; r.run_if = run_if;
    r.done_count = done_count;
    hipLaunchKernelGGL(vjf_gram_reduce_kernel, dim3(njobs + (sc_mask ? 1 : 0)), dim3(VJF_REDUCE_THREADS), 0, st, P, r);
    VJF_HIP(hipGetLastError());
    return 0;
}

// which: 0 whole prep grid, 1 RLS operand rows only, 2 SGD + scalars only
int launch_prep(vjf_ctx* c, int32_t B_total, float* loss4, uint32_t flags, const float* red, int which, hipStream_t st,
                const unsigned* run_word = nullptr, unsigned run_epoch = 0, const unsigned* start_count = nullptr, unsigned start_target = 0,
                int replay = 0,                                 // replay: 1 first pass with a replay behind it, 2 the pass behind the replay
                const unsigned* wait_count = nullptr, unsigned wait_target = 0) {   // the operand kernel waits in-kernel for *wait_count
    const VjfPlan& P = c->plan;
    VjfPrepArgs p{};
    p.state = c->state; p.red = red; p.gbuf = (float*)(c->ws + c->cv.work);
    p.aux = (float*)(c->ws + c->cv.aux);
    p.loss4 = loss4; p.B_total = B_total; p.flags = flags;
    p.n_rowblk = (P.n + VJF_PREP_ROWS - 1) / VJF_PREP_ROWS;
    p.n_sgdblk = (P.train_len + 1023) / 1024;
    p.run_word = run_word; p.run_epoch = run_epoch; p.start_count = start_count; p.start_target = start_target;
    p.wait_count = wait_count; p.wait_target = wait_target;
    if (replay) {
        p.replay_mask = (unsigned*)(c->ws + c->cv.flags) + kReplayMaskWord; p.replay_rho = (float*)(c->ws + c->cv.flags) + kReplayRhoWord;
        p.replay_pass = replay == 2 ? 1 : 0;
    }
    if (which != 2 && P.dz > 16) {                             // (the matrix-core operand kernel holds one 16-column tile of W)
        p.bid0 = 0;
        const int grid = which == 1 ? p.n_rowblk : p.n_rowblk + p.n_sgdblk + 1;
        hipLaunchKernelGGL(vjf_prep_kernel, dim3(grid), dim3(256), 0, st, P, p);
        VJF_HIP(hipGetLastError());
        return 0;
    }
    if (which != 2) {                                          // RLS operands: g and P += G/v, 16 rows per workgroup
        const size_t lds = vjf_prepg_lds_bytes(P);
        hipLaunchKernelGGL(vjf_prepg_kernel, dim3((P.n + 15) / 16), dim3(256), lds, st, P, p);
        VJF_HIP(hipGetLastError());
        if (which == 1) return 0;
    }
    p.bid0 = p.n_rowblk;                                       // clip + SGD and the scalars
    hipLaunchKernelGGL(vjf_prep_kernel, dim3(p.n_sgdblk + 1), dim3(256), 0, st, P, p);
    VJF_HIP(hipGetLastError());
    return 0;
}

// Cholesky + RLS tail + state-noise update of one step.  `st_post` may differ from `st` (three-stream route): the whole update --
// Cholesky workgroup, y / W workgroup, inverse workgroups -- then goes out as ONE launch on `st`, whose workgroups hand the columns
// of L to each other through flags (all of them belong to one grid; the operand kernel precedes it in `st`, so g is in place).
int launch_rls(vjf_ctx* c, int32_t B_total, uint32_t flags, const float* red, hipStream_t st, bool one_launch, hipEvent_t stop = nullptr,
               bool no_triclean = false, const VjfTrialArgs* ta = nullptr) {
    const VjfPlan& P = c->plan;
    if (!(flags & VJF_FLAG_UPDATE)) return 0;
    VjfCholArgs a{};
    a.state = c->state; a.red = red; a.gbuf = (const float*)(c->ws + c->cv.work); a.B_total = B_total; a.flags = flags;
    a.stamps = c->stamps ? (unsigned long long*)(c->ws + c->cv.work + vjf_serial_work_floats(P) * 4) : nullptr;
    const int nbl = (P.n + 31) / 32;
    float* dinv = (float*)(c->ws + c->cv.post);
    double* rpart = (double*)(c->ws + c->cv.post + (size_t)nbl * 1024 * 4);
    int* okflag = (int*)(c->ws + c->cv.post + (size_t)nbl * 1024 * 4 + VJF_RESID_BLOCKS * 8);
    unsigned* colflags = (unsigned*)(c->ws + c->cv.flags);
    a.post = c->post_kernels ? 1 : 0; a.dinv_out = dinv; a.ok_out = okflag; a.lscr = (float*)(c->ws + c->cv.lscr);
    a.flags_out = colflags; a.epoch = ++c->epoch; a.no_triclean = no_triclean ? 1 : 0;
    a.pscr = (float*)(c->ws + c->cv.pscr);
    const bool rls = !(flags & VJF_FLAG_WARM_UP);
    const bool pair = one_launch && c->post_kernels && rls && P.dz <= 16;
    if (!pair) {
        switch (vjf_chol_dzp(P.dz)) {
            case 4: hipLaunchKernelGGL(vjf_chol_lds_kernel<4>, dim3(1), dim3(VJF_CHOL_THREADS), c->lds_chol, st, P, a); break;
            case 8: hipLaunchKernelGGL(vjf_chol_lds_kernel<8>, dim3(1), dim3(VJF_CHOL_THREADS), c->lds_chol, st, P, a); break;
            case 12: hipLaunchKernelGGL(vjf_chol_lds_kernel<12>, dim3(1), dim3(VJF_CHOL_THREADS), c->lds_chol, st, P, a); break;
            case 16: hipLaunchKernelGGL(vjf_chol_lds_kernel<16>, dim3(1), dim3(VJF_CHOL_THREADS), c->lds_chol, st, P, a); break;
            default: hipLaunchKernelGGL(vjf_chol_lds_kernel<32>, dim3(1), dim3(VJF_CHOL_THREADS), c->lds_chol, st, P, a); break;
        }
        VJF_HIP(hipGetLastError());
    }
    if (!c->post_kernels) return 0;
    if (rls) {
        // inverse column halves + the y / W workgroup, which also carries the state-noise update
        VjfPostArgs pa{};
        pa.state = c->state; pa.dinv = dinv; pa.gbuf = a.gbuf; pa.lscr = a.lscr;
        pa.flags = colflags; pa.epoch = a.epoch; pa.status = c->state + P.off[VJF_SLOT_SCALARS] + VJF_SC_STATUS;
        pa.k1_done = c->mfma_trial ? colflags + 16 : nullptr; pa.k1_target = c->k1_count;
        pa.done = colflags + 32; pa.started = colflags + 24; c->post_count += (unsigned)(2 * nbl + 1);
        c->start_count += (unsigned)(2 * nbl + 1);
        pa.red = red; pa.B_total = B_total; pa.fold_sigma = 1; pa.acquire = c->handoff_acquire ? 1 : 0; pa.stamps = a.stamps;
        // Fewer trials than features on a rank that holds them all (the rows of [Phi | dx] are in the workspace): the new weights
        // reproduce dx almost exactly, and the quadratic form of the statistics, sum|dx|^2 - 2 tr(W^T Phi^T dx) + tr(W^T G W),
        // loses the residual under the fp32 rounding of its terms (sigma 1e-5 off where the reference's arithmetic is at 1e-6).
        // The residual is then formed as the reference forms it, dx - Phi W (vjf/model.py:373-374), behind the update: three
        // short launches on a route that is launch-bound anyway.
        const bool direct = ta && !pair && B_total < P.n && ta->B == B_total;
        if (direct) pa.fold_sigma = 0;
        if (pair) {
            pa.role = 2;
            const dim3 grid(2 + 2 * nbl);
            switch (vjf_chol_dzp(P.dz)) {
                case 4: VJF_LAUNCH(vjf_rls_pair_kernel<4>, grid, dim3(VJF_CHOL_THREADS), c->lds_chol, st, stop, P, a, pa); break;
                case 8: VJF_LAUNCH(vjf_rls_pair_kernel<8>, grid, dim3(VJF_CHOL_THREADS), c->lds_chol, st, stop, P, a, pa); break;
                case 12: VJF_LAUNCH(vjf_rls_pair_kernel<12>, grid, dim3(VJF_CHOL_THREADS), c->lds_chol, st, stop, P, a, pa); break;
                default: VJF_LAUNCH(vjf_rls_pair_kernel<16>, grid, dim3(VJF_CHOL_THREADS), c->lds_chol, st, stop, P, a, pa); break;
            }
        } else {
            VJF_LAUNCH(vjf_rls_post_kernel, dim3(2 * nbl + 1), dim3(VJF_POST_THREADS), c->lds_post, st, direct ? nullptr : stop, P, pa);
        }
        VJF_HIP(hipGetLastError());
        if (direct) {
            VjfWideGemm g{};
            float* R = ta->DEL;                                    // (free: the gradient sums -- and a replay's -- are formed)
            g.A = ta->E; g.lda = P.ldE; g.Bm = c->state + P.off[VJF_SLOT_W_MEAN]; g.ldb = P.dz; g.C = R; g.ldc = P.dz;
            g.M = ta->B; g.N = P.dz; g.K = P.n; g.epi = WEPI_NONE;
            launch_wide_gemm(g, st);
            VjfResidArgs ra{};
            ra.state = c->state; ra.red = red; ra.partial = rpart; ra.B_total = B_total; ra.flags = flags;
            hipLaunchKernelGGL(vjf_resid_direct_kernel, dim3(VJF_RESID_BLOCKS), dim3(256), 0, st, P, ra, (const float*)ta->E, (const float*)R, ta->B);
            VJF_LAUNCH(vjf_sigma_kernel, dim3(1), dim3(64), 0, st, stop, P, ra, (const int*)nullptr, 1);
            VJF_HIP(hipGetLastError());
        }
    } else {
        VjfResidArgs ra{};
        ra.state = c->state; ra.red = red; ra.partial = rpart; ra.B_total = B_total; ra.flags = flags;
        hipLaunchKernelGGL(vjf_resid_kernel, dim3(VJF_RESID_BLOCKS), dim3(256), 0, st, P, ra);
        VJF_LAUNCH(vjf_sigma_kernel, dim3(1), dim3(64), 0, st, stop, P, ra, (const int*)nullptr, 0);
        VJF_HIP(hipGetLastError());
    }
    return 0;
}

// K1 + Gram + slab reduce.  `aux_fresh`: the transposed weight copies are known to match the state blob.
int launch_local(vjf_ctx* c, int32_t B, const float* y, const float* u, const float* mu_s, const float* lv_s,
                 const float* eps_s, const float* eps_t, float* mu_t, float* lv_t, uint32_t flags, bool aux_fresh) {
    int rc = check_step_args(c, B, y, u, mu_s, lv_s, eps_s, eps_t, mu_t, lv_t);
    if (rc) return rc;
    c->on_mega = false;
    if (c->mfma_trial && !aux_fresh) { rc = refresh_aux(c); if (rc) return rc; }
    rc = launch_trial(c, trial_args(c, B, y, u, mu_s, lv_s, eps_s, eps_t, mu_t, lv_t, flags), 0, c->stream);
    if (rc) return rc;
    return launch_gram(c, B, 0, c->njobs, kScAll, (float*)(c->ws + c->cv.red), c->stream);
}

int ensure_stream2(vjf_ctx* c) {
    if (c->stream2) return 0;
    VJF_HIP(hipSetDevice(c->cfg.device));
    VJF_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    VJF_HIP(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
    VJF_HIP(hipEventCreate(&c->ev_c));
    VJF_HIP(hipEventCreate(&c->ev_s));          // (default flags: the events are attached to kernel launches)
    for (int i = 0; i < 2; ++i) {
        VJF_HIP(hipEventCreateWithFlags(&c->ev_f[i], hipEventDisableTiming));
        VJF_HIP(hipEventCreateWithFlags(&c->ev_r[i], hipEventDisableTiming));
        VJF_HIP(hipEventCreateWithFlags(&c->ev_b[i], hipEventDisableTiming));
        VJF_HIP(hipEventCreateWithFlags(&c->ev_g[i], hipEventDisableTiming));
    }
    return 0;
}

// ---- the one-launch route (single rank; sgd + update, no warm-up): one launch of vjf_mega_kernel -- a grid that is resident as a
//      whole -- per chunk of steps
int filter_seq_mega(vjf_ctx* c, int32_t T, int32_t B, const float* y, const float* u, const float* eps, const float* mu0,
                    const float* lv0, float* mu, float* lv, float* loss, uint32_t flags) {
    const VjfPlan& P = c->plan;
    const size_t sz = (size_t)B * P.dz;
    int rc = check_step_args(c, B, y, u, mu0, lv0, eps, eps + sz, mu, lv);
    if (rc) return rc;
    MegaShape m{};
    if (!mega_shape(P, B, c->ncu, flags, &m)) return fail(-26, "vjf_filter_seq: %d compute units are too few for the one-launch route", c->ncu);
    VJF_HIP(hipSetDevice(c->cfg.device));
    // (parameters that fit the trial role's LDS: it reads the image the SGD role builds at the start of the launch; else the state
    //  and its transposed copies, refreshed here)
    if (!vjf_mega_trial_lds(P, (int)(kMegaLds / 4) - 8).theta) { rc = refresh_aux(c); if (rc) return rc; }
    // every counter and flag of the launch starts at 0: the launch before it zeroed this block as its first act (the context's first
    // launch finds both blocks zeroed by vjf_ctx_create) -- no memset in front of the launch, no dispatch gap behind it
    unsigned* cnt = (unsigned*)(c->ws + c->cv.mg_cnt) + (size_t)(c->mega_launches & 1u) * MG_C_WORDS;
    unsigned* cnt_next = (unsigned*)(c->ws + c->cv.mg_cnt) + (size_t)((c->mega_launches + 1u) & 1u) * MG_C_WORDS;
    const int nbl = (P.n + 31) / 32;
    const unsigned npost = (unsigned)(2 * nbl + 1);
    float* rede[2] = {(float*)(c->ws + c->cv.red2), (float*)(c->ws + c->cv.red3)};
    float* stw = c->state + P.off[VJF_SLOT_SCALARS] + VJF_SC_STATUS;
    VjfMegaArgs A{};
    A.T = T; A.B = B; A.ntiles = m.ntiles;
    A.n_rls = m.n_rls; A.n_trial = m.n_trial; A.n_gram = m.n_gram; A.n_prep = m.n_prep; A.n_sgd = m.n_sgd;
    A.n_sgd_live = (flags & VJF_FLAG_SGD) ? m.n_sgd : 1;
    A.n_mom = m.n_mom; A.mom = (float*)(c->ws + c->cv.mg_mom);
    A.y = y; A.u = u; A.eps = eps; A.mu0 = mu0; A.lv0 = lv0; A.mu = mu; A.lv = lv; A.loss = loss;
    A.state = c->state; A.aux = (float*)(c->ws + c->cv.aux);
    A.img = (const float*)(c->ws + c->cv.mg_img);
    A.pmsave = (float*)(c->ws + c->cv.mg_pmsave);
    A.slab_early = (float*)(c->ws + c->cv.mg_early); A.slab_late = (float*)(c->ws + c->cv.mg_late); A.gslab = (float*)(c->ws + c->cv.mg_gslab);
    A.red0 = rede[0]; A.red1 = rede[1];
    A.gbuf = (float*)(c->ws + c->cv.work);
    A.xt = (const float*)(c->ws + c->cv.mg_xt);
    A.cnt = cnt; A.cnt_next = cnt_next; A.flags = flags;
    const bool acq = c->handoff_acquire;                                   // (VJF_HANDOFF_ACQUIRE, read when the context is created)
    if (acq) A.flags |= VJF_FLAG_HANDOFF_ACQUIRE;
    A.slab_len = vjf_mega_slab_layout(P).len;
    A.early_len = ((P.n + 3) & ~3) * 16 + 8; A.late_len = A.slab_len + 8 * VJF_MG_RING;
    A.gram_rows = m.gram_rows;
    A.lds_floats = (int)(kMegaLds / 4) - 8;               // (a few static words beside the dynamic region)
    A.sl_pidx = (const int*)(c->ws + c->cv.mg_pidx); A.sl_cidx = (const int*)(c->ws + c->cv.mg_cidx); A.sl_grp = (const int*)(c->ws + c->cv.mg_grp);
    A.stamps = c->stamps ? (unsigned long long*)(c->ws + c->cv.mg_stamps) : nullptr;
    VjfCholArgs C{};
    C.state = c->state; C.red = rede[0]; C.red2 = rede[1]; C.gbuf = A.gbuf; C.B_total = B; C.flags = flags | (acq ? VJF_FLAG_HANDOFF_ACQUIRE : 0u);
    C.stamps = c->stamps ? (unsigned long long*)(c->ws + c->cv.work + vjf_serial_work_floats(P) * 4) : nullptr;
    float* dinv = (float*)(c->ws + c->cv.post);
    C.post = 1; C.dinv_out = dinv; C.ok_out = (int*)(c->ws + c->cv.post + (size_t)nbl * 1024 * 4 + VJF_RESID_BLOCKS * 8);
    C.lscr = (float*)(c->ws + c->cv.lscr); C.flags_out = cnt + MG_C_COLFLAGS; C.epoch = 1; C.no_triclean = 1;
    C.pscr = (float*)(c->ws + c->cv.pscr); C.self_prep = 1; C.src_state = 1;
    C.wait_count = cnt + MG_C_PDONE; C.wait_target = 0; C.wait_stride = npost;
    C.stat_count = cnt + MG_C_STAT; C.stat_target = (unsigned)m.n_gram; C.stat_stride = (unsigned)m.n_gram;
    C.nsteps = T; C.step0 = 0;
    C.sig_word = (const unsigned long long*)(cnt + MG_C_SIGW);
    { const char* ie = getenv("VJF_DEBUG_INJECT"); C.inject_epoch = ie ? (unsigned)atoi(ie) : 0u; }   // (test hook: a hand-off of step k - 1 reports a time-out)
    VjfPostArgs Q{};
    Q.state = c->state; Q.dinv = dinv; Q.gbuf = A.gbuf; Q.lscr = C.lscr; Q.flags = cnt + MG_C_COLFLAGS; Q.epoch = 1; Q.status = stw;
    Q.k1_done = cnt + MG_C_K1; Q.k1_target = (unsigned)m.n_trial; Q.k1_stride = (unsigned)m.n_trial;
    Q.done = cnt + MG_C_PDONE; Q.started = cnt + MG_C_STARTED;
    Q.red = rede[0]; Q.red2 = rede[1]; Q.B_total = B; Q.fold_sigma = 1; Q.stamps = C.stamps; Q.undo_P = 1;
    Q.sig_word = (unsigned long long*)(cnt + MG_C_SIGW); Q.acquire = acq ? 1 : 0;
    Q.prep_count = cnt + MG_C_PREP; Q.prep_target = (unsigned)m.n_prep; Q.prep_stride = (unsigned)m.n_prep;
    Q.nsteps = T; Q.step0 = 0; Q.role = 2;
    Q.xt = (float*)(c->ws + c->cv.mg_xt); Q.xt_count = cnt + MG_C_XT;
    VjfPlan Pk = P;
    const bool ungated = !(flags & (VJF_FLAG_SGD | VJF_FLAG_UPDATE));       // (lite kernel: builders and moments workgroups are the same ones)
    const int grid = m.n_rls + m.n_trial + m.n_gram + m.n_prep + (ungated ? (m.n_mom > m.n_sgd ? m.n_mom : m.n_sgd) : m.n_sgd + m.n_mom);
#ifdef VJF_CHAOS
    {
        static bool told = false;
        if (!told) fprintf(stderr, "vjf chaos build: roles rls %d trial %d gram %d operand %d sgd %d\n", m.n_rls, m.n_trial, m.n_gram, m.n_prep, m.n_sgd);
        told = true;
    }
#endif
    // One resident grid: every wait in it is for a workgroup of the SAME launch, so the whole grid must be on the device at once.
    // The check is the one hipLaunchCooperativeKernel makes -- workgroups per compute unit (occupancy query, made once when the
    // context is created) x compute units >= grid -- and the launch itself is a plain one: identical residency (MI355X guide,
    // "Residency and cooperative launch"), and no cooperative queue.  That queue is why the API is avoided: a process that has
    // made ONE cooperative launch faults in the HIP runtime's exit handler when it runs under rocprofv3 (hsa queue teardown behind
    // the profiler's finalisation; tools/coop_exit_repro.hip shows it with 20 lines and no other library) -- every profile of
    // round 2 ended in SIGSEGV for this reason.  When the grid does not fit -- compute units masked off, a smaller part -- the
    // context leaves this route for good and the caller's entry point goes on with the per-step kernels (nothing of the state
    // has been touched yet).
    const char* refuse = getenv("VJF_DEBUG_REFUSE_COOP");                  // (test hook)
    const bool full = m.n_rls > 0;                                         // (else: trial and SGD roles only, vjf_mega_lite_kernel)
    const int per_cu = full ? c->mega_wg_per_cu : c->lite_wg_per_cu;
    if ((refuse && atoi(refuse)) || per_cu < 1 || grid > per_cu * c->ncu) {
        c->mega_ok = false;
        return kMegaRefused;
    }
    { const char* ab = getenv("VJF_DEBUG_ABSENT"); A.alive_extra = ab ? atoi(ab) : 0; }
    if (DevShared* dm = dev_shared(c->cfg.device)) A.host_word = dm->mirror_d ? dm->mirror_d + VJF_MIRROR_SLOT(stw) : nullptr;   // (test hook: the grid waits for workgroups that never come)
    // the chain of resident grids of this process and device (DevShared): behind the previous one's completion, whichever context's
    DevShared* d = dev_shared(c->cfg.device);
    std::unique_lock<std::mutex> chain;
    hipEvent_t done = nullptr;
    if (d && d->chained) {
        chain = std::unique_lock<std::mutex>(d->mu);
        if (d->last_valid && d->last_stream != c->stream) VJF_HIP(hipStreamWaitEvent(c->stream, d->last, 0));
        done = d->last;
    }
    if (full) VJF_LAUNCH(vjf_mega_kernel, dim3(grid), dim3(VJF_MG_THREADS), kMegaLds, c->stream, done, Pk, A, C, Q);
    else VJF_LAUNCH(vjf_mega_lite_kernel, dim3(grid), dim3(VJF_MG_THREADS), kMegaLds, c->stream, done, Pk, A);
    const hipError_t le = hipGetLastError();
    if (done && le == hipSuccess) { d->last_stream = c->stream; d->last_valid = true; }
    c->on_mega = true;
    if (le == hipErrorLaunchOutOfResources) {
        c->mega_ok = false;
        return kMegaRefused;
    }
    if (le == hipSuccess) ++c->mega_launches;
    VJF_HIP(le);
    return 0;
}

// ---- the three-stream route (trials sharded over ranks, RCCL communicators in the context).  Step t's work splits into
//   chain A (caller's stream): K1 backward half(t) -> gradient Gram -> [all-reduce] -> clip + SGD -> K1 forward half(t+1)
//   chain B (second stream):   [gate: forward half(t)] E^T E Gram(t) -> [all-reduce] -> [gate: RLS(t-1)] P += G/v, g -> the RLS
//                              update of step t as ONE launch (Cholesky workgroup, y / W workgroup, inverse workgroups)
// K1's backward half(t+1) needs W, w_chol, sigma of step t, nothing else on chain A does; chain B(t+1) needs only the forward
// half's rows.  So a step costs max(A, B) instead of A + B.  Every kernel that waits in-kernel (the gates, the backward half,
// the y / W and inverse workgroups) waits for work that the host enqueued BEFORE it: whatever hardware queues the streams share,
// the producers are dispatched first and run to completion.  Results are those of the one-stream order bit for bit (same
// kernels, same sums).  RLS statistics alternate between two reduce buffers.
int filter_seq_streams(vjf_ctx* c, int32_t T, int32_t B, const float* y, const float* u, const float* eps, const float* mu0,
                       const float* lv0, float* mu, float* lv, float* loss, uint32_t flags) {
    int rc = ensure_stream2(c);
    if (rc) return rc;
    c->on_mega = false;
    const VjfPlan& P = c->plan;
    const size_t sy = (size_t)B * P.dy, su = (size_t)B * P.du, sz = (size_t)B * P.dz;
    hipStream_t sa = c->stream, sb = c->stream2, sc = c->stream3;
    float* redg = (float*)(c->ws + c->cv.red);                             // gradients + loss sums (chain A)
    float* rede[2] = {(float*)(c->ws + c->cv.red2), (float*)(c->ws + c->cv.red3)};   // RLS statistics of even / odd steps (chain B)
    const int fw = c->fake_world;
    const int Bt = B * c->world * fw;                                      // trials of all ranks
    const bool exact = (flags & VJF_FLAG_EXACT_NONFINITE) && (flags & VJF_FLAG_SGD) && c->comm_a != nullptr;
    auto args = [&](int t) {
        return trial_args(c, B, y + t * sy, u ? u + t * su : nullptr, t ? mu + (t - 1) * sz : mu0, t ? lv + (t - 1) * sz : lv0,
                          eps + (size_t)t * 2 * sz, eps + (size_t)t * 2 * sz + sz, mu + t * sz, lv + t * sz, flags, t & 1);
    };
    auto all_reduce = [&](float* p, size_t nfl, void* comm, hipStream_t st) -> int {
        if (!comm) return 0;
        VJF_NCCL(nccl().all_reduce(p, p, nfl, kNcclFloat, kNcclSum, comm, st));
        if (fw > 1) hipLaunchKernelGGL(vjf_scale_kernel, dim3(64), dim3(256), 0, st, p, (float)fw, (int)nfl);   // (test hook: fw identical ranks)
        return 0;
    };
    rc = check_step_args(c, B, y, u, mu0, lv0, eps, eps + sz, mu, lv);
    if (rc) return rc;
    if (c->comm_a) {
        // The ranks enter the sequence together: kernels of this route wait in-kernel (bounded, seconds) for kernels that sit behind
        // an all-reduce, and an all-reduce waits for the slowest rank -- one that is late with this CALL by more than the bound
        // (data loading, a first call) must not run its peers' waits out.  One tiny all-reduce and a host synchronisation per
        // call; inside the sequence the per-step collectives keep the ranks in step.
        // (Both communicators: the first collective on one sets its channels up, which can take longer than the bound.)
        float* tok = (float*)(c->ws + c->cv.flags) + 60;                    // (words of the flag block no hand-off uses)
        VJF_HIP(hipMemsetAsync(tok, 0, 8, sa));
        VJF_NCCL(nccl().all_reduce(tok, tok, 1, kNcclFloat, kNcclSum, c->comm_a, sa));
        VJF_HIP(hipStreamSynchronize(sa));
        if (c->comm_b) {
            VJF_NCCL(nccl().all_reduce(tok + 1, tok + 1, 1, kNcclFloat, kNcclSum, c->comm_b, sc));   // (comm_b lives on sc)
            VJF_HIP(hipStreamSynchronize(sc));
        }
    }
    rc = refresh_aux(c);
    if (rc) return rc;
    const int ne = c->n_ejobs, ng = c->njobs - ne;
    unsigned* fl = (unsigned*)(c->ws + c->cv.flags);
    unsigned* fdone = fl + 48;
    const unsigned* pdone = fl + 32;
    float* stw = c->state + P.off[VJF_SLOT_SCALARS] + VJF_SC_STATUS;
    if ((rc = launch_trial(c, args(0), 1, sa, true))) return rc;          // prologue: forward half of step 0
    for (int t = 0; t < T; ++t) {
        // sb: RLS statistics of step t as soon as its forward half is done (a one-wavefront gate on the workgroup count: no
        //     cross-stream event inside the loop), then -- behind W, sigma of t-1 -- P += G/v, g, and the RLS update
        // sc: the statistics have a stream of their own -- they need the forward half of step t only, and sb is still inside the
        //     update of step t-1 when that is done (behind it they cost the chain sb a sixth of its step: 15 of 96 us)
        hipLaunchKernelGGL(vjf_gate_kernel, dim3(1), dim3(64), 0, sc, (const unsigned*)fdone, c->fwd_count, stw);
        // (the operand kernel of sb waits in-kernel for the statistics -- a cross-stream event costs 6-13 us on this stack: for the
        //  reduction's own workgroups on a single rank, for one more launch behind the sum over ranks otherwise)
        if ((rc = launch_gram(c, B, 0, ne, kScRls, rede[t & 1], sc, t & 1, nullptr, c->comm_b ? nullptr : fl + kStatsWord))) return rc;
        if (c->comm_b) {
            if ((rc = all_reduce(rede[t & 1] + P.red_G, (size_t)(P.red_len - P.red_G), c->comm_b, sc))) return rc;   // [G | FDX | sums]
            hipLaunchKernelGGL(vjf_count_kernel, dim3(1), dim3(64), 0, sc, fl + kStatsWord);
            ++c->stats_count;
        } else c->stats_count += (unsigned)(ne + 1);
        // sa: backward half(t) waits in-kernel for the RLS update of step t-1, behind the reloads of its forward half's rows
        if ((rc = launch_trial(c, args(t), 2, sa, false, t > 0 ? pdone : nullptr, c->post_count))) return rc;
        if (t == 0) {
            // the inverse workgroups write only the block-upper half of w_chol (block-lower of w_pchol): the other halves are
            // cleared once per blob (VJF_SC_TRI_CLEAN), here behind the backward half that may still read a full w_chol
            hipLaunchKernelGGL(vjf_triclean_kernel, dim3(64), dim3(256), 0, sa, P, c->state);
            hipLaunchKernelGGL(vjf_triclean_done_kernel, dim3(1), dim3(1), 0, sa, P, c->state);
            VJF_HIP(hipGetLastError());
        }
        // (P += G/v and g behind W, sigma of step t-1: the update of t-1 precedes them in sb)
        if ((rc = launch_prep(c, Bt, nullptr, flags, rede[t & 1], 1, sb, nullptr, 0, nullptr, 0, 0, fl + kStatsWord, c->stats_count))) return rc;
        if (exact) c->k1_count += (unsigned)trial_blocks(c, B);           // (the replayed backward half of this step reads W, w_chol, sigma too)
        if ((rc = launch_rls(c, Bt, flags, rede[t & 1], sb, true, nullptr, true))) return rc;
        if ((rc = launch_gram(c, B, ne, ng, kScAll & ~kScRls, redg, sa, t & 1))) return rc;
        if (c->comm_a) {                                                   // sum the gradients and the loss sums over ranks
            if ((rc = all_reduce(redg, (size_t)P.red_SCA + 4, c->comm_a, sa))) return rc;   // [grad | loss sums]: ONE collective
        }
        // (the scalar workgroup ends once the RLS workgroups of step t are resident: the next backward half spins on their results
        //  and must not take the CUs they need before they are placed)
        rc = launch_prep(c, Bt, loss ? loss + 4 * (size_t)t : nullptr, flags, redg, 2, sa, fl + VJF_CHOL_MAXBLK + 2, c->epoch, fl + 24, c->start_count,
                         exact ? 1 : 0);
        if (rc) return rc;
        if (exact) {
            // VJF_FLAG_EXACT_NONFINITE: the verdict on the step's loss (the same on every rank: it is taken on the summed loss terms) is
            // in the flag block now.  The backward half again with the dropped components' seeds at zero, its gradient sums, their sum
            // over ranks, the SGD pass from them -- every launch returns at once on an ordinary step; the collective runs regardless.
            VjfTrialArgs ar = args(t);
            ar.replay = 1;
            ar.replay_mask = (const unsigned*)(c->ws + c->cv.flags) + kReplayMaskWord;
            ar.replay_rho = (const float*)(c->ws + c->cv.flags) + kReplayRhoWord;
            if ((rc = launch_trial(c, ar, 2, sa, true))) return rc;
            if ((rc = launch_gram(c, B, ne, ng, 0u, redg, sa, t & 1, ar.replay_mask))) return rc;
            if ((rc = all_reduce(redg, (size_t)P.red_SCA, c->comm_a, sa))) return rc;
            if ((rc = launch_prep(c, Bt, nullptr, flags, redg, 2, sa, nullptr, 0, nullptr, 0, 2))) return rc;
        }
        if (t + 1 < T && (rc = launch_trial(c, args(t + 1), 1, sa, true))) return rc;
    }
    VJF_HIP(hipEventRecord(c->ev_c, sb));
    VJF_HIP(hipStreamWaitEvent(sa, c->ev_c, 0));                           // join: the caller's stream sees the final state
    return 0;
}

}  // namespace

#ifdef VJF_CHAOS
// diagnostic build: which workgroups are held, and where (vjf_plan.h), from the environment at every entry
static void chaos_refresh(const vjf_ctx* c) {
    const int range[6] = {getenv("VJF_CHAOS_LO") ? atoi(getenv("VJF_CHAOS_LO")) : 0, getenv("VJF_CHAOS_HI") ? atoi(getenv("VJF_CHAOS_HI")) : 1 << 30,
                          getenv("VJF_CHAOS_SITE") ? atoi(getenv("VJF_CHAOS_SITE")) : -1, getenv("VJF_CHAOS_KIND") ? atoi(getenv("VJF_CHAOS_KIND")) : 0,
                          getenv("VJF_CHAOS_TICKS") && atoi(getenv("VJF_CHAOS_TICKS")) > 0 ? atoi(getenv("VJF_CHAOS_TICKS")) : 20000,
                          getenv("VJF_CHAOS_MASK") ? atoi(getenv("VJF_CHAOS_MASK")) : 7};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(vjf_chaos_range), range, sizeof(range));
    const unsigned* base = (const unsigned*)(c->ws + c->cv.mg_cnt) + (size_t)(c->mega_launches & 1u) * MG_C_WORDS;   // (the block the next launch counts in)
    (void)hipMemcpyToSymbol(HIP_SYMBOL(vjf_chaos_base), &base, sizeof(base));
}
#define VJF_CHAOS_REFRESH(c) chaos_refresh(c)
#else
#define VJF_CHAOS_REFRESH(c) ((void)0)
#endif

int vjf_filter_local(vjf_ctx* c, int32_t B, const float* y, const float* u, const float* mu_s, const float* lv_s,
                     const float* eps_s, const float* eps_t, float* mu_t, float* lv_t, uint32_t flags) {
    if (!c) return fail(-1, "vjf_filter_local: null context");
    DeviceGuard on_device(c->cfg.device);                   // (every launch below goes to the context's device, whatever is current)
    VJF_CHAOS_REFRESH(c);
    VJF_HIP(hipSetDevice(c->cfg.device));
    if (int rp = refuse_if_poisoned(c, "vjf_filter_local")) return rp;
    return launch_local(c, B, y, u, mu_s, lv_s, eps_s, eps_t, mu_t, lv_t, flags, false);
}

namespace {
// The RLS update for feature counts beyond one compute unit's LDS (vjf_rlsb_kernels.h) and the state-noise update, on stream `st`
// `ta`: the trial-parallel half's arguments when this rank holds every trial, else null
// `before_write`: an event the stream waits for before the update's first store to the state (W, then w_chol, w_pchol, P, sigma) --
// the readers of the previous values on another stream; `resid`: B x dz floats for Phi W (default: the trial chain's DEL rows)
// `resident`: the column sequence as one resident launch (vjf_rlsc_loop_kernel) -- for an update that runs beside other streams' kernels
int launch_rlsb(vjf_ctx* c, int32_t B_total, uint32_t flags, const float* red, hipStream_t st, const VjfTrialArgs* ta = nullptr,
                hipEvent_t before_write = nullptr, float* resid = nullptr, bool resident = false) {
    const VjfPlan& P = c->plan;
    const int nbl = (P.n + 31) / 32;
    float* work = (float*)(c->ws + c->cv.work);
    VjfRlsbArgs a{};
    a.state = c->state; a.red = red; a.Lw = (float*)(c->ws + c->cv.lscr);
    a.X = work; a.gbuf = work + (size_t)P.n * P.n; a.ybuf = a.gbuf + (size_t)P.n * P.dz;
    a.Dinv = (float*)(c->ws + c->cv.post);
    a.Ld = (float*)(c->ws + c->cv.tbig);
    a.Pacc = a.Ld + (size_t)nbl * 1024;
    a.ok = (int*)(c->ws + c->cv.post + (size_t)nbl * 1024 * 4 + VJF_RESID_BLOCKS * 8);
    { const char* ab = getenv("VJF_DEBUG_RLSC_ABSENT"); a.absent_wg = ab ? atoi(ab) : 0; }
    const bool rls = !(flags & VJF_FLAG_WARM_UP);
    if (rls) {
        const int gx = 512;
        auto gemm = [&](const float* A_, int lda, int ta, const float* Bm, int ldb, float* C_, int ldc, int M, int N, int K, const int* ok) {
            VjfWideGemm g{};
            g.A = A_; g.lda = lda; g.ta = ta; g.Bm = Bm; g.ldb = ldb; g.C = C_; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.nt = 0;
            g.epi = WEPI_NONE; g.ok = ok;
            launch_wide_gemm(g, st);
        };
        const float* Sx = c->state;
        gemm(Sx + P.off[VJF_SLOT_W_PREC], P.n, 0, Sx + P.off[VJF_SLOT_W_MEAN], P.dz, a.gbuf, P.dz, P.n, P.dz, P.n, nullptr);   // P W
        hipLaunchKernelGGL(vjf_rlsb_prep_kernel, dim3(gx), dim3(256), 0, st, P, a);
        // block column k of L and block row k - 1 of X = L^-1 per launch (the block-upper part of X stays zero: the solves
        // below read all of it)
        VJF_HIP(hipMemsetAsync(a.X, 0, (size_t)P.n * P.n * 4, st));
        // (alone on the chip the launches are the faster form: 937 against 1017 us a step at config E, a step barrier costs more than
        //  a dispatch; beside the trial chain both give 770-780 us, the resident form with a third of the host's enqueue time)
        static const bool per_column = getenv("VJF_RLS_COLUMN_LAUNCHES") != nullptr;   // (A/B)
        if (resident && !per_column && 2 * nbl - 1 <= c->ncu)
            hipLaunchKernelGGL(vjf_rlsc_loop_kernel, dim3(2 * nbl - 1), dim3(VJF_RLSC_THREADS), 0, st, P, a, (unsigned*)(a.ok + 4));
        else
        for (int k = 0; k <= nbl; ++k) {
            a.k = k;
            const int ncol = nbl - k, grid = ncol + (ncol > 1 ? ncol - 1 : 0) + (k > 1 ? k - 1 : 0);
            hipLaunchKernelGGL(vjf_rlsc_col_kernel, dim3(grid), dim3(VJF_RLSC_THREADS), 0, st, P, a);
        }
        gemm(a.X, P.n, 0, a.gbuf, P.dz, a.ybuf, P.dz, P.n, P.dz, P.n, a.ok);                                      // y = X g
        if (before_write) VJF_HIP(hipStreamWaitEvent(st, before_write, 0));
        gemm(a.X, P.n, 1, a.ybuf, P.dz, c->state + P.off[VJF_SLOT_W_MEAN], P.dz, P.n, P.dz, P.n, a.ok);          // W = X^T y
        hipLaunchKernelGGL(vjf_rlsb_final_kernel, dim3(gx), dim3(256), 0, st, P, a);
        VJF_HIP(hipGetLastError());
    }
    else if (before_write) VJF_HIP(hipStreamWaitEvent(st, before_write, 0));
    VjfResidArgs ra{};
    ra.state = c->state; ra.red = red; ra.partial = (double*)(c->ws + c->cv.post + (size_t)nbl * 1024 * 4);
    ra.B_total = B_total; ra.flags = flags;
    if (ta) {
        // state-noise update (model.py:373-377) from the residual itself: R = Phi W with the GEMM kernel into the DEL buffer (free
        // once the gradient sums -- and a replay's -- are formed), then sum (dx - R)^2
        VjfWideGemm g{};
        float* R = resid ? resid : ta->DEL;
        g.A = ta->E; g.lda = P.ldE; g.Bm = c->state + P.off[VJF_SLOT_W_MEAN]; g.ldb = P.dz; g.C = R; g.ldc = P.dz;
        g.M = ta->B; g.N = P.dz; g.K = P.n; g.epi = WEPI_NONE;
        launch_wide_gemm(g, st);
        hipLaunchKernelGGL(vjf_resid_direct_kernel, dim3(VJF_RESID_BLOCKS), dim3(256), 0, st, P, ra, (const float*)ta->E, (const float*)R, ta->B);
    } else {
        // ranks holding shards: the same sum as the quadratic form of the reduced statistics, T = G W with the GEMM kernel (y's
        // buffer is free again), contraction in fp64
        VjfWideGemm g{};
        g.A = red + P.red_G; g.lda = P.n; g.Bm = c->state + P.off[VJF_SLOT_W_MEAN]; g.ldb = P.dz; g.C = a.ybuf; g.ldc = P.dz;
        g.M = P.n; g.N = P.dz; g.K = P.n; g.epi = WEPI_NONE;
        launch_wide_gemm(g, st);
        hipLaunchKernelGGL(vjf_resid_dot_kernel, dim3(VJF_RESID_BLOCKS), dim3(256), 0, st, P, ra, (const float*)a.ybuf);
    }
    hipLaunchKernelGGL(vjf_sigma_kernel, dim3(1), dim3(64), 0, st, P, ra, (const int*)nullptr, ta ? 1 : 0);
    VJF_HIP(hipGetLastError());
    return 0;
}

// After the SGD pass of a one-rank step: the backward half with the seeds of the dropped loss components at zero, the gradient
// sums, and the SGD pass from them -- every launch returns at once unless the first pass found a non-finite component
// (vjf/model.py:138-149; the one-launch route does the same inside its grid).
int launch_replay(vjf_ctx* c, const VjfTrialArgs& a0, int32_t B_total, uint32_t flags, hipStream_t st, int gen = 0) {
    VjfTrialArgs a = a0;
    a.replay = 1;
    a.replay_mask = (const unsigned*)(c->ws + c->cv.flags) + kReplayMaskWord;
    a.replay_rho = (const float*)(c->ws + c->cv.flags) + kReplayRhoWord;
    int rc = launch_trial(c, a, 2, st);
    if (rc) return rc;
    float* red = (float*)(c->ws + c->cv.red);
    if ((rc = launch_gram(c, a.B, c->n_ejobs, c->njobs - c->n_ejobs, 0u, red, st, gen, a.replay_mask))) return rc;
    return launch_prep(c, B_total, nullptr, flags, red, 2, st, nullptr, 0, nullptr, 0, 2);
}

// the serial half of a step; `ta`: the trial-parallel half's arguments when this rank holds ALL trials (then a step with a non-finite
// loss component is replayed as the reference defines it), else null
int filter_global_impl(vjf_ctx* c, int32_t B_total, float* loss4, uint32_t flags, const VjfTrialArgs* ta) {
    const bool replay = ta && (flags & VJF_FLAG_SGD) && (c->fast_chol || c->plan.n > 32 * VJF_CHOL_MAXBLK);
    if (c->fast_chol) {
        const float* red = (const float*)(c->ws + c->cv.red);
        int rc = launch_prep(c, B_total, loss4, flags, red, 0, c->stream, nullptr, 0, nullptr, 0, replay ? 1 : 0);
        if (rc) return rc;
        if (replay && (rc = launch_replay(c, *ta, B_total, flags, c->stream))) return rc;
        return launch_rls(c, B_total, flags, red, c->stream, false, nullptr, false, ta);
    }
    if (c->plan.n > 32 * VJF_CHOL_MAXBLK) {
        // feature counts beyond one CU's LDS: clip + SGD and scalars in the prep kernel, then the RLS update as a sequence of
        // chip-wide launches on the matrix in global memory (vjf_rlsb_kernels.h)
        const float* red = (const float*)(c->ws + c->cv.red);
        int rc = launch_prep(c, B_total, loss4, flags, red, 2, c->stream, nullptr, 0, nullptr, 0, replay ? 1 : 0);
        if (rc) return rc;
        if (replay && (rc = launch_replay(c, *ta, B_total, flags, c->stream))) return rc;
        if (!(flags & VJF_FLAG_UPDATE)) return 0;
        return launch_rlsb(c, B_total, flags, red, c->stream, ta);
    }
    VjfSerialArgs s{};
    s.state = c->state; s.red = (const float*)(c->ws + c->cv.red); s.work = (float*)(c->ws + c->cv.work);
    s.loss4 = loss4; s.B_total = B_total; s.flags = flags;
    s.E = (ta && ta->B == B_total && B_total < c->plan.n) ? ta->E : nullptr;
    hipLaunchKernelGGL(vjf_serial_kernel, dim3(1), dim3(VJF_K2_THREADS), c->lds_k2, c->stream, c->plan, s);
    VJF_HIP(hipGetLastError());
    return 0;
}
// ---- the same sums as ONE collective per step ("one ncclAllReduce(sum, fp32) per step between K1 and K2 on the packed
//      buffer"): the trial-parallel half of step t, one all-reduce of the whole reduce buffer [grad | loss sums | G | Phi^T dx | sums],
//      the serial half -- on the caller's stream, in the one-stream order.  Fewer collectives (one latency of the ring per step
//      instead of two on two chains), no overlap of the RLS chain with the trial chain: which of the two wins at 8 ranks is for the
//      first 8-GPU run to say (bench.py --collectives 1|2).  Same kernels and sums as vjf_filter_local / vjf_filter_global around
//      a caller's all-reduce, bit for bit.
int filter_seq_packed(vjf_ctx* c, int32_t T, int32_t B, const float* y, const float* u, const float* eps, const float* mu0,
                      const float* lv0, float* mu, float* lv, float* loss, uint32_t flags) {
    const VjfPlan& P = c->plan;
    const size_t sy = (size_t)B * P.dy, su = (size_t)B * P.du, sz = (size_t)B * P.dz;
    const int fw = c->fake_world;
    const int Bt = B * c->world * fw;
    float* red = (float*)(c->ws + c->cv.red);
    const float* ms = mu0; const float* ls = lv0;
    c->on_mega = false;
    for (int t = 0; t < T; ++t) {
        const bool fresh = t > 0 && c->fast_chol;
        int rc = launch_local(c, B, y + t * sy, u ? u + t * su : nullptr, ms, ls, eps + (size_t)t * 2 * sz, eps + (size_t)t * 2 * sz + sz,
                              mu + t * sz, lv + t * sz, flags, fresh);
        if (rc) return rc;
        VJF_NCCL(nccl().all_reduce(red, red, (size_t)P.red_len, kNcclFloat, kNcclSum, c->comm_a, c->stream));
        if (fw > 1) hipLaunchKernelGGL(vjf_scale_kernel, dim3(64), dim3(256), 0, c->stream, red, (float)fw, (int)P.red_len);   // (test hook)
        rc = filter_global_impl(c, Bt, loss ? loss + 4 * (size_t)t : nullptr, flags, nullptr);
        if (rc) return rc;
        ms = mu + t * sz; ls = lv + t * sz;
    }
    return 0;
}

}  // namespace

int vjf_filter_global(vjf_ctx* c, int32_t B_total, float* loss4, uint32_t flags) {
    if (!c) return fail(-1, "vjf_filter_global: null context");
    DeviceGuard on_device(c->cfg.device);                   // (every launch below goes to the context's device, whatever is current)
    VJF_CHAOS_REFRESH(c);
    if (B_total < 1) return fail(-20, "vjf_filter_global: B_total=%d", B_total);
    return filter_global_impl(c, B_total, loss4, flags, nullptr);   // (the caller's ranks hold shards: no replay, see vjf_hip.h)
}

namespace {
// Plans whose RLS update is a sequence of launches (n_rbf beyond one compute unit's LDS: BASELINE config E has 1000 features, 33
// column launches a step), single rank, T > 1: the update of step t on a stream of its own beside the trial chain.  Nothing in it
// reads what the backward half of step t or the forward half of step t + 1 writes, and those read none of its results:
//   sa (the caller's stream):  [W, w_chol, sigma of t-1 there] predictive moments, losses, backward half(t) -> gradient sums ->
//                              clip + SGD (+ the replay of a step with a non-finite loss component) -> forward half(t+1)
//   sc:                        [forward half(t) there] G, Phi^T dx
//   sb:                        [G, Phi^T dx there; the update of t-1 done: stream order] P W, P + G/v -> the column launches -> y,
//                              [backward half(t) done: it read the previous W, w_chol, sigma] W, w_chol, w_pchol, P, state-noise update
// Cross-stream order through events only (recorded before the wait that names them, in host order); the rows of E alternate
// between two buffers (the update's residual Phi W reads step t's rows while step t + 1 writes its own), the statistics of the
// two chains have buffers of their own, Phi W of the residual too.  Same kernels, same arithmetic as the one-stream order.
int filter_seq_two(vjf_ctx* c, int32_t T, int32_t B, const float* y, const float* u, const float* eps, const float* mu0,
                   const float* lv0, float* mu, float* lv, float* loss, uint32_t flags) {
    int rc = ensure_stream2(c);
    if (rc) return rc;
    const VjfPlan& P = c->plan;
    const size_t sy = (size_t)B * P.dy, su = (size_t)B * P.du, sz = (size_t)B * P.dz;
    hipStream_t sa = c->stream, sb = c->stream2;
    float* redg = (float*)(c->ws + c->cv.red);
    float* rede[2] = {(float*)(c->ws + c->cv.red2), (float*)(c->ws + c->cv.red3)};
    float* resid = (float*)(c->ws + c->cv.resid);
    auto args = [&](int t) {
        return trial_args(c, B, y + t * sy, u ? u + t * su : nullptr, t ? mu + (t - 1) * sz : mu0, t ? lv + (t - 1) * sz : lv0,
                          eps + (size_t)t * 2 * sz, eps + (size_t)t * 2 * sz + sz, mu + t * sz, lv + t * sz, flags, t & 1);
    };
    rc = check_step_args(c, B, y, u, mu0, lv0, eps, eps + sz, mu, lv);
    if (rc) return rc;
    const int ne = c->n_ejobs, ng = c->njobs - ne;
    const bool replay = (flags & VJF_FLAG_SGD) != 0;
    VJF_HIP(hipEventRecord(c->ev_s, sa));                                  // (sb: behind whatever the caller's stream holds already)
    VJF_HIP(hipStreamWaitEvent(sb, c->ev_s, 0));
    hipStream_t sc = c->stream3;                                            // (its first launch waits for an event of sa behind this point)
    if ((rc = refresh_aux(c, sa))) return rc;
    if ((rc = launch_trial(c, args(0), 1, sa))) return rc;
    VJF_HIP(hipEventRecord(c->ev_f[0], sa));
    // VJF_DEBUG_TWO_TIMELINE=1 (diagnostic): timing events around the phases of every step, printed to stderr behind a synchronisation
    const bool tl = getenv("VJF_DEBUG_TWO_TIMELINE") != nullptr;
    enum { TL_A0, TL_A1, TL_A2, TL_A3, TL_G0, TL_G1, TL_R0, TL_R2, TL_N };
    std::vector<hipEvent_t> tle;
    auto mark = [&](int t, int k, hipStream_t st) -> int {
        if (!tl) return 0;
        VJF_HIP(hipEventRecord(tle[(size_t)t * TL_N + k], st));
        return 0;
    };
    if (tl) {
        tle.resize((size_t)T * TL_N + 1);
        for (auto& e : tle) VJF_HIP(hipEventCreate(&e));
        VJF_HIP(hipEventRecord(tle[(size_t)T * TL_N], sa));
    }
    for (int t = 0; t < T; ++t) {
        const int g = t & 1;
        const VjfTrialArgs ta = args(t);
        // (the statistics on a stream of their own: they need the forward half only, not the previous update, which sb may still be in)
        VJF_HIP(hipStreamWaitEvent(sc, c->ev_f[g], 0));
        if ((rc = mark(t, TL_G0, sc))) return rc;
        if ((rc = launch_gram(c, B, 0, ne, 0u, rede[g], sc, g))) return rc;
        if ((rc = mark(t, TL_G1, sc))) return rc;
        VJF_HIP(hipEventRecord(c->ev_g[g], sc));
        VJF_HIP(hipStreamWaitEvent(sb, c->ev_g[g], 0));
        if (t > 0) VJF_HIP(hipStreamWaitEvent(sa, c->ev_r[g ^ 1], 0));
        if ((rc = mark(t, TL_A0, sa))) return rc;
        if ((rc = launch_trial(c, ta, 2, sa))) return rc;
        if ((rc = mark(t, TL_A1, sa))) return rc;
        if ((rc = launch_gram(c, B, ne, ng, kScAll, redg, sa, g))) return rc;
        if ((rc = launch_prep(c, B, loss ? loss + 4 * (size_t)t : nullptr, flags, redg, 2, sa, nullptr, 0, nullptr, 0, replay ? 1 : 0))) return rc;
        if (replay && (rc = launch_replay(c, ta, B, flags, sa, g))) return rc;
        VJF_HIP(hipEventRecord(c->ev_b[g], sa));
        if ((rc = mark(t, TL_A2, sa))) return rc;
        if (t + 1 < T) {                                                   // (enqueued before the update's ~40 launches: the host must not
            if (c->mfma_trial && (rc = refresh_aux(c, sa))) return rc;    //  hold the trial chain back; this route's SGD pass does not keep
            if ((rc = launch_trial(c, args(t + 1), 1, sa))) return rc;    //  the transposed copies)
            VJF_HIP(hipEventRecord(c->ev_f[g ^ 1], sa));
        }
        if ((rc = mark(t, TL_A3, sa))) return rc;
        if ((rc = mark(t, TL_R0, sb))) return rc;
        if ((rc = launch_rlsb(c, B, flags, rede[g], sb, &ta, c->ev_b[g], resid, true))) return rc;
        if ((rc = mark(t, TL_R2, sb))) return rc;
        VJF_HIP(hipEventRecord(c->ev_r[g], sb));
    }
    VJF_HIP(hipEventRecord(c->ev_c, sb));
    VJF_HIP(hipStreamWaitEvent(sa, c->ev_c, 0));                           // join: the caller's stream sees the final state
    if (tl) {
        VJF_HIP(hipStreamSynchronize(sa));
        static const char* nm[TL_N] = {"sa part2 starts", "sa part2 done", "sa sgd(+replay) done", "sa part1(t+1) done", "sc stats start", "sc stats done",
                                       "sb update starts", "sb update done"};
        for (int t = 0; t < T; ++t)
            for (int k = 0; k < TL_N; ++k) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, tle[(size_t)T * TL_N], tle[(size_t)t * TL_N + k]) == hipSuccess)
                    fprintf(stderr, "two-timeline %10.1f us  [%d] %s\n", ms * 1e3, t, nm[k]);
            }
        for (auto& e : tle) (void)hipEventDestroy(e);
    }
    return 0;
}

// single rank, the step as the reference runs it (model.py:206-216: gradient step and closed-form updates) on a plan the
// one-launch route serves
// the multi-launch RLS plans: their update on a second stream beside the trial chain (filter_seq_two)
bool two_route(const vjf_ctx* c, uint32_t flags) {
    return c->overlap && !c->comm_a && c->world == 1 && !c->fast_chol && c->plan.n > 32 * VJF_CHOL_MAXBLK && !c->stamps &&
           (flags & (VJF_FLAG_UPDATE | VJF_FLAG_WARM_UP)) == VJF_FLAG_UPDATE;
}
bool mega_route(const vjf_ctx* c, uint32_t flags) {
    // (every flag set of VJF.filter: sgd + update is the training step; warm-up and update=False drop the RLS, Gram and operand
    //  roles from the grid; sgd=False the backward pass and the gradient steps)
    const bool rls = (flags & (VJF_FLAG_UPDATE | VJF_FLAG_WARM_UP)) == VJF_FLAG_UPDATE;
    if (rls && !(flags & VJF_FLAG_SGD)) return false;                      // (update without sgd, no warm-up: the per-step kernels)
    if (!rls && c->lite_wg_per_cu < 1) return false;
    return c->mega_ok && c->overlap && !c->comm_a && !c->force_streams && (!c->stamps || c->stamps_keep_overlap);
}
int seq_chunk() {
    // Long sequences go in chunks: the workgroups of one launch stay resident for its whole length, and a compute kernel that
    // stays on the device for a minute is what drivers' lockup timers are for (16384 steps ~ 1 s at config B).
    const char* ce = getenv("VJF_SEQ_CHUNK");                              // (tests)
    return ce && atoi(ce) >= 1 ? atoi(ce) : 16384;
}
}  // namespace

int vjf_filter_step(vjf_ctx* c, int32_t B, const float* y, const float* u, const float* mu_s, const float* lv_s,
                    const float* eps_s, const float* eps_t, float* mu_t, float* lv_t, float* loss4, uint32_t flags) {
    if (!c) return fail(-1, "vjf_filter_step: null context");
    DeviceGuard on_device(c->cfg.device);                   // (every launch below goes to the context's device, whatever is current)
    VJF_CHAOS_REFRESH(c);
    if (int rp = refuse_if_poisoned(c, "vjf_filter_step")) return rp;
    if (mega_route(c, flags) && eps_s && eps_t && eps_t == eps_s + (size_t)B * c->plan.dz) {        // (the sequence layout of eps: (2, B, dz))
        const int rc = filter_seq_mega(c, 1, B, y, u, eps_s, mu_s, lv_s, mu_t, lv_t, loss4, flags);
        if (rc != kMegaRefused) return rc;
    }
    if (mega_route(c, flags)) {
        // the two draws are separate tensors: the sequence entry point wants them adjacent -- stage them in the workspace
        int rc = check_step_args(c, B, y, u, mu_s, lv_s, eps_s, eps_t, mu_t, lv_t);
        if (rc) return rc;
        float* st = (float*)(c->ws + c->cv.DEL);                           // (B, ldD >= 2 dz) floats, unused by this route
        const size_t sz = (size_t)B * c->plan.dz;
        VJF_HIP(hipMemcpyAsync(st, eps_s, sz * 4, hipMemcpyDeviceToDevice, c->stream));
        VJF_HIP(hipMemcpyAsync(st + sz, eps_t, sz * 4, hipMemcpyDeviceToDevice, c->stream));
        rc = filter_seq_mega(c, 1, B, y, u, st, mu_s, lv_s, mu_t, lv_t, loss4, flags);
        if (rc != kMegaRefused) return rc;
    }
    int rc = vjf_filter_local(c, B, y, u, mu_s, lv_s, eps_s, eps_t, mu_t, lv_t, flags);
    if (rc) return rc;
    const VjfTrialArgs ta = trial_args(c, B, y, u, mu_s, lv_s, eps_s, eps_t, mu_t, lv_t, flags);
    return filter_global_impl(c, B, loss4, flags, c->world > 1 ? nullptr : &ta);
}

int vjf_route(vjf_ctx* c, uint32_t flags) {
    if (!c) return fail(-1, "vjf_route: null context");
    if (mega_route(c, flags)) return 1;
    if (c->comm_a && c->collectives == 1) return 4;
    const bool streams = (c->comm_a || c->force_streams) && c->overlap && (flags & VJF_FLAG_UPDATE) && !(flags & VJF_FLAG_WARM_UP) &&
                         c->fast_chol && c->post_kernels && c->mfma_trial && (!c->stamps || c->stamps_keep_overlap);
    if (streams) return 3;
    return two_route(c, flags) ? 2 : 0;
}

int vjf_filter_seq(vjf_ctx* c, int32_t T, int32_t B, const float* y, const float* u, const float* eps, const float* mu0,
                   const float* lv0, float* mu, float* lv, float* loss, uint32_t flags) {
    if (!c) return fail(-1, "vjf_filter_seq: null context");
    DeviceGuard on_device(c->cfg.device);                   // (every launch below goes to the context's device, whatever is current)
    VJF_CHAOS_REFRESH(c);
    if (T < 1) return fail(-23, "vjf_filter_seq: T=%d", T);
    if (int rp = refuse_if_poisoned(c, "vjf_filter_seq")) return rp;
    if (!y || !eps || !mu || !lv) return fail(-1, "vjf_filter_seq: null tensor");
    const size_t sy = (size_t)B * c->plan.dy, su = (size_t)B * c->plan.du, sz = (size_t)B * c->plan.dz;
    const bool streams = (c->comm_a || c->force_streams) && c->overlap && T > 1 && (flags & VJF_FLAG_UPDATE) && !(flags & VJF_FLAG_WARM_UP) &&
                         c->fast_chol && c->post_kernels && c->mfma_trial && (!c->stamps || c->stamps_keep_overlap);
    if (c->comm_a && c->collectives == 1)                                  // (communicators, ONE sum over ranks per step: any flags, any T)
        return filter_seq_packed(c, T, B, y, u, eps, mu0, lv0, mu, lv, loss, flags);
    if (mega_route(c, flags) || streams) {
        const int32_t chunk = seq_chunk();
        for (int32_t t0 = 0; t0 < T; t0 += chunk) {
            int32_t n = T - t0 < chunk ? T - t0 : chunk;
            if (streams && T - t0 - n == 1) n += 1;                        // (no chunk of a single step on the three-stream route)
            auto route = streams ? filter_seq_streams : filter_seq_mega;
            int rc = route(c, n, B, y + t0 * sy, u ? u + t0 * su : nullptr, eps + (size_t)t0 * 2 * sz,
                           t0 ? mu + (size_t)(t0 - 1) * sz : mu0, t0 ? lv + (size_t)(t0 - 1) * sz : lv0,
                           mu + (size_t)t0 * sz, lv + (size_t)t0 * sz, loss ? loss + 4 * (size_t)t0 : nullptr, flags);
            if (rc == kMegaRefused)                                        // (the context has left the one-launch route: the rest per step)
                return vjf_filter_seq(c, T - t0, B, y + t0 * sy, u ? u + t0 * su : nullptr, eps + (size_t)t0 * 2 * sz,
                                      t0 ? mu + (size_t)(t0 - 1) * sz : mu0, t0 ? lv + (size_t)(t0 - 1) * sz : lv0,
                                      mu + (size_t)t0 * sz, lv + (size_t)t0 * sz, loss ? loss + 4 * (size_t)t0 : nullptr, flags);
            if (rc) return rc;
            if (n > chunk) break;
        }
        return 0;
    }
    if (two_route(c, flags) && T > 1)
        return filter_seq_two(c, T, B, y, u, eps, mu0, lv0, mu, lv, loss, flags);
    if (c->world > 1)
        return fail(-24, "vjf_filter_seq: with communicators only the multi-stream schedule exists (update, no warm-up, T > 1, "
                         "fast kernels); use vjf_filter_local / vjf_filter_global around your own all-reduce otherwise");
    const VjfPlan& P = c->plan;
    const float* ms = mu0; const float* ls = lv0;
    VJF_HIP(hipSetDevice(c->cfg.device));
    for (int t = 0; t < T; ++t) {
        // the prep kernel keeps the transposed weight copies current inside a sequence; the generic
        // serial kernel does not, so that path refreshes them every step
        const bool fresh = t > 0 && c->fast_chol;
        int rc = launch_local(c, B, y + t * sy, u ? u + t * su : nullptr, ms, ls, eps + (size_t)t * 2 * sz,
                              eps + (size_t)t * 2 * sz + sz, mu + t * sz, lv + t * sz, flags, fresh);
        if (rc) return rc;
        const VjfTrialArgs ta = trial_args(c, B, y + t * sy, u ? u + t * su : nullptr, ms, ls, eps + (size_t)t * 2 * sz,
                                           eps + (size_t)t * 2 * sz + sz, mu + t * sz, lv + t * sz, flags);
        rc = filter_global_impl(c, B, loss ? loss + 4 * (size_t)t : nullptr, flags, &ta);
        if (rc) return rc;
        ms = mu + t * sz; ls = lv + t * sz;
        (void)P;
    }
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// stand-alone operators
// ------------------------------------------------------------------------------------------------
namespace {
inline dim3 grid1d(size_t n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }
// slot of the loss kernels' partial-sum table for this call (handed out in turn: VJF_LOSS_SLOTS calls may be in flight)
inline int loss_slot() { static std::atomic<unsigned> next{0}; return (int)(next.fetch_add(1u, std::memory_order_relaxed) % VJF_LOSS_SLOTS); }

struct VjfPredArgs {
    const float* x; const float* c; const float* logw; const float* w_mean; const float* w_chol;
    float* mean; float* logvar; int B, n, d, dout;
};
// 16 trials per workgroup: features feature-major in LDS ([feature][17], as the fused kernels hold them), then Phi W (mean)
// and the row norm of Phi w_chol (logvar) as 16 x 16 output tiles on v_mfma_f32_16x16x4_f32 (mma_tile: the matrices are k-major
// for these products), one tile per wavefront and round.
__global__ __launch_bounds__(VJF_K1_THREADS) void vjf_blr_predict_kernel(VjfPredArgs A) {
    constexpr int TB = 16, LD = VJF_LDT, NW = VJF_K1_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_phi = smem;                 // n x LD
    float* s_red = s_phi + A.n * LD;     // NW x TB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b0 = blockIdx.x * TB, nb = min(TB, A.B - b0);
    for (int i = tid; i < TB * A.n; i += VJF_K1_THREADS) {
        const int k = i / TB, b = i - k * TB;
        float ph = 0.f;
        if (b < nb) {
            float d2 = 0.f;
            for (int j = 0; j < A.d; ++j) { const float t = A.x[(size_t)(b0 + b) * A.d + j] - A.c[(size_t)k * A.d + j]; d2 = fmaf(t, t, d2); }
            const float w = expf(A.logw[k]);
            ph = expf(-0.5f * d2 / (w * w));
        }
        s_phi[k * LD + b] = ph;
    }
    __syncthreads();
    const int col = lane & 15, r4 = 4 * (lane >> 4);     // accumulator: row = r4 + r (output), column = trial
    if (A.mean)
        for (int t = wave; t * 16 < A.dout; t += NW) {
            vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            mma_tile(acc, A.w_mean, A.dout, A.dout, t * 16, s_phi, A.n, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = t * 16 + r4 + r;
                if (j < A.dout && col < nb) A.mean[(size_t)(b0 + col) * A.dout + j] = acc[r];
            }
        }
    if (!A.logvar) return;
    float v2 = 0.f;                                      // this lane's share of sum_j Z[trial col][j]^2
    for (int t = wave; t * 16 < A.n; t += NW) {
        vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mma_tile(acc, A.w_chol, A.n, A.n, t * 16, s_phi, A.n, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) v2 = fmaf(acc[r], acc[r], v2);      // (rows beyond n are exact zeros)
    }
    v2 += __shfl_xor(v2, 16, 64);                        // the four row groups of a column, fixed order
    v2 += __shfl_xor(v2, 32, 64);
    if (lane < 16) s_red[wave * TB + lane] = v2;
    __syncthreads();
    for (int i = tid; i < nb * A.dout; i += VJF_K1_THREADS) {
        const int b = i / A.dout;
        float v = s_red[b];
        for (int w = 1; w < NW; ++w) v += s_red[w * TB + b];
        A.logvar[(size_t)(b0 + b) * A.dout + (i - b * A.dout)] = logf(v);
    }
}

struct VjfRecArgs {
    const float* y; const float* u; const float* mu_s; const float* lv_s;
    const float* W[VJF_MAX_HIDDEN]; const float* b[VJF_MAX_HIDDEN];
    const float* mean_W; const float* lv_W; const float* lv_b;
    float* mu_t; float* lv_t;
    int B, dy, du, dz, L; int h[VJF_MAX_HIDDEN];
};
// Recognition.forward for 16 trials per workgroup: activations feature-major in LDS (ping-pong), every layer as 16 x 16 output
// tiles on v_mfma_f32_16x16x4_f32 with the weights read as torch stores them (mma_tile<true>).
__global__ __launch_bounds__(VJF_K1_THREADS) void vjf_recognition_kernel(VjfRecArgs A, int hmax) {
    constexpr int TB = 16, LD = VJF_LDT, NW = VJF_K1_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int din = A.dy + A.du + 2 * A.dz;
    float* s_in = smem;                  // din x LD
    float* s_a = s_in + din * LD;        // hmax x LD
    float* s_b = s_a + hmax * LD;        // hmax x LD
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b0 = blockIdx.x * TB, nb = min(TB, A.B - b0);
    for (int i = tid; i < TB * din; i += VJF_K1_THREADS) {
        const int b = i / din, c = i - b * din;
        float v = 0.f;
        if (b < nb) {
            const size_t g = (size_t)(b0 + b);
            if (c < A.dy) v = A.y[g * A.dy + c];
            else if (c < A.dy + A.du) v = A.u[g * A.du + c - A.dy];
            else if (c < A.dy + A.du + A.dz) v = A.mu_s[g * A.dz + c - A.dy - A.du];
            else v = A.lv_s[g * A.dz + c - A.dy - A.du - A.dz];
        }
        s_in[c * LD + b] = v;
    }
    __syncthreads();
    const int col = lane & 15, r4 = 4 * (lane >> 4);     // accumulator: row = r4 + r (output unit), column = trial
    const float* xin = s_in; int kin = din;
    float* cur = s_a; float* nxt = s_b;
    for (int l = 0; l < A.L; ++l) {
        const int hl = A.h[l];
        const float* bias = A.b[l];
        for (int t = wave; t * 16 < hl; t += NW) {
            vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            mma_tile<true>(acc, A.W[l], kin, hl, t * 16, xin, kin, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = t * 16 + r4 + r;
                if (f < hl) cur[f * LD + col] = tanhf(acc[r] + bias[f]);
            }
        }
        __syncthreads();
        xin = cur; kin = hl;
        float* t = cur; cur = nxt; nxt = t;
    }
    const int nt = (A.dz + 15) / 16;                         // tiles per head; the wavefronts take mean tiles, then log-variance tiles
    for (int t = wave; t < 2 * nt; t += NW) {
        const bool lvh = t >= nt;
        const int f0 = (lvh ? t - nt : t) * 16;
        vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mma_tile<true>(acc, lvh ? A.lv_W : A.mean_W, kin, A.dz, f0, xin, kin, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = f0 + r4 + r;
            if (f < A.dz && col < nb) {
                if (lvh) A.lv_t[(size_t)(b0 + col) * A.dz + f] = acc[r] + A.lv_b[f];
                else A.mu_t[(size_t)(b0 + col) * A.dz + f] = acc[r];
            }
        }
    }
}

// features + target rows of the stand-alone RLS:  E[b] = [Phi(x_b) | target_b | 0]
__global__ void vjf_rls_rows_kernel(const float* x, const float* c, const float* logw, const float* target, float* E,
                                    int B, int n, int d, int dout, int ldE) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * ldE) return;
    const int b = (int)(i / ldE), k = (int)(i - (size_t)b * ldE);
    float v = 0.f;
    if (k < n) {
        float d2 = 0.f;
        for (int j = 0; j < d; ++j) { const float t = x[(size_t)b * d + j] - c[(size_t)k * d + j]; d2 = fmaf(t, t, d2); }
        const float w = expf(logw[k]);
        v = expf(-0.5f * d2 / (w * w));
    } else if (k < n + dout) {
        v = target[(size_t)b * dout + (k - n)];
    }
    E[i] = v;
}

// plan for the stand-alone RLS (only the fields the Gram kernels read for kind-0 jobs)
void rls_plan(int n, int dout, VjfPlan* P) {
    memset(P, 0, sizeof *P);
    P->n = n; P->dz = dout;
    P->ldE = (int)vjf_align(n + dout, VJF_TILE);
    P->red_SCA = 0; P->red_G = 0; P->red_FDX = n * n; P->red_SC = (int)vjf_align((int64_t)n * n + (int64_t)n * dout, 4);
    P->red_len = P->red_SC + RS_N;
}
struct RlsCarve { size_t E, slabs, red, work, jobs, partial, total; int njobs, nsplit; };
RlsCarve rls_carve(int B, int n, int dout, std::vector<VjfJob>* jobs_out) {
    VjfPlan P; rls_plan(n, dout, &P);
    std::vector<VjfJob> jobs; build_jobs(P, jobs);
    // build_jobs also emits gradient jobs from the (zeroed) plan: keep kind 0 only
    std::vector<VjfJob> k0;
    for (auto& j : jobs) if (j.kind == 0) k0.push_back(j);
    RlsCarve c{};
    c.njobs = (int)k0.size(); c.nsplit = split_for(B);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = (o + bytes + 255) / 256 * 256; return at; };
    c.E = take((size_t)B * P.ldE * 4);
    c.slabs = take((size_t)c.njobs * c.nsplit * 1024 * 4);
    c.red = take((size_t)P.red_len * 4);
    VjfPlan Q = P;
    c.work = take(vjf_serial_work_floats(Q) * 4);
    c.jobs = take(k0.size() * sizeof(VjfJob));
    c.partial = take(RS_N * 4);
    c.total = o;
    if (jobs_out) *jobs_out = k0;
    return c;
}
}  // namespace

extern "C" {

int vjf_rbf_forward(const float* x, const float* centroid, const float* logwidth, float* out, int32_t B, int32_t n, int32_t d,
                    void* stream) {
    if (!x || !centroid || !logwidth || !out) return fail(-1, "vjf_rbf_forward: null tensor");
    if (B < 1 || n < 1 || d < 1) return fail(-20, "vjf_rbf_forward: bad shape");
    const size_t lds = (size_t)16 * d * 4;
    if (lds > kMaxLds - 1024) return fail(-11, "vjf_rbf_forward: d=%d too large", d);
    allow_lds(vjf_rbf_kernel, lds);
    hipLaunchKernelGGL(vjf_rbf_kernel, dim3((n + 255) / 256, (B + 15) / 16), dim3(256), lds, (hipStream_t)stream, x, centroid, logwidth, out, B, n, d);
    VJF_HIP(hipGetLastError());
    return 0;
}

int vjf_blr_predict(const float* x, const float* centroid, const float* logwidth, const float* w_mean, const float* w_chol,
                    float* mean, float* logvar, int32_t B, int32_t n, int32_t d, int32_t dout, void* stream) {
    if (!x || !centroid || !logwidth || !w_mean || !w_chol) return fail(-1, "vjf_blr_predict: null tensor");
    if (B < 1 || n < 1 || d < 1 || dout < 1) return fail(-20, "vjf_blr_predict: bad shape");
    const size_t lds = ((size_t)n * VJF_LDT + 64) * 4;
    if (lds > kMaxLds - 1024) return fail(-11, "vjf_blr_predict: n=%d too large", n);
    allow_lds(vjf_blr_predict_kernel, lds);
    VjfPredArgs a{x, centroid, logwidth, w_mean, w_chol, mean, logvar, B, n, d, dout};
    hipLaunchKernelGGL(vjf_blr_predict_kernel, dim3((B + 15) / 16), dim3(VJF_K1_THREADS), lds, (hipStream_t)stream, a);
    VJF_HIP(hipGetLastError());
    return 0;
}

int vjf_blr_sample(const float* x, const float* centroid, const float* logwidth, const float* w_mean, const float* w_chol,
                   const float* noise, float* out, float* w_scratch, int32_t B, int32_t n, int32_t d, int32_t dout, void* stream) {
    if (!x || !centroid || !logwidth || !w_mean || !w_chol || !noise || !out || !w_scratch) return fail(-1, "vjf_blr_sample: null tensor");
    if (B < 1 || n < 1 || d < 1 || dout < 1) return fail(-20, "vjf_blr_sample: bad shape");
    hipStream_t s = (hipStream_t)stream;
    // w = w_mean + w_chol @ noise   (module.py:71)
    {
        VjfWideGemm g{};
        g.A = w_chol; g.lda = n; g.Bm = noise; g.ldb = dout; g.C = w_scratch; g.ldc = dout; g.M = n; g.N = dout; g.K = n;
        g.epi = WEPI_ADD_SRC; g.src = w_mean; g.lds = dout; g.src_scale = 1.f;
        launch_wide_gemm(g, s);
    }
    VJF_HIP(hipGetLastError());
    const size_t lds = ((size_t)n * VJF_LDT + 64) * 4;
    if (lds > kMaxLds - 1024) return fail(-11, "vjf_blr_sample: n=%d too large", n);
    allow_lds(vjf_blr_predict_kernel, lds);
    VjfPredArgs a{x, centroid, logwidth, w_scratch, w_chol, out, nullptr, B, n, d, dout};
    hipLaunchKernelGGL(vjf_blr_predict_kernel, dim3((B + 15) / 16), dim3(VJF_K1_THREADS), lds, s, a);
    VJF_HIP(hipGetLastError());
    return 0;
}

int vjf_rls_scratch_size(int32_t B, int32_t n, int32_t dout, int64_t* bytes) {
    if (!bytes || B < 1 || n < 1 || dout < 1) return fail(-20, "vjf_rls_scratch_size: bad argument");
    *bytes = (int64_t)rls_carve(B, n, dout, nullptr).total;
    return 0;
}

int vjf_blr_rls(const float* x, const float* target, const float* v, float shrink, const float* centroid, const float* logwidth,
                float* w_mean, float* w_chol, float* w_precision, float* w_pchol, void* scratch, uint32_t* status, int32_t B,
                int32_t n, int32_t d, int32_t dout, void* stream) {
    if (!x || !target || !v || !centroid || !logwidth || !w_mean || !w_chol || !w_precision || !w_pchol || !scratch)
        return fail(-1, "vjf_blr_rls: null tensor");
    if (B < 1 || n < 1 || d < 1 || dout < 1) return fail(-20, "vjf_blr_rls: bad shape");
    hipStream_t s = (hipStream_t)stream;
    VjfPlan P; rls_plan(n, dout, &P);
    const size_t lds = vjf_serial_lds_floats(P) * 4;
    if (lds > kMaxLds - 1024) return fail(-11, "vjf_blr_rls: n=%d too large for the single-workgroup RLS kernel", n);
    std::vector<VjfJob> jobs;
    RlsCarve c = rls_carve(B, n, dout, &jobs);
    char* ws = (char*)scratch;
    VJF_HIP(hipMemcpyAsync(ws + c.jobs, jobs.data(), jobs.size() * sizeof(VjfJob), hipMemcpyHostToDevice, s));
    VJF_HIP(hipStreamSynchronize(s));     // host vector goes out of scope
    VJF_HIP(hipMemsetAsync(ws + c.partial, 0, RS_N * 4, s));
    float* E = (float*)(ws + c.E);
    hipLaunchKernelGGL(vjf_rls_rows_kernel, grid1d((size_t)B * P.ldE), dim3(256), 0, s, x, centroid, logwidth, target, E, B, n, d, dout, P.ldE);
    VJF_HIP(hipGetLastError());
    VjfGramArgs g{};
    g.jobs = (const VjfJob*)(ws + c.jobs); g.E = E; g.ACT = E; g.DEL = E; g.slabs = (float*)(ws + c.slabs);
    g.B = B; g.nsplit = c.nsplit; g.rows_per_split = ((B + c.nsplit - 1) / c.nsplit + 7) / 8 * 8;
    hipLaunchKernelGGL(vjf_gram_kernel, dim3(c.njobs * c.nsplit), dim3(VJF_GRAM_THREADS), 0, s, P, g);
    VJF_HIP(hipGetLastError());
    VjfReduceArgs r{};
    r.jobs = g.jobs; r.slabs = g.slabs; r.partial = (const float*)(ws + c.partial); r.red = (float*)(ws + c.red);
    r.njobs = c.njobs; r.nsplit = c.nsplit; r.nblocks_k1 = 1;
    hipLaunchKernelGGL(vjf_gram_reduce_kernel, dim3(c.njobs), dim3(VJF_REDUCE_THREADS), 0, s, P, r);   // sc_mask = 0: no loss sums here
    VJF_HIP(hipGetLastError());
    allow_lds(vjf_rls_kernel, lds);
    VjfRlsArgs a{};
    a.Pm = w_precision; a.Wm = w_mean; a.Wc = w_chol; a.Lm = w_pchol;
    a.G = r.red + P.red_G; a.FDX = r.red + P.red_FDX; a.v = v; a.work = (float*)(ws + c.work); a.status = status;
    a.n = n; a.dout = dout; a.shrink = shrink;
    hipLaunchKernelGGL(vjf_rls_kernel, dim3(1), dim3(VJF_K2_THREADS), lds, s, a);
    VJF_HIP(hipGetLastError());
    return 0;
}

int vjf_kalman_scratch_size(int32_t B, int32_t n, int32_t dout, int64_t* bytes) {
    if (!bytes || B < 1 || n < 1 || dout < 1) return fail(-20, "vjf_kalman_scratch_size: bad argument");
    const size_t nn = ((size_t)n * (n > dout ? n : dout) * 4 + 255) / 256 * 256;
    *bytes = (int64_t)(rls_carve(B, n, dout, nullptr).total + 6 * nn);
    return 0;
}

int vjf_blr_kalman(const float* x, const float* target, const float* v, float diffusion, const float* centroid, const float* logwidth,
                   float* w_mean, float* w_chol, void* scratch, uint32_t* status, int32_t B, int32_t n, int32_t d, int32_t dout,
                   void* stream) {
    if (!x || !target || !v || !centroid || !logwidth || !w_mean || !w_chol || !scratch) return fail(-1, "vjf_blr_kalman: null tensor");
    if (B < 1 || n < 1 || d < 1 || dout < 1) return fail(-20, "vjf_blr_kalman: bad shape");
    if (!(diffusion >= 0.f)) return fail(-25, "vjf_blr_kalman: diffusion needs to be non-negative");   // module.py:127
    hipStream_t s = (hipStream_t)stream;
    VjfPlan P; rls_plan(n, dout, &P);
    const size_t lds = vjf_serial_lds_floats(P) * 4;
    if (lds > kMaxLds - 1024) return fail(-11, "vjf_blr_kalman: n=%d too large for the single-workgroup kernel", n);
    std::vector<VjfJob> jobs;
    RlsCarve c = rls_carve(B, n, dout, &jobs);
    char* ws = (char*)scratch;
    VJF_HIP(hipMemcpyAsync(ws + c.jobs, jobs.data(), jobs.size() * sizeof(VjfJob), hipMemcpyHostToDevice, s));
    VJF_HIP(hipStreamSynchronize(s));     // host vector goes out of scope
    VJF_HIP(hipMemsetAsync(ws + c.partial, 0, RS_N * 4, s));
    float* E = (float*)(ws + c.E);
    hipLaunchKernelGGL(vjf_rls_rows_kernel, grid1d((size_t)B * P.ldE), dim3(256), 0, s, x, centroid, logwidth, target, E, B, n, d, dout, P.ldE);
    VJF_HIP(hipGetLastError());
    VjfGramArgs g{};
    g.jobs = (const VjfJob*)(ws + c.jobs); g.E = E; g.ACT = E; g.DEL = E; g.slabs = (float*)(ws + c.slabs);
    g.B = B; g.nsplit = c.nsplit; g.rows_per_split = ((B + c.nsplit - 1) / c.nsplit + 7) / 8 * 8;
    hipLaunchKernelGGL(vjf_gram_kernel, dim3(c.njobs * c.nsplit), dim3(VJF_GRAM_THREADS), 0, s, P, g);
    VJF_HIP(hipGetLastError());
    VjfReduceArgs r{};
    r.jobs = g.jobs; r.slabs = g.slabs; r.partial = (const float*)(ws + c.partial); r.red = (float*)(ws + c.red);
    r.njobs = c.njobs; r.nsplit = c.nsplit; r.nblocks_k1 = 1;
    hipLaunchKernelGGL(vjf_gram_reduce_kernel, dim3(c.njobs), dim3(VJF_REDUCE_THREADS), 0, s, P, r);
    VJF_HIP(hipGetLastError());
    allow_lds(vjf_kalman_kernel, lds);
    const size_t nn = ((size_t)n * (n > dout ? n : dout) * 4 + 255) / 256 * 256;
    VjfKalmanArgs a{};
    a.Wm = w_mean; a.Wc = w_chol; a.G = r.red + P.red_G; a.Fy = r.red + P.red_FDX; a.v = v;
    for (int q = 0; q < 6; ++q) a.T[q] = (float*)(ws + c.total + (size_t)q * nn);
    a.Dinv = (float*)(ws + c.work);
    a.status = status; a.n = n; a.dout = dout; a.diffusion = diffusion;
    hipLaunchKernelGGL(vjf_kalman_kernel, dim3(1), dim3(VJF_K2_THREADS), lds, s, a);
    VJF_HIP(hipGetLastError());
    return 0;
}

int vjf_recognition_forward(const float* y, const float* u, const float* mu_s, const float* lv_s, const float* const* rec_W,
                            const float* const* rec_b, const float* mean_W, const float* lv_W, const float* lv_b, float* mu_t,
                            float* lv_t, int32_t B, int32_t ydim, int32_t udim, int32_t xdim, int32_t n_hidden,
                            const int32_t* hidden, void* stream) {
    if (!y || !mu_s || !lv_s || !rec_W || !rec_b || !mean_W || !lv_W || !lv_b || !mu_t || !lv_t || !hidden)
        return fail(-1, "vjf_recognition_forward: null tensor");
    if (udim > 0 && !u) return fail(-21, "vjf_recognition_forward: u is required when udim > 0");
    if (n_hidden < 1 || n_hidden > VJF_MAX_HIDDEN) return fail(-3, "vjf_recognition_forward: n_hidden=%d", n_hidden);
    if (B < 1) return fail(-20, "vjf_recognition_forward: bad shape");
    VjfRecArgs a{};
    a.y = y; a.u = u; a.mu_s = mu_s; a.lv_s = lv_s; a.mean_W = mean_W; a.lv_W = lv_W; a.lv_b = lv_b; a.mu_t = mu_t; a.lv_t = lv_t;
    a.B = B; a.dy = ydim; a.du = udim; a.dz = xdim; a.L = n_hidden;
    int hmax = 0;
    for (int l = 0; l < n_hidden; ++l) { a.W[l] = rec_W[l]; a.b[l] = rec_b[l]; a.h[l] = hidden[l]; if (hidden[l] > hmax) hmax = hidden[l]; }
    const size_t lds = (size_t)VJF_LDT * (ydim + udim + 2 * xdim + 2 * hmax) * 4;
    if (lds > kMaxLds - 1024) return fail(-10, "vjf_recognition_forward: layer widths do not fit LDS");
    allow_lds(vjf_recognition_kernel, lds);
    hipLaunchKernelGGL(vjf_recognition_kernel, dim3((B + 15) / 16), dim3(VJF_K1_THREADS), lds, (hipStream_t)stream, a, hmax);
    VJF_HIP(hipGetLastError());
    return 0;
}

int vjf_gaussian_loss(const float* m1, const float* lv1, const float* m2, const float* lv2, const float* logvar, float* out,
                      int32_t B, int32_t d, void* stream) {
    if (!m1 || !m2 || !logvar || !out) return fail(-1, "vjf_gaussian_loss: null tensor");
    if (B < 1 || d < 1) return fail(-20, "vjf_gaussian_loss: bad shape");
    hipLaunchKernelGGL(vjf_loss_kernel, dim3(VJF_LOSS_BLOCKS), dim3(256), 0, (hipStream_t)stream, 0, m1, lv1, m2, lv2, logvar, out, B, d, loss_slot());
    VJF_HIP(hipGetLastError());
    return 0;
}

int vjf_gaussian_entropy(const float* lv, float* out, int32_t B, int32_t d, void* stream) {
    if (!lv || !out) return fail(-1, "vjf_gaussian_entropy: null tensor");
    if (B < 1 || d < 1) return fail(-20, "vjf_gaussian_entropy: bad shape");
    hipLaunchKernelGGL(vjf_loss_kernel, dim3(VJF_LOSS_BLOCKS), dim3(256), 0, (hipStream_t)stream, 1, lv, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, out, B, d, loss_slot());
    VJF_HIP(hipGetLastError());
    return 0;
}

int vjf_poisson_loss(const float* eta, const float* target, float* out, int32_t B, int32_t d, void* stream) {
    if (!eta || !target || !out) return fail(-1, "vjf_poisson_loss: null tensor");
    if (B < 1 || d < 1) return fail(-20, "vjf_poisson_loss: bad shape");
    hipLaunchKernelGGL(vjf_loss_kernel, dim3(VJF_LOSS_BLOCKS), dim3(256), 0, (hipStream_t)stream, 2, eta, (const float*)nullptr, target,
                       (const float*)nullptr, (const float*)nullptr, out, B, d, loss_slot());
    VJF_HIP(hipGetLastError());
    return 0;
}

int vjf_linear_forward(const float* x, const float* W, const float* b, float* out, int32_t B, int32_t din, int32_t dout, void* stream) {
    if (!x || !W || !out) return fail(-1, "vjf_linear_forward: null tensor");
    if (B < 1 || din < 1 || dout < 1) return fail(-20, "vjf_linear_forward: bad shape");
    VjfWideGemm g{};                                       // out = x W^T + b on the matrix cores (vjf_trial_wide.h)
    g.A = x; g.lda = din; g.Bm = W; g.ldb = din; g.nt = 1; g.C = out; g.ldc = dout; g.M = B; g.N = dout; g.K = din;
    g.epi = b ? WEPI_BIAS : WEPI_NONE; g.bias = b;
    launch_wide_gemm(g, (hipStream_t)stream);
    VJF_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
