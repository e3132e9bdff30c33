/*
 * vjf_hip.h -- C ABI of the MI355X (gfx950) implementation of VJF's online filtering step.
 *
 * This is the drop-in boundary for ONE hot path of catniplab/vjf: `VJF.filter`
 * (vjf/model.py:179-221) batched over independent trials, plus the stand-alone operators it is
 * built from.  The reference has no FFI (it is pure Python on torch-CPU); the entry points below
 * are what a binding for that path calls -- the Python side (vjf_amd/_native.py, ctypes) is the
 * reference-shaped host code, see INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (never throws); vjf_last_error() gives text;
 *   - all tensor pointers are DEVICE pointers to row-major contiguous fp32 unless stated;
 *   - the library never allocates device memory: the caller owns `state` and `workspace`
 *     (sizes from vjf_state_size / vjf_workspace_size) and may alias `state` as tensors;
 *   - calls are asynchronous on the context's HIP stream and ordered on it; one context is not
 *     re-entrant, several contexts may coexist (SURVEY.md 8b "Threading"): the launches of the one-launch route -- a grid that
 *     must be resident as a whole -- are chained across the contexts of a process (each waits for the completion of the
 *     previous one, whichever context and stream it came from), and a grid that shares the device with ANOTHER process's
 *     ends within a quarter of a second with VJF_STATUS_NOT_RESIDENT, the state untouched;
 *   - besides the caller's `state` and `workspace` the library holds one page of pinned HOST memory per process and device:
 *     the words through which a launch that has given up a wait tells the host (read at the start of the next call).
 */
#ifndef VJF_HIP_H
#define VJF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VJF_ABI_VERSION 1
#define VJF_MAX_HIDDEN 8

/* likelihood kinds (vjf/likelihood.py:9, :43) */
#define VJF_LIK_GAUSSIAN 0
#define VJF_LIK_POISSON 1

/* per-call flags of VJF.filter (vjf/model.py:180) */
#define VJF_FLAG_SGD 1u      /* sgd=True    : backward, clip to +-1, SGD step (model.py:206-214) */
#define VJF_FLAG_UPDATE 2u   /* update=True : closed-form updates (model.py:215-216, 156-177)   */
#define VJF_FLAG_WARM_UP 4u  /* warm_up=True: no dynamics term / no RLS (model.py:148, 370)     */
#define VJF_FLAG_EXACT_NONFINITE 8u   /* trials sharded over ranks (communicators in the context): a step whose loss has a non-finite
                                       * component is REPLAYED as on one rank (model.py:138-149: the component becomes the constant 0,
                                       * the gradient is that of the others) -- at the price of a second sum over ranks in EVERY step
                                       * (its launches return at once on ordinary steps, the collective itself cannot be skipped:
                                       * every rank must enter it).  Without the flag such a step's SGD update is skipped and
                                       * VJF_STATUS_NONFINITE_* raised.  No effect on one rank (always replayed there). */

/* sticky status bits, read with vjf_get_status */
#define VJF_STATUS_NONFINITE_RECON 1u   /* l_recon non-finite -> replaced by 0 (model.py:138-139) */
#define VJF_STATUS_NONFINITE_DYN 2u     /* l_dynamics non-finite (model.py:141-142)               */
#define VJF_STATUS_NONFINITE_ENT 4u     /* entropy non-finite (model.py:144-145)                  */
#define VJF_STATUS_RLS_FAILED 8u        /* Cholesky pivot <= 0 in rls: RLS state left unchanged   */
/* detail bits beside VJF_STATUS_RLS_FAILED: which bounded in-kernel wait of vjf_filter_seq / vjf_filter_step ran out (a role or
   kernel waited for another one that did not deliver in time; results of the call are then not to be used) */
#define VJF_STATUS_WAIT_RESIDENT 0x100u  /* SGD role / kernel: the trial role's late slabs                         */
#define VJF_STATUS_WAIT_OPERAND 0x200u   /* operand role / kernel: its inputs (early slabs, statistics, previous RLS update) */
#define VJF_STATUS_WAIT_STATS 0x400u     /* Cholesky loop: statistics reduced                                      */
#define VJF_STATUS_WAIT_SIGMA 0x800u     /* Cholesky loop: previous step's sigma / RLS update                      */
#define VJF_STATUS_WAIT_GATE 0x1000u     /* trial role: parameters of the previous step; a gate kernel of the per-step route */
#define VJF_STATUS_WAIT_GATE2 0x2000u    /* Gram role: posterior, previous slab sum                                */
#define VJF_STATUS_WAIT_G 0x4000u        /* y / W loop: operands (g)                                               */
#define VJF_STATUS_WAIT_COLUMN 0x8000u   /* y / W and inverse loops: a column of L, or the trial role's readers    */
#define VJF_STATUS_WAIT_K1 0x10000u      /* trial role / kernel: previous step's RLS update                        */
#define VJF_STATUS_NOT_RESIDENT 0x20000u /* one-launch route: not every workgroup of the grid was placed on the device in time (another
                                          * process's kernels hold compute units): the launch ended before it touched the state; the
                                          * context leaves the one-launch route */
#define VJF_STATUS_WAIT_MASK 0x3ff00u    /* any of the above: the outputs of the call that raised it are not to be used.  The NEXT
                                          * vjf_filter_* call of the context fails (it would chain invalid results) until
                                          * vjf_get_status has been called. */

/* Slots of the state blob, in the reference's state_dict order followed by the plain-attribute
 * RLS tensors and the scalars the reference keeps as Python numbers. */
enum vjf_slot {
    VJF_SLOT_PRIOR_MEAN = 0,   /* (xdim)            VJF.mean            model.py:66  */
    VJF_SLOT_PRIOR_LOGVAR = 1, /* (xdim)            VJF.logvar          model.py:67  */
    VJF_SLOT_LIK_LOGVAR = 2,   /* (1) Gaussian only likelihood.logvar   likelihood.py:16 */
    VJF_SLOT_TR_LOGVAR = 3,    /* (1)               transition.logvar   model.py:331 */
    VJF_SLOT_CENTROID = 4,     /* (n_rbf, xdim+udim) feature.centroid   module.py:20 */
    VJF_SLOT_LOGWIDTH = 5,     /* (n_rbf)           feature.logwidth    module.py:21 */
    VJF_SLOT_REC_W0 = 6,       /* +2k: (h_k, h_{k-1}) recognition.mlp.{2k}.weight  recognition.py:20-26 */
    VJF_SLOT_REC_B0 = 7,       /* +2k: (h_k)          recognition.mlp.{2k}.bias    */
    VJF_SLOT_MEAN_W = 22,      /* (xdim, h_L)  recognition.mean.weight    recognition.py:27 */
    VJF_SLOT_LV_W = 23,        /* (xdim, h_L)  recognition.logvar.weight  recognition.py:28 */
    VJF_SLOT_LV_B = 24,        /* (xdim)       recognition.logvar.bias    */
    VJF_SLOT_DEC_W = 25,       /* (ydim, xdim) decoder.decode.weight      model.py:24 */
    VJF_SLOT_DEC_B = 26,       /* (ydim)       decoder.decode.bias        */
    VJF_SLOT_W_MEAN = 27,      /* (n_rbf, xdim)  velocity.w_mean       module.py:50  */
    VJF_SLOT_W_CHOL = 28,      /* (n_rbf, n_rbf) velocity.w_chol       module.py:52  (upper: inv(L^T)) */
    VJF_SLOT_W_PREC = 29,      /* (n_rbf, n_rbf) velocity.w_precision  module.py:53  */
    VJF_SLOT_W_PCHOL = 30,     /* (n_rbf, n_rbf) velocity.w_pchol      module.py:54  (lower L) */
    VJF_SLOT_SCALARS = 31,     /* (16) see vjf_scalar */
    VJF_N_SLOTS = 32
};

/* indices into the SCALARS slot (all stored as fp32; the counters are exact integers < 2^24) */
enum vjf_scalar {
    VJF_SC_N_LIK = 0,   /* likelihood.n_sample  likelihood.py:17 */
    VJF_SC_N_TR = 1,    /* transition.n_sample  model.py:332     */
    VJF_SC_LR_LIK = 2,  /* optimizer.param_groups[0..3]['lr']  model.py:69-77 */
    VJF_SC_LR_DEC = 3,
    VJF_SC_LR_TR = 4,
    VJF_SC_LR_REC = 5,
    VJF_SC_FREEZE_DEC = 6, /* 1.0 after decoder.requires_grad_(False)  model.py:283 */
    VJF_SC_STATUS = 7,     /* status bits as an integer-valued float */
    VJF_SC_TRI_CLEAN = 8,  /* device-internal: 1 once the zero halves of w_chol / w_pchol have been cleared;
                              write 0 after storing a dense matrix into either tensor from the host */
    VJF_N_SCALARS = 16
};

typedef struct vjf_config {
    int32_t ydim, xdim, udim, n_rbf;
    int32_t n_hidden;                /* 1..VJF_MAX_HIDDEN */
    int32_t hidden[VJF_MAX_HIDDEN];  /* hidden_sizes of Recognition (recognition.py:17) */
    int32_t likelihood;              /* VJF_LIK_* */
    int32_t max_batch;               /* largest B (trials per call on this device) the workspace serves */
    int32_t device;                  /* HIP device ordinal */
} vjf_config;

typedef struct vjf_ctx vjf_ctx;

int vjf_abi_version(void);
/* Text of the last error raised on the calling thread ("" if none). */
const char* vjf_last_error(void);

/* ---- memory plan ------------------------------------------------------------------------- */
/* Number of fp32 elements of the state blob for `cfg`. */
int vjf_state_size(const vjf_config* cfg, int64_t* n_floats);
/* offsets[slot], sizes[slot] (fp32 elements) for the VJF_N_SLOTS slots; unused slots have size 0. */
int vjf_state_layout(const vjf_config* cfg, int64_t* offsets, int64_t* sizes);
/* Bytes of scratch the context needs for cfg->max_batch trials. */
int vjf_workspace_size(const vjf_config* cfg, int64_t* bytes);

/* ---- context ------------------------------------------------------------------------------ */
/* `state` (fp32, vjf_state_size floats) and `workspace` stay owned by the caller and must outlive
 * the context.  `stream` is a hipStream_t (NULL = the null stream). */
int vjf_ctx_create(const vjf_config* cfg, float* state, void* workspace, int64_t workspace_bytes,
                   void* stream, vjf_ctx** out);
int vjf_ctx_destroy(vjf_ctx* ctx);
int vjf_set_stream(vjf_ctx* ctx, void* stream);
/* Synchronises the stream, returns and clears the sticky status bits. */
int vjf_get_status(vjf_ctx* ctx, uint32_t* status);

/* Which schedule vjf_filter_seq / vjf_filter_step use on a single rank (results agree to summation order; the one-launch and
 * three-stream routes are bit-identical to each other):
 *   1 (default)  the one-launch route: ONE launch carries the whole call; the trial, Gram, operand, SGD and RLS roles are
 *                workgroups of one grid, resident as a whole (checked against the occupancy query before it is launched), that
 *                hand over through counters in memory (plans it serves: see vjf_route);
 *                plans whose RLS update is a sequence of launches (n_rbf > 224) run that update on a second internal stream,
 *                beside the backward half of its step and the forward half of the next (bit-identical to the one-stream order);
 *   3            the per-step route on three internal streams (the route the RCCL path uses), for A/B measurements;
 *   0            the per-step kernels in the one-stream order (also what tools that serialise kernels need).
 * Returns the resulting setting, or a negative error code. */
int vjf_set_overlap(vjf_ctx* ctx, int enable);

/* The route a vjf_filter_seq call with these flags would take now: 1 one-launch, 3 three-stream per-step (with communicators:
 * the RCCL route), 2 per-step with the multi-launch RLS update on a second stream (T > 1), 4 per-step with ONE all-reduce per
 * step (vjf_set_collectives), 0 one-stream per-step.
 * Negative on error. */
int vjf_route(vjf_ctx* ctx, uint32_t flags);

/* Trials sharded over ranks, one process per GPU: with communicators attached, vjf_filter_seq sums the RLS statistics and
 * the gradients over ranks itself (two RCCL all-reduces per step, one on each chain of its schedule) and B in its
 * arguments is the LOCAL batch.  RCCL is resolved at run time from the process (torch's copy) or librccl.so.
 * vjf_comm_unique_id: rank 0 fills 256 bytes (two ncclUniqueId) that the caller broadcasts to every rank;
 * vjf_comm_init: every rank, collectively.  Replaces the caller-side all-reduce between vjf_filter_local and
 * vjf_filter_global (vjf/model.py has no multi-GPU path: SURVEY 8e). */
int vjf_comm_unique_id(void* ids256);
int vjf_comm_init(vjf_ctx* ctx, const void* ids256, int32_t rank, int32_t world);
/* How many sums over ranks a step of vjf_filter_seq makes on the in-library route: 2 (default) -- [gradients | loss sums] on the
 * trial chain and [Phi^T Phi | Phi^T dx | sum |dx|^2] on the statistics chain of the three-stream schedule, overlapping each other
 * and the other chain's kernels -- or 1: the whole reduce buffer in ONE all-reduce between the trial-parallel and the serial half
 * of the step (SURVEY.md 8e's packed layout; any flags, one stream, no overlap of the two chains).  Same results to summation
 * order.  Also: VJF_COLLECTIVES=1 in the environment when the context is created. */
int vjf_set_collectives(vjf_ctx* ctx, int32_t per_step);
/* What RCCL itself says about the context's two communicators: ranks[0], ranks[1] = ncclCommCount of the gradient chain's and of
 * the statistics chain's communicator (0, 0 without communicators).  A benchmark line quotes these, not the launcher's
 * environment. */
int vjf_comm_ranks(vjf_ctx* ctx, int32_t* ranks2);

/* Diagnostic: enable/disable s_memtime phase stamps in the serial kernel and (out32 != NULL) copy the
 * 32 stamp words of the last step to the host.  Not part of the reference surface. */
int vjf_debug_stamps(vjf_ctx* ctx, int enable, uint64_t* out32);

/* ---- the hot path: VJF.filter (vjf/model.py:179-221) ---------------------------------------- */
/* One filtering step on B trials.
 *   y (B,ydim); u (B,udim) or NULL; mu_s, lv_s (B,xdim) previous posterior, both NULL => prior
 *   (model.py:107-108); eps_s, eps_t (B,xdim) the two reparametrisation draws in the reference's
 *   order xs then xt (util.py:11-13); outputs mu_t, lv_t (B,xdim) = qt, loss4 (4) =
 *   {loss, -l_recon, -l_dynamics, entropy} (model.py:152).  Equivalent to
 *   vjf_filter_local + vjf_filter_global with B_total = B. */
int vjf_filter_step(vjf_ctx* ctx, int32_t B, const float* y, const float* u, const float* mu_s,
                    const float* lv_s, const float* eps_s, const float* eps_t, float* mu_t, float* lv_t,
                    float* loss4, uint32_t flags);

/* Trial-parallel half of a step: everything per trial plus this device's partial sums, written
 * to the reduce buffer (gradient sums, RLS statistics Phi^T Phi, Phi^T dx, loss sums).  With
 * trials sharded over devices the caller all-reduces (sum) that buffer between the two halves. */
int vjf_filter_local(vjf_ctx* ctx, int32_t B, const float* y, const float* u, const float* mu_s,
                     const float* lv_s, const float* eps_s, const float* eps_t, float* mu_t, float* lv_t,
                     uint32_t flags);
/* Device pointer / length (fp32 elements) of the reduce buffer (lives inside `workspace`). */
int vjf_reduce_buffer(vjf_ctx* ctx, float** ptr, int64_t* n_floats);
/* Serial half: finite guards, clip, SGD, likelihood running variance, RLS (Cholesky, solve,
 * triangular inverse), state-noise running variance, from the (all-reduced) buffer.
 * B_total = trials summed over all devices.  (A step with a non-finite loss component: this half alone cannot re-form the
 * gradient without the dropped component -- the SGD step is skipped, the status bit raised; vjf_filter_step / vjf_filter_seq
 * on one rank replay the backward half, see vjf_filter_seq.) */
int vjf_filter_global(vjf_ctx* ctx, int32_t B_total, float* loss4, uint32_t flags);

/* T successive steps, each fed the previous posterior: the inner loop of VJF.fit
 * (model.py:252-261).  y (T,B,ydim); u (T,B,udim) or NULL; eps (T,2,B,xdim);
 * mu0/lv0 (B,xdim) or NULL => prior; outputs mu, lv (T,B,xdim), loss (T,4).
 * On a single rank with sgd + update and no warm-up (plans the one-launch route serves: see vjf_route) the whole call -- per
 * chunk of 16384 steps -- is ONE kernel launch (a grid resident as a whole); the call stays asynchronous.  With communicators (vjf_comm_init)
 * the steps run as per-step kernels on three internal streams with the sums over ranks done by RCCL; otherwise step by step
 * on the caller's stream.  Every in-kernel wait is bounded: one that runs out raises VJF_STATUS_RLS_FAILED plus a
 * VJF_STATUS_WAIT_* detail bit (vjf_get_status), ends the other waits of the call at once, and the outputs of the call are
 * not to be used.  A loss component that is not finite (VJF_STATUS_NONFINITE_*) is handled as vjf/model.py:138-149 does
 * (the component becomes 0 and the step's gradient is that of the others) on the one-launch route and on the one-stream
 * per-step route of a single rank (there the backward half, the gradient sums and the SGD pass are launched again behind the
 * first SGD pass; they return at once on ordinary steps).  Where trials are sharded over ranks (communicators, or
 * vjf_filter_local / vjf_filter_global around the caller's all-reduce: a replay would need a second sum over ranks), on the
 * three-stream route of one rank and on plans served by the generic single-workgroup serial kernel the SGD step of such a
 * step is skipped. */
int vjf_filter_seq(vjf_ctx* ctx, int32_t T, int32_t B, const float* y, const float* u, const float* eps,
                   const float* mu0, const float* lv0, float* mu, float* lv, float* loss, uint32_t flags);

/* ---- stand-alone operators (the reference's vjf.module / vjf.functional surface) ------------ */
/* functional.rbf (vjf/functional.py:11-22): out(B,n) = exp(-1/2 |x-c|^2 / exp(logw)^2). */
int vjf_rbf_forward(const float* x, const float* centroid, const float* logwidth, float* out,
                    int32_t B, int32_t n, int32_t d, void* stream);
/* LinearRegression.forward(sampling=False) (vjf/module.py:64-77): mean(B,dout), logvar(B,dout). */
int vjf_blr_predict(const float* x, const float* centroid, const float* logwidth, const float* w_mean,
                    const float* w_chol, float* mean, float* logvar, int32_t B, int32_t n, int32_t d,
                    int32_t dout, void* stream);
/* LinearRegression.forward(sampling=True) (vjf/module.py:70-73): out = feat @ (w_mean + w_chol @ noise). */
int vjf_blr_sample(const float* x, const float* centroid, const float* logwidth, const float* w_mean,
                   const float* w_chol, const float* noise, float* out, float* w_scratch, int32_t B,
                   int32_t n, int32_t d, int32_t dout, void* stream);
/* LinearRegression.rls (vjf/module.py:79-112), in place on w_mean/w_chol/w_precision/w_pchol.
 * v: device scalar.  scratch: >= vjf_rls_scratch_size(B,n,dout) bytes.  status: device uint32
 * (0 ok, VJF_STATUS_RLS_FAILED when the state was left unchanged). */
int vjf_rls_scratch_size(int32_t B, int32_t n, int32_t dout, int64_t* bytes);
int vjf_blr_rls(const float* x, const float* target, const float* v, float shrink, const float* centroid,
                const float* logwidth, float* w_mean, float* w_chol, float* w_precision, float* w_pchol,
                void* scratch, uint32_t* status, int32_t B, int32_t n, int32_t d, int32_t dout, void* stream);

/* LinearRegression.kalman (vjf/module.py:114-142 over vjf/kalman.py:15-50, 102-145), A = I, Q = diffusion I, R = v I, in
 * n x n algebra on the Gram statistics: the reference's (samples x samples) innovation covariance is never formed (its
 * update -- S^-1 enters the gain twice, kalman.py:134-135 -- is reproduced as it is).  w_mean (n, dout) and w_chol (n, n) in
 * place; w_chol leaves as the lower Cholesky factor of the covariance.  status[0] (device, may
 * be NULL): 0 or VJF_STATUS_RLS_FAILED (state untouched). */
int vjf_kalman_scratch_size(int32_t B, int32_t n, int32_t dout, int64_t* bytes);
int vjf_blr_kalman(const float* x, const float* target, const float* v, float diffusion, const float* centroid,
                   const float* logwidth, float* w_mean, float* w_chol, void* scratch, uint32_t* status, int32_t B,
                   int32_t n, int32_t d, int32_t dout, void* stream);
/* Recognition.forward (vjf/recognition.py:31-42). Wb: pointers to layer weights/biases on device. */
int vjf_recognition_forward(const float* y, const float* u, const float* mu_s, const float* lv_s,
                            const float* const* rec_W, const float* const* rec_b, const float* mean_W,
                            const float* lv_W, const float* lv_b, float* mu_t, float* lv_t, int32_t B,
                            int32_t ydim, int32_t udim, int32_t xdim, int32_t n_hidden,
                            const int32_t* hidden, void* stream);
/* functional.gaussian_loss (vjf/functional.py:32-75): lv1/lv2 NULL for point arguments;
 * logvar: device scalar; out: device scalar. */
int vjf_gaussian_loss(const float* m1, const float* lv1, const float* m2, const float* lv2,
                      const float* logvar, float* out, int32_t B, int32_t d, void* stream);
/* functional.gaussian_entropy (vjf/functional.py:25-29). */
int vjf_gaussian_entropy(const float* lv, float* out, int32_t B, int32_t d, void* stream);
/* PoissonLikelihood.loss (vjf/likelihood.py:51-62). */
int vjf_poisson_loss(const float* eta, const float* target, float* out, int32_t B, int32_t d, void* stream);
/* LinearDecoder.forward, Tensor branch (vjf/model.py:28-30): out(B,ydim) = x @ W^T + b. */
int vjf_linear_forward(const float* x, const float* W, const float* b, float* out, int32_t B, int32_t din,
                       int32_t dout, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VJF_HIP_H */
