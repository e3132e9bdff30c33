"""ctypes binding of libvjf_hip.so (include/vjf_hip.h).  No torch types cross this boundary:
only raw device pointers, sizes and flags.  There is NO fallback: if the library is missing or a
call fails, an exception is raised.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# VJF_LIB=chaos loads the diagnostic build of the same sources (-DVJF_CHAOS, vjf_plan.h: workgroups are held at random in front of
# their hand-offs; tools/chaos_handoffs.py) -- same ABI, same kernels otherwise
#  (any other VJF_LIB=name: libvjf_hip_name.so, a hand-built experiment of the same sources)
_variant = os.environ.get("VJF_LIB", "")
LIB_PATH = os.path.join(HERE, f"libvjf_hip_{_variant}.so" if _variant.isalnum() else "libvjf_hip.so")

ABI_VERSION = 1
MAX_HIDDEN = 8
LIK_GAUSSIAN, LIK_POISSON = 0, 1
FLAG_SGD, FLAG_UPDATE, FLAG_WARM_UP, FLAG_EXACT_NONFINITE = 1, 2, 4, 8
STATUS_NONFINITE_RECON, STATUS_NONFINITE_DYN, STATUS_NONFINITE_ENT, STATUS_RLS_FAILED = 1, 2, 4, 8
STATUS_NOT_RESIDENT, STATUS_WAIT_MASK = 0x20000, 0x3ff00

# enum vjf_slot
SLOT_PRIOR_MEAN, SLOT_PRIOR_LOGVAR, SLOT_LIK_LOGVAR, SLOT_TR_LOGVAR, SLOT_CENTROID, SLOT_LOGWIDTH = range(6)
SLOT_REC_W0, SLOT_REC_B0 = 6, 7
SLOT_MEAN_W, SLOT_LV_W, SLOT_LV_B, SLOT_DEC_W, SLOT_DEC_B = 22, 23, 24, 25, 26
SLOT_W_MEAN, SLOT_W_CHOL, SLOT_W_PREC, SLOT_W_PCHOL, SLOT_SCALARS = 27, 28, 29, 30, 31
N_SLOTS = 32
# enum vjf_scalar
SC_N_LIK, SC_N_TR, SC_LR_LIK, SC_LR_DEC, SC_LR_TR, SC_LR_REC, SC_FREEZE_DEC, SC_STATUS, SC_TRI_CLEAN = range(9)
N_SCALARS = 16


class VjfError(RuntimeError):
    """A C-ABI call returned an error code."""


class VjfConfig(C.Structure):
    _fields_ = [("ydim", C.c_int32), ("xdim", C.c_int32), ("udim", C.c_int32), ("n_rbf", C.c_int32),
                ("n_hidden", C.c_int32), ("hidden", C.c_int32 * MAX_HIDDEN), ("likelihood", C.c_int32),
                ("max_batch", C.c_int32), ("device", C.c_int32)]


_P = C.c_void_p
_I = C.c_int32
_U = C.c_uint32
_F = C.c_float

# name -> argtypes; every function returns int except vjf_last_error / vjf_abi_version
SIGNATURES = {
    "vjf_abi_version": [],
    "vjf_state_size": [C.POINTER(VjfConfig), C.POINTER(C.c_int64)],
    "vjf_state_layout": [C.POINTER(VjfConfig), C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "vjf_workspace_size": [C.POINTER(VjfConfig), C.POINTER(C.c_int64)],
    "vjf_ctx_create": [C.POINTER(VjfConfig), _P, _P, C.c_int64, _P, C.POINTER(_P)],
    "vjf_ctx_destroy": [_P],
    "vjf_set_stream": [_P, _P],
    "vjf_get_status": [_P, C.POINTER(_U)],
    "vjf_set_overlap": [_P, _I],
    "vjf_route": [_P, _U],
    "vjf_comm_unique_id": [_P],
    "vjf_comm_init": [_P, _P, _I, _I],
    "vjf_comm_ranks": [_P, C.POINTER(_I)],
    "vjf_set_collectives": [_P, _I],
    "vjf_debug_stamps": [_P, _I, C.POINTER(C.c_uint64)],
    "vjf_filter_step": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U],
    "vjf_filter_local": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _U],
    "vjf_reduce_buffer": [_P, C.POINTER(_P), C.POINTER(C.c_int64)],
    "vjf_filter_global": [_P, _I, _P, _U],
    "vjf_filter_seq": [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _U],
    "vjf_rbf_forward": [_P, _P, _P, _P, _I, _I, _I, _P],
    "vjf_blr_predict": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vjf_blr_sample": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vjf_rls_scratch_size": [_I, _I, _I, C.POINTER(C.c_int64)],
    "vjf_blr_rls": [_P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vjf_kalman_scratch_size": [_I, _I, _I, C.POINTER(C.c_int64)],
    "vjf_blr_kalman": [_P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vjf_recognition_forward": [_P, _P, _P, _P, C.POINTER(_P), C.POINTER(_P), _P, _P, _P, _P, _P, _I, _I, _I, _I, _I,
                                C.POINTER(_I), _P],
    "vjf_gaussian_loss": [_P, _P, _P, _P, _P, _P, _I, _I, _P],
    "vjf_gaussian_entropy": [_P, _P, _I, _I, _P],
    "vjf_poisson_loss": [_P, _P, _P, _I, _I, _P],
    "vjf_linear_forward": [_P, _P, _P, _P, _I, _I, _I, _P],
}

_lib = None


def lib():
    """Load libvjf_hip.so once.  Raises ImportError with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: the HIP library is not built. Run `python -m vjf_amd._build` "
                          "(needs hipcc, ROCm >= 7.0). vjf_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError here = header / library mismatch
        fn.argtypes = args
        fn.restype = C.c_int
    L.vjf_last_error.argtypes = []
    L.vjf_last_error.restype = C.c_char_p
    if L.vjf_abi_version() != ABI_VERSION:
        raise ImportError(f"libvjf_hip.so ABI {L.vjf_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = lib().vjf_last_error().decode("utf-8", "replace")
        raise VjfError(f"{what or 'vjf call'} failed ({rc}): {msg}")


def make_config(ydim, xdim, udim, n_rbf, hidden, likelihood, max_batch, device=0):
    hidden = [int(h) for h in hidden]
    if not 1 <= len(hidden) <= MAX_HIDDEN:
        raise ValueError(f"hidden_sizes must have 1..{MAX_HIDDEN} entries")
    cfg = VjfConfig()
    cfg.ydim, cfg.xdim, cfg.udim, cfg.n_rbf = int(ydim), int(xdim), int(udim), int(n_rbf)
    cfg.n_hidden = len(hidden)
    for i in range(MAX_HIDDEN):
        cfg.hidden[i] = hidden[i] if i < len(hidden) else 0
    cfg.likelihood = int(likelihood)
    cfg.max_batch = int(max_batch)
    cfg.device = int(device)
    return cfg


def state_layout(cfg):
    """-> (n_floats, offsets[N_SLOTS], sizes[N_SLOTS])"""
    L = lib()
    n = C.c_int64()
    check(L.vjf_state_size(C.byref(cfg), C.byref(n)), "vjf_state_size")
    off = (C.c_int64 * N_SLOTS)()
    siz = (C.c_int64 * N_SLOTS)()
    check(L.vjf_state_layout(C.byref(cfg), off, siz), "vjf_state_layout")
    return n.value, list(off), list(siz)


def workspace_size(cfg):
    b = C.c_int64()
    check(lib().vjf_workspace_size(C.byref(cfg), C.byref(b)), "vjf_workspace_size")
    return b.value


def ptr(t):
    """Device pointer of a contiguous fp32 torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())
