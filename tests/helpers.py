"""GPU-test helpers: move state between a vjf_amd.VJF model (device blob) and the oracle / fixtures."""
import numpy as np
import torch

from oracle import vjf_oracle as orc
from tests import goldenio as gio
from tests.margins import check_close


def _set(t, a):
    t.copy_(torch.as_tensor(np.asarray(a, np.float32)).reshape(t.shape).to(t.device))


def model_arrays(model):
    """{fixture key: device tensor view} for a vjf_amd.VJF model."""
    vel = model.transition.velocity
    d = {"prior_mean": model.mean, "prior_logvar": model.logvar, "tr_logvar": model.transition.logvar,
         "centroid": vel.feature.centroid, "logwidth": vel.feature.logwidth, "w_mean": vel.w_mean, "w_chol": vel.w_chol,
         "w_precision": vel.w_precision, "w_pchol": vel.w_pchol, "mean_W": model.recognition.mean.weight,
         "lv_W": model.recognition.logvar.weight, "lv_b": model.recognition.logvar.bias,
         "dec_W": model.decoder.decode.weight, "dec_b": model.decoder.decode.bias}
    if hasattr(model.likelihood, "logvar"):
        d["lik_logvar"] = model.likelihood.logvar
    for k, lin in enumerate(model.recognition.linears()):
        d[f"rec_W{k}"], d[f"rec_b{k}"] = lin.weight, lin.bias
    return d


def load_fixture_state(model, z, prefix):
    """Copy the `prefix.*` arrays of a golden fixture into the model's device blob."""
    for k, t in model_arrays(model).items():
        _set(t, z[f"{prefix}.{k}"])
    if f"{prefix}.n_lik" in z.files:
        model.likelihood.n_sample = int(z[f"{prefix}.n_lik"])
    model.transition.n_sample = int(z[f"{prefix}.n_tr"])
    for g, lr in zip(model.optimizer.param_groups, z[f"{prefix}.lr"]):
        g["lr"] = float(lr)


def load_oracle_state(model, dtype=np.float64):
    """OracleState holding a copy of the model's current device state."""
    lik = orc.GAUSSIAN if hasattr(model.likelihood, "logvar") else orc.POISSON
    s = orc.OracleState(model.ydim, model.xdim, model.udim, model.n_rbf, tuple(model.hidden_sizes), lik)
    a = {k: v.detach().cpu().numpy().astype(dtype) for k, v in model_arrays(model).items()}
    s.prior_mean, s.prior_logvar = a["prior_mean"], a["prior_logvar"]
    s.lik_logvar = a.get("lik_logvar")
    s.n_lik = model.likelihood.n_sample if lik == orc.GAUSSIAN else 0
    s.tr_logvar, s.n_tr = a["tr_logvar"], model.transition.n_sample
    s.centroid, s.logwidth = a["centroid"], a["logwidth"]
    s.w_mean, s.w_chol, s.w_precision, s.w_pchol = a["w_mean"], a["w_chol"], a["w_precision"], a["w_pchol"]
    L = len(model.hidden_sizes)
    s.rec_W = [a[f"rec_W{k}"] for k in range(L)]
    s.rec_b = [a[f"rec_b{k}"] for k in range(L)]
    s.mean_W, s.lv_W, s.lv_b, s.dec_W, s.dec_b = a["mean_W"], a["lv_W"], a["lv_b"], a["dec_W"], a["dec_b"]
    s.lr = [float(g["lr"]) for g in model.optimizer.param_groups]
    s.freeze_decoder = bool(model._scalars[6].item())
    return s


LOOSE = ("w_mean", "w_chol", "w_pchol", "w_precision")


def state_close(model, ref, *, rtol, atol, rls_rtol=None, rls_atol=None, prefix=None):
    """Compare the model's device state with an OracleState (ref) or fixture arrays (ref=z, prefix)."""
    import inspect
    import os
    fr = inspect.stack()[1]
    site = f"{os.path.basename(fr.filename)}:{fr.lineno}"
    got = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in model_arrays(model).items()}
    if prefix is None:
        want = gio.state_arrays(ref)
        want["prior_mean"], want["prior_logvar"] = ref.prior_mean, ref.prior_logvar
    else:
        want = {k: ref[f"{prefix}.{k}"] for k in got}
    for k in got:
        if k not in want or want[k] is None:
            continue
        rt = (rls_rtol or rtol) if k in LOOSE else rtol
        at = (rls_atol or atol) if k in LOOSE else atol
        check_close(got[k], np.asarray(want[k], np.float64).reshape(got[k].shape), rtol=rt, atol=at, err_msg=f"state tensor {k}",
                    what=f"{site} state tensor {k}" + (" [rls]" if k in LOOSE else ""))
