"""Parity margins: every tolerance comparison of the GPU suite also records what was ACHIEVED -- the largest absolute error, the
largest error relative to the reference's magnitude, and the share of the tolerance it used -- per compared tensor and test.
`pytest -m gpu` writes them to gpurun_out/parity_margins.json (conftest.py); a copy per round lives under profiles/.  The stated
tolerances are set from these numbers, not the other way round."""
import inspect
import json
import os

import numpy as np

RECORDS = {}


def _test_id():
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]


def check_close(a, b, *, rtol=1e-7, atol=0.0, err_msg="", what=None):
    """np.testing.assert_allclose(a, b) + a record of the achieved errors."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if what is None:
        fr = inspect.stack()[2] if len(inspect.stack()) > 2 else None
        what = f"{os.path.basename(fr.filename)}:{fr.lineno} {(fr.code_context or [''])[0].strip()[:90]}" if fr else "?"
    if a.shape == b.shape or a.size == b.size or b.size == 1 or a.size == 1:
        try:
            aa, bb = np.broadcast_arrays(a.reshape(b.shape) if a.size == b.size else a, b)
            fin = np.isfinite(aa) & np.isfinite(bb)
            if fin.any():
                d = np.abs(aa - bb)[fin]
                mag = np.abs(bb)[fin]
                rec = {"what": what, "n": int(aa.size), "max_abs_err": float(d.max()), "ref_scale": float(mag.max()),
                       "max_rel_err": float((d / np.maximum(mag, 1e-30))[mag > max(atol, 1e-12)].max()) if (mag > max(atol, 1e-12)).any() else None,
                       "rtol": float(rtol), "atol": float(atol),
                       "used": float((d / (atol + rtol * mag + 1e-300)).max())}       # <= 1 passes
                RECORDS.setdefault(_test_id(), []).append(rec)
        except Exception:
            pass
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=err_msg)


def dump(path):
    if not RECORDS:
        return
    worst = {}
    for tid, recs in RECORDS.items():
        for r in recs:
            k = (r["rtol"], r["atol"])
            w = worst.setdefault(k, {"rtol": r["rtol"], "atol": r["atol"], "comparisons": 0, "max_used": 0.0, "max_abs_err": 0.0, "max_rel_err": 0.0})
            w["comparisons"] += 1
            w["max_used"] = max(w["max_used"], r["used"])
            w["max_abs_err"] = max(w["max_abs_err"], r["max_abs_err"])
            if r["max_rel_err"] is not None:
                w["max_rel_err"] = max(w["max_rel_err"], r["max_rel_err"])
    out = {"note": "achieved errors of every tolerance comparison of `pytest -m gpu` (device fp32 against fixtures / the fp64 oracle / other routes); "
                   "`used` = max |a-b| / (atol + rtol |b|), <= 1 passes",
           "by_tolerance": sorted(worst.values(), key=lambda w: (w["rtol"], w["atol"])),
           "by_test": RECORDS}
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
