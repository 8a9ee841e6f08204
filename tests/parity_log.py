"""Observed parity errors of the GPU tests, relative to each test's own bar (1.0 = at the bar).

tests/ record the maxima here; conftest.py writes them to gpurun_out/parity_errors.json when the session ends (the
GPU box merges gpurun_out/ back; the copy kept for the judge is profiles/r02_parity_errors.json)."""
import json
import os

_MAX = {}


def record(key, value, bar):
    """value = observed |error|, bar = the allowed |error| for it (both arrays or scalars of one shape)."""
    import numpy as np
    value = np.asarray(value, np.float64); bar = np.asarray(bar, np.float64)
    if value.size == 0:
        return
    with np.errstate(all="ignore"):
        rel = np.where(bar > 0, value / bar, np.where(value > 0, np.inf, 0.0))
    e = _MAX.setdefault(key, {"max_error_over_bar": 0.0, "max_abs_error": 0.0, "n": 0})
    e["max_error_over_bar"] = max(e["max_error_over_bar"], float(np.nanmax(rel)))
    e["max_abs_error"] = max(e["max_abs_error"], float(np.nanmax(value)))
    e["n"] += int(value.size)


def dump(root):
    if not _MAX:
        return
    d = os.path.join(root, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    out = {"bar": "1e-12 relative to the sum of |terms| of each window (sums), 1e-12 * sum|num terms| / sum den (window Fst); "
                  "max_error_over_bar = worst observed error as a fraction of that bar", "cases": _MAX}
    with open(os.path.join(d, "parity_errors.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
