import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


_OBSERVED = {}


def _note(v):
    """Worst observed error per test, written to gpurun_out/parity_observed.json at the end of the session."""
    t = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    _OBSERVED[t] = max(_OBSERVED.get(t, 0.0), float(v))


def rel_err(a, b):
    """max element-wise relative error."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    v = float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0
    _note(v)
    return v


def relclose(a, b, tol):
    """max |a - b| <= tol * max |b|  (error relative to the scale of the reference array: its entries include exact zeros)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        print(f"relclose: shapes {a.shape} vs {b.shape}")
        return False
    v = float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300)) if a.size else 0.0
    _note(v)
    if not v <= tol:
        print(f"relclose: observed {v:.3e} > tol {tol:.1e}")
    return v <= tol


def pytest_sessionfinish(session, exitstatus):
    if not _OBSERVED:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_observed.json"), "w") as f:
            json.dump(dict(sorted(_OBSERVED.items())), f, indent=1)
    except OSError:
        pass
