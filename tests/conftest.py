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


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    a = np.asarray(a).ravel()
    b = np.asarray(b).ravel()
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def circ_err(xa, xb, L):
    """max distance on the circle of circumference L (x=0 and x=L-eps are neighbours)."""
    d = np.abs(np.asarray(xa, dtype=float).ravel() - np.asarray(xb, dtype=float).ravel())
    return float(np.max(np.minimum(d, L - d)))


# Measured error margins of the GPU tests: every call appends "name: value" to gpurun_out/measured_r4.json on the
# box (gpurun merges that directory back), so that the asserted bounds can be quoted against what was measured.
_MEASURED = {}


def record_measure(name, value):
    import json
    _MEASURED[name] = float(value)
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "measured_r4.json"), "w") as f:
            json.dump(_MEASURED, f, indent=1, sort_keys=True)
    except OSError:
        pass
