import os
import sys

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda", 0)


# Measured parity maxima: tests call `measured("name", value, tolerance)`; the values are printed (pytest -s / -rP)
# and written to gpurun_out/parity_maxima.json at the end of the session, from where a round's numbers are copied
# to profiles/.
_MEASURED = {}


def measured(name, value, tol=None, unit=""):
    v = float(value)
    _MEASURED[name] = {"max": v, "tol": tol, "unit": unit}
    print(f"[parity] {name}: {v:.3e}{(' ' + unit) if unit else ''}" + (f" (tolerance {tol:g})" if tol is not None else ""))
    return v


def pytest_sessionfinish(session, exitstatus):
    if not _MEASURED:
        return
    import json
    out = os.path.join(REPO, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_maxima.json"), "w") as f:
            json.dump(_MEASURED, f, indent=1, sort_keys=True)
    except OSError:
        pass
