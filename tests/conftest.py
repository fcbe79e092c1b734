import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# libdpc reads its A/B kernel-selection switches (DPC_UNFUSED_ATTN, ...) only in processes that also set DPC_DEBUG=1 (csrc/common.h:
# debug_switch); a few GPU tests compare the fused kernels with their unfused compositions through those switches
os.environ.setdefault("DPC_DEBUG", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def note_error(name, err):
    """Measured deviations behind the asserted tolerances: with DPC_TOL_LOG=<file> every parity test appends (test id, quantity,
    measured error) -- the tolerances in the tests are set from such a log (worst measured x 3, never above SURVEY.md 8d's bars)."""
    path = os.environ.get("DPC_TOL_LOG")
    if path:
        import json
        test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps({"test": test, "name": name, "err": float(err)}) + "\n")
    return err


@pytest.fixture(scope="session")
def golden():
    return load_golden


def pytest_collection_modifyitems(config, items):
    """GPU tests fail loudly (not skip) when selected with -m gpu on a box without a GPU/extension."""
    return
