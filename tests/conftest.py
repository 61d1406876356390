"""pytest configuration: the `gpu` marker, and import paths for the package and the oracle (tests only)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. a plain `pytest tests/` here."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """npz -> nested dict ('P/u_x' -> d['P']['u_x'])."""
    import numpy as np
    raw = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in raw.files:
        if "/" in k:
            a, b = k.split("/", 1)
            out.setdefault(a, {})[b] = raw[k]
        else:
            out[k] = raw[k]
    return out
