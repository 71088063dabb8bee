import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(ROOT, "3d-denoising-diffusion-model_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) where no device is visible, so the
    # CPU tier can still be run with no -m filter.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def rel_err(a, b):
    """max |a-b| / max |b|  (the 'rel fp32' figure the parity bar is stated in)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rel_err_per_channel(a, b, axis=1):
    """max over channels c of  max|a_c - b_c| / max|b_c|  -- each channel (e.g. the eps and the
    variance channel of the model output) is held to the bar on its own scale, so a small channel
    cannot hide behind a large one."""
    a = np.moveaxis(np.asarray(a, dtype=np.float64), axis, 0)
    b = np.moveaxis(np.asarray(b, dtype=np.float64), axis, 0)
    a = a.reshape(a.shape[0], -1)
    b = b.reshape(b.shape[0], -1)
    num = np.abs(a - b).max(axis=1)
    den = np.maximum(np.abs(b).max(axis=1), 1e-30)
    return float((num / den).max())
