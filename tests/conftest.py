import glob
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
    # the library is built in-tree by __graft_entry__.build(); a checkout without it (fresh clone) gets it
    # compiled here once (hipcc cross-compiles without a GPU).  The product itself never builds implicitly.
    from exemplars_vc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from exemplars_vc_amd.csrc.build import build
        build()


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_golden(path):
    with np.load(path, allow_pickle=False) as d:
        return {k: d[k] for k in d.files}


def rel_err(got, want):
    """max |got-want| / |want| over entries where want != 0, and max |got| where want == 0."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    nz = want != 0
    r = np.max(np.abs(got[nz] - want[nz]) / np.abs(want[nz])) if nz.any() else 0.0
    z = np.max(np.abs(got[~nz])) if (~nz).any() else 0.0
    return float(r), float(z)
