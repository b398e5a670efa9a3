import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # safety net for a fresh checkout: build the HIP library in-tree (hipcc cross-compiles gfx950
    # without a GPU); normally __graft_entry__.build() has already done it
    from pygcn_amd import build as native_build
    if native_build.needs_build():
        native_build.build(verbose=False)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def assert_normwise(got, ref, rel=1e-5, what=""):
    """The parity metric of BASELINE.md §3 / SURVEY §7: max|got-ref| <= rel * max|ref|
    (elementwise-relative error is meaningless at cancelling outputs)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    scale = np.abs(ref).max() if ref.size else 0.0
    err = np.abs(got - ref).max() if ref.size else 0.0
    assert err <= rel * scale + 1e-30, f"{what}: max|d|={err:.3e} > {rel:g}*{scale:.3e}"


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand if the .so is missing."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "liboracle_spmm.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gcn_oracle
    return gcn_oracle
