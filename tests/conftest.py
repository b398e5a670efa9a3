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


# ---- parity ledger: every normwise comparison of a run, with the error it measured.  Written to
# $PYGCN_LEDGER (a JSON file) at the end of the session; with PYGCN_LEDGER_ONLY=1 a comparison that
# misses its gate is recorded instead of failing, so ONE run prices every gate (how the gates of
# DESIGN §2 were set: profiles/r04_parity_ledger.md).
_LEDGER = []
_CURRENT = {"test": ""}


@pytest.fixture(autouse=True)
def _ledger_test_name(request):
    _CURRENT["test"] = request.node.nodeid
    yield


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get("PYGCN_LEDGER")
    if path and _LEDGER:
        import json
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(_LEDGER, f, indent=0)


def _record(what, err, scale, rel, extra=0.0):
    ratio = float(err / scale) if scale > 0 else 0.0
    _LEDGER.append({"test": _CURRENT["test"], "what": what, "err_over_scale": ratio, "gate": float(rel),
                    "allowance": float(extra), "ok": bool(err <= rel * scale + extra * scale + 1e-30)})
    return ratio


def assert_normwise(got, ref, rel=1e-5, what=""):
    """The parity metric of BASELINE.md §3 / SURVEY §7: max|got-ref| <= rel * max|ref|
    (elementwise-relative error is meaningless at cancelling outputs)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    scale = np.abs(ref).max() if ref.size else 0.0
    err = np.abs(got - ref).max() if ref.size else 0.0
    _record(what, err, scale, rel)
    if os.environ.get("PYGCN_LEDGER_ONLY") == "1":
        return
    assert err <= rel * scale + 1e-30, f"{what}: max|d|={err:.3e} > {rel:g}*{scale:.3e}"


def assert_parity(got, ref32, ref64, what="", rel=1e-5):
    """The north-star gate for a result that is a LONG float32 reduction (a weight / bias gradient
    summed over the graph's vertices), where the float32 reference itself carries rounding error of
    the order of the contract:
      primary   max|got - ref64| <= rel * max|ref64|         (ref64: the same step in float64)
      secondary max|got - ref32| <= (rel + e_ref) * max|ref32|,  e_ref = max|ref32 - ref64| / max|ref64|
    i.e. within the contract of exact arithmetic, and within the contract PLUS THE REFERENCE'S OWN
    measured rounding error of the float32 reference (oracle = the reference's CPU arithmetic)."""
    got = np.asarray(got, np.float64)
    ref32, ref64 = np.asarray(ref32, np.float64), np.asarray(ref64, np.float64)
    assert got.shape == ref32.shape == ref64.shape, f"{what}: shapes {got.shape} {ref32.shape} {ref64.shape}"
    scale = np.abs(ref64).max() if ref64.size else 0.0
    e_ref = (np.abs(ref32 - ref64).max() / scale) if scale > 0 else 0.0
    err64 = np.abs(got - ref64).max() if ref64.size else 0.0
    err32 = np.abs(got - ref32).max() if ref64.size else 0.0
    _record(what + " [vs float64]", err64, scale, rel)
    _record(what + " [vs float32 reference]", err32, scale, rel, extra=e_ref)
    if os.environ.get("PYGCN_LEDGER_ONLY") == "1":
        return
    assert err64 <= rel * scale + 1e-30, f"{what}: vs float64 max|d|={err64:.3e} > {rel:g}*{scale:.3e}"
    assert err32 <= (rel + e_ref) * scale + 1e-30, \
        f"{what}: vs float32 reference max|d|={err32:.3e} > ({rel:g}+{e_ref:.2e})*{scale:.3e}"


@pytest.fixture
def h2_scheme():
    """The scaled two-part fp16 scheme (the one that HAS bounds of max|operand|), restored afterwards."""
    from pygcn_amd import spmm as S
    before = S.gemm_scheme()
    S.set_gemm_scheme("h2")
    yield
    S.set_gemm_scheme(before)


@pytest.fixture(params=["bf16x3", "h2"])
def gemm_scheme(request):
    """Runs a test under both decompositions of the 256-wide fp32 GEMMs: "bf16x3" (the default,
    fp32-equivalent) and "h2" (scaled two-part fp16, opt-in)."""
    from pygcn_amd import spmm as S
    before = S.gemm_scheme()
    S.set_gemm_scheme(request.param)
    yield request.param
    S.set_gemm_scheme(before)


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
