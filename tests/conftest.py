import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The suite runs with the library's DEFAULT options, the automatic self test included (option self_test = auto: the
# first host-pointer solve or reserve of a kernel set that the general or the exact kernels serve checks every kernel
# variant against the one-lane kernel first -- a short run, ~10 ms per handle and kernel set).  PIK_SELF_TEST=off
# in the environment switches it off for a faster local run; tests/test_gpu_self_test.py sets what it needs.


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O


# ---- the two exact builds ------------------------------------------------------------------------
# Tests that compare an exact build with the oracle at tolerance zero run twice:
#   "portable": the verification library (libpick_ik_amd_strict.so: plain IEEE arithmetic) against the
#               oracle's math mode "portable";
#   "fma":      the PRODUCT library's exact kernels (option arithmetic = exact: fused multiply-adds at stated
#               places) against the oracle's math mode "fma".
# The tests are written for the first pair (pk.Solver(..., strict=True), O.math_mode("portable")); under "fma"
# this fixture maps both onto the second.
_exact_solver_class = None


def _exact_solver():
    global _exact_solver_class
    if _exact_solver_class is None:
        import pick_ik_amd as pk

        class ExactSolver(pk.Solver):
            def __init__(self, chain, device=0, strict=False, exact=False):
                super().__init__(chain, device=device, strict=False, exact=bool(strict or exact))

        _exact_solver_class = ExactSolver
    return _exact_solver_class


@pytest.fixture(params=["portable", "fma"])
def exact_flavour(request, monkeypatch):
    if request.param == "fma":
        import pick_ik_amd as pk
        from oracle import oracle as O
        real_mode = O.math_mode
        monkeypatch.setattr(pk, "Solver", _exact_solver())
        monkeypatch.setattr(O, "math_mode", lambda m: real_mode("fma" if m == "portable" else m))
    return request.param
