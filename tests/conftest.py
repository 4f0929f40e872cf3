import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The library's automatic self test (option self_test = auto: every kernel variant against the one-lane kernel,
# once per parameter set served by the general or the exact kernels) is switched off for the suite -- the tests
# compare the variants themselves, and hundreds of handles x parameter sets would each pay a dozen extra solves.
# tests/test_gpu_self_test.py switches it back on.
os.environ.setdefault("PIK_SELF_TEST", "off")


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
