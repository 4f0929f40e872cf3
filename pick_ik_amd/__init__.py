"""pick_ik_amd -- MI355X-native batched IK solver behind pick_ik's solver interface.

The package holds only the hot path BASELINE.json names: ``csrc/`` (HIP kernels + the C ABI of
include/pick_ik_amd.h), ``solver.py`` (ctypes mirror of the reference's solver interface),
``robots.py`` (serial-chain tables) and ``build.py``.
"""
from . import robots  # noqa: F401
from .solver import (APPROXIMATE, NO_IK_SOLUTION, SUCCESS, Params, PickIkAmdError,  # noqa: F401
                     Solver, default_params, ik_gradient, ik_memetic)
