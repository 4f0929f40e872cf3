"""CPU: the C-ABI library loads, exports every symbol include/pick_ik_amd.h declares, mirrors the
yaml defaults, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from pick_ik_amd import solver
    return solver


def test_exports_every_declared_symbol(built):
    header = open(os.path.join(ROOT, "include", "pick_ik_amd.h")).read()
    declared = set(re.findall(r"\b(pikamd_[a-z_]+)\s*\(", header))
    assert declared == set(built.EXPORTED_SYMBOLS)
    L = built.lib()
    for name in declared:
        assert getattr(L, name) is not None


def test_defaults_mirror_yaml_and_oracle(built, oracle_mod):
    p = built.default_params()
    o = oracle_mod.default_params()
    assert [f[0] for f in p._fields_] == [f[0] for f in o._fields_]
    for name, _ in p._fields_:
        assert getattr(p, name) == getattr(o, name), name
    # src/pick_ik_parameters.yaml
    assert (p.mode, p.gd_step_size, p.gd_max_iters, p.gd_min_cost_delta) == (0, 1e-4, 100, 1e-12)
    assert (p.position_threshold, p.orientation_threshold, p.cost_threshold) == (1e-3, 1e-3, 1e-3)
    assert (p.position_scale, p.rotation_scale) == (1.0, 0.5)
    assert (p.memetic_population_size, p.memetic_elite_size) == (16, 4)
    assert (p.memetic_wipeout_fitness_tol, p.memetic_max_generations, p.memetic_gd_max_iters) == (
        1e-5, 100, 25)
    assert p.stop_optimization_on_valid_solution == 1 and p.memetic_num_threads == 1


def test_struct_sizes_match_header(built, tmp_path):
    """the ctypes mirrors against the header as a C compiler lays it out (and the header is valid
    strict C99 and C++11)"""
    import subprocess
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "include/pick_ik_amd.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu\\n", sizeof(pikamd_params), sizeof(pikamd_stats), '
                   'sizeof(pikamd_batch), sizeof(pikamd_urdf_model)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + ROOT, str(src), "-o", str(exe)],
                   check=True)
    subprocess.run(["g++", "-std=c++11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + ROOT, "-x",
                    "c++", str(src)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [C.sizeof(built.Params), built.STATS_DTYPE.itemsize, C.sizeof(built.Batch), C.sizeof(built.UrdfModel)]
    assert sizes[:2] == [144, 24]


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="this checks the no-GPU behaviour")
def test_no_cpu_fallback(built):
    import pick_ik_amd as pk
    with pytest.raises(pk.PickIkAmdError, match="no HIP device"):
        pk.Solver(pk.robots.panda())


def test_bad_chain_rejected(built):
    import numpy as np
    import pick_ik_amd as pk
    ch = pk.robots.panda()
    import dataclasses
    bad = dataclasses.replace(ch, axis=np.zeros((7, 3)))
    with pytest.raises(pk.PickIkAmdError, match="zero joint axis"):
        pk.Solver(bad)
