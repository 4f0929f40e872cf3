"""The MoveIt plugin shim (pick_ik_amd/host/pick_ik_plugin_shim.cpp) meets a compiler and a GPU.

ROS 2 / MoveIt / pluginlib / Eigen are absent from this image, so the shim is compiled against the
declaration stubs of tests/native/ros2_stubs/ (README there): on the CPU the translation unit must
compile warning-free and the device-independent behaviour of initialize() is checked; on the GPU
tests/native/shim_check.cpp drives initialize / searchPositionIK -- parameter mapping, restarts,
solution callback, approximate-solution gate, every overload -- onto the real C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "shim_check.cpp")
EXE = os.path.join(ROOT, "tests", "native", "shim_check")
STUBS = os.path.join(ROOT, "tests", "native", "ros2_stubs")
HOST = os.path.join(ROOT, "pick_ik_amd", "host")


def _deps():
    out = [SRC, os.path.join(HOST, "pick_ik_plugin_shim.cpp"), os.path.join(HOST, "pick_ik_amd.hpp"),
           os.path.join(ROOT, "oracle", "pik_oracle.h")]
    for d, _, files in os.walk(STUBS):
        out += [os.path.join(d, f) for f in files]
    return out


@pytest.fixture(scope="module")
def exe():
    import __graft_entry__ as g
    g.build()
    lib_dir = os.path.join(ROOT, "pick_ik_amd")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(map(os.path.getmtime, _deps())):
        oracle_dir = os.path.join(ROOT, "oracle")  # (the checker of the arithmetic = exact cases)
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + STUBS, "-I" + HOST, SRC,
                        "-o", EXE, "-L" + lib_dir, "-lpick_ik_amd", "-Wl,-rpath," + lib_dir,
                        "-L" + oracle_dir, "-lpik_oracle", "-Wl,-rpath," + oracle_dir,
                        "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return EXE


def test_shim_compiles_warning_free_against_the_stubs():
    """the shim translation unit alone (as a ROS 2 workspace would build it), -Wall -Wextra -Werror"""
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + STUBS, "-I" + HOST,
                    os.path.join(HOST, "pick_ik_plugin_shim.cpp")], check=True)


def test_shim_overrides_every_reference_virtual():
    """the six searchPositionIK overloads + the other virtuals of include/pick_ik/pick_ik_plugin.hpp:24-102"""
    text = open(os.path.join(HOST, "pick_ik_plugin_shim.cpp")).read()
    assert text.count("bool searchPositionIK(") == 6
    assert text.count(") const override") >= 10  # 6 + getJointNames, getLinkNames, getPositionFK, getPositionIK
    for name in ("memetic_num_threads", "memetic_stop_on_first_solution", "approximate_solution_cost_threshold",
                 "approximate_solution_joint_threshold", "gd_max_iters", "memetic_gd_max_iters"):
        assert f'"{name}"' in text, name


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="the device-independent part")
def test_shim_initialize_error_behaviour_without_device(exe):
    r = subprocess.run([exe, "cpu"], capture_output=True, text=True)
    assert r.returncode == 0 and "shim checks without a device OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_shim_drives_the_c_abi_on_the_gpu(exe):
    r = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "plugin shim checks OK" in r.stdout, r.stdout + r.stderr
