"""ctypes binding of libpick_ik_amd.so (include/pick_ik_amd.h) -- the host-side mirror of pick_ik's
solver interface for batch callers.

Names follow the reference: ``ik_memetic`` / ``ik_gradient`` (include/pick_ik/ik_memetic.hpp:97-104,
include/pick_ik/ik_gradient.hpp:43-49), ``Params`` carries the fields of
src/pick_ik_parameters.yaml.  There is no CPU implementation behind this module: if the HIP
library is missing or no gfx950 device is present every call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PIK_LIB") or os.path.join(_HERE, "libpick_ik_amd.so")  # (PIK_LIB: A/B experiments)
#: verification build (no FMA contraction, generic rotations); see pick_ik_amd/build.py
LIB_STRICT_PATH = os.environ.get("PIK_LIB_STRICT") or os.path.join(_HERE, "libpick_ik_amd_strict.so")

SUCCESS = 1
APPROXIMATE = 2
NO_IK_SOLUTION = -31
MAX_SLOTS = 128
MAX_BATCHES = 64
MAX_HOST_JOBS = 16


class PickIkAmdError(RuntimeError):
    pass


class Params(C.Structure):
    """pikamd_params: src/pick_ik_parameters.yaml names and defaults, iteration budgets only."""

    _fields_ = [
        ("mode", C.c_int32),
        ("gd_step_size", C.c_double),
        ("gd_max_iters", C.c_int32),
        ("gd_min_cost_delta", C.c_double),
        ("position_threshold", C.c_double),
        ("orientation_threshold", C.c_double),
        ("cost_threshold", C.c_double),
        ("position_scale", C.c_double),
        ("rotation_scale", C.c_double),
        ("center_joints_weight", C.c_double),
        ("avoid_joint_limits_weight", C.c_double),
        ("minimal_displacement_weight", C.c_double),
        ("stop_optimization_on_valid_solution", C.c_int32),
        ("memetic_num_threads", C.c_int32),
        ("memetic_stop_on_first_solution", C.c_int32),
        ("memetic_population_size", C.c_int32),
        ("memetic_elite_size", C.c_int32),
        ("memetic_wipeout_fitness_tol", C.c_double),
        ("memetic_max_generations", C.c_int32),
        ("memetic_gd_max_iters", C.c_int32),
        ("return_approximate_solution", C.c_int32),
    ]


class _Chain(C.Structure):
    _fields_ = [
        ("dof", C.c_int32),
        ("origin_xyz_rpy", C.POINTER(C.c_double)),
        ("axis", C.POINTER(C.c_double)),
        ("joint_type", C.POINTER(C.c_int32)),
        ("tip_xyz_rpy", C.POINTER(C.c_double)),
        ("qmin", C.POINTER(C.c_double)),
        ("qmax", C.POINTER(C.c_double)),
        ("vmax", C.POINTER(C.c_double)),
        ("bounded", C.POINTER(C.c_uint8)),
    ]


class _Tip(C.Structure):
    _fields_ = [
        ("n_joints", C.c_int32),
        ("variable", C.POINTER(C.c_int32)),
        ("origin_xyz_rpy", C.POINTER(C.c_double)),
        ("axis", C.POINTER(C.c_double)),
        ("joint_type", C.POINTER(C.c_int32)),
        ("tip_xyz_rpy", C.POINTER(C.c_double)),
    ]


class _MultiChain(C.Structure):
    _fields_ = [
        ("dof", C.c_int32),
        ("n_tips", C.c_int32),
        ("tips", C.POINTER(_Tip)),
        ("qmin", C.POINTER(C.c_double)),
        ("qmax", C.POINTER(C.c_double)),
        ("vmax", C.POINTER(C.c_double)),
        ("bounded", C.POINTER(C.c_uint8)),
    ]


class Batch(C.Structure):
    """pikamd_batch: one batch of a multi-batch call (device or host pointers, see the header)."""

    _fields_ = [
        ("B", C.c_int64),
        ("goal_pos_quat", C.c_void_p),
        ("seed", C.c_void_p),
        ("initial_guess", C.c_void_p),
        ("problem_offset", C.c_int64),
        ("solution", C.c_void_p),
        ("status", C.c_void_p),
        ("final_cost", C.c_void_p),
        ("stats", C.c_void_p),
        ("completed", C.c_void_p),
    ]


MAX_DOF, MAX_TIPS, MAX_NAME, MAX_MIMIC = 16, 8, 64, 4  # (PIKAMD_MAX_MIMIC: per tip path)


class _UrdfTip(C.Structure):
    _fields_ = [("n_joints", C.c_int32), ("variable", C.c_int32 * MAX_DOF),
                ("origin_xyz_rpy", C.c_double * 6 * MAX_DOF), ("axis", C.c_double * 3 * MAX_DOF),
                ("joint_type", C.c_int32 * MAX_DOF), ("tip_xyz_rpy", C.c_double * 6)]


class MimicJointC(C.Structure):
    """pikamd_mimic_joint / pko_mimic_joint"""
    _fields_ = [("tip", C.c_int32), ("after_variable", C.c_int32), ("master_variable", C.c_int32), ("joint_type", C.c_int32),
                ("origin_xyz_rpy", C.c_double * 6), ("axis", C.c_double * 3), ("multiplier", C.c_double), ("offset", C.c_double)]


def mimic_array(chain):
    """the chain's MimicJoint records as a C array (None when it has none)"""
    ms = tuple(getattr(chain, "mimic", ()) or ())
    if not ms:
        return None
    arr = (MimicJointC * len(ms))()
    for i, m in enumerate(ms):
        arr[i] = MimicJointC(int(m.tip), int(m.after_variable), int(m.master_variable), int(m.joint_type),
                             (C.c_double * 6)(*[float(x) for x in m.origin_xyz_rpy]), (C.c_double * 3)(*[float(x) for x in m.axis]),
                             float(m.multiplier), float(m.offset))
    return arr


class UrdfModel(C.Structure):
    """pikamd_urdf_model: what pikamd_urdf_extract found between base_link and the tip link(s)."""

    _fields_ = [("dof", C.c_int32), ("n_tips", C.c_int32), ("variable_names", (C.c_char * MAX_NAME) * MAX_DOF),
                ("qmin", C.c_double * MAX_DOF), ("qmax", C.c_double * MAX_DOF), ("vmax", C.c_double * MAX_DOF),
                ("bounded", C.c_uint8 * MAX_DOF), ("tips", _UrdfTip * MAX_TIPS),
                ("n_mimic", C.c_int32), ("mimic", MimicJointC * (MAX_TIPS * MAX_MIMIC))]


STATS_DTYPE = np.dtype(
    [("cost_evals", "<i8"), ("generations", "<i4"), ("wipeouts", "<i4"),
     ("pool_erasures", "<i4"), ("reserved", "<i4")])

#: every symbol include/pick_ik_amd.h declares
EXPORTED_SYMBOLS = (
    "pikamd_default_params", "pikamd_create", "pikamd_destroy", "pikamd_variables",
    "pikamd_fk_batch", "pikamd_cost_batch", "pikamd_gd_step_batch", "pikamd_solve_batch",
    "pikamd_solve_batch_device", "pikamd_fk_batch_device", "pikamd_last_error", "pikamd_version",
    "pikamd_kernel_name", "pikamd_reserve", "pikamd_create_multi", "pikamd_n_tips",
    "pikamd_solve_batches_device", "pikamd_solve_batches_async", "pikamd_wait", "pikamd_solve_batches",
    "pikamd_urdf_extract", "pikamd_create_from_urdf", "pikamd_set_option", "pikamd_solve_batch_host", "pikamd_set_mimic_joints",
    "pikamd_shard_bounds", "pikamd_solve_batch_sharded", "pikamd_self_test", "pikamd_self_test_cost",
)

_libs = {}


# double cost_fn(const double* q, int32_t dof, int32_t pose_index, void* user) -- pikamd_cost_fn
COST_FN = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_int32, C.c_int32, C.c_void_p)


def cost_callback(fn):
    """Python callable fn(q: ndarray[dof], pose_index) -> float as a pikamd_cost_fn"""
    def trampoline(q, dof, pose, _user):
        return float(fn(np.ctypeslib.as_array(q, shape=(dof,)).copy(), int(pose)))
    return COST_FN(trampoline)


def lib(strict: bool = False):
    """Load the HIP library; fails loudly when it has not been built (no fallback)."""
    if strict in _libs:
        return _libs[strict]
    path = LIB_STRICT_PATH if strict else LIB_PATH
    if not os.path.exists(path):
        raise PickIkAmdError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). pick_ik_amd has no CPU fallback.")
    L = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    vp = C.c_void_p
    L.pikamd_default_params.argtypes = [C.POINTER(Params)]
    L.pikamd_default_params.restype = None
    L.pikamd_create.argtypes = [C.POINTER(_Chain), C.c_int32, C.POINTER(vp)]
    L.pikamd_create_multi.argtypes = [C.POINTER(_MultiChain), C.c_int32, C.POINTER(vp)]
    L.pikamd_create_multi.restype = C.c_int32
    L.pikamd_n_tips.argtypes = [vp]
    L.pikamd_n_tips.restype = C.c_int32
    L.pikamd_destroy.argtypes = [vp]
    L.pikamd_destroy.restype = None
    L.pikamd_variables.argtypes = [vp, dp]
    L.pikamd_fk_batch.argtypes = [vp, C.c_int64, dp, dp]
    L.pikamd_cost_batch.argtypes = [vp, C.POINTER(Params), C.c_int64, dp, dp, dp, dp, ip]
    L.pikamd_gd_step_batch.argtypes = [vp, C.POINTER(Params), C.c_int64, dp, dp, dp, dp, dp, dp,
                                       dp, ip]
    L.pikamd_solve_batch.argtypes = [vp, C.POINTER(Params), C.c_int64, dp, dp, C.c_uint64,
                                     C.c_int64, dp, ip, dp, vp]
    L.pikamd_solve_batch_device.argtypes = [vp, C.POINTER(Params), C.c_int64, vp, vp, C.c_uint64,
                                            C.c_int64, vp, vp, vp, vp, vp, C.c_int32]
    L.pikamd_fk_batch_device.argtypes = [vp, C.c_int64, vp, vp, vp]
    L.pikamd_solve_batches_device.argtypes = [vp, C.POINTER(Params), C.c_int32, C.POINTER(Batch),
                                              C.c_uint64, vp, C.c_int32]
    L.pikamd_solve_batches_async.argtypes = [vp, C.POINTER(Params), C.c_int32, C.POINTER(Batch),
                                             C.c_uint64, C.c_int32]
    L.pikamd_wait.argtypes = [vp, C.c_int32]
    L.pikamd_solve_batches.argtypes = [vp, C.POINTER(Params), C.c_int32, C.POINTER(Batch), C.c_uint64]
    L.pikamd_reserve.argtypes = [vp, C.POINTER(Params), C.c_int64, C.c_int32, vp]
    L.pikamd_reserve.restype = C.c_int32
    L.pikamd_urdf_extract.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int32,
                                      C.POINTER(UrdfModel)]
    L.pikamd_urdf_extract.restype = C.c_int32
    L.pikamd_create_from_urdf.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int32, C.c_int32,
                                          C.POINTER(vp)]
    L.pikamd_create_from_urdf.restype = C.c_int32
    L.pikamd_shard_bounds.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.pikamd_shard_bounds.restype = None
    L.pikamd_solve_batch_sharded.argtypes = [C.POINTER(vp), C.c_int32, C.POINTER(Params), C.c_int64, dp, dp, dp,
                                             C.c_uint64, C.c_int64, dp, ip, dp, vp]
    L.pikamd_solve_batch_sharded.restype = C.c_int32
    L.pikamd_self_test.argtypes = [vp, C.POINTER(Params), C.c_int32, C.POINTER(C.c_uint32)]
    L.pikamd_self_test.restype = C.c_int32
    L.pikamd_self_test_cost.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.pikamd_self_test_cost.restype = C.c_int32
    L.pikamd_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.pikamd_set_option.restype = C.c_int32
    L.pikamd_solve_batch_host.argtypes = [vp, C.POINTER(Params), C.c_int64, dp, dp, dp, C.c_uint64, C.c_int64, COST_FN,
                                          vp, dp, ip, dp, vp]
    L.pikamd_solve_batch_host.restype = C.c_int32
    L.pikamd_set_mimic_joints.argtypes = [vp, C.c_int32, C.POINTER(MimicJointC)]
    L.pikamd_set_mimic_joints.restype = C.c_int32
    L.pikamd_last_error.restype = C.c_char_p
    L.pikamd_version.restype = C.c_char_p
    L.pikamd_kernel_name.restype = C.c_char_p
    L.pikamd_kernel_name.argtypes = [vp, C.POINTER(Params)]
    for name in ("pikamd_create", "pikamd_variables", "pikamd_fk_batch", "pikamd_cost_batch",
                 "pikamd_gd_step_batch", "pikamd_solve_batch", "pikamd_solve_batch_device",
                 "pikamd_fk_batch_device", "pikamd_solve_batches_device",
                 "pikamd_solve_batches_async", "pikamd_wait", "pikamd_solve_batches"):
        getattr(L, name).restype = C.c_int32
    _libs[strict] = L
    return L


def _check(rc: int, L=None):
    if rc != 0:
        L = L or lib()
        raise PickIkAmdError(f"pick_ik_amd error {rc}: {L.pikamd_last_error().decode()}")


def default_params(**kw) -> Params:
    p = Params()
    lib().pikamd_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _urdf_text(urdf: str) -> bytes:
    text = urdf if urdf.lstrip().startswith("<") else open(urdf).read()
    return text.encode()


def urdf_extract(urdf: str, base_link: str, tip_links, strict: bool = False):
    """pikamd_urdf_extract (no GPU needed): the chain the NATIVE reader finds, as a robots.Chain (one
    tip) or robots.MultiChain, plus the variable names (= the order of the joint vector).
    `urdf` is the XML text or a path; tip_links a name or a sequence of names."""
    from .robots import Chain, multi_chain
    L = lib(strict)
    tips = [tip_links] if isinstance(tip_links, str) else list(tip_links)
    arr = (C.c_char_p * len(tips))(*[t.encode() for t in tips])
    m = UrdfModel()
    _check(L.pikamd_urdf_extract(_urdf_text(urdf), base_link.encode(), arr, len(tips), C.byref(m)), L)
    d = m.dof
    names = [m.variable_names[i].value.decode() for i in range(d)]
    lim = [np.array(m.qmin[:d]), np.array(m.qmax[:d]), np.array(m.vmax[:d]), np.array(m.bounded[:d], dtype=np.uint8)]

    def tip_arrays(t):
        n = t.n_joints
        return (list(t.variable[:n]), np.array([list(r) for r in t.origin_xyz_rpy[:n]]).reshape(n, 6),
                np.array([list(r) for r in t.axis[:n]]).reshape(n, 3), np.array(t.joint_type[:n], dtype=np.int32),
                np.array(t.tip_xyz_rpy[:]))

    from .robots import MimicJoint
    import dataclasses
    mim = tuple(MimicJoint(after_variable=x.after_variable, master_variable=x.master_variable,
                           origin_xyz_rpy=tuple(x.origin_xyz_rpy), axis=tuple(x.axis), multiplier=x.multiplier,
                           offset=x.offset, joint_type=x.joint_type, tip=x.tip) for x in m.mimic[:m.n_mimic])
    if m.n_tips == 1:
        _, o, a, jt, tip = tip_arrays(m.tips[0])
        return Chain(name="urdf", origin_xyz_rpy=o, axis=a, joint_type=jt, tip_xyz_rpy=tip, qmin=lim[0],
                     qmax=lim[1], vmax=lim[2], bounded=lim[3], mimic=mim), names
    paths = [tip_arrays(m.tips[k]) for k in range(m.n_tips)]
    mc = multi_chain("urdf", paths, *lim)
    return (dataclasses.replace(mc, mimic=mim) if mim else mc), names


class Solver:
    """One solver handle = one serial chain (robots.Chain) or one multi-tip chain
    (robots.MultiChain: goals and FK results hold n_tips poses per problem) on one GPU
    (PickIKPlugin::initialize's role, reference src/pick_ik_plugin.cpp:22-71)."""

    def __init__(self, chain, device: int = 0, strict: bool = False, exact=None):
        """strict: the verification library (plain IEEE arithmetic, oracle math mode "portable");
        exact: None / True = the product library's DEFAULT, option arithmetic = exact (its exact kernels
        with fused multiply-adds at stated places, oracle math mode "fma": joint vectors identical to the
        reference algorithm's); False = arithmetic = fast (the Denavit-Hartenberg kernels, opt-in)."""
        self._L = lib(strict)
        self.strict = strict
        self.exact = (exact is None or bool(exact)) and not strict
        self.chain = chain
        self.dof = int(chain.dof)
        self.device = int(device)
        self.n_tips = int(getattr(chain, "n_tips", 1))
        h = C.c_void_p()
        lim = [_f64(chain.qmin), _f64(chain.qmax), _f64(chain.vmax),
               np.ascontiguousarray(chain.bounded, dtype=np.uint8)]
        if hasattr(chain, "tips"):  # robots.MultiChain: several tip frames
            keep, tips = [lim], (_Tip * self.n_tips)()
            for i, t in enumerate(chain.tips):
                a = [np.ascontiguousarray(t.variable, dtype=np.int32), _f64(t.origin_xyz_rpy),
                     _f64(t.axis), np.ascontiguousarray(t.joint_type, dtype=np.int32),
                     _f64(t.tip_xyz_rpy)]
                keep.append(a)
                tips[i] = _Tip(len(a[0]), _ip(a[0]), _dp(a[1]), _dp(a[2]), _ip(a[3]), _dp(a[4]))
            self._keep = (keep, tips)
            c = _MultiChain(self.dof, self.n_tips, tips, _dp(lim[0]), _dp(lim[1]), _dp(lim[2]),
                            lim[3].ctypes.data_as(C.POINTER(C.c_uint8)))
            self._chk(self._L.pikamd_create_multi(C.byref(c), self.device, C.byref(h)))
        else:
            k = [_f64(chain.origin_xyz_rpy), _f64(chain.axis),
                 np.ascontiguousarray(chain.joint_type, dtype=np.int32), _f64(chain.tip_xyz_rpy)]
            self._keep = (k, lim)
            c = _Chain(self.dof, _dp(k[0]), _dp(k[1]), _ip(k[2]), _dp(k[3]), _dp(lim[0]),
                       _dp(lim[1]), _dp(lim[2]), lim[3].ctypes.data_as(C.POINTER(C.c_uint8)))
            self._chk(self._L.pikamd_create(C.byref(c), self.device, C.byref(h)))
        self._h = h
        arr = mimic_array(chain)
        if arr is not None:
            self._chk(self._L.pikamd_set_mimic_joints(self._h, len(arr), arr))
        if not self.strict and exact is not None:  # (None: whatever the library defaults to -- exact)
            self.set_option("arithmetic", "exact" if exact else "fast")
        self._env_options()

    @classmethod
    def from_urdf(cls, urdf: str, base_link: str, tip_links, device: int = 0, strict: bool = False):
        """pikamd_create_from_urdf: the library reads the robot description itself (native reader).
        The handle's `chain` / `variable_names` are what pikamd_urdf_extract reports."""
        chain, names = urdf_extract(urdf, base_link, tip_links, strict)
        self = cls.__new__(cls)
        self._L = lib(strict)
        self.strict, self.chain, self.variable_names = strict, chain, names
        self.exact = not strict  # (the library's default arithmetic)
        self.dof, self.device, self.n_tips = int(chain.dof), int(device), int(getattr(chain, "n_tips", 1))
        tips = [tip_links] if isinstance(tip_links, str) else list(tip_links)
        arr = (C.c_char_p * len(tips))(*[t.encode() for t in tips])
        h = C.c_void_p()
        self._keep = None
        _check(self._L.pikamd_create_from_urdf(_urdf_text(urdf), base_link.encode(), arr, len(tips), self.device,
                                               C.byref(h)), self._L)
        self._h = h
        return self

    def _chk(self, rc: int):
        _check(rc, self._L)

    # ---- scheduling options (pikamd_set_option) -------------------------------------------
    def set_option(self, name: str, value) -> None:
        """Pin one scheduling choice of this handle ("lanes_per_elite", "lanes_per_elite_schedule",
        "passes", "two_per_simd", "regime"; None / "" restores the default).  Results never depend on
        these."""
        v = b"" if value is None else str(value).encode()
        self._chk(self._L.pikamd_set_option(self._h, name.encode(), v))

    def self_test(self, params: Params, n: int = 64) -> int:
        """pikamd_self_test: every kernel variant against the one-lane kernel on n generated targets of this
        chain; returns the mask of variants that disagreed (and are now switched off for this handle)"""
        m = C.c_uint32(0)
        self._chk(self._L.pikamd_self_test(self._h, C.byref(params), n, C.byref(m)))
        return int(m.value)

    def self_test_cost(self):
        """pikamd_self_test_cost: (automatic self tests run on this handle, their wall-clock time in ms)"""
        runs, ms = C.c_int32(0), C.c_double(0.0)
        self._chk(self._L.pikamd_self_test_cost(self._h, C.byref(runs), C.byref(ms)))
        return int(runs.value), float(ms.value)

    #: Test / experiment hook of THIS binding (the library itself never reads the environment): these
    #: variables are turned into handle options before a solve whenever they have changed.
    ENV_OPTIONS = (("PIK_LPE", "lanes_per_elite"), ("PIK_LPE_SCHED", "lanes_per_elite_schedule"),
                   ("PIK_PASSES", "passes"), ("PIK_OCC2", "two_per_simd"), ("PIK_REGIME", "regime"),
                   ("PIK_SPECIALISED", "specialised"), ("PIK_SHARD_CHUNKS", "shard_chunks"),
                   ("PIK_SELF_TEST", "self_test"))
    # (no environment form of "joint_layout": it changes what the arrays mean, callers set it explicitly)

    def _env_options(self) -> None:
        cur = tuple(os.environ.get(k) for k, _ in self.ENV_OPTIONS)
        seen = getattr(self, "_env_seen", (None,) * len(cur))
        if cur == seen:
            return
        for (_, name), new, old in zip(self.ENV_OPTIONS, cur, seen):
            if new != old:
                self.set_option(name, new)
        self._env_seen = cur

    def close(self):
        if getattr(self, "_h", None):
            self._L.pikamd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parity hooks -------------------------------------------------------------------
    def variables(self) -> np.ndarray:
        out = np.empty((self.dof, 7))
        self._chk(self._L.pikamd_variables(self._h, _dp(out)))
        return out

    def fk(self, q) -> np.ndarray:
        """make_fk_fn: tip pose [n][7] = x y z qw qx qy qz for joint vectors q [n][dof]."""
        q = _f64(q).reshape(-1, self.dof)
        out = np.empty((q.shape[0], 7) if self.n_tips == 1 else (q.shape[0], self.n_tips, 7))
        self._chk(self._L.pikamd_fk_batch(self._h, q.shape[0], _dp(q), _dp(out)))
        return out

    def cost(self, params: Params, goal_pos_quat, seed, q):
        """cost_fn and solution_fn of n (goal, seed, q) triples (broadcast goal/seed if 1-D)."""
        q = _f64(q).reshape(-1, self.dof)
        n = q.shape[0]
        g7 = 7 * self.n_tips
        goal = np.ascontiguousarray(np.broadcast_to(_f64(goal_pos_quat).reshape(-1, g7), (n, g7)))
        seed = np.ascontiguousarray(np.broadcast_to(_f64(seed).reshape(-1, self.dof),
                                                    (n, self.dof)))
        cost = np.empty(n)
        sol = np.empty(n, dtype=np.int32)
        self._chk(self._L.pikamd_cost_batch(self._h, C.byref(params), n, _dp(goal), _dp(seed), _dp(q),
                                       _dp(cost), _ip(sol)))
        return cost, sol

    def gd_step(self, params: Params, goal_pos_quat, seed, local, best, local_cost, best_cost):
        """One step() of src/ik_gradient.cpp:24-94 on n GradientIk states."""
        local = _f64(local).reshape(-1, self.dof).copy()
        n = local.shape[0]
        best = _f64(best).reshape(n, self.dof).copy()
        goal = _f64(goal_pos_quat).reshape(n, 7 * self.n_tips)
        seed = _f64(seed).reshape(n, self.dof)
        lc = _f64(local_cost).reshape(n).copy()
        bc = _f64(best_cost).reshape(n).copy()
        grad = np.empty((n, self.dof))
        imp = np.empty(n, dtype=np.int32)
        self._chk(self._L.pikamd_gd_step_batch(self._h, C.byref(params), n, _dp(goal), _dp(seed),
                                          _dp(local), _dp(best), _dp(lc), _dp(bc), _dp(grad),
                                          _ip(imp)))
        return local, best, lc, bc, grad, imp

    # ---- solvers ------------------------------------------------------------------------
    def _host_batch(self, goal_pos_quat, seed, initial_guess=None, problem_offset=0):
        """(pikamd_batch with host pointers, arrays to keep alive, outputs)"""
        goal = _f64(goal_pos_quat).reshape(-1, 7 * self.n_tips)
        B = goal.shape[0]
        seed = _f64(seed).reshape(B, self.dof)
        guess = None if initial_guess is None else _f64(initial_guess).reshape(B, self.dof)
        sol = np.empty((B, self.dof))
        status = np.empty(B, dtype=np.int32)
        cost = np.empty(B)
        stats = np.zeros(B, dtype=STATS_DTYPE)
        b = Batch(B, goal.ctypes.data, seed.ctypes.data, None if guess is None else guess.ctypes.data,
                  problem_offset, sol.ctypes.data, status.ctypes.data, cost.ctypes.data,
                  stats.ctypes.data, None)
        return b, (goal, seed, guess), (sol, status, cost, stats)

    def solve_batch(self, params: Params, goal_pos_quat, seed, rng_seed: int = 0,
                    problem_offset: int = 0, initial_guess=None):
        """ik_memetic / ik_gradient (params.mode) for B problems given as host arrays.
        seed = ik_seed_state (minimal-displacement reference, returned on failure); initial_guess =
        where the search starts (None = seed), src/pick_ik_plugin.cpp:199-245."""
        self._env_options()
        if initial_guess is None:
            goal = _f64(goal_pos_quat).reshape(-1, 7 * self.n_tips)
            B = goal.shape[0]
            seed = _f64(seed).reshape(B, self.dof)
            sol = np.empty((B, self.dof))
            status = np.empty(B, dtype=np.int32)
            cost = np.empty(B)
            stats = np.zeros(B, dtype=STATS_DTYPE)
            self._chk(self._L.pikamd_solve_batch(self._h, C.byref(params), B, _dp(goal), _dp(seed),
                                                 C.c_uint64(rng_seed), problem_offset, _dp(sol),
                                                 _ip(status), _dp(cost),
                                                 stats.ctypes.data_as(C.c_void_p)))
            return sol, status, cost, stats
        return self.solve_batches(params, [(goal_pos_quat, seed, initial_guess, problem_offset)],
                                  rng_seed=rng_seed)[0]

    def solve_batch_host(self, params: Params, goal_pos_quat, seed, cost_fn, rng_seed: int = 0, problem_offset: int = 0,
                         initial_guess=None):
        """pikamd_solve_batch_host: queries with a host cost function (kinematics::KinematicsBase::IKCostFn) -- one
        more goal of weight 1 per tip pose INSIDE the search (src/pick_ik_plugin.cpp:130-135) --, solved on the host
        with the exact kernels' arithmetic.  cost_fn(q: ndarray[dof], pose_index) -> float."""
        goal = _f64(goal_pos_quat).reshape(-1, 7 * self.n_tips)
        B = goal.shape[0]
        seed = _f64(seed).reshape(B, self.dof)
        guess = None if initial_guess is None else _f64(initial_guess).reshape(B, self.dof)
        sol = np.empty((B, self.dof))
        status = np.empty(B, dtype=np.int32)
        cost = np.empty(B)
        stats = np.zeros(B, dtype=STATS_DTYPE)
        cb = cost_fn if isinstance(cost_fn, COST_FN) else cost_callback(cost_fn)
        self._chk(self._L.pikamd_solve_batch_host(self._h, C.byref(params), B, _dp(goal), _dp(seed),
                                                  None if guess is None else _dp(guess), C.c_uint64(rng_seed),
                                                  problem_offset, cb, None, _dp(sol), _ip(status), _dp(cost),
                                                  stats.ctypes.data_as(C.c_void_p)))
        return sol, status, cost, stats

    def solve_batches(self, params: Params, batches, rng_seed: int = 0, job: int | None = None):
        """Several batches as one pool (pikamd_solve_batches).  batches: sequence of
        (goal_pos_quat, seed, initial_guess or None, problem_offset); returns a list of
        (solution, status, cost, stats).  With `job` the call is asynchronous
        (pikamd_solve_batches_async): the results are valid after wait(job)."""
        self._env_options()
        made = [self._host_batch(g, sd, ig, off) for g, sd, ig, off in batches]
        arr = (Batch * len(made))(*[m[0] for m in made])
        if job is None:
            self._chk(self._L.pikamd_solve_batches(self._h, C.byref(params), len(made), arr,
                                                   C.c_uint64(rng_seed)))
        else:
            self._chk(self._L.pikamd_solve_batches_async(self._h, C.byref(params), len(made), arr,
                                                         C.c_uint64(rng_seed), job))
            self._jobs = getattr(self, "_jobs", {})
            self._jobs[job] = made  # outputs must stay alive until wait()
        return [m[2] for m in made]

    def wait(self, job: int):
        self._chk(self._L.pikamd_wait(self._h, job))
        getattr(self, "_jobs", {}).pop(job, None)

    def solve_batch_device(self, params: Params, B: int, d_goal: int, d_seed: int, d_solution: int,
                           d_status: int, d_cost: int = 0, d_stats: int = 0, rng_seed: int = 0,
                           problem_offset: int = 0, stream: int = 0, slot: int = 0):
        """Enqueue a solve on HBM-resident buffers (raw device addresses, e.g. tensor.data_ptr());
        returns immediately -- the caller synchronises the stream."""
        self._env_options()
        self._chk(self._L.pikamd_solve_batch_device(
            self._h, C.byref(params), B, d_goal, d_seed, C.c_uint64(rng_seed), problem_offset,
            d_solution, d_status, d_cost or None, d_stats or None, stream or None, slot))

    def solve_batches_device(self, params: Params, batches, rng_seed: int = 0, stream: int = 0,
                             slot: int = 0):
        """Enqueue several HBM-resident batches as ONE pool (pikamd_solve_batches_device).
        batches: sequence of dicts / Batch with raw device addresses."""
        self._env_options()
        arr = (Batch * len(batches))(*[b if isinstance(b, Batch) else Batch(**b) for b in batches])
        self._chk(self._L.pikamd_solve_batches_device(self._h, C.byref(params), len(batches), arr,
                                                      C.c_uint64(rng_seed), stream or None, slot))

    def reserve(self, params: Params, B: int, slot: int = 0, stream: int = 0):
        """Allocate a slot's scratch + upload constants ahead of the first solve on it."""
        self._env_options()
        self._chk(self._L.pikamd_reserve(self._h, C.byref(params), B, slot, stream or None))

    def fk_device(self, n: int, d_q: int, d_pos_quat: int, stream: int = 0):
        self._chk(self._L.pikamd_fk_batch_device(self._h, n, d_q, d_pos_quat, stream or None))

    def kernel_name(self, params: Params) -> str:
        return self._L.pikamd_kernel_name(self._h, C.byref(params)).decode()


def shard_bounds(total: int, rank: int, world: int, strict: bool = False):
    """pikamd_shard_bounds: the contiguous range [lo, hi) device `rank` of `world` solves"""
    lo, hi = C.c_int64(), C.c_int64()
    lib(strict).pikamd_shard_bounds(total, rank, world, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def solve_batch_sharded(solvers, params: Params, goal_pos_quat, seed, rng_seed: int = 0, problem_offset: int = 0,
                        initial_guess=None):
    """pikamd_solve_batch_sharded: one batch over several handles (one per GPU), host arrays in and out."""
    s0 = solvers[0]
    for s in solvers:
        s._env_options()
    goal = _f64(goal_pos_quat).reshape(-1, 7 * s0.n_tips)
    B = goal.shape[0]
    seed = _f64(seed).reshape(B, s0.dof)
    guess = None if initial_guess is None else _f64(initial_guess).reshape(B, s0.dof)
    sol = np.empty((B, s0.dof))
    status = np.empty(B, dtype=np.int32)
    cost = np.empty(B)
    stats = np.zeros(B, dtype=STATS_DTYPE)
    arr = (C.c_void_p * len(solvers))(*[s._h for s in solvers])
    _check(s0._L.pikamd_solve_batch_sharded(arr, len(solvers), C.byref(params), B, _dp(goal), _dp(seed),
                                            None if guess is None else _dp(guess), C.c_uint64(rng_seed),
                                            problem_offset, _dp(sol), _ip(status), _dp(cost),
                                            stats.ctypes.data_as(C.c_void_p)), s0._L)
    return sol, status, cost, stats


def ik_memetic(solver: Solver, initial_guess, goal_pos_quat, params: Params | None = None,
               approx_solution: bool = False, rng_seed: int = 0):
    """Batch form of pick_ik::ik_memetic (src/ik_memetic.cpp:285-373): returns
    (solutions [B][dof], has_value [B]) where has_value mirrors std::optional."""
    p = params or default_params()
    p.mode = 0
    p.return_approximate_solution = int(approx_solution)
    sol, status, _, _ = solver.solve_batch(p, goal_pos_quat, initial_guess, rng_seed=rng_seed)
    return sol, status > 0


def ik_gradient(solver: Solver, initial_guess, goal_pos_quat, params: Params | None = None,
                approx_solution: bool = False):
    """Batch form of pick_ik::ik_gradient (src/ik_gradient.cpp:96-139)."""
    p = params or default_params()
    p.mode = 1
    p.return_approximate_solution = int(approx_solution)
    sol, status, _, _ = solver.solve_batch(p, goal_pos_quat, initial_guess)
    return sol, status > 0
