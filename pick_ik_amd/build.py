"""Builds pick_ik_amd/libpick_ik_amd.so (+ the strict verification build) from csrc/ with hipcc for
gfx950, in-tree so that the .so travels with the repository snapshot to the GPU box.

The library is one translation unit for the C ABI (pik_amd.hip) plus one per supported chain length
(pik_inst.hip compiled with -DPIK_INST_D=1..16): the kernels are templates over the number of
joints, and a single translation unit took 2.5 minutes; the seventeen objects build in parallel and
only the ones whose inputs changed are rebuilt.  Objects live in pick_ik_amd/_build/ (git-ignored).
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import re
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libpick_ik_amd.so")
# Verification build of the SAME sources: no FMA contraction, generic joint rotations -- IEEE
# arithmetic in the reference's operation order, bit-comparable with the CPU oracle's portable
# math mode (tests/test_gpu_strict_parity.py).  ~2x slower.
LIB_STRICT = os.path.join(_HERE, "libpick_ik_amd_strict.so")
HEADERS = ["pik_kernels.hpp", "pik_math.hpp", "pik_host.hpp", "pik_solver.hpp", "pik_launch.hpp", "pik_exact.hpp",
           "pik_host_solve.hpp"]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "pick_ik_amd.h")
DOFS = tuple(range(1, 17))
BUILD_DIR = os.path.join(_HERE, "_build")


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


COMMON_FLAGS = ["-DPIK_COMMON=1"]  # third flavour: the kernels specialised for the common configuration (pik_math.hpp)
# ... and the same with the joint goals left in (center / avoid-limits / minimal-displacement weights)
COMMON_GOALS_FLAGS = ["-DPIK_COMMON=1", "-DPIK_NO_GOALS=0"]


def _flavor_flags(strict: bool):
    # product build: -ffp-contract=on, i.e. a * b + c is fused where the SOURCE writes it as one
    # expression and nowhere else.  hipcc's default (fast) lets the backend fuse across statements
    # depending on the surrounding code, so the same inlined function could round differently in two
    # kernel variants (seen when the one-lane kernel gained a second evaluation path: its pose cost
    # was contracted differently from the 2..16-lane kernels').  1-2 % slower than fast, and the
    # bit-identity of the variants holds by construction instead of by luck.
    flags = ["-DPIK_STRICT=1", "-ffp-contract=off"] if strict else ["-ffp-contract=on"]
    return flags + os.environ.get("PIK_EXTRA_HIPCC_FLAGS", "").split()  # (experiments only)


def _objects(strict: bool):
    """(object path, source, extra flags) of every translation unit of one flavor"""
    d = os.path.join(BUILD_DIR, "strict" if strict else "fast")
    only = os.environ.get("PIK_ONLY_D")  # experiments: kernels for these chain lengths only, e.g. "6,7"
    keep = {int(x) for x in only.split(",")} if only else set(DOFS)
    objs = [(os.path.join(d, "pik_amd.o"), "pik_amd.hip", [])]
    # the host solver for queries with a host cost function: an exact flavour's arithmetic for the host
    # (product library: the fused one, pik_exact_host_solve; verification library: its own, pik_strict_host_solve)
    objs.append((os.path.join(d, "pik_host_solve.o"), "pik_host_solve.hip",
                 ["--cuda-host-only"] + ([] if strict else EXACT_FLAGS)))
    for n in DOFS:
        extra = [f"-DPIK_INST_D={n}"] + ([] if n in keep else ["-DPIK_INST_STUB=1"])
        objs.append((os.path.join(d, f"pik_inst_d{n}.o"), "pik_inst.hip", extra))
    return objs


def _is_strict_obj(o) -> bool:
    return os.sep + "strict" + os.sep in o[0]


# the exact flavour with fused multiply-adds at stated places (pik_math.hpp PIK_XF, namespace pik_exact): the
# literal kernels the PRODUCT library links (option arithmetic = exact; chains with a floating joint)
EXACT_FLAGS = ["-DPIK_STRICT=1", "-DPIK_EXACT_FMA=1", "-ffp-contract=off"]


def _exact_objects():
    d = os.path.join(BUILD_DIR, "exact")
    only = os.environ.get("PIK_ONLY_D")
    keep = {int(x) for x in only.split(",")} if only else set(DOFS)
    return [(os.path.join(d, f"pik_inst_d{n}.o"), "pik_inst.hip",
             [f"-DPIK_INST_D={n}"] + (EXACT_FLAGS if n in keep else ["-DPIK_INST_STUB=1"] + EXACT_FLAGS)) for n in DOFS]


def _is_exact_obj(o) -> bool:
    return os.sep + "exact" + os.sep in o[0]


def _common_objects(goals: bool = False):
    """the per-length objects of the common-configuration flavours (fast flags + -DPIK_COMMON=1 [-DPIK_NO_GOALS=0])"""
    d = os.path.join(BUILD_DIR, "common_goals" if goals else "common")
    fl = COMMON_GOALS_FLAGS if goals else COMMON_FLAGS
    only = os.environ.get("PIK_ONLY_D")
    keep = {int(x) for x in only.split(",")} if only else set(DOFS)
    return [(os.path.join(d, f"pik_inst_d{n}.o"), "pik_inst.hip",
             [f"-DPIK_INST_D={n}"] + (fl if n in keep else ["-DPIK_INST_STUB=1"] + fl)) for n in DOFS]


def _cmd(obj, src, extra, strict):
    # (the exact objects carry their whole flavour in `extra`: no -ffp-contract=on in front of it)
    bare = "-DPIK_EXACT_FMA=1" in extra
    # -O2, not -O3: measured identical throughput (4.13 M vs 4.13 M solves/s, 15.86 vs 15.82 ms), and
    # -O3 miscompiles the heaviest strict kernel (multi-tip, nine joints: wrong solutions / counters
    # that came and went with unrelated edits; -O1 and -O2 builds of the same source are bit-exact)
    # --offload-compress: the device code of an object is stored compressed (the library is 2.3x smaller; the
    # HIP runtime unpacks a code object when it is first used)
    flavour = os.environ.get("PIK_EXTRA_HIPCC_FLAGS", "").split() if bare else _flavor_flags(strict)
    # -Rpass-analysis=kernel-resource-usage: registers / spills / scratch / occupancy of every kernel go to
    # stderr (kept beside the object as <obj>.res: the resource ledger, see write_ledger)
    return [hipcc(), "--offload-arch=gfx950", "--offload-compress", "-O2", "-std=c++17", "-fPIC", "-c", *flavour,
            "-Rpass-analysis=kernel-resource-usage", *extra, "-o", obj, os.path.join(CSRC, src)]


def _stamp(cmd):
    return hashlib.sha1(" ".join(cmd).encode()).hexdigest()


def _obj_stale(obj, src, extra, strict) -> bool:
    if not os.path.exists(obj) or not os.path.exists(obj + ".cmd"):
        return True
    if open(obj + ".cmd").read() != _stamp(_cmd(obj, src, extra, strict)):
        return True
    deps = _deps(src, "-DPIK_STRICT=1" in _cmd(obj, src, extra, strict))
    # what the object was compiled from, comments and blank space apart (a reworded comment in a header every
    # translation unit includes is not 84 recompilations)
    if os.path.exists(obj + ".deps"):
        return open(obj + ".deps").read() != _code_hash(deps)
    t = os.path.getmtime(obj)
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    with open(obj + ".deps", "w") as f:  # (an object from before this rule: by modification time, once)
        f.write(_code_hash(deps))
    return False


def _strip_comments(text: str) -> str:
    """C / C++ source without its comments (string and character literals respected), blank space collapsed"""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


_code_hash_cache = {}


def _code_hash(files) -> str:
    h = hashlib.sha1()
    for f in files:
        key = (f, os.path.getmtime(f))
        if key not in _code_hash_cache:
            _code_hash_cache[key] = hashlib.sha1(_strip_comments(open(f).read()).encode()).hexdigest()
        h.update(_code_hash_cache[key].encode())
    return h.hexdigest()


_INCLUDE = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _deps(src, strict_flags: bool):
    """the files one translation unit includes, found by following its #include "..." lines; pik_exact.hpp is only
    read by the exact flavours (it sits behind #if defined(PIK_STRICT) in pik_kernels.hpp), so an edit of it leaves
    the objects of the fast flavours alone"""
    seen, todo = [], [os.path.join(CSRC, src)]
    while todo:
        f = os.path.realpath(todo.pop())
        if f in seen or not os.path.exists(f):
            continue
        seen.append(f)
        for inc in _INCLUDE.findall(open(f).read()):
            if os.path.basename(inc) == "pik_exact.hpp" and not strict_flags:
                continue
            todo.append(os.path.join(os.path.dirname(f), inc))
    return seen


def _sources():
    return ([os.path.join(CSRC, f) for f in ("pik_amd.hip", "pik_inst.hip", "pik_urdf.hpp", *HEADERS)] +
            [HEADER, os.path.abspath(__file__)])


def _lib_stamp(strict: bool) -> str:
    """what a library was linked from: flavour flags + the chain lengths with real kernels"""
    flags = _flavor_flags(strict) + ([] if strict else ["+exact:"] + EXACT_FLAGS + ["+common:"] + COMMON_FLAGS + ["+common_goals:"] + COMMON_GOALS_FLAGS)
    return _stamp(flags + ["only=" + os.environ.get("PIK_ONLY_D", "all")])


def is_stale(lib: str = LIB) -> bool:
    """A library is up to date when it is newer than every source AND was linked with the default
    flags for all chain lengths (an experiment build -- PIK_ONLY_D / PIK_EXTRA_HIPCC_FLAGS -- leaves a
    library that is newer than the sources but is not the product; its stamp gives it away).  The
    objects in _build/ are only a cache (they do not travel to the GPU box: the prebuilt .so files do,
    and must not be rebuilt there just because the cache is absent)."""
    if not os.path.exists(lib) or not os.path.exists(lib + ".stamp"):
        return True
    if open(lib + ".stamp").read() != _lib_stamp(lib == LIB_STRICT):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _sources())


def _compile(obj, src, extra, strict, verbose):
    os.makedirs(os.path.dirname(obj), exist_ok=True)
    cmd = _cmd(obj, src, extra, strict)
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)}\n{r.stderr[-4000:]}")
    with open(obj + ".res", "w") as f:
        f.write(r.stderr)
    with open(obj + ".cmd", "w") as f:
        f.write(_stamp(cmd))
    with open(obj + ".deps", "w") as f:
        f.write(_code_hash(_deps(src, "-DPIK_STRICT=1" in cmd)))


def _link(lib, objs, verbose):
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib + ".tmp", *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(lib + ".tmp", lib)


#: kernel namespace (what pikamd_kernel_name reports in front of "::") -> the flags of that flavour's objects
FLAVOUR_FLAGS = {
    "pik": ["-ffp-contract=on"],
    "pik_common": ["-ffp-contract=on"] + COMMON_FLAGS,
    "pik_common_goals": ["-ffp-contract=on"] + COMMON_GOALS_FLAGS,
    "pik_exact": EXACT_FLAGS,
    "pik_strict": ["-DPIK_STRICT=1", "-ffp-contract=off"],
}
_flavour_sha_cache = {}


def flavour_sha(namespace: str, dof: int = 7) -> str:
    """Hash of the DEVICE source of one kernel flavour: pik_inst.hip preprocessed with the flavour's flags
    (`hipcc -E`, device side), reduced to the lines that come from this repository's own files (csrc/, include/).
    Comments and the branches the flavour's macros switch off are not in it, so an edit of pik_exact.hpp or of a
    PIK_STRICT block leaves the hash of the fast flavours alone.  The flags are part of the hash.  bench.py compares it
    with the hash stored beside the PMC counters of a flavour (profiles/roofline_inputs.json: roofline.inputs_stale)."""
    key = (namespace, dof)
    if key in _flavour_sha_cache:
        return _flavour_sha_cache[key]
    flags = FLAVOUR_FLAGS[namespace]
    cmd = [hipcc(), "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-E", *flags, f"-DPIK_INST_D={dof}",
           os.path.join(CSRC, "pik_inst.hip"), "-o", "-"]
    r = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)}\n{r.stderr[-2000:]}")
    own = (os.path.realpath(CSRC) + os.sep, os.path.realpath(HEADER))
    h = hashlib.sha256(" ".join(flags).encode())
    keep = False
    for line in r.stdout.splitlines():
        if line.startswith("# ") and '"' in line:  # line marker: # <n> "<file>" <flags>
            f = line.split('"')[1]
            rp = os.path.realpath(f if os.path.isabs(f) else os.path.join(CSRC, f))
            keep = rp.startswith(own[0]) or rp == own[1]
            continue
        if keep and line.strip():
            h.update(line.strip().encode())
            h.update(b"\n")
    _flavour_sha_cache[key] = h.hexdigest()[:16]
    return _flavour_sha_cache[key]


LEDGER = os.path.join(BUILD_DIR, "kernel_resources.csv")


def ledger_rows():
    """(flavour, kernel, {field: value}) of every kernel of every object the two libraries link, from the
    compiler's resource remarks kept beside the objects; None when an object has none (not built here)"""
    from . import kernel_resources as KR
    rows = []
    for o in _objects(False) + _exact_objects() + _common_objects() + _common_objects(True) + _objects(True):
        res = o[0] + ".res"
        if o[1] != "pik_inst.hip":
            continue
        if not os.path.exists(res):
            return None
        flavour = os.path.basename(os.path.dirname(o[0]))
        rows += KR.rows_of(open(res).read(), flavour)
    return rows


def write_ledger(path: str = LEDGER):
    from . import kernel_resources as KR
    rows = ledger_rows()
    if rows is None:
        return None
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("flavour,kernel," + ",".join(KR.FIELDS) + "\n")
        for fl, name, v in sorted(rows, key=lambda r: (r[0], r[1])):
            f.write(f'{fl},"{name}",' + ",".join(str(v[k]) for k in KR.FIELDS) + "\n")
    return path


def build_library(force: bool = False, verbose: bool = False, strict_too: bool = True) -> str:
    """Builds the product library and the strict-arithmetic verification library.

    The product library also links the EXACT kernels (the per-length objects of the exact flavour with fused
    multiply-adds, namespace pik_exact): they serve the option arithmetic = exact and the chains the
    Denavit-Hartenberg form cannot express -- a floating joint (pik_amd.hip ops_of)."""
    flavors = [(LIB, False)] + ([(LIB_STRICT, True)] if strict_too else [])
    jobs, relink = [], []
    for lib, strict in flavors:
        if not (force or is_stale(lib) or os.environ.get("PIK_ONLY_D") or os.environ.get("PIK_EXTRA_HIPCC_FLAGS")):
            continue
        objs = _objects(strict)
        if not strict:  # + the exact kernels + the common-configuration kernels
            objs = objs + _exact_objects() + _common_objects() + _common_objects(True)
        stale = [o for o in objs if force or _obj_stale(*o, _is_strict_obj(o))]
        jobs += [(o, _is_strict_obj(o)) for o in stale if (o, _is_strict_obj(o)) not in jobs]
        relink.append((lib, [o[0] for o in objs]))
    if jobs:
        # the per-length objects take longest for the long chains: start those first
        def length_of(job):  # (-DPIK_INST_D=<n> of a per-length object; the other objects first)
            for fl in job[0][2]:
                if fl.startswith("-DPIK_INST_D="):
                    return -int(fl.split("=")[1])
            return -100
        jobs.sort(key=length_of)
        with cf.ThreadPoolExecutor(max_workers=max(1, os.cpu_count() or 1)) as ex:
            for f in [ex.submit(_compile, *o, strict, verbose) for o, strict in jobs]:
                f.result()
    for lib, objs in relink:
        _link(lib, objs, verbose)
        with open(lib + ".stamp", "w") as f:
            f.write(_lib_stamp(lib == LIB_STRICT))
    if relink and not os.environ.get("PIK_ONLY_D"):
        write_ledger()
    return LIB


if __name__ == "__main__":
    import sys
    if "--ledger" in sys.argv:  # the committed copy of the resource ledger: python -m pick_ik_amd.build --ledger <csv>
        build_library()
        print(write_ledger(sys.argv[sys.argv.index("--ledger") + 1]))
    else:
        print(build_library(force="--force" in sys.argv, verbose=True, strict_too="--fast-only" not in sys.argv))
