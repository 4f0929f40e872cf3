"""pick_ik_amd/build.py: the rebuild rule.  An object is stale when the CODE of a file it includes changed -- comments
and blank space apart -- and the files a translation unit includes are found by following its #include lines
(pik_exact.hpp only for the exact flavours)."""
import os

from pick_ik_amd import build as B


def test_comments_and_blank_space_do_not_count_as_code():
    a = 'int a = 1; // one\n/* two */ const char* s = "x//y /*z*/"; char q = \'"\';   // "\nint b;'
    b = 'int a = 1;\nconst char* s = "x//y /*z*/";\n\n   char q = \'"\'; int b; /* trailing */'
    assert B._strip_comments(a) == B._strip_comments(b)
    assert B._strip_comments(a) != B._strip_comments(a.replace("int b", "long b"))
    assert '"x//y /*z*/"' in B._strip_comments(a)  # literals are not comments


def test_dependencies_follow_the_includes():
    base = lambda files: {os.path.basename(f) for f in files}
    fast = base(B._deps("pik_inst.hip", False))
    exact = base(B._deps("pik_inst.hip", True))
    assert {"pik_inst.hip", "pik_launch.hpp", "pik_kernels.hpp", "pik_math.hpp", "pik_host.hpp", "pick_ik_amd.h"} <= fast
    assert "pik_exact.hpp" not in fast and exact == fast | {"pik_exact.hpp"}
    abi = base(B._deps("pik_amd.hip", False))
    assert "pik_urdf.hpp" in abi and "pik_kernels.hpp" not in abi and "pik_host_solve.hpp" not in abi
    assert "pik_host_solve.hpp" in base(B._deps("pik_host_solve.hip", True))


def test_every_flavour_has_a_source_hash():
    # (hipcc -E of the device side: needs the compiler, no GPU)
    shas = {ns: B.flavour_sha(ns) for ns in B.FLAVOUR_FLAGS}
    assert all(len(v) == 16 for v in shas.values()) and len(set(shas.values())) == len(shas)
