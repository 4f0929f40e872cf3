#!/usr/bin/env python3
"""Static "profile" of one kernel of the product build: instructions per source line (needs only
hipcc, no GPU).

usage: tools/line_profile.py <kernel-substring> <first line> <last line> [--spills] [--top N]
  e.g. tools/line_profile.py memetic_kernelILi7ELi1ELb0ELi2E 330 545     (the one-lane gradient descent)

Compiles pik_inst.hip device-only to assembly with line tables (-gline-tables-only; PIK_ISA_D picks the
chain length, default 7), and attributes every instruction of the named kernel to the INNERMOST source
line of its inlined-at chain -- restricted to instructions whose chain passes through
pik_kernels.hpp:[first, last], i.e. to one loop of the kernel.  Prints the lines with the most
instructions and their vector / scalar / memory split.  --spills counts only v_readlane_b32 (SGPR
spill reloads).

It is a STATIC count (code on untaken paths is included; a line inside a loop counts once), but it is
what found this round's arithmetic savings: the library sqrt, the if-converted 2-pi fold, the clamp,
the atan2 select chains and the constants copied into vector registers all showed up as lines with
more vector instructions than arithmetic."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pick_ik_amd import build  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
if len(args) < 3:
    sys.exit(__doc__)
kern, flo, fhi = args[0], int(args[1]), int(args[2])
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 50
if "--top" in sys.argv:
    args = [a for a in args if a != sys.argv[sys.argv.index("--top") + 1]]
spills_only = "--spills" in sys.argv
out = "/tmp/isa/pik_fast_g.s"
os.makedirs("/tmp/isa", exist_ok=True)
if "--reuse" not in sys.argv:
    cmd = [build.hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", *build._flavor_flags(False),
           "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-gline-tables-only", "-o", out,
           "-DPIK_INST_D=" + os.environ.get("PIK_ISA_D", "7"),
           os.path.join(ROOT, "pick_ik_amd", "csrc", "pik_inst.hip")]
    cmd += os.environ.get("PIK_EXTRA_HIPCC_FLAGS", "").split()  # e.g. -DPIK_COMMON=1: the specialised flavour
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN\d+pik\w*" + re.escape(kern) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
src = {f: open(os.path.join(ROOT, "pick_ik_amd", "csrc", f)).read().splitlines()
       for f in ("pik_math.hpp", "pik_kernels.hpp")}
cur = None
inner = collections.Counter()
kinds = collections.defaultdict(collections.Counter)
total = 0
for l in lines[start:end]:
    if re.match(r"\s+\.loc\s", l):
        cur = l
        continue
    s = l.strip()
    if not s or s.startswith((".", ";")) or s.endswith(":") or cur is None:
        continue
    chain = re.findall(r"(pik_\w+\.hpp|__clang_hip_math\.h|amd_\w+\.h):(\d+)", cur)
    if not any(f == "pik_kernels.hpp" and flo <= int(n) <= fhi for f, n in chain):
        continue
    op = s.split()[0]
    if spills_only and not op.startswith("v_readlane"):
        continue
    total += 1
    kind = "vector" if op.startswith("v_") else "scalar" if op.startswith("s_") else "memory"
    inner[chain[0]] += 1
    kinds[chain[0]][kind] += 1
print(f"{kern}: {total} instructions attributed to pik_kernels.hpp:{flo}-{fhi}")
for (f, n), c in inner.most_common(top):
    text = src.get(f, [])
    line = text[int(n) - 1].strip()[:88] if 0 < int(n) <= len(text) else ""
    k = kinds[(f, n)]
    print(f"{c:5d}  v{k['vector']:4d} s{k['scalar']:4d} m{k['memory']:3d}  {f[:16]:16s}:{n:>5s} | {line}")
