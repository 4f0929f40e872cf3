#!/usr/bin/env python3
"""Static instruction statistics of one kernel of the fast build (needs only hipcc, no GPU).

usage: tools/isa_stats.py [kernel-substring] [--strict]
Compiles pik_amd.hip device-only to assembly, cuts out the named kernel, and prints instruction
class counts for the whole kernel and for its largest depth-2 loop (the gradient-descent body of
memetic_kernel).  Used to track VALU overhead (v_readlane = SGPR spill reloads, v_accvgpr = VGPR
spills, v_mov/v_cndmask) against the FP64 arithmetic while tuning."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
name = args[0] if args else "memetic_kernelILi7ELi1E"
strict = "--strict" in sys.argv
out = "/tmp/isa/pik_strict.s" if strict else "/tmp/isa/pik_fast.s"
os.makedirs("/tmp/isa", exist_ok=True)
if "--reuse" not in sys.argv:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=on",
           "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out,
           "-DPIK_INST_D=" + os.environ.get("PIK_ISA_D", "7"),
           os.path.join(ROOT, "pick_ik_amd", "csrc", "pik_inst.hip")]
    if strict:
        cmd[5:6] = ["-DPIK_STRICT", "-ffp-contract=off"]
    cmd += os.environ.get("PIK_EXTRA_HIPCC_FLAGS", "").split()
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN\d+pik\w*" + re.escape(name) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end]


def classify(op):
    if op.startswith("v_readlane") or op.startswith("v_writelane"):
        return "sgpr-spill (v_read/writelane)"
    if op.startswith("v_accvgpr"):
        return "vgpr-spill (v_accvgpr)"
    if re.match(r"v_(fma|fmac|mul|add|max|min)_f64|v_pk_", op):
        return "fp64 arith"
    if re.match(r"v_(rcp|rsq|sqrt|div_|ldexp|rndne|cvt|frexp|trig|fract|cmp_class).*", op):
        return "fp64 special"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith("v_mov"):
        return "v_mov"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if op.startswith("v_"):
        return "valu other (int/bit)"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "vmem"
    return "other"


def stats(seg, title):
    c = collections.Counter()
    for l in seg:
        m = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
        if m:
            c[classify(m.group(1))] += 1
    valu = sum(v for k, v in c.items() if k.startswith(("fp64", "v_", "valu", "sgpr", "vgpr")))
    print(f"== {title}: {sum(c.values())} instructions, {valu} VALU")
    for k, v in c.most_common():
        print(f"   {k:34s} {v:6d}  {100.0 * v / max(valu, 1):5.1f}% of VALU")


stats(body, name)
# depth-2 loops: segments between consecutive depth<=2 loop headers; the largest one is the
# gradient-descent body (approximation: the tail after the loop's back edge up to the next header
# is counted with it)
heads = [i for i, l in enumerate(body) if re.search(r"=>\s*This (Inner )?Loop Header: Depth=[12]\b", l)]
segs = [(heads[k], heads[k + 1] if k + 1 < len(heads) else len(body)) for k in range(len(heads))
        if "Depth=2" in body[heads[k]]]
for a, b in sorted(segs, key=lambda ab: ab[0] - ab[1])[:3]:
    stats(body[a:b], f"depth-2 loop segment (lines {a}-{b})")
for l in lines:
    if name in l and ".amdhsa_kernel" in l:
        i = lines.index(l)
        for k in lines[i:i + 80]:
            if re.search(r"next_free_vgpr|next_free_sgpr|private_segment_fixed|group_segment_fixed", k):
                print("  ", k.strip())
        break
