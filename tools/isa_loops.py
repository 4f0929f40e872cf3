#!/usr/bin/env python3
"""Instruction classes of one device function, loop by loop (needs only hipcc, no GPU).

usage: tools/isa_loops.py <flavour> <D> <function-substring> [--reuse] [--trips a,b,c...] [--lines Ln [--top N]] [extra hipcc flags]
  flavour = exact | plain | fast | common         (the flags of pick_ik_amd/build.py's objects)
  function-substring matches the DEMANGLED name, spaces removed, e.g. "gradient_descent_exact<7,0,1,2,3>"

The function's assembly is cut into its natural loops (a backward branch to a label closes a loop; nesting by
containment) and every loop's OWN instructions (its nested loops apart) are counted per class.  --trips gives the
trip count of every loop in the order printed (default 1 for the function body), so that the last table is a
DYNAMIC estimate per call: classes x trips, the figure the static whole-function count cannot give (the rolled joint
loops of the fork run 21 + 7 times per descent step).  Loops whose trip count is data dependent are given their mean.
--lines Ln: the instructions of loop Ln (its nested loops included; "body" = the whole function) per innermost SOURCE
line of their inlined-at chain (the assembly is then compiled with -gline-tables-only), classes beside them.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.environ.get("PIK_ISA_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # (PIK_ISA_ROOT: another source tree, e.g. `git archive` of an earlier round)
TAG = os.environ.get("PIK_ISA_TAG", "")
FLAVOURS = {
    "exact": ["-DPIK_STRICT=1", "-DPIK_EXACT_FMA=1", "-ffp-contract=off"],
    "plain": ["-DPIK_STRICT=1", "-ffp-contract=off"],
    "fast": ["-ffp-contract=on"],
    "common": ["-ffp-contract=on", "-DPIK_COMMON=1"],
}


def classify(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane r/w (sgpr spill)"
    if op.startswith("v_accvgpr"):
        return "v_accvgpr (vgpr spill)"
    if re.match(r"v_(fma|fmac|mul|add|max|min)_f64|v_pk_(fma|mul|add)_f32", op):
        return "fp64 arith"
    if re.match(r"v_(rcp|rsq|sqrt|div_|ldexp|rndne|cvt|frexp|trig|fract|cmp_class|floor|ceil|trunc).*", op):
        return "fp64 special"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith("v_mov"):
        return "v_mov"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if op.startswith("v_"):
        return "valu int/bit"
    if op.startswith(("s_load", "s_buffer")):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_getpc")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "vmem"
    return "other"


VALU = ("fp64", "v_", "valu", "lane")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flavour, D, want = args[0], args[1], args[2].replace(" ", "")
    extra = [a for a in args[3:]]
    trips = None
    if "--trips" in sys.argv:
        trips = [float(x) for x in sys.argv[sys.argv.index("--trips") + 1].split(",")]
        extra = [a for a in extra if a != sys.argv[sys.argv.index("--trips") + 1]]
    os.makedirs("/tmp/isa", exist_ok=True)
    by_line = None
    if "--lines" in sys.argv:
        by_line = sys.argv[sys.argv.index("--lines") + 1]
        extra = [a for a in extra if a != by_line] + ["-gline-tables-only"]
    top = 40
    if "--top" in sys.argv:
        top = int(sys.argv[sys.argv.index("--top") + 1])
        extra = [a for a in extra if a != str(top)]
    out = f"/tmp/isa/{TAG}{flavour}_d{D}{'_g' if by_line else ''}.s"
    if "--reuse" not in sys.argv or not os.path.exists(out):
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", *FLAVOURS[flavour],
               "--cuda-device-only", "-S", "-o", out, f"-DPIK_INST_D={D}", *extra,
               os.path.join(ROOT, "pick_ik_amd", "csrc", "pik_inst.hip")]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
    syms = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    dem = subprocess.run(["c++filt"], input="\n".join(s for _, s in syms), capture_output=True, text=True).stdout.splitlines()
    hit = [(i, s, d) for (i, s), d in zip(syms, dem) if want in d.replace(" ", "")]
    if not hit:
        raise SystemExit(f"no function matches {want}")
    start, sym, name = hit[0]
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith((".Lfunc_end", "s_endpgm")) and i > start)
    # instructions with their label positions
    insts, labels, locs, cur = [], {}, [], None
    for l in lines[start + 1:end + 1]:
        if re.match(r"\s+\.loc\s", l):
            cur = l
            continue
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\s*(.*)", l)
        if m and not l.strip().startswith("."):
            insts.append((m.group(1), m.group(2)))
            locs.append(cur)
    loops = set()
    for k, (op, rest) in enumerate(insts):
        if op.startswith(("s_cbranch", "s_branch")):
            t = rest.split()[0].rstrip(",") if rest else ""
            if t in labels and labels[t] <= k:
                loops.add((labels[t], k))
    # loops sharing a header: keep the widest (several back edges of one loop)
    by_head = {}
    for a, b in loops:
        by_head[a] = max(by_head.get(a, b), b)
    loops = sorted(by_head.items(), key=lambda ab: (ab[0], -ab[1]))
    nodes = [(-1, len(insts))] + [(a, b) for a, b in loops]  # node 0 = the function body
    depth = []
    for n, (a, b) in enumerate(nodes):
        depth.append(sum(1 for (c, d) in nodes[:n] if c <= a and b <= d and (c, d) != (a, b)))
    own = []
    for n, (a, b) in enumerate(nodes):
        inner = [(c, d) for m, (c, d) in enumerate(nodes) if m != n and a <= c and d <= b and (c, d) != (a, b)]
        c = collections.Counter()
        for k in range(max(a, 0), min(b + 1, len(insts))):
            if any(c0 <= k <= d0 for c0, d0 in inner):
                continue
            c[classify(insts[k][0])] += 1
        own.append(c)
    print(f"== {name.split('(')[0]}: {len(insts)} instructions, {len(nodes) - 1} loops")
    classes = sorted({k for c in own for k in c}, key=lambda k: -sum(c[k] for c in own))
    print("   loop (instruction range)        depth   own  " + "  ".join(f"{k[:12]:>12s}" for k in classes))
    for n, ((a, b), c) in enumerate(zip(nodes, own)):
        tag = "body" if n == 0 else f"L{n}"
        print(f"   {tag:5s} [{max(a, 0):6d},{min(b, len(insts)):6d}]  {depth[n]:5d} {sum(c.values()):6d}  " +
              "  ".join(f"{c[k]:12d}" for k in classes))
    if by_line:
        n = 0 if by_line == "body" else int(by_line[1:])
        a, b = nodes[n]
        src = {}
        per = collections.defaultdict(collections.Counter)
        for k in range(max(a, 0), min(b + 1, len(insts))):
            chain = re.findall(r"(pik_\w+\.hpp|__clang_hip_\w+\.h|amd_\w+\.h|\w+\.h):(\d+)", locs[k] or "")
            key = chain[0] if chain else ("?", "0")
            per[key][classify(insts[k][0])] += 1
        print(f"== source lines of {by_line} ({min(b, len(insts)) - max(a, 0)} instructions), innermost inlined frame")
        for (f, ln), c in sorted(per.items(), key=lambda kv: -sum(kv[1].values()))[:top]:
            path = os.path.join(ROOT, "pick_ik_amd", "csrc", f)
            if f not in src:
                src[f] = open(path).read().splitlines() if os.path.exists(path) else []
            text = src[f][int(ln) - 1].strip()[:70] if 0 < int(ln) <= len(src[f]) else ""
            tot = sum(c.values())
            fp = c["fp64 arith"] + c["fp64 special"]
            rest = ", ".join(f"{k.split(' ')[0]} {v}" for k, v in c.most_common() if not k.startswith("fp64"))
            print(f"{tot:5d}  fp64 {fp:4d}  {f[:14]:14s}:{ln:>5s} | {text:70s} | {rest}")
    if trips:
        trips = (trips + [1.0] * len(nodes))[:len(nodes)]
        tot = collections.Counter()
        for t, c in zip(trips, own):
            for k, v in c.items():
                tot[k] += t * v
        total = sum(tot.values())
        valu = sum(v for k, v in tot.items() if k.startswith(VALU))
        print(f"== dynamic estimate (trips {','.join(str(t) for t in trips)}): {total:.0f} instructions, {valu:.0f} vector")
        for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
            print(f"   {k:28s} {v:9.0f}  {100.0 * v / total:5.1f}% of all  {100.0 * v / max(valu, 1):5.1f}% of vector")


if __name__ == "__main__":
    main()
