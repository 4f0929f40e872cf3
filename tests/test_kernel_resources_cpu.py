"""Kernel resource ledger (CPU; needs only the build).  Every object of the two libraries is compiled with
-Rpass-analysis=kernel-resource-usage and pick_ik_amd/build.py keeps the remarks; the ledger of every shipped kernel
(registers, spills, scratch bytes per lane, occupancy, LDS) is committed as profiles/r06_kernel_resources.csv.  A
kernel whose scratch or spill counts GROW past the committed figures fails here -- the kernels for long chains sit
at the register cap and have come out of the compiler wrong four times after edits elsewhere;
what is committed is what the GPU tests have seen pass.  Re-take the ledger with
`python -m pick_ik_amd.build --ledger profiles/r06_kernel_resources.csv` after a deliberate change."""
import csv
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMITTED = os.path.join(ROOT, "profiles", "r06_kernel_resources.csv")
GROW = ("scratch_bytes_per_lane", "vgpr_spills", "sgpr_spills")


def _committed():
    return {(r["flavour"], r["kernel"]): r for r in csv.DictReader(open(COMMITTED))}


def _current():
    import __graft_entry__ as g
    g.build()
    from pick_ik_amd import build as B
    rows = B.ledger_rows()
    if rows is None:
        pytest.skip("no compiler remarks beside the objects (libraries built elsewhere)")
    return {(fl, k): v for fl, k, v in rows}


def test_no_shipped_kernel_grew_past_the_committed_ledger():
    want, have = _committed(), _current()
    missing = sorted(set(have) - set(want))
    assert not missing, f"kernels without a committed record (re-take the ledger): {missing[:5]}"
    grown = []
    for key, v in have.items():
        for f in GROW:
            if int(v[f]) > int(want[key][f]):
                grown.append((key, f, int(want[key][f]), int(v[f])))
    assert not grown, f"grew past the committed ledger: {grown[:8]}"


def test_two_per_simd_kernels_keep_their_occupancy():
    """memetic_kernel<D, 1, false, 2> exists for two wavefronts per SIMD: 256 registers, NO AGPR (a single one
    in a shared callee put the exact flavour's build back to one wavefront), occupancy 2 -- every flavour,
    D = 1..9"""
    rows = _committed()
    seen = 0
    for (fl, k), r in rows.items():
        if "memetic_kernel<" in k and k.endswith(",1,false,2>"):
            seen += 1
            assert int(r["occupancy"]) >= 2 and int(r["agprs"]) == 0, (fl, k, r)
    assert seen == 5 * 9


def test_ledger_covers_every_flavour_and_length():
    rows = _committed()
    for fl, ns in (("fast", "pik"), ("common", "pik_common"), ("common_goals", "pik_common_goals"),
                   ("exact", "pik_exact"), ("strict", "pik_strict")):
        for d in range(1, 17):
            assert (fl, f"{ns}::memetic_kernel<{d},1,false,1>") in rows, (fl, d)
            assert (fl, f"{ns}::ik_gradient_kernel<{d},false>") in rows, (fl, d)
    # the exact flavours' kernels carry a stack (their evaluations are calls); the product's one-per-SIMD
    # kernels carry none
    for (fl, k), r in rows.items():
        if fl in ("fast", "common", "common_goals") and "memetic_kernel<" in k and k.endswith(",1>"):
            assert int(r["scratch_bytes_per_lane"]) == 0 or int(k.split("<")[1].split(",")[0]) >= 10, (fl, k, r)
