"""Small batches on the boundaries of a wavefront (0, 1, 2, 63, 64, 65 problems) x every lanes-per-elite
request x both joint layouts x memetic / local / species / joint-goal calls, one and several tip frames: the
answers never depend on the execution shape, whichever kernel flavour serves the call."""
import numpy as np
import pytest

import pick_ik_amd as pk
from tests.common import ARITHMETIC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("exact", ARITHMETIC)
@pytest.mark.parametrize("name", ["panda", "torso_dual_arm"])
def test_small_batches_every_shape(name, exact):
    import __graft_entry__ as g
    g.build()
    ch = pk.robots.by_name(name)
    s = pk.Solver(ch, device=0, exact=exact)
    rng = np.random.default_rng(1)
    flavours = set()
    try:
        for kw in (dict(memetic_population_size=32), dict(mode=1), dict(memetic_population_size=32, memetic_num_threads=2),
                   dict(memetic_population_size=32, minimal_displacement_weight=0.01, cost_threshold=0.05)):
            p = pk.default_params(**kw)
            flavours.add(s.kernel_name(p).split("::")[0])
            for B in (0, 1, 2, 63, 64, 65):
                q = rng.uniform(ch.qmin, ch.qmax, size=(B, ch.dof))
                goal = s.fk(q) if B else np.zeros((0, 7 * s.n_tips))
                seed = np.clip(q + 0.1, ch.qmin, ch.qmax)
                ref = None
                for lanes in ("1", "2", "8", "16", None):
                    for layout in (None, "soa"):
                        s.set_option("lanes_per_elite", lanes)
                        s.set_option("joint_layout", layout)
                        out = s.solve_batch(p, goal, np.ascontiguousarray(seed.T) if layout else seed, rng_seed=3)
                        sol = out[0].reshape(ch.dof, B).T if layout else out[0]
                        if ref is None:
                            ref = (sol, out[1], out[2])
                        what = f"{name} {kw} B {B} lanes {lanes} layout {layout}"
                        np.testing.assert_array_equal(sol, ref[0], err_msg=what)
                        np.testing.assert_array_equal(out[1], ref[1], err_msg=what)
                        np.testing.assert_array_equal(out[2], ref[2], err_msg=what)
                s.set_option("joint_layout", None)
                s.set_option("lanes_per_elite", None)
        assert flavours == ({"pik_common", "pik", "pik_common_goals"} if exact is False else {"pik_exact"}), flavours
    finally:
        s.close()
