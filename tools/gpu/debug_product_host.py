import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pick_ik_amd as pk
from pick_ik_amd import robots
from tests.test_gpu_fuzz import random_chain, random_params
import tests.test_gpu_product_arithmetic as T
i = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(0x9A0 + i)
ch = random_chain(rng, 2 + i)
while any(t not in (robots.REVOLUTE, robots.PRISMATIC) for t in ch.joint_type):
    ch = random_chain(rng, 2 + i)
kw = random_params(rng)
kw.pop("memetic_num_threads", None); kw.pop("memetic_stop_on_first_solution", None)
print("jt", ch.joint_type, "bounded", ch.bounded, "axis", ch.axis.tolist())
print(kw)
for over in (dict(memetic_max_generations=1, memetic_gd_max_iters=0), dict(memetic_max_generations=1, memetic_gd_max_iters=1),
             dict(memetic_max_generations=1), dict(memetic_max_generations=2), dict()):
    k2 = dict(kw, **over)
    try:
        T._compare(ch, k2, 20, None, 100 + i, 3, str(over))
        print(over, "identical")
    except AssertionError as e:
        print(over, "DIFFERENT", str(e).splitlines()[3][:200] if len(str(e).splitlines()) > 3 else "")
# primitive: cost of random candidates, host vs GPU
rng2 = np.random.default_rng(5)
lo = np.where(ch.bounded == 1, ch.qmin, -3.0); hi = np.where(ch.bounded == 1, ch.qmax, 3.0)
q = rng2.uniform(lo, hi, size=(64, ch.dof)); cand = rng2.uniform(lo, hi, size=(64, ch.dof))
s = pk.Solver(ch, device=0)
goal = s.fk(q)
gc, gs = s.cost(pk.default_params(**kw), goal, cand, cand)
hsol, hst, hcost, _ = T._host(ch, dict(kw, mode=2), goal, cand, 0, 0)
print("cost identical:", np.array_equal(gc, hcost), "verdict:", np.array_equal(gs, hst), "max rel", np.abs(gc - hcost).max() / np.abs(gc).max())
print(np.c_[gc, hcost][:5])
