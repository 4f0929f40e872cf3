import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pick_ik_amd as pk
from pick_ik_amd import robots
from tests.test_gpu_fuzz import random_chain, random_params
import tests.test_gpu_product_arithmetic as T
def check(ch, tag):
    rng2 = np.random.default_rng(5)
    lo = np.where(ch.bounded == 1, ch.qmin, -3.0); hi = np.where(ch.bounded == 1, ch.qmax, 3.0)
    q = rng2.uniform(lo, hi, size=(64, ch.dof)); cand = rng2.uniform(lo, hi, size=(64, ch.dof))
    s = pk.Solver(ch, device=0)
    goal = s.fk(q)
    for kw in (dict(), dict(center_joints_weight=0.1, avoid_joint_limits_weight=0.1, minimal_displacement_weight=0.1)):
        gc, gs = s.cost(pk.default_params(**kw), goal, cand, cand)
        hsol, hst, hcost, _ = T._host(ch, dict(kw, mode=2), goal, cand, 0, 0)
        print(tag, "goals" if kw else "plain", "cost identical:", np.array_equal(gc, hcost), "n diff", int((gc != hcost).sum()), "jt", list(ch.joint_type), "bounded", list(ch.bounded),
              "axes", np.round(ch.axis, 2).tolist() if ch.dof <= 3 else "")
    s.close()
check(robots.panda(), "panda"); check(robots.ur5(), "ur5")
for i in range(8):
    rng = np.random.default_rng(0x9A0 + i)
    ch = random_chain(rng, 2 + i)
    while any(t not in (robots.REVOLUTE, robots.PRISMATIC) for t in ch.joint_type):
        ch = random_chain(rng, 2 + i)
    check(ch, f"chain{i}")
