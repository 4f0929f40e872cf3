import sys, os
sys.path.insert(0, ".")
import numpy as np
import pick_ik_amd as pk
from oracle import oracle as O
from tests.test_gpu_multi_tip import CHAINS, problems
ch = CHAINS["torso_dual_arm"]()
o, q, goal, sd = problems(O, ch, 96, 3)
s = pk.Solver(ch, device=0, strict=True)
kw = dict(memetic_population_size=24, memetic_max_generations=12)
os.environ["PIK_PASSES"] = "none"
with O.math_mode("portable"):
    a = s.solve_batch(pk.default_params(**kw), goal, sd, rng_seed=11, problem_offset=7)
    b = o.solve_batch(O.default_params(**kw), goal, sd, rng_seed=11, problem_offset=7, num_threads=8)
bad = a[3]["cost_evals"] != b[3]["cost_evals"]
print(os.environ.get("PIK_LIB_STRICT"), "bad evals", bad.sum(), [hex(int(x)) for x in a[3]["cost_evals"][:4]])
