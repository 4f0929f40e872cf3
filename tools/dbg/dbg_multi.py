import sys, os
sys.path.insert(0, ".")
import numpy as np
import pick_ik_amd as pk
from oracle import oracle as O
from tests.test_gpu_multi_tip import CHAINS, problems
for name in CHAINS:
    ch = CHAINS[name]()
    o, q, goal, sd = problems(O, ch, 96, 3)
    for strict in (True, False):
        s = pk.Solver(ch, device=0, strict=strict)
        kw = dict(memetic_population_size=24, memetic_max_generations=12)
        for marks in ("none", "1,2,3,5,8"):
            os.environ["PIK_PASSES"] = marks
            with O.math_mode("portable"):
                a = s.solve_batch(pk.default_params(**kw), goal, sd, rng_seed=11, problem_offset=7)
                b = o.solve_batch(O.default_params(**kw), goal, sd, rng_seed=11, problem_offset=7, num_threads=8)
            bad = a[3]["cost_evals"] != b[3]["cost_evals"]
            print(name, "strict" if strict else "fast", marks, "bad evals", bad.sum(), "of", len(bad),
                  [hex(int(x)) for x in a[3]["cost_evals"][:4]], [hex(int(x)) for x in b[3]["cost_evals"][:4]],
                  "gens eq", (a[3]["generations"] == b[3]["generations"]).mean())
        s.close()
