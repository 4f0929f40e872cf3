import numpy as np, pick_ik_amd as pk
chain = pk.robots.panda()
s = pk.Solver(chain)
goal = s.fk(np.random.uniform(chain.qmin, chain.qmax, (4096, 7)))
seed = np.tile(pk.robots.PANDA_HOME, (4096, 1))
sol, status, cost, stats = s.solve_batch(pk.default_params(memetic_population_size=128), goal, seed, rng_seed=1)
print((status == pk.SUCCESS).mean())
f = pk.Solver(chain, exact=False)
e = s
sol, status, cost, stats = e.solve_batch_host(pk.default_params(memetic_population_size=32), goal[:8], seed[:8],
                                              lambda q, pose: 0.5 * (q[2] - 0.5) ** 2)
print(status, sol[:, 2])
