#!/usr/bin/env python3
"""bench.py -- converged IK solves/s of the MI355X-native memetic solver.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the benchmark OWNS its
launch: started plainly (no WORLD_SIZE in the environment) it refuses to run unless N GPUs are
visible and then starts N ranks of itself through torch.distributed.run on 127.0.0.1 (one process per
GPU, RCCL); started BY a launcher (WORLD_SIZE set) it checks WORLD_SIZE == N.  Either way every rank
asserts dist.get_world_size() == N and the line reports n_gpus = N.  A "step" is one pass of the hot path (ik_memetic) over one
batch of synthetic targets: BASELINE.json configs[1] -- Panda 7-DOF, population 128, batch 4096
random reachable targets per GPU (weak scaling: every rank solves its own 4096-problem shard per
step; random streams are keyed by the global problem index, so the sharded job computes exactly what
one big call would).  Inputs are resident in HBM before the timed region.  The K steps are handed to
the library `--pool` batches per call (pikamd_solve_batches_device: the problems of the batches of
one call are solved as ONE pool by the persistent wavefronts, every batch getting the answers a call
of its own would give), calls round-robin on `--streams` HIP streams; for N > 1 the solutions and
status words of all K steps are then gathered to every rank with one RCCL all-gather each (the only
collective of the path, inside the timed region); the region is closed by device synchronise +
barrier.

**Which arithmetic is the headline.**  `--arithmetic exact` (the default): the product library's literal kernels
(option arithmetic = exact, namespace pik_exact) -- the reference's algorithm evaluation for evaluation, whose
converged joint vectors are IDENTICAL to the CPU oracle's on the same seeds and targets (checked inside this run
on a sample: `parity.identical_to_oracle_on_sample`).  `value`, `config`, `roofline` are that flavour's.  The
fast flavour (Denavit-Hartenberg frames, frame-based gradient probes; agrees with the oracle statistically,
DESIGN.md section 3) is timed on the same batches as the leg `fast`, with its own `roofline` and legs.
`--arithmetic fast` swaps the two (the leg is then called `exact`).

`--config 5` runs BASELINE.json configs[4] instead: 1 048 576 targets at population 512 in N
contiguous shards (strong scaling), one call per step, the same final gather.  `--config 3` / `--config 4` run
configs[2] (UR5, population 256, 65 536 targets per GPU per step, joint-centring + minimal-displacement costs)
and configs[3] (Panda, approximate-solution mode, 65 536 unreachable targets per GPU per step; `value` counts
answers -- every problem returns its best individual), one call per step; the N = 1 line of the default
config carries both as legs (`config3`, `config4`).

After the timed headline region rank 0 (N = 1) appends further legs to the same JSON line, each timed
on its own: `sustained` (512 steps in pools of 64 on 4 streams: the throughput regime), `single_batch`
(one isolated 4096-target call at a time, median of 24), `config3` / `config4`, `other_robots` (arithmetic = exact on a
nine-variable chain, a two-tip tree and a chain on a floating base, each checked against the oracle), the other flavour (`fast`, with
the same legs inside), `plain_ieee` (the verification library: no fused operation anywhere),
`value_incl_h2d_d2h` (host-pointer entry point) and `cpu_baseline` (the oracle on the host cores).

Prints ONE JSON line on rank 0 (fields documented in DESIGN.md "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Independent calls are overlapped on separate HIP streams.  The HIP runtime multiplexes streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4; measured: N queues -> N/2 kernels in flight),
# so the limit has to be raised before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "256")

import numpy as np  # noqa: E402

# FP64 work model of one cost evaluation of the REFERENCE algorithm: FK 924
# flop + 7 sincos, pose cost 60 flop + 3 sqrt + 1 atan2; sincos/atan2 counted as 80 flop, sqrt/div 8.
FLOP_PER_EVAL = {7: 1650.0, 6: 1450.0}
PEAK_FP64_VALU_TFLOPS = 78.6  # MI355X vector FP64 = 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec
# vector instruction issue, round-2 model: one wave64 instruction per 4 cycles per SIMD whatever its class
PEAK_VALU_WAVE_INSTR_PER_S = 256 * 4 * 2.4e9 / 4
ROOFLINE_INPUTS = os.path.join(ROOT, "profiles", "roofline_inputs.json")
# ... and the calibrated one: SIMD cycles per wave64 instruction of each class with two wavefronts per SIMD,
# MEASURED by tools/valu_rates.hip on this chip (profiles/r03_valu_rates.json; FP64 arithmetic takes 4.46
# cycles, not 4; simple 32-bit operations 2.1 -- the guide's SIMD-32 picture holds for those, the 16-lane
# picture for FP64 and everything 64 bits wide).  The instructions no counter classifies (moves, selects,
# lane reads, FP64 compares / max) cost between 2.1 and 4.5: both prices are reported.
VALU_RATES = os.path.join(ROOT, "profiles", "r03_valu_rates.json")
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9


def issue_roof(classes, problems_per_s):
    """fraction of the chip's SIMD cycles the vector instructions of `classes` (per problem) need"""
    try:
        rows = {(r["instr"], r["form"], r["waves_per_simd"]): r["simd_cycles_per_instr"]
                for r in json.load(open(VALU_RATES))["rows"]}
        c = lambda name: rows[(name, "thr", 2)]  # noqa: E731
        price = {"fp64_arith": c("v_fma_f64"), "fp64_trans": c("v_rsq_f64"), "int64": c("v_mad_u64_u32"),
                 "cvt": c("v_cvt_f64_u32"), "int32": c("v_add_u32"), "fp32": c("v_fma_f32")}
        lo_rest, hi_rest = c("v_mov_b32"), c("v_fma_f64")
    except Exception:
        return None
    known = sum(classes.get(k, 0.0) for k in price)
    rest = max(0.0, classes.get("all", 0.0) - known)
    cyc = sum(classes.get(k, 0.0) * v for k, v in price.items())
    lo, hi = cyc + rest * lo_rest, cyc + rest * hi_rest
    return {"frac_low": lo * problems_per_s / SIMD_CYCLES_PER_S, "frac_high": hi * problems_per_s / SIMD_CYCLES_PER_S,
            "simd_cycles_per_problem_low": lo, "simd_cycles_per_problem_high": hi,
            "cycles_per_instruction": dict(price, unclassified_low=lo_rest, unclassified_high=hi_rest),
            "classes_per_problem": classes, "rates_from": "profiles/r03_valu_rates.json (tools/valu_rates.hip, 2 waves per SIMD)",
            "clock_ghz_assumed": 2.4}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 512 (config 5: 8)")
    ap.add_argument("--warmup", type=int, default=None, help="default 64 (config 5: 2)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5),
                    help="BASELINE.json config: 2 = Panda P=128, 4096 targets per GPU per step (the "
                         "metric's config); 3 = UR5 P=256, 65 536 targets per GPU per step, joint goals on; "
                         "4 = Panda approximate mode, 65 536 unreachable targets per GPU per step; "
                         "5 = Panda P=512, 1 048 576 targets over all GPUs per step")
    ap.add_argument("--batch", type=int, default=0, help="problems per GPU per step (0 = the config's)")
    ap.add_argument("--population", type=int, default=0)
    ap.add_argument("--elites", type=int, default=4)
    ap.add_argument("--robot", default="panda")
    ap.add_argument("--pool", type=int, default=int(os.environ.get("PIK_BENCH_POOL", "0")),
                    help="batches handed to the library per call (0 = choose from --steps)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("PIK_BENCH_STREAMS", "0")),
                    help="HIP streams the calls are spread over; 0 = choose from --steps")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="problems timed on the CPU oracle (rank 0, N=1); 0 disables")
    ap.add_argument("--max-generations", type=int, default=100)
    ap.add_argument("--arithmetic", choices=("fast", "exact"), default="exact",
                    help="the flavour of the HEADLINE region.  exact (default): the product library's literal kernels "
                         "(option arithmetic = exact), whose joint vectors are identical to the oracle's; fast: the "
                         "Denavit-Hartenberg / frame-probe kernels.  The other flavour is timed as a leg of the line")
    ap.add_argument("--no-strict", action="store_true", help="skip the other flavour's leg and the plain-IEEE library's timing")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-pointer (PCIe-inclusive) timing")
    ap.add_argument("--no-legs", action="store_true", help="skip the sustained / single-batch / config 3, 4 legs")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, form the process group (gloo without GPUs), all-reduce a "
                         "one per rank and report the world size: the launch path without the solver")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 8 if args.config != 2 else 512
    if args.warmup is None:
        args.warmup = 2 if args.config != 2 else 64
    if args.config in BIG_CONFIGS:
        args.robot = BIG_CONFIGS[args.config]["robot"]
    return args


# BASELINE.json configs[2] and configs[3]: one 65 536-target call per step
BIG_CONFIGS = {
    3: dict(robot="ur5", population=256, batch=65536, unreachable=False,
            kw=dict(center_joints_weight=0.01, minimal_displacement_weight=0.001, cost_threshold=0.01),
            what="UR5 6-DOF, population 256, joint-centring (0.01) + minimal-displacement (0.001) costs, cost threshold 0.01"),
    4: dict(robot="panda", population=128, batch=65536, unreachable=True, kw=dict(return_approximate_solution=1),
            what="Panda 7-DOF, population 128, approximate-solution mode, targets pushed out to 1.0-1.5 m (unreachable)"),
}

METRIC = {2: "converged IK solves/sec (7-DOF Panda, batched random targets)",
          5: "converged IK solves/sec (7-DOF Panda, batched random targets)",
          3: "converged IK solves/sec (6-DOF UR5, joint costs on, batched random targets)",
          4: "IK answers/sec (7-DOF Panda, approximate-solution mode, unreachable targets)"}


def unit_of(config):
    """config 4 (approximate mode) counts ANSWERS -- every problem returns its best individual, SUCCESS or
    APPROXIMATE --, the other configs converged solves; every leg of a config uses its headline's unit"""
    return "answers/s" if config == 4 else "solves/s"


def flavour_sha_of(kernel_name):
    """hash of the device source of the flavour that serves a call (pick_ik_amd/build.py flavour_sha; the kernel
    namespace names the flavour); None when it cannot be computed here (no hipcc)"""
    try:
        from pick_ik_amd import build
        ns = kernel_name.split("::")[0].strip()
        return build.flavour_sha(ns) if ns in build.FLAVOUR_FLAGS else None
    except Exception:
        return None


class RooflineInputs:
    """profiles/roofline_inputs.json: executed work per solved problem from rocprofv3 PMC passes of bench.py's own
    commands, one record per benchmarked shape AND flavour (`driver_cmd`, `driver_cmd_exact`, ...)."""

    def __init__(self):
        try:
            self.all = json.load(open(ROOFLINE_INPUTS)) if os.path.exists(ROOFLINE_INPUTS) else {}
        except Exception:
            self.all = {}

    @staticmethod
    def key(shape, flavour):
        return None if shape is None else shape + ("_exact" if flavour == "exact" else "")

    def record(self, shape, flavour):
        return self.all.get(self.key(shape, flavour)) if shape else None

    @staticmethod
    def stale(rec, kernel_name):
        """True when the flavour's device source changed since the record's counters were taken"""
        if not rec or not rec.get("flavour_sha"):
            return None
        now = flavour_sha_of(kernel_name)
        return None if now is None else (now != rec["flavour_sha"])

    def leg(self, shape, flavour, problems_per_s, kernel_name):
        """the compact roofline of a leg: executed FP64 flop and vector-issue fractions at `problems_per_s`"""
        r = self.record(shape, flavour)
        if not r:
            return None
        fl, vi = r.get("executed_fp64_flop_per_problem"), r.get("valu_wave_instructions_per_problem")
        cal = issue_roof(r["valu_classes_per_problem"], problems_per_s) if r.get("valu_classes_per_problem") else None
        return {"bound": "fp64_valu", "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                "achieved": fl * problems_per_s / 1e12 if fl else None,
                "frac": fl * problems_per_s / 1e12 / PEAK_FP64_VALU_TFLOPS if fl else None,
                "valu_issue_frac": vi * problems_per_s / PEAK_VALU_WAVE_INSTR_PER_S if vi else None,
                "valu_issue_calibrated": ({"frac_low": cal["frac_low"], "frac_high": cal["frac_high"]} if cal else None),
                "executed_fp64_flop_per_problem": fl, "hbm_bytes_per_problem": r.get("hbm_bytes_per_problem"),
                "work_per_problem_from": self.key(shape, flavour), "source": r.get("source"),
                "inputs_stale": self.stale(r, kernel_name)}


def self_launch(args) -> int:
    """`bench.py --gpus N` started without a launcher: start the N ranks (one per GPU) ourselves."""
    import socket
    import subprocess
    if not args.launch_check:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible; refusing to run "
                             f"{args.gpus} ranks on fewer devices")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, PIK_BENCH_SELF_LAUNCHED="1")
    return subprocess.run(cmd, env=env).returncode


def launch_check(args, world, rank, local_rank):
    """The launch path alone: N ranks, one process group, one all-reduce."""
    import torch
    import torch.distributed as dist
    cuda = torch.cuda.is_available()
    if cuda:
        torch.cuda.set_device(local_rank)
    if "MASTER_ADDR" not in os.environ:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29511"
    dist.init_process_group("nccl" if cuda else "gloo", rank=rank, world_size=world)
    assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    one = torch.ones(1, device=torch.device("cuda", local_rank) if cuda else "cpu")
    dist.all_reduce(one)
    assert int(one.item()) == args.gpus, (one.item(), args.gpus)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks": dist.get_world_size(),
                          "all_reduce_of_ones": int(one.item()), "backend": dist.get_backend(),
                          "self_launched": os.environ.get("PIK_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
    dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: this process becomes the launcher of the N ranks
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: the launcher's "
                         f"rank count and --gpus must agree")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_check:
        return launch_check(args, world, rank, local_rank)
    import torch
    import torch.distributed as dist

    use_dist = world > 1 or os.environ.get("PIK_BENCH_FORCE_DIST") == "1"  # (1-rank smoke test)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: pick_ik_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29511"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if use_dist:
        dist.barrier()
    import pick_ik_amd as pk
    from pick_ik_amd import distributed as pkd
    from pick_ik_amd.solver import Batch

    HOMES = {"panda": pk.robots.PANDA_HOME, "ur5": pk.robots.UR5_HOME,
             "dual_ur5": np.concatenate([pk.robots.UR5_HOME] * 2),
             "panda_on_torso": pk.robots.PANDA_ON_TORSO_HOME, "floating_panda": pk.robots.FLOATING_PANDA_HOME}
    flavour = args.arithmetic
    other = "fast" if flavour == "exact" else "exact"
    chain = pk.robots.by_name(args.robot)
    D = chain.dof
    home = HOMES.get(args.robot, np.zeros(D))
    n_tips = int(getattr(chain, "n_tips", 1))  # (non-default robots: several tip frames)
    solver = pk.Solver(chain, device=local_rank, exact=(flavour == "exact"))
    if args.config == 5:
        total = 1048576
        lo, hi = pkd.shard_range(total, rank, world)
        B = args.batch or (hi - lo)
        population = args.population or 512
        scaling = "strong"
    elif args.config in BIG_CONFIGS:
        B = args.batch or BIG_CONFIGS[args.config]["batch"]
        population = args.population or BIG_CONFIGS[args.config]["population"]
        scaling = "weak"
    else:
        B = args.batch or 4096
        population = args.population or 128
        scaling = "weak"
    extra_kw = BIG_CONFIGS[args.config]["kw"] if args.config in BIG_CONFIGS else {}
    unreachable = args.config in BIG_CONFIGS and BIG_CONFIGS[args.config]["unreachable"]
    params = pk.default_params(memetic_population_size=population, memetic_elite_size=args.elites,
                               memetic_max_generations=args.max_generations, **extra_kw)
    K, W = args.steps, args.warmup
    # Pools and streams: the critical path of a pool is 100 generations long whatever its size (the
    # ~1 % of targets that are never reached), so a short run wants everything in ONE pool (nothing
    # is left to overlap a second pool's tail with), a long run a few pools in flight.
    # Measured (MI355X, 4096-problem steps): 20 steps as 1 x 20 / 2 x 10 / 4 x 5 / 20 x 1 pools:
    # 2.20 / 2.21 / 1.95 / 1.04 M solves/s; 512 steps as pools of 16 / 32 / 64 on 2 streams: 2.8 / 3.5 /
    # 4.2 M solves/s; pools of 64 on 4 streams: 4.7 M.
    pool = args.pool if args.pool > 0 else (1 if args.config != 2 else min(K, pk.solver.MAX_BATCHES))
    pool = max(1, min(pool, pk.solver.MAX_BATCHES, max(K, 1)))
    n_calls = (K + pool - 1) // pool
    # (with >= 4 calls in flight the library switches from latency-greedy to efficiency-greedy kernel
    #  variants: 4.7 M instead of 4.3 M solves/s at 512 steps)
    S = args.streams if args.streams > 0 else min(4, n_calls)
    S = max(1, min(S, pk.solver.MAX_SLOTS, n_calls))

    def counted(st, config=args.config):
        """what `value` counts: converged solves, or (config 4) answers"""
        return (st > 0) if config == 4 else (st == pk.SUCCESS)

    # ---- synthetic inputs, resident in HBM: distinct batches for every step -----------------
    # (the sustained leg re-uses the steps' buffers: it needs SUS_K + SUS_W of them)
    SUS_K, SUS_W, SUS_POOL, SUS_S, SINGLE_REPS = 512, 64, 64, 4, 24
    legs = world == 1 and not use_dist and not args.no_legs and args.config == 2
    n_steps = max(K + W, (SUS_K + SUS_W) if legs else 0, SINGLE_REPS if legs else 0)
    rng = np.random.default_rng(0x5049434B + rank)
    f64 = dict(dtype=torch.float64, device=dev)

    def new_outputs(n, b, d):
        return ([torch.empty(b, d, **f64) for _ in range(n)],
                [torch.zeros(b, dtype=torch.int32, device=dev) for _ in range(n)],
                [torch.empty(b, **f64) for _ in range(n)],
                [torch.zeros(b, 3, dtype=torch.int64, device=dev) for _ in range(n)])  # pikamd_stats = 24 bytes

    goals, seeds = [], []
    seed_t = torch.from_numpy(np.tile(home, (B, 1))).to(dev)
    for _ in range(n_steps):
        q = torch.from_numpy(rng.uniform(chain.qmin, chain.qmax, size=(B, D))).to(dev)
        g = torch.empty(B, 7 * n_tips, **f64)
        solver.fk_device(B, q.data_ptr(), g.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if unreachable:  # (config 4: the reachable pose's direction, 1.0-1.5 m out)
            d3 = g[:, :3] / g[:, :3].norm(dim=1, keepdim=True)
            g[:, :3] = d3 * torch.from_numpy(rng.uniform(1.0, 1.5, size=(B, 1))).to(dev)
        goals.append(g)
        seeds.append(seed_t)
    main_out = new_outputs(n_steps, B, D)
    sols, status, costs, stats_ = main_out
    # final gather buffers: every rank receives the solutions / status of the whole job
    gathered = None
    if use_dist:
        gathered = (torch.empty(world * K * B, D, **f64),
                    torch.empty(world * K * B, dtype=torch.int32, device=dev))
    torch.cuda.synchronize()

    def offset_of(i):  # global problem index of step i's first problem on this rank
        return (i * world + rank) * B

    def records(first, count, out=None):
        o = out or main_out
        return [Batch(B, goals[i].data_ptr(), seeds[i].data_ptr(), None, offset_of(i),
                      o[0][i].data_ptr(), o[1][i].data_ptr(), o[2][i].data_ptr(), o[3][i].data_ptr(), None)
                for i in range(first, first + count)]

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(S, SUS_S if legs else 0))]

    def run_steps(slv, first, count, events=None, out=None, pool=pool, S=S):
        """enqueue steps [first, first + count) `pool` batches per call, calls round-robin on S streams"""
        c = 0
        for f in range(first, first + count, pool):
            n = min(pool, first + count - f)
            slot = c % S
            st = streams[slot]
            with torch.cuda.stream(st):
                if events is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                slv.solve_batches_device(params, records(f, n, out), rng_seed=1234, stream=st.cuda_stream,
                                         slot=slot)
                if events is not None:
                    e1.record(st)
                    events.append((e0, e1))
            c += 1

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def final_gather():
        # the only collective of the path: gather the shard results (RCCL over xGMI)
        torch.cuda.synchronize()
        pkd.gather_results(dist, torch.cat(sols[W:W + K]), torch.cat(status[W:W + K]), gathered[0], gathered[1])

    # Reserve every slot's scratch (untimed): allocation + constant upload must not land in the timed
    # region.
    for slot in range(S):
        solver.reserve(params, B * pool, slot=slot, stream=streams[slot].cuda_stream)
    torch.cuda.synchronize()
    run_steps(solver, 0, W)
    if use_dist:
        # warm-up of the collective as well (RCCL sets up its channels on the first call: ~7 ms that
        # belong to no step): the same gather, on the warm-up steps' buffers
        torch.cuda.synchronize()
        k0 = min(W, K)
        if k0 > 0:
            pkd.gather_results(dist, torch.cat(sols[:k0]), torch.cat(status[:k0]),
                               gathered[0][: world * k0 * B], gathered[1][: world * k0 * B])
    fence()
    ev = []
    t0 = time.perf_counter()
    run_steps(solver, W, K, events=ev)
    t_enqueued = time.perf_counter() - t0
    if use_dist:
        final_gather()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], **f64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- results of the timed steps ---------------------------------------------------------
    st_all = torch.stack(status[W:W + K])
    converged = counted(st_all).sum().to(torch.float64)
    succeeded = st_all.eq(pk.SUCCESS).sum().to(torch.float64)
    evals = torch.stack(stats_[W:W + K])[:, :, 0].sum().to(torch.float64)
    gens = (torch.stack(stats_[W:W + K])[:, :, 1] & 0xFFFFFFFF).to(torch.float64).mean()
    totals = torch.stack([converged, evals, gens, succeeded])
    if use_dist:
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
        # the gathered copy must hold exactly this rank's results at this rank's position
        lo_ = rank * K * B
        assert torch.equal(gathered[1][lo_:lo_ + K * B], torch.cat(status[W:W + K]))
    converged_total, evals_total = float(totals[0]), float(totals[1])
    mean_gens = float(totals[2]) / world
    call_ms = [a.elapsed_time(b) for a, b in ev]

    result_line = None
    if rank == 0:
        problems = K * B * world
        # Per-launch duration.  A "launch" is one call (a pool of `pool` batches, all its passes).
        # `raw` = HIP-event time from the first kernel of a call to its last; with S streams calls
        # overlap, so the duration one launch effectively occupies the chip for is wall / launches
        # (= raw when S = 1); the roofline uses that one.
        raw_launch_s = float(np.mean(call_ms)) * 1e-3
        avg_launch_s = elapsed / n_calls
        alg_flop_per_launch = evals_total / (n_calls * world) * FLOP_PER_EVAL.get(D, 236.0 * D)
        alg_tflops = alg_flop_per_launch / avg_launch_s / 1e12
        alg_bytes_per_problem = 56 * n_tips + 8 * D + 8 * D + 4 + 8
        # executed FP64 work and HBM traffic per problem: PMC counters of this command under this flavour,
        # tools/profile_driver_cmd.sh -> tools/read_prof.py (same seeds => same work per problem).
        # (the work per problem depends on which kernel variants run: one record per benchmarked shape and flavour)
        rins = RooflineInputs()
        if args.config == 5:
            shape = "config5"
        elif args.config in BIG_CONFIGS:
            shape = f"config{args.config}"
        else:
            shape = ("driver_cmd" if n_calls == 1 and pool > 1 else "single_batch" if pool == 1 and S == 1
                     else "default_run" if (n_calls >= 8 and S >= 4) else None)
        config_as_profiled = (args.max_generations == 100 and
                              ((args.config == 2 and args.robot == "panda" and B == 4096 and population == 128) or
                               (args.config == 5 and args.robot == "panda" and population == 512) or
                               (args.config in BIG_CONFIGS and B == BIG_CONFIGS[args.config]["batch"] and
                                population == BIG_CONFIGS[args.config]["population"])))
        rin = rins.record(shape, flavour) if config_as_profiled else None
        kname = solver.kernel_name(params)
        exec_flop_pp = rin.get("executed_fp64_flop_per_problem") if rin else None
        traffic_pp = rin.get("hbm_bytes_per_problem") if rin else None
        valu_pp = rin.get("valu_wave_instructions_per_problem") if rin else None
        per_launch = problems / (n_calls * world)
        exec_tflops = (exec_flop_pp * per_launch / avg_launch_s / 1e12) if exec_flop_pp else None
        out = {
            "metric": METRIC[args.config],
            "value": converged_total / elapsed,
            "unit": unit_of(args.config),
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.robot} {D}-DOF memetic IK (BASELINE configs[{args.config - 1}]): "
                            f"population {population}, elites {args.elites}, batch {B} random "
                            f"reachable targets per GPU per step, seed = ready pose, max_generations "
                            f"{args.max_generations}, gd_max_iters {params.memetic_gd_max_iters}",
                "robot": args.robot,
                "dof": D,
                "tip_frames": n_tips,
                "batch_per_gpu": B,
                "batches_per_call": pool,
                "calls": n_calls,
                "streams": S,
                "problems_in_flight_per_gpu": min(S, n_calls) * pool * B,
                "host_enqueue_ms_per_step": t_enqueued / K * 1e3,
                "success_rate": float(totals[3]) / problems,
                **({"what": BIG_CONFIGS[args.config]["what"]} if args.config in BIG_CONFIGS else {}),
                "mean_generations": mean_gens,
                "mean_cost_evals_per_solve": evals_total / problems,
                "parallelism": f"shard{world}",
                "arithmetic": flavour,
                "arithmetic_note": ("exact: the literal algorithm (MoveIt's chain product, 2D+3 cost evaluations per "
                                    "gradient step, IEEE sqrt / divide, fused multiply-adds at stated places); joint "
                                    "vectors identical to the CPU oracle's (math mode 'fma'), see `parity`"
                                    if flavour == "exact" else
                                    "fast: Denavit-Hartenberg frames + frame-based gradient probes; agrees with the "
                                    "oracle statistically (success rate, costs), not joint vector by joint vector"),
            },
            "roofline": {
                "bound": "fp64_valu",
                "kernel": kname,
                "achieved": exec_tflops,
                "peak": PEAK_FP64_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": (exec_tflops / PEAK_FP64_VALU_TFLOPS) if exec_tflops else None,
                "avg_launch_ms": avg_launch_s * 1e3,
                "raw_event_launch_ms": raw_launch_s * 1e3,
                "executed_fp64_flop_per_problem": exec_flop_pp,
                "problems_per_launch": per_launch,
                "source": (rin or {}).get("source"),
                "work_per_problem_from": rins.key(shape, flavour) if rin else None,
                "inputs_flavour_sha": (rin or {}).get("flavour_sha"), "flavour_sha": flavour_sha_of(kname),
                "inputs_stale": rins.stale(rin, kname),
                "note": "launch = one call (a pool of batches_per_call batches, all its compaction "
                        "passes). achieved = EXECUTED FP64 flop (rocprofv3 PMC: (ADD + MUL + TRANS + "
                        "2 FMA)_F64 wave instructions x 64 lanes, per problem, profiles/"
                        "roofline_inputs.json, the record of this shape AND flavour) x problems per launch / "
                        "avg_launch_ms; avg_launch_ms = "
                        "wall / launches (calls on S streams overlap), raw_event_launch_ms = HIP-event "
                        "time of one call while sharing the chip. `algorithmic` prices the REFERENCE's "
                        "cost_fn evaluations (literal counter kept by the kernel) at 1.65 kflop each: the exact "
                        "flavour performs them all but re-uses the accept evaluation's prefix frames in the "
                        "probes (about half the flop), the fast flavour replaces the probes by frame "
                        "arithmetic (~4x fewer), so that figure is a speed-of-algorithm number and may exceed 1. "
                        "inputs_stale: the flavour's preprocessed device source changed since the counters were taken.",
                "algorithmic": {"achieved": alg_tflops, "frac": alg_tflops / PEAK_FP64_VALU_TFLOPS,
                                "flop_per_launch": alg_flop_per_launch},
                "hbm": {"achieved": alg_bytes_per_problem * per_launch / avg_launch_s / 1e9,
                        "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": alg_bytes_per_problem * per_launch / avg_launch_s / 1e9 / PEAK_HBM_GBS,
                        "algorithmic_bytes_per_problem": alg_bytes_per_problem},
                "traffic": (traffic_pp * per_launch) if traffic_pp else None,
                # the resource the kernels are bound by: every vector instruction of a wavefront
                # (FP64 or not) takes one 4-cycle issue slot of its SIMD
                "valu_issue": ({"achieved": valu_pp * per_launch / avg_launch_s,
                                "peak": PEAK_VALU_WAVE_INSTR_PER_S, "unit": "wave instructions/s",
                                "frac": valu_pp * per_launch / avg_launch_s / PEAK_VALU_WAVE_INSTR_PER_S,
                                "valu_wave_instructions_per_problem": valu_pp,
                                "fp64_share": (rin or {}).get("fp64_share_of_valu_instructions"),
                                "model": "round-2 model: 4 cycles per instruction of any class (kept for "
                                         "comparison); `calibrated` prices every class with its measured cycles",
                                "calibrated": (issue_roof(rin.get("valu_classes_per_problem"), per_launch / avg_launch_s)
                                               if rin.get("valu_classes_per_problem") else None)}
                               if valu_pp else None),
            },
        }
        out["n_ranks"] = dist.get_world_size() if use_dist else 1
        out["launch"] = ("self (bench.py started its ranks)" if os.environ.get("PIK_BENCH_SELF_LAUNCHED") == "1"
                         else "external launcher" if "WORLD_SIZE" in os.environ else "single process")

        profiled_cfg2 = args.max_generations == 100 and args.robot == "panda" and B == 4096 and population == 128

        def oracle_sample_check(o, mode, first_step=W, n=256):
            """joint vectors + status + cost of the first n problems of one timed batch against the CPU oracle
            (math mode `mode`) -- the checker, after the timed regions"""
            try:
                from oracle import oracle as O
                n = min(B, n)
                with O.math_mode(mode):
                    ref = O.Oracle(chain).solve_batch(
                        O.default_params(memetic_population_size=population, memetic_elite_size=args.elites,
                                         memetic_max_generations=args.max_generations, **extra_kw),
                        goals[first_step].cpu().numpy()[:n], np.tile(home, (n, 1)), rng_seed=1234,
                        problem_offset=offset_of(first_step), num_threads=O.max_threads())
                same = bool(np.array_equal(o[0][first_step].cpu().numpy()[:n], ref[0]) and
                            np.array_equal(o[1][first_step].cpu().numpy()[:n], ref[1]) and
                            np.array_equal(o[2][first_step].cpu().numpy()[:n], ref[2]))
                agree = float(np.mean(np.all(np.abs(o[0][first_step].cpu().numpy()[:n] - ref[0]) <= 1e-6, axis=1)))
                return {"identical_to_oracle_on_sample": same, "joint_vectors_within_1e-6_rad": agree,
                        "oracle_math_mode": mode,
                        "sample": f"first {n} problems of the first timed batch, joint vectors + status + cost"}
            except Exception as e:  # the checker is optional for a measurement
                return {"identical_to_oracle_on_sample": None, "oracle_math_mode": mode, "sample": f"oracle unavailable: {e}"}

        def time_pool(slv, o, fl):
            """the headline's shape (same steps, pool, streams) on another solver handle"""
            for slot in range(S):
                slv.reserve(params, B * pool, slot=slot, stream=streams[slot].cuda_stream)
            torch.cuda.synchronize()
            run_steps(slv, 0, min(W, pool), out=o)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            run_steps(slv, W, K, out=o)
            torch.cuda.synchronize()
            dts = time.perf_counter() - ts
            st = torch.stack(o[1][W:W + K])
            n_counted = float(counted(st).sum().item())
            kn = slv.kernel_name(params)
            return {"value": n_counted / dts, "unit": unit_of(args.config), "ms_per_step": dts / K * 1e3,
                    "success_rate": float(st.eq(pk.SUCCESS).double().mean().item()), "kernel": kn,
                    "roofline": rins.leg(shape, fl, K * B / dts, kn) if config_as_profiled and fl != "strict" else None}

        def sustained_and_single(slv, o, fl, sus_k=SUS_K):
            """legs of the config-2 line: the throughput regime, and one isolated batch at a time"""
            d = {}
            kn = slv.kernel_name(params)
            if slv is solver and K >= SUS_K and pool == SUS_POOL and S >= SUS_S:
                d["sustained"] = {"value": out["value"], "unit": "solves/s", "steps": K, "ms_per_step": out["ms_per_step"],
                                  "batches_per_call": pool, "streams": S, "success_rate": converged_total / problems,
                                  "same_as": "the headline region (this run already has the sustained shape)",
                                  "roofline": rins.leg("default_run", fl, problems / elapsed, kn) if profiled_cfg2 else None}
            else:
                for slot in range(SUS_S):
                    slv.reserve(params, B * SUS_POOL, slot=slot, stream=streams[slot].cuda_stream)
                torch.cuda.synchronize()
                run_steps(slv, 0, SUS_W, out=o, pool=SUS_POOL, S=SUS_S)
                torch.cuda.synchronize()
                tsu = time.perf_counter()
                run_steps(slv, SUS_W, sus_k, out=o, pool=SUS_POOL, S=SUS_S)
                torch.cuda.synchronize()
                dsu = time.perf_counter() - tsu
                su_conv = float((torch.stack(o[1][SUS_W:SUS_W + sus_k]) == pk.SUCCESS).sum().item())
                d["sustained"] = {"value": su_conv / dsu, "unit": "solves/s", "steps": sus_k, "warmup": SUS_W,
                                  "ms_per_step": dsu / sus_k * 1e3, "batches_per_call": SUS_POOL, "streams": SUS_S,
                                  "success_rate": su_conv / (sus_k * B),
                                  "roofline": rins.leg("default_run", fl, sus_k * B / dsu, kn) if profiled_cfg2 else None}
            # ---- single-batch leg: one isolated call at a time (what a planner with ONE batch sees) --
            ms = []
            sb_conv = 0.0
            st0 = streams[0]
            for r_ in range(SINGLE_REPS + 2):
                i = r_ % n_steps
                torch.cuda.synchronize()
                tb = time.perf_counter()
                with torch.cuda.stream(st0):
                    slv.solve_batches_device(params, records(i, 1, o), rng_seed=1234, stream=st0.cuda_stream, slot=0)
                torch.cuda.synchronize()
                if r_ >= 2:  # (two untimed calls first)
                    ms.append((time.perf_counter() - tb) * 1e3)
                    sb_conv += float((o[1][i] == pk.SUCCESS).sum().item())
            ms.sort()
            med = ms[len(ms) // 2]
            d["single_batch"] = {"median_ms": med, "min_ms": ms[0], "max_ms": ms[-1], "repetitions": len(ms),
                                 "value": sb_conv / len(ms) / (med * 1e-3), "unit": "solves/s",
                                 "batch": B, "success_rate": sb_conv / (len(ms) * B),
                                 "what": "one pikamd_solve_batches_device call of ONE batch, device synchronised "
                                         "before and after, nothing else in flight",
                                 "roofline": rins.leg("single_batch", fl, B / (med * 1e-3), kn) if profiled_cfg2 else None}
            return d

        def big_config_legs(fl):
            """BASELINE configs 3 and 4 as legs: their own robot / parameters / targets, one 65 536-target call per
            step, three timed steps behind one untimed"""
            d = {}
            for cfg_id, cfg in BIG_CONFIGS.items():
                ch2 = pk.robots.by_name(cfg["robot"])
                home2 = HOMES[cfg["robot"]]
                sv = pk.Solver(ch2, device=local_rank, exact=(fl == "exact"))
                p2 = pk.default_params(memetic_population_size=cfg["population"], memetic_elite_size=args.elites,
                                       memetic_max_generations=args.max_generations, **cfg["kw"])
                B2, D2, n2 = cfg["batch"], ch2.dof, 4
                rng2 = np.random.default_rng(0xC0F + cfg_id)
                seed2 = torch.from_numpy(np.tile(home2, (B2, 1))).to(dev)
                g2 = []
                for _ in range(n2):
                    q2 = torch.from_numpy(rng2.uniform(ch2.qmin, ch2.qmax, size=(B2, D2))).to(dev)
                    g = torch.empty(B2, 7, **f64)
                    sv.fk_device(B2, q2.data_ptr(), g.data_ptr(), torch.cuda.current_stream().cuda_stream)
                    if cfg["unreachable"]:
                        d3 = g[:, :3] / g[:, :3].norm(dim=1, keepdim=True)
                        g[:, :3] = d3 * torch.from_numpy(rng2.uniform(1.0, 1.5, size=(B2, 1))).to(dev)
                    g2.append(g)
                so2, st2, co2, sa2 = new_outputs(n2, B2, D2)
                torch.cuda.synchronize()
                sv.reserve(p2, B2, slot=0, stream=streams[0].cuda_stream)

                def one(i):
                    with torch.cuda.stream(streams[0]):
                        sv.solve_batches_device(p2, [Batch(B2, g2[i].data_ptr(), seed2.data_ptr(), None, i * B2,
                                                           so2[i].data_ptr(), st2[i].data_ptr(), co2[i].data_ptr(),
                                                           sa2[i].data_ptr(), None)],
                                                rng_seed=1234, stream=streams[0].cuda_stream, slot=0)
                one(0)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for i in range(1, n2):
                    one(i)
                torch.cuda.synchronize()
                d2 = time.perf_counter() - t2
                stt = torch.stack(st2[1:])
                n_counted = float(counted(stt, cfg_id).sum().item())
                kn2 = sv.kernel_name(p2)
                leg = {"value": n_counted / d2, "unit": unit_of(cfg_id), "steps": n2 - 1,
                       "warmup": 1, "ms_per_step": d2 / (n2 - 1) * 1e3, "batch": B2, "workload": cfg["what"],
                       "success_rate": float((stt == pk.SUCCESS).float().mean().item()),
                       "mean_generations": float((torch.stack(sa2[1:])[:, :, 1] & 0xFFFFFFFF).double().mean().item()),
                       "kernel": kn2,
                       "roofline": (rins.leg(f"config{cfg_id}", fl, (n2 - 1) * B2 / d2, kn2)
                                    if args.max_generations == 100 else None)}
                # the leg's own parity object: the first 128 problems of its first timed step against the CPU oracle
                # (the checker, after the timed region; exact flavour: every bit -- oracle math mode "fma")
                if world == 1 and not use_dist and not args.no_strict:
                    try:
                        from oracle import oracle as O
                        n_chk, mode2 = min(B2, 128), ("fma" if fl == "exact" else "libm")
                        with O.math_mode(mode2):
                            ref = O.Oracle(ch2).solve_batch(
                                O.default_params(memetic_population_size=cfg["population"], memetic_elite_size=args.elites,
                                                 memetic_max_generations=args.max_generations, **cfg["kw"]),
                                g2[1].cpu().numpy()[:n_chk], np.tile(home2, (n_chk, 1)), rng_seed=1234,
                                problem_offset=B2, num_threads=O.max_threads())
                        got = (so2[1].cpu().numpy()[:n_chk], st2[1].cpu().numpy()[:n_chk], co2[1].cpu().numpy()[:n_chk])
                        leg["parity"] = {
                            "identical_to_oracle_on_sample": bool(all(np.array_equal(a_, b_) for a_, b_ in zip(got, ref[:3]))),
                            "joint_vectors_within_1e-6_rad": float(np.mean(np.all(np.abs(got[0] - ref[0]) <= 1e-6, axis=1))),
                            "verdicts_equal": float(np.mean(got[1] == ref[1])), "oracle_math_mode": mode2,
                            "sample": f"first {n_chk} problems of the first timed step, joint vectors + status + cost"}
                    except Exception as e:  # the checker is optional for a measurement
                        leg["parity"] = {"identical_to_oracle_on_sample": None, "sample": f"oracle unavailable: {e}"}
                if cfg_id == 4:
                    fc = torch.cat(co2[1:])
                    leg["final_cost_median"] = float(fc.median().item())
                    leg["final_cost_p95"] = float(fc.quantile(0.95).item()) if fc.numel() <= 16_000_000 else None
                d[f"config{cfg_id}"] = leg
                sv.close()
                del g2, so2, st2, co2, sa2
            return d

        # what `arithmetic = exact` costs on robots that are not of the Panda / UR5 kind (VERDICT r05 "missing 4"): a
        # chain of class 2 with nine variables (the specialised forms reach ten since round 6), a tree with two tip
        # frames (memoised descent for several tips), a chain on a floating base (fork form) -- one pool of eight
        # 4096-target steps each, population 128, behind two untimed ones; 64 problems against the CPU oracle
        OTHER_ROBOTS = ("panda_on_torso", "torso_dual_arm", "floating_panda")

        def other_robot_legs(fl):
            d = {}
            for name in OTHER_ROBOTS:
                try:
                    ch3 = pk.robots.by_name(name)
                    home3 = HOMES.get(name, np.zeros(ch3.dof))
                    nt3 = int(getattr(ch3, "n_tips", 1))
                    sv = pk.Solver(ch3, device=local_rank, exact=(fl == "exact"))
                    p3 = pk.default_params(memetic_population_size=128, memetic_elite_size=args.elites,
                                           memetic_max_generations=args.max_generations)
                    B3, D3, w3, k3 = 4096, ch3.dof, 2, 8
                    rng3 = np.random.default_rng(0x0B07 + len(name))
                    seed3 = torch.from_numpy(np.tile(home3, (B3, 1))).to(dev)
                    g3 = []
                    for _ in range(w3 + k3):
                        q3 = torch.from_numpy(rng3.uniform(ch3.qmin, ch3.qmax, size=(B3, D3))).to(dev)
                        g = torch.empty(B3, 7 * nt3, **f64)
                        sv.fk_device(B3, q3.data_ptr(), g.data_ptr(), torch.cuda.current_stream().cuda_stream)
                        g3.append(g)
                    so3, st3, co3, sa3 = new_outputs(w3 + k3, B3, D3)
                    torch.cuda.synchronize()
                    sv.reserve(p3, B3 * k3, slot=0, stream=streams[0].cuda_stream)

                    def pool3(first, count):
                        with torch.cuda.stream(streams[0]):
                            sv.solve_batches_device(p3, [Batch(B3, g3[i].data_ptr(), seed3.data_ptr(), None, i * B3,
                                                               so3[i].data_ptr(), st3[i].data_ptr(), co3[i].data_ptr(),
                                                               sa3[i].data_ptr(), None) for i in range(first, first + count)],
                                                    rng_seed=1234, stream=streams[0].cuda_stream, slot=0)
                    pool3(0, w3)
                    torch.cuda.synchronize()
                    t3 = time.perf_counter()
                    pool3(w3, k3)
                    torch.cuda.synchronize()
                    d3 = time.perf_counter() - t3
                    stt = torch.stack(st3[w3:])
                    leg = {"value": float((stt == pk.SUCCESS).sum().item()) / d3, "unit": "solves/s", "steps": k3, "warmup": w3,
                           "ms_per_step": d3 / k3 * 1e3, "batch": B3, "dof": D3, "tip_frames": nt3, "population": 128,
                           "success_rate": float((stt == pk.SUCCESS).float().mean().item()), "kernel": sv.kernel_name(p3)}
                    if world == 1 and not use_dist and not args.no_strict and fl == "exact":
                        try:
                            from oracle import oracle as O
                            n_chk = 64
                            with O.math_mode("fma"):
                                ref = O.Oracle(ch3).solve_batch(
                                    O.default_params(memetic_population_size=128, memetic_elite_size=args.elites,
                                                     memetic_max_generations=args.max_generations),
                                    g3[w3].cpu().numpy()[:n_chk].reshape((n_chk, nt3, 7) if nt3 > 1 else (n_chk, 7)),
                                    np.tile(home3, (n_chk, 1)), rng_seed=1234, problem_offset=w3 * B3,
                                    num_threads=O.max_threads())
                            got = (so3[w3].cpu().numpy()[:n_chk], st3[w3].cpu().numpy()[:n_chk], co3[w3].cpu().numpy()[:n_chk])
                            leg["parity"] = {
                                "identical_to_oracle_on_sample": bool(all(np.array_equal(a_, b_) for a_, b_ in zip(got, ref[:3]))),
                                "oracle_math_mode": "fma",
                                "sample": f"first {n_chk} problems of the first timed step, joint vectors + status + cost"}
                        except Exception as e:  # the checker is optional for a measurement
                            leg["parity"] = {"identical_to_oracle_on_sample": None, "sample": f"oracle unavailable: {e}"}
                    d[name] = leg
                    sv.close()
                    del g3, so3, st3, co3, sa3
                except Exception as e:  # (a leg never takes the headline down with it)
                    d[name] = {"error": repr(e)}
            return {"other_robots": d}

        # ---- legs of the headline flavour ---------------------------------------------------------
        if legs:
            out.update(sustained_and_single(solver, main_out, flavour))
            out.update(big_config_legs(flavour))
            if flavour == "exact" and args.config == 2:
                out.update(other_robot_legs(flavour))
        # ---- parity of the headline flavour: the exact kernels' results ARE the oracle's ------------
        if world == 1 and not use_dist and not args.no_strict:
            out["parity"] = oracle_sample_check(main_out, "fma" if flavour == "exact" else "libm")
            out["parity"]["what"] = (
                "exact flavour: every output of the sampled problems equal to the oracle's, bit for bit "
                "(tests/test_gpu_strict_parity.py and tests/test_gpu_full_size.py assert the same at tolerance 0)"
                if flavour == "exact" else
                "fast flavour: not expected to be identical -- the descent is chaotic and its arithmetic differs from "
                "the oracle's in the last bit (DESIGN.md section 3); joint_vectors_within_1e-6_rad sits on the chaos floor")
        # ---- the other flavour on the same batches, with the same legs ------------------------------
        if world == 1 and not use_dist and not args.no_strict:
            o_out = new_outputs(n_steps, B, D)
            osv = pk.Solver(chain, device=local_rank, exact=(other == "exact"))
            leg = time_pool(osv, o_out, other)
            leg["build"] = ("libpick_ik_amd.so, option arithmetic = exact (namespace pik_exact)" if other == "exact" else
                            "libpick_ik_amd.so, default arithmetic (namespaces pik_common / pik_common_goals / pik)")
            leg["parity"] = oracle_sample_check(o_out, "fma" if other == "exact" else "libm")
            if legs:
                leg.update(sustained_and_single(osv, o_out, other))
            osv.close()
            if legs:
                leg.update(big_config_legs(other))
            out[other] = leg
            # ... and the verification library: the same literal kernels without any fused operation
            if flavour == "exact" or other == "exact":
                strict = pk.Solver(chain, device=local_rank, strict=True)
                pi = time_pool(strict, o_out, "strict")
                pi["build"] = ("libpick_ik_amd_strict.so: the literal kernels without any fused operation "
                               "(-ffp-contract=off); bit-identical to the oracle's math mode 'portable'")
                pi["parity"] = oracle_sample_check(o_out, "portable")
                strict.close()
                out["plain_ieee"] = pi
            del o_out
        # ---- the same steps through the host-pointer entry point (PCIe in the timed region) -----
        if world == 1 and not args.no_pcie:
            hg = [goals[i].cpu().numpy() for i in range(W, W + K)]
            hs = np.tile(home, (B, 1))
            J = max(1, min(4, n_calls))
            hpool = min(pool, pk.solver.MAX_BATCHES)
            # one untimed round: staging buffers and the jobs' streams are created on first use
            for j in range(J):
                solver.solve_batches(params, [(hg[0], hs, None, 0)] * min(hpool, K), rng_seed=1234, job=j)
            for j in range(J):
                solver.wait(j)
            th = time.perf_counter()
            outs, c = [], 0
            for f in range(0, K, hpool):
                n = min(hpool, K - f)
                j = c % J
                if c >= J:
                    solver.wait(j)
                outs.append(solver.solve_batches(
                    params, [(hg[i], hs, None, offset_of(W + i)) for i in range(f, f + n)], rng_seed=1234,
                    job=j))
                c += 1
            for j in range(J):
                solver.wait(j)
            dth = time.perf_counter() - th
            h_conv = float(sum(counted(r[1]).sum() for o in outs for r in o))
            same = all(np.array_equal(r[1], status[W + i].cpu().numpy())
                       for i, r in enumerate(r for o in outs for r in o))
            out["value_incl_h2d_d2h"] = h_conv / dth
            out["config"]["host_pointer_path"] = {
                "entry_point": "pikamd_solve_batches_async / pikamd_wait (pageable host arrays -> "
                               "pinned staging -> async H2D, kernels, async D2H)",
                "jobs_in_flight": J, "batches_per_job": hpool, "ms_per_step": dth / K * 1e3,
                "bytes_per_step_h2d": B * (56 * n_tips + 8 * D), "bytes_per_step_d2h": B * (8 * D + 4 + 8 + 24),
                "status_identical_to_device_path": bool(same)}
        # ---- config 5 through the native multi-device front end of the C ABI -------------------
        if world == 1 and args.config == 5 and not args.no_pcie:
            from pick_ik_amd import solver as pks
            hg5 = goals[W].cpu().numpy()
            hs5 = np.tile(home, (B, 1))
            # (one untimed call of the same size: pinned staging buffers and scratch are sized on first use)
            pks.solve_batch_sharded([solver], params, hg5, hs5, rng_seed=1234, problem_offset=offset_of(W))
            tn = time.perf_counter()
            r5 = pks.solve_batch_sharded([solver], params, hg5, hs5, rng_seed=1234, problem_offset=offset_of(W))
            dtn = time.perf_counter() - tn
            out["native_front_end"] = {
                "entry_point": "pikamd_solve_batch_sharded (host arrays in and out; one host thread per device, "
                               "two staged chunks per shard; H2D / D2H inside the time)",
                "devices": 1, "problems": int(B), "ms": dtn * 1e3,
                "value": float((r5[1] == pk.SUCCESS).sum()) / dtn, "unit": "solves/s",
                "status_identical_to_device_path": bool(np.array_equal(r5[1], status[W].cpu().numpy()))}
        # ---- CPU baseline: the oracle (a port), all host cores, bounded sample -------------------
        if world == 1 and args.cpu_sample != 0:
            from oracle import oracle as O
            cores = O.max_threads()
            try:
                o = O.Oracle(chain, timing_build=True)
                how = "gcc -O3 -march=native -ffp-contract=fast (oracle/_native, compiled on this host)"
            except Exception as e:
                o = O.Oracle(chain)
                how = f"portable -O3 build (native build failed: {e})"
            n = args.cpu_sample if args.cpu_sample > 0 else min(K * B, 256 * cores)
            g = torch.cat(goals[W:W + K]).cpu().numpy()[:n]
            n = len(g)
            sd = np.tile(home, (n, 1))
            po = O.default_params(memetic_population_size=population, memetic_elite_size=args.elites,
                                  memetic_max_generations=args.max_generations, **extra_kw)
            tc = time.perf_counter()
            _, ost, _, ostats = o.solve_batch(po, g, sd, rng_seed=1234, problem_offset=offset_of(W),
                                              num_threads=cores)
            dt = time.perf_counter() - tc
            out["cpu_baseline"] = {
                "value": float(counted(ost).sum()) / dt,
                "unit": unit_of(args.config),
                "cores": cores,
                "kind": "port",
                "sample": f"first {n} problems of the timed steps ({n // cores} per core, dynamic "
                          f"scheduling), oracle/pik_oracle.c (plain C restatement of pick_ik's "
                          f"algorithm without its mutex / std::function / allocation overheads), "
                          f"{how}, {dt:.1f} s wall",
                "success_rate": float((ost == O.SUCCESS).mean()),
                "mean_generations": float(ostats["generations"].mean()),
            }
            gpu_ok = (torch.cat(status[W:W + K])[:n] == pk.SUCCESS).float().mean().item()
            out["config"]["success_rate_vs_cpu_sample"] = gpu_ok / max(1e-9, float((ost == 1).mean()))
        result_line = json.dumps(out)
    solver.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL leaves a version banner in the C stdio buffer of stdout, which would otherwise be
        # flushed at exit, AFTER the result: push it out first so the JSON is the last line
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
