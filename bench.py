#!/usr/bin/env python3
"""bench.py -- converged IK solves/s of the MI355X-native memetic solver.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by
torch.distributed.run, one rank per GPU.  A "step" is one pass of the hot path (ik_memetic) over one
batch of synthetic targets: BASELINE.json configs[1] -- Panda 7-DOF, population 128, batch 4096
random reachable targets per GPU (weak scaling: every rank solves its own 4096-problem shard; random
streams are keyed by the global problem index, so the sharded job computes exactly what one big
call would).  Inputs are resident in HBM before the timed region; K steps are enqueued on
`--streams` HIP streams (independent batches overlap on the GPU, as a server feeding 4096-target
batches would run them); for N > 1 the solutions and status words of all K steps are then gathered
to every rank with one RCCL all-gather each (the only collective of the path, inside the timed
region), and the region is closed by a device synchronise + barrier.

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

# Independent batches are overlapped on separate HIP streams.  The HIP runtime multiplexes streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4; measured: N queues -> N/2 kernels in flight),
# so the limit has to be raised before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "256")

import numpy as np  # noqa: E402

# FP64 work model of one cost evaluation: FK 924 flop + 7 sincos, pose
# cost 60 flop + 3 sqrt + 1 atan2; sincos/atan2 counted as 80 flop, sqrt/div as 8.
FLOP_PER_EVAL = {7: 1650.0, 6: 1450.0}
# algorithmic HBM bytes per solve: goal 56 + seed 8D in, solution 8D + status 4 + cost 8 out
PEAK_FP64_VALU_TFLOPS = 78.6  # MI355X vector FP64 (AMD spec); = half the 157.3 TF FP32 vector
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--batch", type=int, default=4096, help="problems per GPU per step")
    ap.add_argument("--population", type=int, default=128)
    ap.add_argument("--elites", type=int, default=4)
    ap.add_argument("--robot", default="panda")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("PIK_BENCH_STREAMS", "0")),
                    help="HIP streams the steps are spread over; 0 = choose from --steps")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="problems timed on the CPU oracle (rank 0, N=1); 0 disables")
    ap.add_argument("--max-generations", type=int, default=100)
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = world > 1 or os.environ.get("PIK_BENCH_FORCE_DIST") == "1"  # (1-rank smoke test)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: pick_ik_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29511"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if use_dist:
        dist.barrier()
    import pick_ik_amd as pk

    chain = pk.robots.by_name(args.robot)
    D = chain.dof
    home = {"panda": pk.robots.PANDA_HOME, "ur5": pk.robots.UR5_HOME,
            "dual_ur5": np.concatenate([pk.robots.UR5_HOME] * 2)}.get(args.robot, np.zeros(D))
    n_tips = int(getattr(chain, "n_tips", 1))  # (non-default robots: several tip frames)
    solver = pk.Solver(chain, device=local_rank)
    params = pk.default_params(memetic_population_size=args.population,
                               memetic_elite_size=args.elites,
                               memetic_max_generations=args.max_generations)
    B, K, W = args.batch, args.steps, args.warmup
    # Streams: enough independent batches in flight to cover the ~45-85 ms latency of one batch (its
    # critical path is 100 generations long whatever its size) without phase-aligning too many of
    # them at start-up; measured on MI355X: 16 streams for short runs, 64 once K >> 64.
    auto_streams = 64 if K >= 256 else (32 if K >= 96 else 16)
    S = max(1, min(args.streams if args.streams > 0 else auto_streams, pk.solver.MAX_SLOTS, max(K, 1)))

    # ---- synthetic inputs, resident in HBM: distinct batches for every step -----------------
    n_steps = K + W
    rng = np.random.default_rng(0x5049434B + rank)
    f64 = dict(dtype=torch.float64, device=dev)
    goals, seeds, sols, stats_, costs, status = [], [], [], [], [], []
    seed_t = torch.from_numpy(np.tile(home, (B, 1))).to(dev)
    for _ in range(n_steps):
        q = torch.from_numpy(rng.uniform(chain.qmin, chain.qmax, size=(B, D))).to(dev)
        g = torch.empty(B, 7 * n_tips, **f64)
        solver.fk_device(B, q.data_ptr(), g.data_ptr(), torch.cuda.current_stream().cuda_stream)
        goals.append(g)
        seeds.append(seed_t)
        sols.append(torch.empty(B, D, **f64))
        status.append(torch.zeros(B, dtype=torch.int32, device=dev))
        costs.append(torch.empty(B, **f64))
        stats_.append(torch.zeros(B, 3, dtype=torch.int64, device=dev))  # pikamd_stats = 24 bytes
    # final gather buffers: every rank receives the solutions / status of the whole job
    gathered = None
    if use_dist:
        gathered = (torch.empty(world * K * B, D, **f64),
                    torch.empty(world * K * B, dtype=torch.int32, device=dev))
    torch.cuda.synchronize()

    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(n_steps)]

    def run_step(i):
        slot = i % S
        st = streams[slot]
        with torch.cuda.stream(st):
            ev[i][0].record(st)
            solver.solve_batch_device(
                params, B, goals[i].data_ptr(), seeds[i].data_ptr(), sols[i].data_ptr(),
                status[i].data_ptr(), costs[i].data_ptr(), stats_[i].data_ptr(), rng_seed=1234,
                problem_offset=(i * world + rank) * B, stream=st.cuda_stream, slot=slot)
            ev[i][1].record(st)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def final_gather():
        # the only collective of the path: gather the shard results (RCCL over xGMI)
        torch.cuda.synchronize()
        dist.all_gather_into_tensor(gathered[0], torch.cat(sols[W:W + K]))
        dist.all_gather_into_tensor(gathered[1], torch.cat(status[W:W + K]))

    # Reserve every slot's scratch (untimed): allocation + constant upload must not land in the timed
    # region when W < streams.
    for slot in range(S):
        solver.reserve(params, B, slot=slot, stream=streams[slot].cuda_stream)
    torch.cuda.synchronize()
    for i in range(W):
        run_step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(W, W + K):
        run_step(i)
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
    converged = (st_all == pk.SUCCESS).sum().to(torch.float64)
    evals = torch.stack(stats_[W:W + K])[:, :, 0].sum().to(torch.float64)
    gens = (torch.stack(stats_[W:W + K])[:, :, 1] & 0xFFFFFFFF).to(torch.float64).mean()
    totals = torch.stack([converged, evals, gens])
    if use_dist:
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
        # the gathered copy must hold exactly this rank's results at this rank's position
        lo = rank * K * B
        assert torch.equal(gathered[1][lo:lo + K * B], torch.cat(status[W:W + K]))
    converged_total, evals_total = float(totals[0]), float(totals[1])
    mean_gens = float(totals[2]) / world
    launch_ms = [ev[i][0].elapsed_time(ev[i][1]) for i in range(W, W + K)]

    result_line = None
    if rank == 0:
        # Per-launch duration.  `raw` = HIP-event time from the first pass to the last of one batch;
        # with S overlapping streams every batch shares the chip with S-1 others, so raw durations
        # overlap S-fold.  The duration one launch effectively occupies the chip for is
        # wall / launches (= raw when S = 1); the roofline uses that one.
        raw_launch_s = float(np.mean(launch_ms)) * 1e-3
        avg_launch_s = elapsed / K
        flop_per_launch = evals_total / (K * world) * FLOP_PER_EVAL.get(D, 236.0 * D)
        achieved_tflops = flop_per_launch / avg_launch_s / 1e12
        hbm_bytes_per_launch = B * (56 + 8 * D + 8 * D + 4 + 8)
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "converged IK solves/sec (7-DOF Panda, batched random targets)",
            "value": converged_total / elapsed,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.robot} {D}-DOF memetic IK (BASELINE configs[1]): population "
                            f"{args.population}, elites {args.elites}, batch {B} random reachable "
                            f"targets per GPU per step, seed = ready pose, max_generations "
                            f"{args.max_generations}, gd_max_iters {params.memetic_gd_max_iters}",
                "batch_per_gpu": B,
                "streams": S,
                "host_enqueue_ms_per_step": t_enqueued / K * 1e3,
                "success_rate": converged_total / (K * B * world),
                "mean_generations": mean_gens,
                "mean_cost_evals_per_solve": evals_total / (K * B * world),
                "parallelism": f"shard{world}",
            },
            "roofline": {
                "bound": "fp64_valu",
                "kernel": solver.kernel_name(params),
                "achieved": achieved_tflops,
                "peak": PEAK_FP64_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved_tflops / PEAK_FP64_VALU_TFLOPS,
                "avg_launch_ms": avg_launch_s * 1e3,
                "raw_event_launch_ms": raw_launch_s * 1e3,
                "flop_per_launch": flop_per_launch,
                "note": "launch = one 4096-problem batch (all its compaction passes). algorithmic "
                        "FP64 flops = reference cost_fn evaluations (literal counter kept by the "
                        "kernel) x 1.65 kflop; the fast build executes fewer (frame-based gradient "
                        "probes). avg_launch_ms = wall / launches (steady-state occupancy of the "
                        "chip by one launch; S streams overlap), raw_event_launch_ms = first-pass-"
                        "to-last-pass HIP-event time of one batch while sharing the chip.",
                "hbm": {"achieved": hbm_bytes_per_launch / avg_launch_s / 1e9, "peak": PEAK_HBM_GBS,
                        "unit": "GB/s",
                        "frac": hbm_bytes_per_launch / avg_launch_s / 1e9 / PEAK_HBM_GBS},
                "traffic": traffic,
            },
        }
        # ---- CPU baseline: the oracle (a port), all host cores, bounded sample ---------------
        if world == 1 and args.cpu_sample != 0:
            from oracle import oracle as O
            cores = O.max_threads()
            n = args.cpu_sample if args.cpu_sample > 0 else min(B, 128 * cores)
            o = O.Oracle(chain)
            g = goals[W].cpu().numpy()[:n]
            sd = np.tile(home, (n, 1))
            po = O.default_params(memetic_population_size=args.population,
                                  memetic_elite_size=args.elites,
                                  memetic_max_generations=args.max_generations)
            tc = time.perf_counter()
            _, ost, _, ostats = o.solve_batch(po, g, sd, rng_seed=1234, problem_offset=W * B,
                                              num_threads=cores)
            dt = time.perf_counter() - tc
            out["cpu_baseline"] = {
                "value": float((ost == O.SUCCESS).sum()) / dt,
                "unit": "solves/s",
                "cores": cores,
                "kind": "port",
                "sample": f"first {n} problems of the first timed batch, oracle/pik_oracle.c "
                          f"(plain C restatement, -O3), {dt:.1f} s wall",
                "success_rate": float((ost == O.SUCCESS).mean()),
                "mean_generations": float(ostats["generations"].mean()),
            }
            gpu_ok = (status[W][:n] == pk.SUCCESS).float().mean().item()
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
