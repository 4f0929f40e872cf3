// pik_host_solve.hpp -- ik_memetic / ik_gradient on the HOST for queries that carry a host cost function.
//
// A MoveIt caller may pass an IKCostFn; in the reference it is one more Goal per tip pose, weight 1,
// INSIDE cost_fn and under cost_threshold^2 in solution_fn (src/pick_ik_plugin.cpp:130-135,
// src/goal.cpp:146-161, 175-182, 188-203): it steers every gradient step and every child of the search.  An
// opaque host closure cannot run inside a GPU kernel, and a host round trip per cost evaluation costs 96 us
// against ~1 us for the evaluation.  Such queries are therefore solved HERE: the
// reference's algorithm as it runs it -- one problem after the other, sequential mating pool, std::sort with
// the tie order every implementation of this repository uses -- on the calling thread, with the arithmetic of
// the exact kernels compiled for the host (pik_math.hpp, this translation unit's flavour) and the library's
// counter-based random streams.  It is the SAME algorithm the exact kernels run, so without a callback it
// returns their bits; with one, the callback takes part in the search exactly as in the reference.
//
// This is not a general CPU path: the C ABI entry point (pikamd_solve_batch_host) refuses a call without a cost
// function -- those belong on the GPU.
//
// Restated from src/ik_gradient.cpp:14-139 and src/ik_memetic.cpp:18-373 (every function cites its lines); the
// evaluation counter is the reference's (one per cost_fn invocation).
//
// Compiled as the FAST flavour (no PIK_STRICT; tests/native/host_product_check.hip only, never part of a
// library) the same loop runs the PRODUCT kernels' arithmetic on the host: Denavit-Hartenberg forward kinematics,
// the gradient from the per-joint frames of the accept evaluation (probe_gradient), the line search by angle
// addition -- the source the kernels compile, expression for expression.  Every operation of that arithmetic is an
// IEEE operation or a correctly rounded square root (pik_math.hpp sqrt_pair), so this sequential execution must
// return the product kernels' bits: the direct whole-solve check of the benchmarked build
// (tests/test_gpu_product_arithmetic.py).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include <chrono>

#include "pik_solver.hpp"

namespace pik {

template <int D>
struct HostProblem {
    const ConstsK<D>* kc = nullptr;
    int n_tips = 1;
    GoalK goal;              // one tip
    GoalSet goals;           // several (pointer to n_tips x 7 doubles)
    double seed[D];          // ik_seed_state: the minimal-displacement reference
    pikamd_cost_fn cb = nullptr;
    void* user = nullptr;
    long long evals = 0;
    // wall-clock limits (options host_max_time / host_gd_max_time; steady-clock seconds, 0 = none)
    double deadline = 0.0, gd_max_time = 0.0;
};

inline double host_now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// the reference's `system_clock::now() < timeout_point` in front of a generation / a step, negated
template <int D>
inline bool host_expired(const HostProblem<D>& pb) {
    return pb.deadline > 0.0 && !(host_now() < pb.deadline);
}

// cost_fn AND the solution_fn verdict of one joint vector (src/goal.cpp:163-203), the verdict from the same
// forward kinematics as in the kernels; the host cost function appended as the reference appends it
template <int D>
inline void host_evaluate(HostProblem<D>& pb, const double (&q)[D], EvalOut& e) {
    const ChainK<D>& c = pb.kc->chain;
    const ParamsK& p = pb.kc->params;
    pb.evals++;
    if (pb.n_tips > 1) {
        double unused[D];
        eval_multi<D, false>(c, p, pb.goals, pb.seed, q, e, nullptr, 0, unused);
    } else {
        double tipt[3], d0[4];
        eval_pose<D, false>(c, p, pb.goal, pb.seed, q, e, tipt, d0, nullptr, 0);
    }
    if (!pb.cb) return;
    // goal_cost = sum over the goals, in the order the plugin pushes them (centre, avoid limits, minimal
    // displacement, then one callback per pose): the joint goals re-accumulated exactly as pose_tail / eval_multi
    // accumulate them, the callbacks behind; cost = pose_cost + goal_cost (src/goal.cpp:188-203)
    double gc = 0.0;
    if (p.goal_mask & 1) gc = gc + e.g0 * p.w_center_sq;
    if (p.goal_mask & 2) gc = gc + e.g1 * p.w_limits_sq;
    if (p.goal_mask & 4) gc = gc + e.g2 * p.w_disp_sq;
    bool ok = e.sol;
    for (int k = 0; k < pb.n_tips; ++k) {
        const double ck = pb.cb(q, D, k, pb.user) * 1.0; // eval * weight^2, weight 1
        gc = gc + ck;
        ok = ok && !(ck >= p.cost_thr_sq); // (src/goal.cpp:175-182 rejects on cost >= threshold: a NaN passes, as there)
    }
    e.cost = e.pc + gc;
    e.sol = ok;
}

// GradientIk -- include/pick_ik/ik_gradient.hpp:25-34 (+ the verdicts that travel with the costs)
template <int D>
struct HostGradientIk {
    double gradient[D], working[D], local[D], best[D];
    double local_cost, best_cost;
    bool local_sol, best_sol;
#if !defined(PIK_STRICT)
    // product arithmetic: what the accept evaluation of `local` leaves for the gradient and the line search
    double bsn[D], bcs[D];            // sines / cosines of the joints at `local`
    double frames[6 * D > 0 ? 6 * D : 1]; // world axis + origin of the joints (one tip: joints 1 .. D-1), stride 1
    double tipt[3], d0[4];
    EvalOut e_local;
    double gr_multi[D]; // several tips: the central differences come with the accept evaluation (eval_multi)
#endif
};

#if defined(PIK_STRICT)
// GradientIk::from -- src/ik_gradient.cpp:14-22
template <int D>
inline void host_gradient_from(HostGradientIk<D>& ik, HostProblem<D>& pb, const double (&guess)[D]) {
    EvalOut e;
    host_evaluate<D>(pb, guess, e);
    for (int i = 0; i < D; ++i) {
        ik.gradient[i] = 0.0;
        ik.working[i] = guess[i];
        ik.local[i] = guess[i];
        ik.best[i] = guess[i];
    }
    ik.local_cost = ik.best_cost = e.cost;
    ik.local_sol = ik.best_sol = e.sol;
}

// step -- src/ik_gradient.cpp:24-94
template <int D>
inline bool host_gd_step(HostGradientIk<D>& ik, HostProblem<D>& pb) {
    const ChainK<D>& c = pb.kc->chain;
    const double h = pb.kc->params.step_size;
    EvalOut e;
    for (int i = 0; i < D; ++i) { // :28-43
        ik.working[i] = ik.local[i] - h;
        host_evaluate<D>(pb, ik.working, e);
        const double p1 = e.cost;
        ik.working[i] = ik.local[i] + h;
        host_evaluate<D>(pb, ik.working, e);
        const double p3 = e.cost;
        ik.working[i] = ik.local[i];
        ik.gradient[i] = p3 - p1;
    }
    double sum = h; // :46-54
    for (int i = 0; i < D; ++i) sum = sum + std::fabs(ik.gradient[i]);
    const double f = 1.0 / sum * h;
    for (int i = 0; i < D; ++i) ik.gradient[i] = ik.gradient[i] * f;
    for (int i = 0; i < D; ++i) ik.working[i] = ik.local[i] - ik.gradient[i]; // :57-66
    host_evaluate<D>(pb, ik.working, e);
    const double p1 = e.cost;
    for (int i = 0; i < D; ++i) ik.working[i] = ik.local[i] + ik.gradient[i];
    host_evaluate<D>(pb, ik.working, e);
    const double p3 = e.cost;
    const double p2 = (p1 + p3) * 0.5;
    const double cost_diff = (p3 - p1) * 0.5; // :69-73
    double joint_diff = p2 / cost_diff;
    if (!std::isfinite(joint_diff)) joint_diff = 0.0;
    for (int i = 0; i < D; ++i) { // :77-81
#if PIK_XF
        const double updated = fma_f64(-ik.gradient[i], joint_diff, ik.local[i]);
#else
        const double updated = ik.local[i] - ik.gradient[i] * joint_diff;
#endif
        ik.working[i] = clamp_joint<D>(c, i, updated);
    }
    for (int i = 0; i < D; ++i) ik.local[i] = ik.working[i]; // :84-85
    host_evaluate<D>(pb, ik.local, e);
    ik.local_cost = e.cost;
    ik.local_sol = e.sol;
    if (ik.local_cost < ik.best_cost) { // :88-93
        for (int i = 0; i < D; ++i) ik.best[i] = ik.local[i];
        ik.best_cost = ik.local_cost;
        ik.best_sol = ik.local_sol;
        return true;
    }
    return false;
}

#else
// ---- product arithmetic (fast flavour): gradient_descent<D, MODE, 1> of pik_kernels.hpp, one lane ----
// the accept evaluation of ik.local: cost + verdict, the joints' sines / cosines and world frames
template <int D>
inline void host_accept_fast(HostGradientIk<D>& ik, HostProblem<D>& pb) {
    const ChainK<D>& c = pb.kc->chain;
    const ParamsK& p = pb.kc->params;
    pb.evals++;
    if (pb.n_tips > 1) {
        eval_multi<D, true>(c, p, pb.goals, pb.seed, ik.local, ik.e_local, ik.frames, 1, ik.gr_multi);
        return;
    }
    eval_pose_sc<D, true, false, 1>(c, p, pb.goal, pb.seed, ik.local, ik.e_local, ik.tipt, ik.d0, ik.frames, 1, ik.local,
                                    ik.bsn, ik.bcs);
}

// GradientIk::from -- src/ik_gradient.cpp:14-22
template <int D>
inline void host_gradient_from(HostGradientIk<D>& ik, HostProblem<D>& pb, const double (&guess)[D]) {
    for (int i = 0; i < D; ++i) {
        ik.gradient[i] = 0.0;
        ik.working[i] = guess[i];
        ik.local[i] = guess[i];
        ik.best[i] = guess[i];
        ik.bsn[i] = ik.bcs[i] = 0.0;
    }
    host_accept_fast<D>(ik, pb);
    ik.local_cost = ik.best_cost = ik.e_local.cost;
    ik.local_sol = ik.best_sol = ik.e_local.sol;
}

// step -- src/ik_gradient.cpp:24-94 as the product kernels take it: the 2D central differences from the frames of
// the accept evaluation (probe_gradient: same mathematics, term by term), the two line-search evaluations by
// angle addition when the step is small (ParamsK::line_delta).  The reference's evaluation counter advances by
// 2D + 3 all the same.
template <int D>
inline bool host_gd_step(HostGradientIk<D>& ik, HostProblem<D>& pb) {
    const ChainK<D>& c = pb.kc->chain;
    const ParamsK& p = pb.kc->params;
    const double h = p.step_size;
    double gr[D];
    if (pb.n_tips > 1) {
        for (int j = 0; j < D; ++j) gr[j] = ik.gr_multi[j];
    } else {
        probe_gradient<D, false>(c, p, pb.goal, pb.seed, ik.local, ik.e_local, ik.tipt, ik.d0, ik.frames, 1, gr);
    }
    pb.evals += 2 * D;
    double sum = h;
    for (int j = 0; j < D; ++j) sum = sum + fabs(gr[j]);
    const double f = 1.0 / sum * h;
    for (int j = 0; j < D; ++j) ik.gradient[j] = gr[j] * f;
    const bool line_delta = PIK_LINE_DELTA(p) && pb.n_tips == 1; // (several tips: full evaluations, as the kernels)
    double p13[2];
    for (int side = 0; side < 2; ++side) {
        for (int j = 0; j < D; ++j) ik.working[j] = side ? ik.local[j] + ik.gradient[j] : ik.local[j] - ik.gradient[j];
        EvalOut e;
        double tipt[3], d0[4];
        pb.evals++;
        if (pb.n_tips > 1) {
            double unused[D];
            eval_multi<D, false>(c, p, pb.goals, pb.seed, ik.working, e, nullptr, 0, unused);
        } else if (line_delta) {
            eval_pose_sc<D, false, true, 2>(c, p, pb.goal, pb.seed, ik.working, e, tipt, d0, nullptr, 0, ik.local, ik.bsn, ik.bcs);
        } else {
            double sn[D], cs[D], fr[6 * D > 0 ? 6 * D : 1];
            eval_pose_sc<D, true, false, 1>(c, p, pb.goal, pb.seed, ik.working, e, tipt, d0, fr, 1, ik.local, sn, cs);
        }
        p13[side] = e.cost;
    }
    const double p1 = p13[0], p3 = p13[1];
    const double p2 = (p1 + p3) * 0.5;
    const double cost_diff = (p3 - p1) * 0.5;
    double joint_diff = p2 / cost_diff;
    if (!std::isfinite(joint_diff)) joint_diff = 0.0;
    for (int j = 0; j < D; ++j) ik.local[j] = clamp_joint<D>(c, j, fma_f64(-ik.gradient[j], joint_diff, ik.local[j]));
    host_accept_fast<D>(ik, pb);
    ik.local_cost = ik.e_local.cost;
    ik.local_sol = ik.e_local.sol;
    if (ik.local_cost < ik.best_cost) {
        for (int i = 0; i < D; ++i) ik.best[i] = ik.local[i];
        ik.best_cost = ik.local_cost;
        ik.best_sol = ik.local_sol;
        return true;
    }
    return false;
}

#endif

struct HostResult {
    bool have = false, valid = false;
    double cost = 0.0;
    int generations = 0, wipeouts = 0, erasures = 0;
};

// ik_gradient -- src/ik_gradient.cpp:96-139 (wall-clock limit: the iteration budget binds)
template <int D>
inline HostResult host_ik_gradient(HostProblem<D>& pb, const double (&guess)[D], bool approx, double (&out)[D]) {
    const ParamsK& p = pb.kc->params;
    HostResult r;
    HostGradientIk<D> ik;
    host_gradient_from<D>(ik, pb, guess); // (also the verdict of the initial guess, :102-104)
    if (p.stop_on_valid && ik.best_sol) {
        pb.evals--; // the reference returns before it evaluates any cost
        std::memcpy(out, guess, sizeof out);
        r.have = r.valid = true;
        r.cost = ik.best_cost;
        return r;
    }
    int num_iterations = 0;
    double previous_cost = 0.0;
    while (!host_expired<D>(pb) && num_iterations < p.local_max_iters) { // :115
        if (host_gd_step<D>(ik, pb)) {
            if (p.stop_on_valid && ik.best_sol) { // :117-121
                std::memcpy(out, ik.best, sizeof out);
                r.have = r.valid = true;
                r.cost = ik.best_cost;
                r.generations = num_iterations + 1;
                return r;
            }
        }
        if (std::fabs(ik.local_cost - previous_cost) <= p.min_cost_delta) break;
        previous_cost = ik.local_cost;
        num_iterations++;
    }
    r.generations = num_iterations;
    if (!p.stop_on_valid && ik.best_sol) { // :130-138
        r.have = r.valid = true;
    } else if (approx) {
        r.have = true;
    }
    if (r.have) {
        std::memcpy(out, ik.best, sizeof out);
        r.cost = ik.best_cost;
    }
    return r;
}

// Individual -- include/pick_ik/ik_memetic.hpp:19-24 (+ the verdict of its genes, + the pre-sort slot: the tie
// order of the reference's unstable std::sort as every implementation here fixes it)
template <int D>
struct HostIndividual {
    double genes[D];
    double fitness;
    double extinction;
    double gradient[D];
    bool sol;
    int slot;
};

// MemeticIk -- include/pick_ik/ik_memetic.hpp:47-85
template <int D>
struct HostMemetic {
    std::vector<HostIndividual<D>> population;
    std::vector<int> mating_pool;
    HostIndividual<D> best, best_curr;
    bool has_previous = false;
    double previous_fitness = 0.0;
    std::vector<double> extinction_grading;
    double inverse_gene_size = 0.0;
    int P = 0, E = 0;
    unsigned long long rng_seed = 0, problem = 0;
    unsigned species = 0;
    unsigned init_epoch = 0;
    int wipeouts = 0, erasures = 0;
    bool returned = false; // this species' ik_memetic_impl has returned a solution
};

// MemeticIk::computeExtinctions -- src/ik_memetic.cpp:57-64
template <int D>
inline void host_compute_extinctions(HostMemetic<D>& ik) {
    const double min_fitness = ik.population[0].fitness;
    const double max_fitness = ik.population[(size_t)ik.P - 1].fitness;
    for (int i = 0; i < ik.P; ++i)
        ik.population[(size_t)i].extinction =
            (ik.population[(size_t)i].fitness + min_fitness * (ik.extinction_grading[(size_t)i] - 1)) / max_fitness;
}

// Robot::set_random_valid_configuration -- src/robot.cpp:87-95, 23-30; repro: the slots of the REPRODUCE stream
// (words 2, 3 of gene j's block), else the INIT stream's (53-bit double j)
template <int D>
inline void host_random_configuration(const ChainK<D>& c, double (&config)[D], unsigned long long seed, uint32_t stream,
                                      unsigned long long problem, uint32_t epoch, uint32_t individual, bool repro) {
    for (int j = 0; j < D; ++j) {
        const uint32_t slot = repro ? (uint32_t)(2 * (1 + j) + 1) : (uint32_t)j;
        const U4 w = rng_block(seed, stream, problem, epoch, individual, slot >> 1);
        const double u = (slot & 1u) ? u01_from_words(w.z, w.w) : u01_from_words(w.x, w.y);
        const bool bounded = (c.bounded_mask >> j) & 1u;
        config[j] = bounded ? uniform_real(c.qmin[j], c.qmax[j], u) : uniform_real(config[j] - M_PI, config[j] + M_PI, u);
    }
}

// MemeticIk::initPopulation -- src/ik_memetic.cpp:93-117
template <int D>
inline void host_init_population(HostMemetic<D>& ik, HostProblem<D>& pb, const double (&initial_guess)[D]) {
    const ChainK<D>& c = pb.kc->chain;
    double guess[D];
    std::memcpy(guess, initial_guess, sizeof guess); // (may alias ik.best.genes)
    const uint32_t epoch = ik.init_epoch++;
    for (int i = 0; i < ik.P; ++i) {
        HostIndividual<D>& ind = ik.population[(size_t)i];
        double genotype[D];
        std::memcpy(genotype, guess, sizeof genotype);
        if (i > 0 && i < ik.E)
            host_random_configuration<D>(c, genotype, ik.rng_seed, STREAM_INIT, ik.problem, epoch,
                                         (uint32_t)i | (ik.species << 20), false);
        std::memcpy(ind.genes, genotype, sizeof genotype);
        ind.extinction = 1.0;
        for (int j = 0; j < D; ++j) ind.gradient[j] = 0.0;
        ind.slot = i;
        if (i < ik.E) { // (:100-104: the elites' fitness is computed once here ...)
            EvalOut e;
            host_evaluate<D>(pb, ind.genes, e);
        }
    }
    for (int i = 0; i < ik.P; ++i) { // (... and everyone's again, :111-113)
        EvalOut e;
        host_evaluate<D>(pb, ik.population[(size_t)i].genes, e);
        ik.population[(size_t)i].fitness = e.cost;
        ik.population[(size_t)i].sol = e.sol;
    }
    host_compute_extinctions<D>(ik);
    ik.has_previous = false;
}

// MemeticIk::MemeticIk / from -- src/ik_memetic.cpp:18-41
template <int D>
inline void host_memetic_from(HostMemetic<D>& ik, HostProblem<D>& pb, const double (&initial_guess)[D],
                              unsigned long long rng_seed, unsigned long long problem, unsigned species) {
    const ParamsK& p = pb.kc->params;
    ik.P = p.population;
    ik.E = p.elites;
    ik.population.assign((size_t)ik.P, HostIndividual<D>{});
    ik.mating_pool.assign((size_t)ik.E, 0);
    ik.extinction_grading.assign((size_t)ik.P, 0.0);
    EvalOut e;
    host_evaluate<D>(pb, initial_guess, e);
    for (int i = 0; i < D; ++i) {
        ik.best.genes[i] = initial_guess[i];
        ik.best.gradient[i] = 0.0;
    }
    ik.best.fitness = e.cost;
    ik.best.sol = e.sol;
    ik.best.extinction = 0.0;
    ik.best.slot = 0;
    ik.best_curr = ik.best;
    for (int i = 0; i < ik.P; ++i) ik.extinction_grading[(size_t)i] = (double)i / (double)(ik.P - 1);
    ik.inverse_gene_size = 1.0 / (double)D;
    ik.rng_seed = rng_seed;
    ik.problem = problem;
    ik.species = species;
}

// MemeticIk::gradientDescent -- src/ik_memetic.cpp:66-91 (5 ms wall budget: the iteration budget binds)
template <int D>
inline void host_gradient_descent(HostMemetic<D>& ik, int i, HostProblem<D>& pb) {
    const ParamsK& p = pb.kc->params;
    HostIndividual<D>& individual = ik.population[(size_t)i];
    HostGradientIk<D> local_ik;
    host_gradient_from<D>(local_ik, pb, individual.genes);
    int num_iterations = 0;
    double previous_cost = 0;
    // :75-78: this descent's own time limit and nothing else -- the reference tests the query's max_time only
    // between generations (:226-228), so an elite's descent may run past it by up to gd_max_time
    const double limit = pb.gd_max_time > 0.0 ? host_now() + pb.gd_max_time : 0.0;
    while (!(limit > 0.0 && !(host_now() < limit)) && num_iterations < p.gd_max_iters) {
        host_gd_step<D>(local_ik, pb);
        if (std::fabs(local_ik.local_cost - previous_cost) <= p.min_cost_delta) break;
        previous_cost = local_ik.local_cost;
        num_iterations++;
    }
    for (int j = 0; j < D; ++j) individual.genes[j] = local_ik.best[j];
    EvalOut e;
    host_evaluate<D>(pb, individual.genes, e); // (:88: the fitness is computed again)
    individual.fitness = e.cost;
    individual.sol = e.sol;
    for (int j = 0; j < D; ++j) individual.gradient[j] = local_ik.gradient[j];
}

// MemeticIk::reproduce -- src/ik_memetic.cpp:119-190 (the draws: DESIGN.md section 2)
template <int D>
inline void host_reproduce(HostMemetic<D>& ik, HostProblem<D>& pb, uint32_t generation) {
    const ChainK<D>& c = pb.kc->chain;
    int pool_size = ik.E;
    for (int i = 0; i < ik.E; ++i) ik.mating_pool[(size_t)i] = i;
    auto erase = [&](int individual_index) {
        for (int k = 0; k < pool_size; ++k)
            if (ik.mating_pool[(size_t)k] == individual_index) {
                for (int m = k; m + 1 < pool_size; ++m) ik.mating_pool[(size_t)m] = ik.mating_pool[(size_t)m + 1];
                pool_size--;
                ik.erasures++;
                return;
            }
    };
    for (int i = ik.E; i < ik.P; ++i) {
        HostIndividual<D>& child = ik.population[(size_t)i];
        const uint32_t ind = (uint32_t)i | (ik.species << 20);
        child.slot = i;
        EvalOut e;
        if (pool_size > 0) {
            const U4 w = rng_block(ik.rng_seed, STREAM_REPRODUCE, ik.problem, generation, ind, 0);
            const uint32_t pool = (uint32_t)pool_size;
            const int idxA = (int)(((uint64_t)w.x * pool) >> 32);
            const double mix_ratio = u01_from_words(w.z, w.w);
            int idxB = idxA;
            if (pool_size > 1) {
                idxB = (int)(((uint64_t)w.y * (pool - 1u)) >> 32);
                idxB += (idxB >= idxA) ? 1 : 0;
            }
            const int ia = ik.mating_pool[(size_t)idxA], ib = ik.mating_pool[(size_t)idxB];
            const HostIndividual<D>& parentA = ik.population[(size_t)ia];
            const HostIndividual<D>& parentB = ik.population[(size_t)ib];
            const double extinction = 0.5 * (parentA.extinction + parentB.extinction);
            const double mutation_prob = extinction * (1.0 - ik.inverse_gene_size) + ik.inverse_gene_size;
            for (int j = 0; j < D; ++j) {
                const U4 wj = rng_block(ik.rng_seed, STREAM_REPRODUCE, ik.problem, generation, ind, (uint32_t)(1 + j));
                double gene = mix_ratio * parentA.genes[j] + (1.0 - mix_ratio) * parentB.genes[j];
                gene += u01_from_word(wj.x) * parentA.gradient[j] + u01_from_word(wj.y) * parentB.gradient[j];
                const double original_gene = gene;
                if (u01_from_word(wj.z) < mutation_prob)
                    gene += extinction * c.hspan[j] * uniform_real(-1.0, 1.0, u01_from_word(wj.w));
                gene = clamp_joint<D>(c, j, gene);
                child.genes[j] = gene;
                child.gradient[j] = gene - original_gene;
            }
            host_evaluate<D>(pb, child.genes, e);
            child.fitness = e.cost;
            child.sol = e.sol;
            const double fa = parentA.fitness, fb = parentB.fitness;
            if (child.fitness < fa) erase(ia);
            if (child.fitness < fb) erase(ib);
        } else {
            host_random_configuration<D>(c, child.genes, ik.rng_seed, STREAM_REPRODUCE, ik.problem, generation, ind, true);
            host_evaluate<D>(pb, child.genes, e);
            child.fitness = e.cost;
            child.sol = e.sol;
            for (int j = 0; j < D; ++j) child.gradient[j] = 0.0;
        }
    }
}

// MemeticIk::sortPopulation -- src/ik_memetic.cpp:200-209 (ties: pre-sort slot)
template <int D>
inline void host_sort_population(HostMemetic<D>& ik) {
    for (int i = 0; i < ik.P; ++i) ik.population[(size_t)i].slot = i;
    std::sort(ik.population.begin(), ik.population.end(), [](const HostIndividual<D>& x, const HostIndividual<D>& y) {
        if (x.fitness < y.fitness) return true;
        if (y.fitness < x.fitness) return false;
        return x.slot < y.slot;
    });
    host_compute_extinctions<D>(ik);
    ik.best_curr = ik.population[0];
    if (ik.best_curr.fitness < ik.best.fitness) ik.best = ik.best_curr;
}

// one pass of the loop body of ik_memetic_impl -- src/ik_memetic.cpp:229-261; true: `return ik.best()` at :252-255
template <int D>
inline bool host_memetic_generation(HostMemetic<D>& ik, HostProblem<D>& pb, uint32_t iter) {
    const ParamsK& p = pb.kc->params;
    for (int i = 0; i < ik.E; ++i) host_gradient_descent<D>(ik, i, pb);
    host_reproduce<D>(ik, pb, iter);
    host_sort_population<D>(ik);
    if (p.stop_on_valid && ik.best.sol) return true;
    // checkWipeout -- src/ik_memetic.cpp:43-55
    bool wipeout = false;
    if (ik.has_previous) wipeout = !(ik.best_curr.fitness < ik.previous_fitness - p.wipeout_tol);
    if (!wipeout) {
        ik.previous_fitness = ik.best_curr.fitness;
        ik.has_previous = true;
    } else {
        ik.wipeouts++;
        host_init_population<D>(ik, pb, ik.best.genes);
    }
    return false;
}

// ik_memetic -- src/ik_memetic.cpp:285-373 with ik_memetic_impl :211-283 (several species: the lock-step schedule
// of the reference's race that the oracle and the kernels run)
template <int D>
inline HostResult host_ik_memetic(HostProblem<D>& pb, const double (&guess)[D], unsigned long long rng_seed,
                                  unsigned long long problem, bool approx, int num_threads, bool stop_on_first,
                                  double (&out)[D]) {
    const ParamsK& p = pb.kc->params;
    HostResult r;
    {
        EvalOut e; // :294-296: the initial guess is accepted when it already is a solution
        host_evaluate<D>(pb, guess, e);
        pb.evals--; // (solution_fn only; the cost is evaluated below when it is not)
        if (p.stop_on_valid && e.sol) {
            std::memcpy(out, guess, sizeof out);
            r.have = r.valid = true;
            r.cost = e.cost;
            return r;
        }
    }
    const int S = num_threads <= 1 ? 1 : num_threads;
    std::vector<HostMemetic<D>> ik((size_t)S);
    for (int s = 0; s < S; ++s) {
        host_memetic_from<D>(ik[(size_t)s], pb, guess, rng_seed, problem, (unsigned)s);
        host_init_population<D>(ik[(size_t)s], pb, guess);
    }
    int iter = 0;
    bool terminate = false;
    while (!host_expired<D>(pb) && iter < p.max_generations && !terminate) { // :228
        int running = 0;
        for (int s = 0; s < S; ++s) {
            if (ik[(size_t)s].returned) continue;
            if (host_memetic_generation<D>(ik[(size_t)s], pb, (uint32_t)iter)) {
                ik[(size_t)s].returned = true;
                if (S == 1 || stop_on_first) terminate = true;
            } else {
                running++;
            }
        }
        if (running == 0) terminate = true;
        iter++;
    }
    r.generations = iter;
    r.wipeouts = ik[0].wipeouts;
    r.erasures = ik[0].erasures;
    double min_cost = 1.7976931348623157e308;
    for (int s = 0; s < S; ++s) { // :299-311 / :337-371
        const HostMemetic<D>& m = ik[(size_t)s];
        bool has_value = false, is_valid = false;
        if (m.returned) {
            has_value = is_valid = true;
        } else if (!p.stop_on_valid && m.best.sol) { // post-loop of ik_memetic_impl, :272-282
            has_value = is_valid = true;
        } else if (approx) {
            has_value = true;
        }
        if (has_value && m.best.fitness < min_cost) {
            min_cost = m.best.fitness;
            std::memcpy(out, m.best.genes, sizeof out);
            r.cost = m.best.fitness;
            r.valid = is_valid;
            r.have = true;
        }
    }
    return r;
}

// the batch: what pikamd_solve_batch does for each problem, on the host (status / cost / stats conventions of
// include/pick_ik_amd.h; solution = ik_seed_state on failure, src/pick_ik_plugin.cpp:213-217)
template <int D>
int host_solve_batch(const pikamd_solver* s, const pikamd_params* p, const ParamsK& pk, long long B, const double* goal,
                     const double* seed, const double* guess, unsigned long long rng_seed, long long problem_offset,
                     pikamd_cost_fn cb, void* user, double* solution, int32_t* status, double* final_cost,
                     pikamd_stats* stats) {
    std::vector<char> mem(sizeof(ConstsK<D>));
    ConstsK<D>& kc = *new (mem.data()) ConstsK<D>;
    std::memset(&kc, 0, sizeof kc);
    kc.chain = make_chain_k<D>(s->chain);
    for (int k = 1; k < s->n_tips; ++k) kc.more[k - 1] = make_chain_k<D>(s->more[k - 1]);
    kc.n_tips = s->n_tips;
    kc.params = pk;
    for (long long b = 0; b < B; ++b) {
        HostProblem<D> pb;
        pb.kc = &kc;
        // max_time is a limit per QUERY in the reference (MemeticIkParams::max_time / GradientIkParams::max_time of one
        // ik_memetic / ik_gradient call): every problem of a batch gets its own.  With a limit set the answers depend
        // on the machine's load, as the reference's do.
        pb.deadline = s->opt.host_max_time > 0.0 ? host_now() + s->opt.host_max_time : 0.0;
        pb.gd_max_time = s->opt.host_gd_max_time;
        pb.n_tips = s->n_tips;
        pb.cb = cb;
        pb.user = user;
        const double* g7 = goal + 7 * (long long)s->n_tips * b;
        if (s->n_tips > 1) pb.goals.ptr = g7;
        else make_goal(g7, pb.goal);
        double ig[D], out[D];
        for (int j = 0; j < D; ++j) {
            pb.seed[j] = seed[b * D + j];
            ig[j] = guess ? guess[b * D + j] : pb.seed[j];
            out[j] = 0.0;
        }
        HostResult r;
        if (p->mode == 0)
            r = host_ik_memetic<D>(pb, ig, rng_seed, (unsigned long long)(problem_offset + b), pk.approx != 0,
                                   p->memetic_num_threads, pk.stop_on_first != 0, out);
        else
            r = host_ik_gradient<D>(pb, ig, pk.approx != 0, out);
        const long long evals = pb.evals;
        for (int j = 0; j < D; ++j) solution[b * D + j] = r.have ? out[j] : pb.seed[j];
        status[b] = r.have ? (r.valid ? 1 : 2) : -31; // SUCCESS / APPROXIMATE / NO_IK_SOLUTION
        if (final_cost) {
            if (r.have) {
                final_cost[b] = r.cost;
            } else {
                EvalOut e; // cost of the initial guess
                host_evaluate<D>(pb, ig, e);
                final_cost[b] = e.cost;
            }
        }
        if (stats) {
            stats[b].cost_evals = evals;
            stats[b].generations = r.generations;
            stats[b].wipeouts = r.wipeouts;
            stats[b].pool_erasures = r.erasures;
            stats[b].reserved = 0;
        }
    }
    return 0;
}

} // namespace pik
