// pik_kernels.hpp -- HIP kernels of the batched IK solver for gfx950 (wave64).
//
// Work decomposition of the memetic kernel (src/ik_memetic.cpp restated for a 64-wide wavefront):
//
//   * one GROUP of GS = pow2ceil(elite_size) lanes per IK problem, 64/GS problems per wavefront;
//     lane e of a group owns elite e (genes, gradient, fitness, extinction in VGPRs) and runs its
//     gradient descent -- where 93% of the reference's cost evaluations are;
//   * the P - E children of a generation are never stored: each round the GS lanes of a group
//     generate + evaluate GS children, the reference's *sequential* mating-pool semantics
//     (src/ik_memetic.cpp:127-179) are recovered exactly by accepting a round only up to the first
//     child that erases a parent and re-speculating the rest against the shrunken pool, and a
//     running top-GS set (keys in VGPRs, genes in LDS) replaces std::sort -- elite selection is
//     done with wavefront shuffles + ballots;
//   * every fitness value carries the solution_fn verdict of the same forward kinematics, so the
//     per-generation `solution_fn(best)` costs nothing;
//   * a wavefront is persistent: groups whose problem finished pull the next problem index from a
//     global atomic counter, so early convergence of some problems does not idle their lanes;
//   * no __syncthreads-scale synchronisation: a workgroup is exactly one wavefront.
//
// Gradient descent (src/ik_gradient.cpp) is one wave-cooperative routine with a single inlined
// evaluation body driven by a wave-uniform phase counter (accept -> [probes] -> line- -> line+ ->
// accept ...); the 2D finite-difference probes are computed from the per-joint world frames of
// the accept evaluation (kept in LDS, one column per lane) instead of 2D forward kinematics.
// The PIK_STRICT build keeps the literal 2D + 3 evaluations so that it can be compared bit for bit
// with the CPU oracle.
//
// HBM traffic is ~180 B per solve (goal + seed in, solution + status + cost out); the kernel is
// bound by FP64 VALU issue (FK chain products, sincos/atan2 polynomials), not by memory.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "pik_math.hpp"

namespace pik {

constexpr int WAVE = 64;
constexpr int PIKAMD_NO_IK_SOLUTION_K = -31; // moveit_msgs MoveItErrorCodes::NO_IK_SOLUTION

struct StatsK {
    long long cost_evals;
    int generations;
    int wipeouts;
    int pool_erasures;
    int reserved;
};

// One batch of a call (device copy of pikamd_batch).  A call solves the problems of n_batches
// batches as ONE pool: virtual problem index v in [0, sum B) belongs to the batch with
// start <= v < start + B.  Random streams are keyed by problem_offset + (v - start), so a batch gets
// the answers it gets when solved by a call of its own.
struct BatchK {
    long long start;         // first virtual index
    long long B;
    const double* goal;      // [B][7] ([B][n_tips][7] for a multi-tip chain)
    const double* seed;      // [B][D] ik_seed_state: displacement reference, returned on failure
    const double* guess;     // [B][D] start of the search (the host passes seed when the caller gave none)
    long long problem_offset;
    double* solution;        // [B][D]
    int* status;             // [B]
    double* cost;            // [B] or null
    StatsK* stats;           // [B] or null
    unsigned* completed;     // null, or a counter incremented (release, system scope) per finished problem
};

struct SolveArgs {
    long long B;             // problems of the whole call (all batches)
    const BatchK* batches;   // [n_batches] in device memory, ascending `start`
    int n_batches;
    int signal;              // some batch of the call has a completion counter
    unsigned long long rng_seed;
    unsigned long long* work_counter;
    int gs_log2;
    int fresh; // 1: problems start from their seeds; 0: they resume from the state arrays
    // ---- compaction passes (memetic mode) ----
    // A solve is cut into passes at fixed generation marks.  A problem still running when it
    // reaches `pause_gen` parks its state in HBM and appends its index to `list_out`; the next
    // pass packs the survivors densely, 64 / (GS * LPE) per wavefront, so the few long-running
    // problems of a batch stop pinning mostly idle wavefronts.  Survivors are sparse in the batch,
    // so the parked state is one contiguous record per problem (field r of problem b at
    // st_d[b * D_ROWS + r]): a park/resume touches ~6 cache lines instead of one line per field
    // (measured: 34 MB -> 12 MB of HBM traffic per 4096-problem batch against the field-major
    // layout).
    const int* list_in;       // problem indices of this pass (null: 0 .. B-1)
    const unsigned* n_in;     // number of entries of list_in (device memory)
    int* list_out;            // survivors
    unsigned* n_out;
    unsigned* done;           // wavefronts of this launch that have finished (see the kernel's end)
    int pause_gen;            // generation count at which a running problem is parked
    int pad_;
    // Device-side choice of the kernel variant of a pass: the host cannot know how many problems
    // survive to a pass, so it enqueues every candidate variant (lanes per elite 16, 8, ... 1) and
    // each one looks at the survivor count: it runs iff sel_lo < *n_in <= sel_hi, otherwise all its
    // wavefronts return at once.  Exactly one variant runs (and re-arms the pass's counters).
    unsigned sel_lo, sel_hi;
    double* st_d;             // [cap][D_ROWS]
    int* st_i;                // [cap][I_ROWS]
    long long* st_l;          // [cap][L_ROWS]
    long long cap;
    // ---- stored population (only for chains with unbounded variables) ----
    // When the mating pool runs empty the reference re-rolls child slot i around the CURRENT
    // content of population_[i] for unbounded variables (src/ik_memetic.cpp:181-184 ->
    // src/robot.cpp:26-28), i.e. around the previous generation's rank-i individual.  Bounded
    // chains never read that, so they stream children without storing them; chains with
    // continuous joints keep fitness + genes of every individual here, double-buffered by
    // generation parity, and rank them at the end of each generation.
    //   per problem and parity: P doubles fitness, P*D doubles genes, P ints order (rank -> slot)
    double* pop;              // null when every variable is bounded
    long long pop_stride;     // doubles per (problem, species, parity)
    // ---- species (memetic_num_threads > 1, src/ik_memetic.cpp:312-371) ----
    // pow2ceil(species) adjacent groups of a wavefront share one problem and advance in lock-step.
    int species;
    int sp_log2;
};

// ---- phase timing (experiments only, -DPIK_PHASE_TIMING; tools/phase_timing.py) ----
#if defined(PIK_PHASE_TIMING)
__device__ unsigned long long pik_phase_cycles[16];
// per-wavefront accumulators in (scalar) registers, flushed once at the end of the kernel; the
// scheduling barriers keep the compiler from moving work across a tick
#define PIK_T0()                                         \
    __builtin_amdgcn_sched_barrier(0);                   \
    pik_t_ = __builtin_readcyclecounter();               \
    __builtin_amdgcn_sched_barrier(0)
#define PIK_TICK(k)                                                   \
    do {                                                              \
        __builtin_amdgcn_sched_barrier(0);                            \
        const unsigned long long n_ = __builtin_readcyclecounter();   \
        pik_acc_[k] += n_ - pik_t_;                                   \
        pik_t_ = n_;                                                  \
        __builtin_amdgcn_sched_barrier(0);                            \
    } while (0)
#define PIK_TIMING_DECL() unsigned long long pik_t_ = 0, pik_acc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PIK_TIMING_FLUSH()                                                                           \
    do {                                                                                             \
        if (threadIdx.x == 0)                                                                        \
            for (int k_ = 0; k_ < 10; ++k_) atomicAdd(&pik_phase_cycles[k_], pik_acc_[k_]);          \
    } while (0)
#else
#define PIK_T0() (void)0
#define PIK_TICK(k) (void)0
#define PIK_TIMING_DECL() (void)0
#define PIK_TIMING_FLUSH() (void)0
#endif

// rows of the parked state
template <int D>
struct StateRows {
    // per elite e (E_MAX = 64 is never reached in practice; rows are addressed e * ELITE + k)
    static constexpr int ELITE = 2 * D + 3; // genes D, gradient D, fitness, extinction, solution flag
    static constexpr int BEST0(int E) { return E * ELITE; }          // best genes D
    static constexpr int SCAL0(int E) { return E * ELITE + D; }      // best_fit best_sol seed_cost prev_fit
    static constexpr int D_ROWS(int E) { return E * ELITE + D + 4; }
    static constexpr int I_ROWS = 10; // gen init_epoch wipeouts erasures has_prev need_init pop_guess act sp_has sp_val
    static constexpr int L_ROWS = 1; // gd_steps
};

// the batch a virtual problem index belongs to (binary search: a handful of cached loads, used
// when a problem starts, resumes and finishes -- never inside a generation)
__device__ __forceinline__ const BatchK* find_batch(const SolveArgs& a, long long v) {
    int lo = 0, hi = a.n_batches - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.batches[mid].start <= v)
            lo = mid;
        else
            hi = mid - 1;
    }
    return a.batches + lo;
}

// A problem's results are in memory: tell whoever polls the batch's completion counter.  Called by
// the ONE lane that stored all of the problem's results, so the release of the atomic orders them
// before the increment (no wave-wide fence: a __threadfence_system() in this place, even behind a
// branch that was never taken, miscompiled the strict multi-tip kernel for 9 joints -- garbage in
// values that had travelled through ds_bpermute, as if a wait for them had gone missing).
__device__ __forceinline__ void signal_completed(const BatchK* bk) {
    if (bk->completed) __hip_atomic_fetch_add(bk->completed, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

#define PIK_CONSTS(kc)                                                                \
    const PIK_CONSTANT ConstsK<D>* const kcc_ = (const PIK_CONSTANT ConstsK<D>*)(kc); \
    CK<D> c = kcc_->chain;                                                            \
    PK p = kcc_->params;                                                              \
    (void)p

template <int D>
__device__ __forceinline__ void load_goal(const double* __restrict__ g7, GoalK& g) {
    make_goal(g7, g);
}

// ---- one tip (GoalK) or several (GoalSet): the kernels are written once over the goal type ----
template <bool MULTI>
struct GoalSel {
    using type = GoalK;
};
template <>
struct GoalSel<true> {
    using type = GoalSet;
};

// the goal of a lane that has no problem (it still executes the evaluations, masked): any valid one
__device__ __forceinline__ void goal_reset(GoalK& g, const double*) {
    g.t[0] = g.t[1] = g.t[2] = 0.0;
    g.q[0] = 1.0;
    g.q[1] = g.q[2] = g.q[3] = 0.0;
}
__device__ __forceinline__ void goal_reset(GoalSet& g, const double* any_goal) { g.ptr = any_goal; }

// problem `prob`'s goal(s): [B][7] or [B][n_tips][7]
template <int D>
__device__ __forceinline__ void load_goals(CK<D>, const double* __restrict__ goal, long long prob, GoalK& g) {
    load_goal<D>(goal + 7 * prob, g);
}
template <int D>
__device__ __forceinline__ void load_goals(CK<D> c, const double* __restrict__ goal, long long prob, GoalSet& g) {
    g.ptr = goal + 7 * (prob * tip_count<D>(c));
}

// cost + verdict of one joint vector.
// Strict build: a REAL function call.  With every evaluation inlined the strict memetic kernels sat
// at 256 VGPRs + ~150 AGPRs + >200 SGPRs spilled into VGPR lanes, and the multi-tip kernel for nine
// joints came out wrong in ways that changed with every unrelated edit (garbage counters, then a
// hang, then different solutions) -- the verification build trades speed for a register budget the
// compiler handles comfortably.
#if defined(PIK_STRICT)
#define PIK_EVAL_FN __device__ __noinline__
#else
#define PIK_EVAL_FN __device__ __forceinline__
#endif
// OCC: the wavefronts per SIMD of the calling kernel.  It changes nothing in the code; as a template argument
// it gives the kernels compiled for two per SIMD their OWN instances of the called functions, whose only
// callers are those kernels -- the compiler then holds the instances to the 256-register budget too (no
// AGPRs; a single AGPR in a shared instance put the two-per-SIMD kernel of the exact flavour back to one).
// Exact flavours, ONE tip frame, chains of up to PIK_XEVAL_INLINE_MAXD variables: inlined after all.  A call costs the
// callee's saves of ~70 callee-saved registers, the candidate and the result through memory and their waits -- per
// child, 124 times a generation (interleaved A/B, seven variables: 58.45 -> 56.4 ms on the driver's pool); these
// kernels are far from the register cap (256 + ~200 of 512), the long chains and the several-tip kernels that
// were the reason for the calls keep them.
#ifndef PIK_XEVAL_INLINE_MAXD
#define PIK_XEVAL_INLINE_MAXD 10
#endif
template <int D, int OCC = 1>
__device__ __forceinline__ void evaluate_impl(CK<D> c_in, PK p_in, const GoalK& g, const double (&seed)[D],
                                              const double (&q)[D], EvalOut& e) {
#if defined(PIK_STRICT)
    CK<D> c = scalar_ref(c_in); // (a call: see scalar_ref)
    PK p = scalar_ref(p_in);
#else
    CK<D> c = c_in;
    PK p = p_in;
#endif
    double tipt[3], d0[4];
#if defined(PIK_STRICT)
    if constexpr (PIK_XUZ_D(D)) {
        const uint32_t xm = (PIK_XUA || c.uniform_z == 1u) ? c.uniform_z : 0u; // (pik_math.hpp UZ / UA: the same evaluation without the per-joint decisions)
        if (xm) {
            double R[9];
            if (xm == 1u) fk_uz<D, 1>(c, q, R, tipt);
            else fk_uz<D, 2>(c, q, R, tipt);
            pose_tail<D>(c, p, g, seed, q, R, tipt, e, d0);
            return;
        }
    }
#endif
    eval_pose<D, false>(c, p, g, seed, q, e, tipt, d0, nullptr, 0);
}
template <int D, int OCC = 1>
PIK_EVAL_FN void evaluate_call(CK<D> c_in, PK p_in, const GoalK& g, const double (&seed)[D], const double (&q)[D],
                               EvalOut& e) {
    evaluate_impl<D, OCC>(c_in, p_in, g, seed, q, e);
}
template <int D, int OCC = 1>
__device__ __forceinline__ void evaluate(CK<D> c_in, PK p_in, const GoalK& g, const double (&seed)[D],
                                         const double (&q)[D], EvalOut& e) {
#if defined(PIK_STRICT)
    if constexpr (D <= PIK_XEVAL_INLINE_MAXD) evaluate_impl<D, OCC>(c_in, p_in, g, seed, q, e);
    else evaluate_call<D, OCC>(c_in, p_in, g, seed, q, e);
#else
    evaluate_impl<D, OCC>(c_in, p_in, g, seed, q, e);
#endif
}
template <int D, int OCC = 1>
PIK_EVAL_FN void evaluate(CK<D> c_in, PK p_in, const GoalSet& g, const double (&seed)[D],
                          const double (&q)[D], EvalOut& e) {
#if defined(PIK_STRICT)
    CK<D> c = scalar_ref(c_in);
    PK p = scalar_ref(p_in);
#else
    CK<D> c = c_in;
    PK p = p_in;
#endif
    double unused[D];
    eval_multi<D, false>(c, p, g, seed, q, e, nullptr, 0, unused);
}

// Every kernel that exchanges data through LDS here runs workgroups of exactly ONE wavefront, whose
// lanes execute each instruction together and whose LDS operations are performed in program order.  A
// lane reading what another lane of its wavefront wrote therefore needs no hardware synchronisation at
// all -- only the compiler must keep the accesses in order.  __syncthreads() is a workgroup-scope fence +
// barrier: the barrier is dropped for 64-thread workgroups, but the fence still drains the LDS queue
// (s_waitcnt lgkmcnt(0)) after every group of writes, a stall of a lone wavefront's critical path at
// every exchange of the cooperative descent.  A wavefront-scope fence orders the accesses and costs no
// instruction.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

#define PIK_POP(a) (PIK_COMMON ? (double*)nullptr : (a).pop)

__device__ __forceinline__ double shfl_f64(double v, int src_lane) { return __shfl(v, src_lane, WAVE); }
__device__ __forceinline__ int shfl_i32(int v, int src_lane) { return __shfl(v, src_lane, WAVE); }

// ------------------------------------------------------------------------------------------
// Gradient descent: GradientIk + step() + the driver loops of MemeticIk::gradientDescent
// (src/ik_memetic.cpp:66-91) and ik_gradient (src/ik_gradient.cpp:96-139).
// All 64 lanes of the wavefront call this together; `active` masks lanes without work.
// ------------------------------------------------------------------------------------------
// the joint update of step() (src/ik_gradient.cpp:77-81) before the clamp, with the FMA spelled out
// so that every variant of the routine rounds it the same way
__device__ __forceinline__ double gd_update(double local, double grad, double joint_diff) {
#if defined(PIK_STRICT) && !PIK_XF
    return local - grad * joint_diff;
#else
    return fma_f64(-grad, joint_diff, local);
#endif
}

template <int D>
struct GdState {
    double local[D], best[D], grad[D];
    double local_cost, best_cost;
    bool best_sol; // solution_fn verdict of `best`
    int steps;     // step() calls made
    int iters;     // the reference's num_iterations counter
    int found;     // GD_LOCAL: 1 = in-loop solution return, 2 = initial guess already a solution
};

enum GdMode { GD_ELITE = 0, GD_LOCAL = 1, GD_SINGLE = 2 };

// LDS rows (64 doubles each, one column per lane) used inside gradient_descent:
//   0 .. 6D-1    per-joint world frames of the last accept evaluation
//   6D .. 7D-1   the accepted joint vector (LPE > 1: probes index it by a per-lane joint)
//   7D .. 8D-1   gradient exchange between the sub-lanes of an elite (LPE > 1)
// LPE >= 8: the cooperative gradient descent (gd_wide) has its own LDS layout, see WideLds
// ONE_TIP_LPE1: the one-lane, one-tip descent of the product build does not store the first joint's
// frame (a chain constant): 6 (D - 1) rows
constexpr int GD_ROWS(int D, int LPE = 2, bool one_tip = false) {
    // (LPE >= 8, several tips: the team block also holds the problem's goals, see WideLdsM)
    const int rows = LPE == 1 ? (one_tip ? 6 * (D - 1) : 6 * D)
                     : LPE < 8 ? 8 * D
                               : ((WAVE / (LPE / 2)) * (14 * D + 12 + 4 * (LPE / 2) + (one_tip ? 0 : 8 * MAX_TIPS)) + WAVE - 1) / WAVE;
#if defined(PIK_STRICT)
    // ... or what the exact flavour's descent keeps (pik_exact.hpp ExactLds), whichever is larger.  One / two lanes
    // per elite: the exact flavours never store the per-joint frames the 6 D / 8 D rows above are for -- 5 D rows (the
    // fork forms: ExactLds 4 D, ExactFloatLds 5 D; the literal routine's probe costs: 2 D), so that eight wavefronts of
    // the two-lane kernel share a CU's LDS (its two-per-SIMD build, pik_launch.hpp)
    const int exact = LPE >= 4 ? 2 * D + 2 + ((15 * D + 24) * (WAVE / LPE) + WAVE - 1) / WAVE : 5 * D;
    if (LPE <= 2) return exact;
    return rows > exact ? rows : exact;
#else
    return rows;
#endif
}

// LPE = lanes per elite.  With LPE > 1 the LPE adjacent lanes [ebase, ebase + LPE) hold the same
// GradientIk state; they split the 2D probes (joint j goes to sub-lane j % LPE), evaluate the two
// line-search probes simultaneously (sub-lane parity picks q - g / q + g) and all repeat the
// accept evaluation, so a step costs 2 evaluations + ceil(D/LPE) probes instead of 3 + D.  Every
// lane performs exactly the arithmetic the LPE = 1 code performs, so results are bit-identical.
template <int D, int MODE, int LPE, typename G>
__device__ __forceinline__ void gradient_descent(CK<D> c, PK p, const G& g,
                                                 const double (&seed)[D],
                                                 const double* __restrict__ seed_gptr,
                                                 GdState<D>& s, bool active, int max_iters,
                                                 double* lds, int lane, int sub) {
    constexpr int PH_ACCEPT = 0, PH_PROBE = 1, PH_LINE1 = 2, PH_LINE2 = 3;
    constexpr int LOC0 = 6 * D, GSH0 = 7 * D;
    double* const fr = lds + lane;
    const int ebase = lane - sub;
    const double h = p.step_size;
    bool done = !active;
    bool first = true;
    int ph = PH_ACCEPT; // wave-uniform
    int probe = 0;      // wave-uniform (strict build)
    int num_iterations = 0;
    double previous_cost = 0.0;
    double p1 = 0.0, p3 = 0.0, pm0 = 0.0;
    double q_eval[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        q_eval[j] = s.local[j];
        s.grad[j] = 0.0;
    }
    s.steps = 0;
    s.iters = 0;
    s.found = 0;
    (void)probe;
    (void)pm0;
    (void)PH_PROBE;
    (void)ebase;
    (void)seed_gptr;
    (void)LOC0;
    (void)GSH0;

    constexpr bool IS_MULTI = std::is_same<G, GoalSet>::value; // several tip frames
    static_assert(!IS_MULTI || LPE <= 2, "several tips: one lane per elite, or two (the line-search pair)");
    double bsn[D], bcs[D]; // sines / cosines of the joints at the accepted point (s.local)
#pragma unroll
    for (int j = 0; j < D; ++j) bsn[j] = bcs[j] = 0.0;
    const bool line_delta = PIK_LINE_DELTA(p);
    (void)line_delta;
    while (__any(!done)) {
        EvalOut e;
        double tipt[3], d0[4];
        double gr_multi[D]; // several tips, fast build: the probes come with the accept evaluation
        (void)gr_multi;
#if defined(PIK_STRICT)
        evaluate<D>(c, p, g, seed, q_eval, e); // (a call, see evaluate)
#else
        if constexpr (IS_MULTI) {
            if (ph == PH_ACCEPT) {
                eval_multi<D, true>(c, p, g, seed, q_eval, e, fr, WAVE, gr_multi);
            } else {
                eval_multi<D, false>(c, p, g, seed, q_eval, e, nullptr, 0, gr_multi);
            }
        } else {
            // the two line-search evaluations sit a tiny step away from the accepted point: their
            // sines / cosines come from the accepted point's (sincos_delta) -- valid for the step
            // sizes anyone uses (<= 1e-3 rad; larger ones take the full evaluation)
            if (line_delta && ph != PH_ACCEPT)
                eval_pose_sc<D, false, true, 2>(c, p, g, seed, q_eval, e, tipt, d0, nullptr, 0, s.local, bsn, bcs);
            else
                eval_pose_sc<D, true, LPE != 1, 1>(c, p, g, seed, q_eval, e, tipt, d0, fr, WAVE, s.local, bsn, bcs);
        }
#endif
        (void)tipt;
        (void)d0;
        if (ph == PH_ACCEPT) {
            if (first) {
                // GradientIk::from -- src/ik_gradient.cpp:14-22
                first = false;
                if (MODE != GD_SINGLE) {
                    s.local_cost = e.cost;
                    s.best_cost = e.cost;
                }
                s.best_sol = e.sol;
                if (!done) {
                    if (MODE == GD_LOCAL && p.stop_on_valid && e.sol) {
                        s.found = 2; // ik_gradient early return, src/ik_gradient.cpp:102-104
                        done = true;
                    } else if (max_iters <= 0) {
                        done = true;
                    }
                }
            } else if (!done) {
                // tail of step(): always accept, update best -- src/ik_gradient.cpp:84-93
                s.local_cost = e.cost;
                s.steps += 1;
                const bool improved = e.cost < s.best_cost;
                if (improved) {
#pragma unroll
                    for (int j = 0; j < D; ++j) s.best[j] = s.local[j];
                    s.best_cost = e.cost;
                    s.best_sol = e.sol;
                }
                if (MODE == GD_SINGLE) {
                    done = true;
                } else if (MODE == GD_LOCAL && improved && p.stop_on_valid && e.sol) {
                    s.found = 1; // src/ik_gradient.cpp:117-121
                    s.iters = num_iterations + 1;
                    done = true;
                } else if (fabs(e.cost - previous_cost) <= p.min_cost_delta) {
                    s.iters = num_iterations;
                    done = true;
                } else {
                    previous_cost = e.cost;
                    num_iterations += 1;
                    s.iters = num_iterations;
                    if (num_iterations >= max_iters) done = true;
                }
            }
            // head of the next step(): gradient direction -- src/ik_gradient.cpp:28-54
#if defined(PIK_STRICT)
            if (!done) {
#pragma unroll
                for (int j = 0; j < D; ++j) s.grad[j] = 0.0;
            }
            probe = 0;
            {
                // LPE > 1: the 2D literal probe evaluations are dealt out to the LPE lanes of the elite,
                // probe `probe + sub` to sub-lane `sub` (a lane beyond 2D evaluates the accepted point)
                const int pr = probe + sub;
                const int ni = pr >> 1;
                const double dh = (pr < 2 * D) ? ((pr & 1) ? h : -h) : 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) q_eval[j] = s.local[j] + ((j == ni) ? dh : 0.0);
            }
            ph = PH_PROBE;
#else
            {
                double gr[D];
                if constexpr (IS_MULTI) {
#pragma unroll
                    for (int j = 0; j < D; ++j) gr[j] = gr_multi[j];
                } else if constexpr (LPE == 1) {
                    probe_gradient<D, false>(c, p, g, seed, s.local, e, tipt, d0, fr, WAVE, gr);
                } else {
                    // this sub-lane's share of the probes, joint index per lane
                    constexpr int KP = (D + LPE - 1) / LPE;
                    PK pf = fresh_after(p, e.cost);
                    ProbeBase pb;
                    make_probe_base(g, tipt, d0, e, pb);
                    const uint32_t prismatic_mask = PIK_PRISMATIC(c), bounded_mask = c.bounded_mask;
#pragma unroll
                    for (int j = 0; j < D; ++j) fr[(LOC0 + j) * WAVE] = s.local[j];
                    wave_sync();
                    // a ROLLED loop (the joint index is per lane anyway, nothing in the body depends on a
                    // compile-time k).  Unrolled, memetic_kernel<10, 2> -- 256 VGPRs + 250 AGPRs + ~400 SGPRs
                    // spilled into VGPR lanes -- came out wrong after an arithmetic simplification inside
                    // probe_joint (results of most problems garbage, a hang with compaction passes; the
                    // same source at LPE 4, and LPE 2 for every other chain length, was right): the third
                    // register-cap miscompile of this code base, caught by the shape-invariance fuzz.
#pragma unroll 1
                    for (int k = 0; k < KP; ++k) {
                        const int j = k * LPE + sub;
                        const bool valid = j < D;
                        const int jj = valid ? j : 0;
                        const double a[3] = {fr[(6 * jj + 0) * WAVE], fr[(6 * jj + 1) * WAVE], fr[(6 * jj + 2) * WAVE]};
                        const double o[3] = {fr[(6 * jj + 3) * WAVE], fr[(6 * jj + 4) * WAVE], fr[(6 * jj + 5) * WAVE]};
                        const double qj = fr[(LOC0 + jj) * WAVE];
                        JointGoalConsts jc;
                        jc.qmin = jc.qmax = jc.mid = jc.hspan = jc.mdf = jc.seed = 0.0;
                        jc.bounded = (bounded_mask >> jj) & 1u;
                        if (PIK_GM(pf)) {
                            CK<D> cf = fresh(c);
                            jc.qmin = cf.qmin[jj];
                            jc.qmax = cf.qmax[jj];
                            jc.mid = cf.mid[jj];
                            jc.hspan = cf.hspan[jj];
                            jc.mdf = cf.mdf[jj];
                            jc.seed = seed_gptr ? seed_gptr[jj] : 0.0;
                        }
                        const double gj = probe_joint(pf, e, pb, tipt, d0, a, o,
                                                      (prismatic_mask >> jj) & 1u, qj, jc);
                        if (valid) fr[(GSH0 + jj) * WAVE] = gj;
                    }
                    wave_sync();
#pragma unroll
                    for (int j = 0; j < D; ++j) gr[j] = lds[(GSH0 + j) * WAVE + ebase + (j % LPE)];
                    wave_sync();
                }
                double sum = h;
#pragma unroll
                for (int j = 0; j < D; ++j) sum = sum + fabs(gr[j]);
                const double f = 1.0 / sum * h;
                if (!done) {
#pragma unroll
                    for (int j = 0; j < D; ++j) s.grad[j] = gr[j] * f;
                }
                if (LPE == 1) {
#pragma unroll
                    for (int j = 0; j < D; ++j) q_eval[j] = s.local[j] - s.grad[j];
                    ph = PH_LINE1;
                } else {
                    // both line probes at once: even sub-lanes q - g, odd sub-lanes q + g
                    const double sg = (sub & 1) ? 1.0 : -1.0;
#pragma unroll
                    for (int j = 0; j < D; ++j) q_eval[j] = s.local[j] + sg * s.grad[j];
                    ph = PH_LINE2;
                }

            }
#endif
        } else if (ph == PH_PROBE) {
#if defined(PIK_STRICT)
            // literal central differences: probe 2i -> c(q - h e_i), probe 2i+1 -> c(q + h e_i)
            if constexpr (LPE > 1) {
                // every sub-lane has evaluated one probe: the costs meet in LDS (row = probe, column =
                // the elite's first lane), and once all 2D are there every lane of the elite forms
                // gradient[i] = p3 - p1 (src/ik_gradient.cpp:41) from the same two numbers
                const int pr = probe + sub;
                if (pr < 2 * D) lds[pr * WAVE + ebase] = e.cost;
                probe += LPE;
                if (probe < 2 * D) {
                    const int pn = probe + sub;
                    const int ni = pn >> 1;
                    const double dh = (pn < 2 * D) ? ((pn & 1) ? h : -h) : 0.0;
#pragma unroll
                    for (int j = 0; j < D; ++j) q_eval[j] = s.local[j] + ((j == ni) ? dh : 0.0);
                } else {
                    wave_sync();
                    double gr[D];
#pragma unroll
                    for (int j = 0; j < D; ++j) gr[j] = lds[(2 * j + 1) * WAVE + ebase] - lds[(2 * j) * WAVE + ebase];
                    wave_sync();
                    if (!done) {
#pragma unroll
                        for (int j = 0; j < D; ++j) s.grad[j] = gr[j];
                    }
                    double sum = h;
#pragma unroll
                    for (int j = 0; j < D; ++j) sum = sum + fabs(s.grad[j]);
                    const double f = 1.0 / sum * h;
                    if (!done) {
#pragma unroll
                        for (int j = 0; j < D; ++j) s.grad[j] = s.grad[j] * f;
                    }
                    // both line probes at once: even sub-lanes q - g, odd sub-lanes q + g
                    const double sg = (sub & 1) ? 1.0 : -1.0;
#pragma unroll
                    for (int j = 0; j < D; ++j) q_eval[j] = s.local[j] + sg * s.grad[j];
                    ph = PH_LINE2;
                }
            } else {
            const int i = probe >> 1;
            if (probe & 1) {
                const double gi = e.cost - pm0;
                if (!done) {
#pragma unroll
                    for (int j = 0; j < D; ++j) s.grad[j] += ((j == i) ? 1.0 : 0.0) * gi;
                }
            } else {
                pm0 = e.cost;
            }
            probe += 1;
            if (probe < 2 * D) {
                const int ni = probe >> 1;
                const double dh = (probe & 1) ? h : -h;
#pragma unroll
                for (int j = 0; j < D; ++j) q_eval[j] = s.local[j] + ((j == ni) ? dh : 0.0);
            } else {
                double sum = h;
#pragma unroll
                for (int j = 0; j < D; ++j) sum = sum + fabs(s.grad[j]);
                const double f = 1.0 / sum * h;
                if (!done) {
#pragma unroll
                    for (int j = 0; j < D; ++j) s.grad[j] = s.grad[j] * f;
                }
#pragma unroll
                for (int j = 0; j < D; ++j) q_eval[j] = s.local[j] - s.grad[j];
                ph = PH_LINE1;
            }
            } // LPE == 1
#endif
        } else if (ph == PH_LINE1) {
            p1 = e.cost;
#pragma unroll
            for (int j = 0; j < D; ++j) q_eval[j] = s.local[j] + s.grad[j];
            ph = PH_LINE2;
        } else {
            // secant step size + clamp -- src/ik_gradient.cpp:66-81
            if (LPE == 1) {
                p3 = e.cost;
            } else {
                p1 = shfl_f64(e.cost, ebase);
                p3 = shfl_f64(e.cost, ebase + 1);
            }
            const double p2 = (p1 + p3) * 0.5;
            const double cost_diff = (p3 - p1) * 0.5;
            double joint_diff = p2 / cost_diff;
            if (!isfinite(joint_diff)) joint_diff = 0.0;
            if (!done) {
                CK<D> cl = fresh_after(c, joint_diff); // limits: reloaded, not hoisted + spilled
#pragma unroll
                for (int j = 0; j < D; ++j)
                    s.local[j] = clamp_joint<D>(cl, j, gd_update(s.local[j], s.grad[j], joint_diff));
            }
#pragma unroll
            for (int j = 0; j < D; ++j) q_eval[j] = s.local[j];
            ph = PH_ACCEPT;
        }
    }
}

#if defined(PIK_STRICT)
} // namespace pik
#include "pik_exact.hpp" // the exact flavour's descent with the accept evaluation's work re-used by the probes
namespace pik {
#endif

// the descent of an elite / a local-mode problem / one step.  Exact flavour: one tip frame and no floating joint
// -> the memoised routine; several tips or a floating joint -> the literal one.
#if defined(PIK_STRICT)
// (the literal routine as a real call as well: see gradient_descent_exact)
template <int D, int MODE, int LPE>
__device__ __noinline__ void gradient_descent_literal(CK<D> c_in, PK p_in, const GoalK& g, const double (&seed)[D],
                                                      const double* seed_gptr, GdState<D>& s, bool active, int max_iters_in,
                                                      double* lds, int lane, int sub) {
    CK<D> c = scalar_ref(c_in);
    PK p = scalar_ref(p_in);
    const int max_iters = scalar_int(max_iters_in);
    gradient_descent<D, MODE, LPE>(c, p, g, seed, seed_gptr, s, active, max_iters, lds, lane, sub);
}
// LITERAL_OK = false: a kernel compiled for two wavefronts per SIMD.  The literal routine needs more than the
// 256 registers such a kernel has; the host never launches that variant for a chain with a floating joint.
template <int D, int MODE, int LPE, bool LITERAL_OK = true>
__device__ __forceinline__ void descent(CK<D> c, PK p, const GoalK& g, const double (&seed)[D], const double* seed_gptr,
                                        GdState<D>& s, bool active, int max_iters, double* lds, int lane, int sub) {
    if (PIK_XUZ_D(D) && c.uniform_z == 1u && p.goal_mask == 0) // (every joint revolute about z: pik_exact.hpp UZ; no joint goal on)
        gradient_descent_exact<D, MODE, LPE, LITERAL_OK ? 1 : 2, PIK_XUZ_D(D) ? 3 : 0>(c, p, g, seed, s, active, max_iters, lds, lane, sub);
    else if (PIK_XUZ_D(D) && c.uniform_z == 1u)
        gradient_descent_exact<D, MODE, LPE, LITERAL_OK ? 1 : 2, PIK_XUZ_D(D) ? 1 : 0>(c, p, g, seed, s, active, max_iters, lds, lane, sub);
    else if (PIK_XUZ_D(D) && PIK_XUA && c.uniform_z == 2u) // (... about x, y or z: UA)
        gradient_descent_exact<D, MODE, LPE, LITERAL_OK ? 1 : 2, PIK_XUZ_D(D) ? 2 : 0>(c, p, g, seed, s, active, max_iters, lds, lane, sub);
    else if (!LITERAL_OK || (c.float_mask == 0u && c.m_count == 0u)) // (a floating or a mimic joint: the literal routine)
        gradient_descent_exact<D, MODE, LPE, LITERAL_OK ? 1 : 2>(c, p, g, seed, s, active, max_iters, lds, lane, sub);
    else {
        // one floating joint, one or two lanes per elite: the fork form (pik_exact.hpp exact_accept_float)
        if constexpr (PIK_XFLOAT_FORK && LPE <= 2 && D >= 7) {
            if (x_float_fork_ok<D>(c)) {
                gradient_descent_exact_fork<D, MODE, LPE, GoalK>(c, p, g, seed, s, active, max_iters, lds, lane, sub);
                return;
            }
        }
        gradient_descent_literal<D, MODE, LPE>(c, p, g, seed, seed_gptr, s, active, max_iters, lds, lane, sub);
    }
}
template <int D, int MODE, int LPE, bool LITERAL_OK = true>
__device__ __forceinline__ void descent(CK<D> c, PK p, const GoalSet& g, const double (&seed)[D], const double* seed_gptr,
                                        GdState<D>& s, bool active, int max_iters, double* lds, int lane, int sub) {
    // several tip frames: the memoised routine (pik_exact.hpp) unless a floating or a mimic joint sits on some path
    if constexpr (PIK_XMULTI_MEMO && LPE <= 2) {
        if (x_multi_memo_ok<D>(c)) {
            gradient_descent_exact_fork<D, MODE, LPE, GoalSet>(c, p, g, seed, s, active, max_iters, lds, lane, sub);
            return;
        }
    }
    gradient_descent<D, MODE, LPE>(c, p, g, seed, seed_gptr, s, active, max_iters, lds, lane, sub);
}
#define PIK_DESCENT(MODE, LPE, ...) descent<D, MODE, LPE>(c, p, __VA_ARGS__)
#define PIK_DESCENT_OCC(MODE, LPE, OCC, ...) descent<D, MODE, LPE, (OCC) == 1>(c, p, __VA_ARGS__)
#else
#define PIK_DESCENT(MODE, LPE, ...) gradient_descent<D, MODE, LPE>(c, p, __VA_ARGS__)
#define PIK_DESCENT_OCC(MODE, LPE, OCC, ...) gradient_descent<D, MODE, LPE>(c, p, __VA_ARGS__)
#endif

// ------------------------------------------------------------------------------------------
// Cooperative ("wide") gradient descent: LPE >= 8 lanes per elite.
//
// The one-lane routine above is a chain of ~1000 dependent FP64 instructions per evaluation; the few
// problems of a batch that run all memetic_max_generations (unreachable targets) execute 100
// generations x 25 iterations x 2 evaluations of it back to back, and that latency -- not
// throughput -- bounds a batch.  Here the LPE lanes of an elite form two TEAMS of C = LPE / 2 lanes
// that evaluate one joint vector together:
//   * the state is TRANSPOSED: lane r of a team owns joint r (+ C, + 2C ...): its value, best
//     value, gradient component, limits -- updates, clamps and probes are lane-local;
//   * forward kinematics: every lane computes the sine / cosine of ITS joint (one sincos deep
//     instead of D), the three lanes r = 0, 1, 2 each carry one ROW of the running frame through
//     the chain (dh_row: a row of R' = R M only needs that row of R), the rows are gathered
//     through LDS and every lane evaluates the pose cost (replicated, no broadcast needed);
//   * the D finite-difference probes are one per lane; the two line-search probes run on the two
//     teams side by side.
// Every lane executes exactly the arithmetic the one-lane code executes for the same quantity
// (shared helpers with explicit FMAs, sums gathered and added in the same order), so the results
// are bit-identical to every other LPE -- asserted by the invariance tests.
template <int D, int C>
struct WideLds {
    static constexpr int KP = (D + C - 1) / C; // joints per lane
    static constexpr int SC0 = 0;              // [D][6]  sin, cos, tz of each joint + its DH constants a, cos / sin alpha
    static constexpr int FR0 = 6 * D;          // [D][6]  world axis + origin of each joint
    static constexpr int RT0 = 12 * D;         // [3][4]  rows of the tip frame (R | t)
    static constexpr int GG0 = 12 * D + 12;    // [D]     probe results
    static constexpr int QQ0 = 13 * D + 12;    // [D]     the evaluated joint vector
    // [C][4] a dump per lane: the lanes of a team that hold no joint / carry no row of the frame store
    // THERE instead of being masked off -- a store under a lane mask costs six scalar / lane-mask
    // instructions around it (the mask itself usually comes back from a spilled scalar register), and the
    // chain has one per joint
    static constexpr int DUM0 = 14 * D + 12;
    static constexpr int STRIDE = 14 * D + 12 + 4 * C; // doubles per team
};

// per-lane constants of the wide routine (loaded once per gradient descent)
template <int KP>
struct WideLane {
    double th0[KP], dd[KP], pm[KP];     // DH angle / shift offsets, prismatic flag (as 0.0 / 1.0)
    double clo[KP], chi[KP]; // clamp limits (ChainK::clo / chi)
    bool bounded[KP], valid[KP];
    int j[KP];
    int sc[KP], qq[KP], gg[KP]; // where this lane stores its joint's sine / cosine / shift, value, probe result
    int frb, frs, fr2, rtb;     // where it stores its row of the joint frames (base, stride per joint, offset
                                // of the origin component) and of the tip frame
    double brow[4]; // this lane's row of the base frame (rows 0..2; lanes r >= 3 shadow row 2)
};

// SCM: the sine / cosine exchange of fk_dh_joints (1: export at qk; 2: qk is qb plus a small step)
template <int D, int C, bool WANT_FRAMES, int SCM>
__device__ __forceinline__ void eval_wide(CK<D> c_in, PK p_in, const GoalK& g, const double (&seed)[D],
                                          const double (&qk)[WideLds<D, C>::KP],
                                          const WideLane<WideLds<D, C>::KP>& wl, int r, double* T,
                                          EvalOut& e, double (&tipt)[3], double (&d0)[4],
                                          const double (&qb)[WideLds<D, C>::KP],
                                          double (&bsn)[WideLds<D, C>::KP], double (&bcs)[WideLds<D, C>::KP]) {
    using L = WideLds<D, C>;
    MT mt = c_in.mt; // (unused: the coefficients are literals)
    // (1) every lane: sine / cosine and axial shift of its joint(s)
#pragma unroll
    for (int k = 0; k < L::KP; ++k) {
        double sn, cs;
        if (SCM == 2) {
            sincos_delta(bsn[k], bcs[k], (qk[k] - qb[k]) * (1.0 - wl.pm[k]), sn, cs);
        } else {
            double qa = qk[k]; // (angles beyond 10^4 revolutions: folded first, as fk_dh_joints does)
            if (!wave_all(fabs(qa) <= 65536.0)) qa = (wl.pm[k] != 0.0) ? qa : fold_2pi(mt, qa);
            sincos_f64<false>(mt, dh_angle(qa, wl.pm[k], wl.th0[k]), sn, cs);
            if (SCM == 1) {
                bsn[k] = sn;
                bcs[k] = cs;
            }
        }
        const double tz = dh_shift(qk[k], wl.pm[k], wl.dd[k]);
        T[wl.sc[k] + 0] = sn; // (a lane without a joint: its dump)
        T[wl.sc[k] + 1] = cs;
        T[wl.sc[k] + 2] = tz;
        T[wl.qq[k]] = qk[k];
    }
    wave_sync();
    // (2) lanes 0..2: one row of the frame through the chain
    double r0 = wl.brow[0], r1 = wl.brow[1], r2 = wl.brow[2], t = wl.brow[3];
    double o[12];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        if (WANT_FRAMES) {
            T[wl.frb + wl.frs * j] = r2;          // world joint axis = third column
            T[wl.frb + wl.frs * j + wl.fr2] = t;  // a point on it
        }
        // the joint's sine / cosine / shift and its constants all come out of LDS (gd_wide put the
        // constants there once): the reads do not depend on the running row, so they are in flight
        // long before they are needed -- as scalar loads pinned behind the sine they were one
        // scalar-cache round trip per joint on the critical path of a lone wavefront
        const double sn = T[L::SC0 + 6 * j + 0], cs = T[L::SC0 + 6 * j + 1], tz = T[L::SC0 + 6 * j + 2];
        const double a_j = T[L::SC0 + 6 * j + 3], ca_j = T[L::SC0 + 6 * j + 4], sa_j = T[L::SC0 + 6 * j + 5];
        if (j == 0) { // the tip transform: requested now, lands while the rows run down the chain
            CK<D> ct = fresh_after(c_in, sn);
#pragma unroll
            for (int i = 0; i < 12; ++i) o[i] = ct.dh_tip[i];
        }
        dh_row(r0, r1, r2, t, sn, cs, tz, a_j, ca_j, sa_j);
    }
    iso_row(r0, r1, r2, t, o);
    T[wl.rtb + 0] = r0;
    T[wl.rtb + 1] = r1;
    T[wl.rtb + 2] = r2;
    T[wl.rtb + 3] = t;
    wave_sync();
    // (3) every lane: the whole tip frame, pose cost + verdict (replicated)
    double R[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        R[i * 3 + 0] = T[L::RT0 + 4 * i + 0];
        R[i * 3 + 1] = T[L::RT0 + 4 * i + 1];
        R[i * 3 + 2] = T[L::RT0 + 4 * i + 2];
        tipt[i] = T[L::RT0 + 4 * i + 3];
    }
    double qfull[D];
#pragma unroll
    for (int j = 0; j < D; ++j) qfull[j] = 0.0;
    if (PIK_GM(p_in)) { // the joint goals sum over all joints, in the order the one-lane code adds them
#pragma unroll
        for (int j = 0; j < D; ++j) qfull[j] = T[L::QQ0 + j];
    }
    pose_tail<D>(c_in, p_in, g, seed, qfull, R, tipt, e, d0);
}

// GradientIk + step() + MemeticIk::gradientDescent's loop (GD_ELITE) for LPE >= 8 lanes per elite;
// MODE = GD_LOCAL: ik_gradient's loop (its early exits, src/ik_gradient.cpp:102-104, 117-121) for LPE
// lanes per problem
template <int D, int LPE, int MODE = GD_ELITE>
__device__ __forceinline__ void gd_wide(CK<D> c, PK p, const GoalK& g, const double (&seed)[D],
                                        const double* __restrict__ seed_gptr, GdState<D>& s, bool active,
                                        int max_iters, double* lds, int lane, int sub) {
    constexpr int C = LPE / 2;
    using L = WideLds<D, C>;
    constexpr int KP = L::KP;
    const int team = sub / C;  // 0: evaluates q - g in the line search, 1: q + g
    const int r = sub % C;
    const int ebase = lane - sub;
    double* const T = lds + (lane / C) * L::STRIDE;
    const double h = p.step_size;

    // ---- per-lane constants and the transposed state ----
    WideLane<KP> wl;
    double loc[KP], bst[KP], grd[KP];
    {
        CK<D> cl = fresh(c);
        const uint32_t prismatic_mask = PIK_PRISMATIC(cl), bounded_mask = cl.bounded_mask;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int j = r + k * C;
            wl.valid[k] = j < D;
            const int jj = wl.valid[k] ? j : 0;
            wl.j[k] = jj;
            wl.th0[k] = cl.dh[jj][0];
            wl.dd[k] = cl.dh[jj][1];
            if (wl.valid[k]) { // the joint's remaining constants go to LDS once (read by eval_wide's chain)
                T[L::SC0 + 6 * jj + 3] = cl.dh[jj][2];
                T[L::SC0 + 6 * jj + 4] = cl.dh[jj][3];
                T[L::SC0 + 6 * jj + 5] = cl.dh[jj][4];
            }
            wl.pm[k] = ((prismatic_mask >> jj) & 1u) ? 1.0 : 0.0;
            wl.clo[k] = cl.clo[jj];
            wl.chi[k] = cl.chi[jj];
            wl.bounded[k] = (bounded_mask >> jj) & 1u;
            const int dump = L::DUM0 + 4 * r;
            wl.sc[k] = wl.valid[k] ? L::SC0 + 6 * jj : dump;
            wl.qq[k] = wl.valid[k] ? L::QQ0 + jj : dump + 3;
            wl.gg[k] = wl.valid[k] ? L::GG0 + jj : dump;
            // this lane's joint value out of the replicated vector: selects between opaque COPIES
            // (a select chain over the array's elements is turned into a dynamically indexed load,
            // which sends the whole GdState to scratch memory: 26 scratch instructions, five stores
            // per iteration of the descent)
            double v = s.local[0];
#pragma unroll
            for (int m = 1; m < D; ++m) {
                double e = s.local[m];
                asm volatile("" : "+v"(e));
                v = (jj == m) ? e : v;
            }
            loc[k] = v;
            bst[k] = v;
            grd[k] = 0.0;
        }
        const int row = r < 3 ? r : 2;
        wl.frb = r < 3 ? L::FR0 + row : L::DUM0 + 4 * r;
        wl.frs = r < 3 ? 6 : 0;
        wl.fr2 = r < 3 ? 3 : 1;
        wl.rtb = r < 3 ? L::RT0 + 4 * row : L::DUM0 + 4 * r;
        wl.brow[0] = cl.dh_base[3 * row + 0];
        wl.brow[1] = cl.dh_base[3 * row + 1];
        wl.brow[2] = cl.dh_base[3 * row + 2];
        wl.brow[3] = cl.dh_base[9 + row];
    }
    bool done = !active;
    bool first = true;
    int num_iterations = 0;
    double previous_cost = 0.0;
    s.steps = 0;
    s.iters = 0;
    s.found = 0;
    double bsn[KP], bcs[KP]; // sine / cosine of this lane's joint(s) at the accepted point
#pragma unroll
    for (int k = 0; k < KP; ++k) bsn[k] = bcs[k] = 0.0;
    const bool line_delta = PIK_LINE_DELTA(p); // (see gradient_descent)

    while (__any(!done)) {
        // ---------------- accept evaluation at `loc` (both teams, redundantly) ----------------
        EvalOut e;
        double tipt[3], d0[4];
        eval_wide<D, C, true, 1>(c, p, g, seed, loc, wl, r, T, e, tipt, d0, loc, bsn, bcs);
        if (first) {
            // GradientIk::from -- src/ik_gradient.cpp:14-22
            first = false;
            s.local_cost = e.cost;
            s.best_cost = e.cost;
            s.best_sol = e.sol;
            if (!done) {
                if (MODE == GD_LOCAL && p.stop_on_valid && e.sol) {
                    s.found = 2; // ik_gradient early return, src/ik_gradient.cpp:102-104
                    done = true;
                } else if (max_iters <= 0) {
                    done = true;
                }
            }
        } else if (!done) {
            // tail of step(): always accept, update best -- src/ik_gradient.cpp:84-93
            s.local_cost = e.cost;
            s.steps += 1;
            const bool improved = e.cost < s.best_cost;
            if (improved) {
#pragma unroll
                for (int k = 0; k < KP; ++k) bst[k] = loc[k];
                s.best_cost = e.cost;
                s.best_sol = e.sol;
            }
            if (MODE == GD_LOCAL && improved && p.stop_on_valid && e.sol) {
                s.found = 1; // src/ik_gradient.cpp:117-121
                s.iters = num_iterations + 1;
                done = true;
            } else if (fabs(e.cost - previous_cost) <= p.min_cost_delta) {
                s.iters = num_iterations;
                done = true;
            } else {
                previous_cost = e.cost;
                num_iterations += 1;
                s.iters = num_iterations;
                if (num_iterations >= max_iters) done = true;
            }
        }
        // ---------------- head of the next step(): gradient direction ----------------
        {
            PK pf = fresh_after(p, e.cost);
            ProbeBase pb;
            make_probe_base(g, tipt, d0, e, pb);
            double gk[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int jj = wl.j[k];
                const double a[3] = {T[L::FR0 + 6 * jj + 0], T[L::FR0 + 6 * jj + 1], T[L::FR0 + 6 * jj + 2]};
                const double o[3] = {T[L::FR0 + 6 * jj + 3], T[L::FR0 + 6 * jj + 4], T[L::FR0 + 6 * jj + 5]};
                JointGoalConsts jc;
                jc.qmin = jc.qmax = jc.mid = jc.hspan = jc.mdf = jc.seed = 0.0;
                jc.bounded = wl.bounded[k];
                if (PIK_GM(pf)) {
                    CK<D> cf = fresh(c);
                    jc.qmin = cf.qmin[jj];
                    jc.qmax = cf.qmax[jj];
                    jc.mid = cf.mid[jj];
                    jc.hspan = cf.hspan[jj];
                    jc.mdf = cf.mdf[jj];
                    jc.seed = seed_gptr[jj];
                }
                gk[k] = probe_joint(pf, e, pb, tipt, d0, a, o, wl.pm[k] != 0.0, loc[k], jc);
                T[wl.gg[k]] = gk[k];
            }
            wave_sync();
            double sum = h;
#pragma unroll
            for (int j = 0; j < D; ++j) sum = sum + fabs(T[L::GG0 + j]);
            const double f = 1.0 / sum * h;
            if (!done) {
#pragma unroll
                for (int k = 0; k < KP; ++k) grd[k] = gk[k] * f;
            }
        }
        // ---------------- line search: q - g on team 0, q + g on team 1 ----------------
        double qe[KP];
        const double sg = team ? 1.0 : -1.0;
#pragma unroll
        for (int k = 0; k < KP; ++k) qe[k] = loc[k] + sg * grd[k];
        EvalOut e2;
        if (line_delta)
            eval_wide<D, C, false, 2>(c, p, g, seed, qe, wl, r, T, e2, tipt, d0, loc, bsn, bcs);
        else
            eval_wide<D, C, false, 0>(c, p, g, seed, qe, wl, r, T, e2, tipt, d0, loc, bsn, bcs);
        // secant step size + clamp -- src/ik_gradient.cpp:66-81
        const double p1 = shfl_f64(e2.cost, ebase);
        const double p3 = shfl_f64(e2.cost, ebase + C);
        const double p2 = (p1 + p3) * 0.5;
        const double cost_diff = (p3 - p1) * 0.5;
        double joint_diff = p2 / cost_diff;
        if (!isfinite(joint_diff)) joint_diff = 0.0;
        if (!done) {
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                loc[k] = clamp_lim(gd_update(loc[k], grd[k], joint_diff), wl.clo[k], wl.chi[k]); // clamp_joint
            }
        }
    }
    // ---- back to the replicated layout: best genes and the last normalised gradient ----
    wave_sync();
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (wl.valid[k]) {
            T[L::SC0 + wl.j[k]] = bst[k];
            T[L::SC0 + D + wl.j[k]] = grd[k];
        }
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < D; ++j) {
        s.best[j] = T[L::SC0 + j];
        s.grad[j] = T[L::SC0 + D + j];
    }
    wave_sync();
}

// ------------------------------------------------------------------------------------------
// The cooperative descent for SEVERAL tip frames (8 / 16 lanes per elite).
//
// An evaluation is a loop over the tips (as eval_multi's is): for tip k the lanes take the sines of
// `q + theta0` of tip k's chain, lanes 0..2 carry the frame rows down that chain, every lane forms tip k's
// pose error, and -- with the accept evaluation -- each lane probes ITS joint against tip k's joint frames and
// adds the result to its gradient component, in the order eval_multi adds them.  The LDS block of a team is
// the single-tip one plus the problem's goals; the constants of the tips' chains (per joint: theta0, d, a,
// cos / sin alpha, on-the-path flag, prismatic flag; base and tip transforms) are copied to LDS ONCE per
// kernel (stage_tip_constants) -- they are indexed by a per-lane joint, which a scalar load cannot do.
// No angle addition in the line search: the one-lane several-tip code has none either.
// ------------------------------------------------------------------------------------------
template <int D, int C>
struct WideLdsM : WideLds<D, C> {
    static constexpr int GL0 = WideLds<D, C>::STRIDE;                   // [MAX_TIPS][8] goals: t[3], q[4]
    static constexpr int STRIDE = WideLds<D, C>::STRIDE + 8 * MAX_TIPS; // doubles per team
    static constexpr int TCS = 7 * D + 24;                              // doubles per tip of the constants block
    static constexpr int TC_ROWS = (MAX_TIPS * TCS + WAVE - 1) / WAVE;
};

// tip-independent per-lane constants of the several-tip routine
template <int KP>
struct WideLaneM {
    double clo[KP], chi[KP];
    bool bounded[KP], valid[KP];
    int j[KP];
    int sc[KP], qq[KP], gg[KP];
    int frb, frs, fr2, rtb, row;
};

// once per kernel: the constants of every tip's chain into the LDS block TC (all lanes, strided)
template <int D>
__device__ __forceinline__ void stage_tip_constants(CK<D> c0, double* TC, int lane) {
    constexpr int TCS = 7 * D + 24;
    const int n_tips = tip_count<D>(c0);
    for (int tip = 0; tip < n_tips; ++tip) {
        CK<D> ck = tip_chain<D>(c0, tip);
        const uint32_t am = ck.active_mask, pmask = PIK_PRISMATIC(ck);
        for (int i = lane; i < TCS; i += WAVE) {
            double v;
            if (i < 7 * D) {
                const int j = i / 7, f = i - 7 * j;
                v = f < 5 ? ck.dh[j][f] : f == 5 ? (((am >> j) & 1u) ? 1.0 : 0.0) : (((pmask >> j) & 1u) ? 1.0 : 0.0);
            } else if (i < 7 * D + 12) {
                v = ck.dh_base[i - 7 * D];
            } else {
                v = ck.dh_tip[i - 7 * D - 12];
            }
            TC[tip * TCS + i] = v;
        }
    }
    wave_sync();
}

template <int D, int C, bool WANT_GRAD>
__device__ __forceinline__ void eval_wide_multi(CK<D> c0, PK p_in, const double (&seed)[D],
                                                const double* __restrict__ seed_gptr,
                                                const double (&qk)[WideLds<D, C>::KP],
                                                const WideLaneM<WideLds<D, C>::KP>& wl, int r, double* T,
                                                const double* TC, int n_tips, EvalOut& e,
                                                double (&gk)[WideLds<D, C>::KP]) {
    using L = WideLdsM<D, C>;
    constexpr int KP = L::KP;
    MT mt = c0.mt; // (unused: the coefficients are literals)
    double pc = 0.0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < KP; ++k) gk[k] = 0.0;
    e.lin = e.ang = e.vn = 0.0;
    (void)r;
#pragma unroll 1
    for (int tip = 0; tip < n_tips; ++tip) { // wave-uniform trip count
        const double* tc = TC + tip * L::TCS;
        // (1) every lane: sine / cosine and axial shift of its joint(s) in this tip's chain
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int jj = wl.j[k];
            const double th0 = tc[7 * jj + 0], dd = tc[7 * jj + 1], act = tc[7 * jj + 5], pm = tc[7 * jj + 6];
            double qa = qk[k]; // (angles beyond 10^4 revolutions: folded first, as fk_dh_joints does)
            if (!wave_all(fabs(qa) <= 65536.0)) qa = (pm != 0.0) ? qa : fold_2pi(mt, qa);
            const double qj = (act != 0.0) ? qa : 0.0; // a variable that is not on this tip's path counts as 0
            double sn, cs;
            sincos_f64<false>(mt, dh_angle(qj, pm, th0), sn, cs);
            const double tz = dh_shift(qj, pm, dd);
            T[wl.sc[k] + 0] = sn; // (a lane without a joint: its dump)
            T[wl.sc[k] + 1] = cs;
            T[wl.sc[k] + 2] = tz;
            T[wl.qq[k]] = qk[k];
        }
        wave_sync();
        // (2) lanes 0..2: one row of the frame through the chain
        {
            const double* tb = tc + 7 * D;
            double r0 = tb[3 * wl.row + 0], r1 = tb[3 * wl.row + 1], r2 = tb[3 * wl.row + 2], t = tb[9 + wl.row];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                if (WANT_GRAD) {
                    T[wl.frb + wl.frs * j] = r2;         // world joint axis = third column
                    T[wl.frb + wl.frs * j + wl.fr2] = t; // a point on it
                }
                const double sn = T[L::SC0 + 6 * j + 0], cs = T[L::SC0 + 6 * j + 1], tz = T[L::SC0 + 6 * j + 2];
                dh_row(r0, r1, r2, t, sn, cs, tz, tc[7 * j + 2], tc[7 * j + 3], tc[7 * j + 4]);
            }
            double o[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) o[i] = tb[12 + i];
            iso_row(r0, r1, r2, t, o);
            T[wl.rtb + 0] = r0;
            T[wl.rtb + 1] = r1;
            T[wl.rtb + 2] = r2;
            T[wl.rtb + 3] = t;
        }
        wave_sync();
        // (3) every lane: this tip's pose error (replicated)
        double R[9], tipt[3], d0[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            R[i * 3 + 0] = T[L::RT0 + 4 * i + 0];
            R[i * 3 + 1] = T[L::RT0 + 4 * i + 1];
            R[i * 3 + 2] = T[L::RT0 + 4 * i + 2];
            tipt[i] = T[L::RT0 + 4 * i + 3];
        }
        PK p = fresh_after(p_in, tipt[0]);
        GoalK g;
        g.t[0] = T[L::GL0 + 8 * tip + 0];
        g.t[1] = T[L::GL0 + 8 * tip + 1];
        g.t[2] = T[L::GL0 + 8 * tip + 2];
        g.q[0] = T[L::GL0 + 8 * tip + 3];
        g.q[1] = T[L::GL0 + 8 * tip + 4];
        g.q[2] = T[L::GL0 + 8 * tip + 5];
        g.q[3] = T[L::GL0 + 8 * tip + 6];
        const double dx = g.t[0] - tipt[0], dy = g.t[1] - tipt[1], dz = g.t[2] - tipt[2];
        EvalOut ek;
        ek.lin = sqrt_pos(dx * dx + dy * dy + dz * dz);
        double qt[4];
        matrix_to_quat(R, qt);
        quat_mul_conj(qt, g.q, d0);
        ek.ang = angle_of(mt, d0, ek.vn);
        PoseErr pe;
        pe.lin = ek.lin;
        pe.ang = ek.ang;
        pc = pc + pose_cost(p, pe);
        ok = ok && (!PIK_POS_TEST(p) || ek.lin <= p.pos_thr) && (!PIK_ORI_TEST(p) || fabs(ek.ang) <= p.ori_thr);
        // (4) with the accept evaluation: each lane's probe of its joint(s) against this tip's frames
        if (WANT_GRAD) {
            ek.g0 = ek.g1 = ek.g2 = 0.0;
            ProbeBase pb;
            make_probe_base(g, tipt, d0, ek, pb);
            JointGoalConsts jc;
            jc.qmin = jc.qmax = jc.mid = jc.hspan = jc.mdf = jc.seed = 0.0;
            jc.bounded = false;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int jj = wl.j[k];
                const double a[3] = {T[L::FR0 + 6 * jj + 0], T[L::FR0 + 6 * jj + 1], T[L::FR0 + 6 * jj + 2]};
                const double o[3] = {T[L::FR0 + 6 * jj + 3], T[L::FR0 + 6 * jj + 4], T[L::FR0 + 6 * jj + 5]};
                const double dj = probe_joint(p, ek, pb, tipt, d0, a, o, tc[7 * jj + 6] != 0.0, qk[k], jc, true, false);
                gk[k] += (tc[7 * jj + 5] != 0.0) ? dj : 0.0;
            }
        }
    }
    // joint goals: once per evaluation, over all variables in the one-lane order (src/goal.cpp:188-203)
    PK p = fresh_after(p_in, pc);
    CK<D> c = fresh_after(c0, pc);
    double cost = pc;
    e.g0 = e.g1 = e.g2 = 0.0;
    if (PIK_GM(p)) {
        double q[D];
#pragma unroll
        for (int j = 0; j < D; ++j) q[j] = T[L::QQ0 + j];
        double gc = 0.0;
        if (PIK_GM(p) & 1) {
            e.g0 = goal_cost_term<D>(c, p, 0, q, seed);
            const double w = e.g0 * p.w_center_sq;
            gc = gc + w;
            ok = ok && (w < p.cost_thr_sq);
        }
        if (PIK_GM(p) & 2) {
            e.g1 = goal_cost_term<D>(c, p, 1, q, seed);
            const double w = e.g1 * p.w_limits_sq;
            gc = gc + w;
            ok = ok && (w < p.cost_thr_sq);
        }
        if (PIK_GM(p) & 4) {
            e.g2 = goal_cost_term<D>(c, p, 2, q, seed);
            const double w = e.g2 * p.w_disp_sq;
            gc = gc + w;
            ok = ok && (w < p.cost_thr_sq);
        }
        cost = cost + gc;
        if (WANT_GRAD) {
            ProbeBase pb0;
            pb0.dt0[0] = pb0.dt0[1] = pb0.dt0[2] = 0.0;
            pb0.aw0 = 0.0;
            pb0.vn2 = 0.0;
            pb0.inv_n2 = 1.0;
            const double z3[3] = {0.0, 0.0, 0.0}, z4[4] = {1.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int jj = wl.j[k];
                CK<D> cf = fresh(c0);
                JointGoalConsts jc;
                jc.bounded = wl.bounded[k];
                jc.qmin = cf.qmin[jj];
                jc.qmax = cf.qmax[jj];
                jc.mid = cf.mid[jj];
                jc.hspan = cf.hspan[jj];
                jc.mdf = cf.mdf[jj];
                jc.seed = seed_gptr[jj];
                gk[k] += probe_joint(p, e, pb0, z3, z4, z3, z3, false, qk[k], jc, false, true);
            }
        }
    }
    e.cost = cost;
    e.sol = ok;
}

// GradientIk + step() + MemeticIk::gradientDescent's loop (GD_ELITE) for several tips, LPE >= 8 lanes per
// elite; MODE = GD_LOCAL: ik_gradient's loop with its early exits, LPE lanes per problem
template <int D, int LPE, int MODE = GD_ELITE>
__device__ __forceinline__ void gd_wide_multi(CK<D> c, PK p, const GoalSet& gs, const double (&seed)[D],
                                              const double* __restrict__ seed_gptr, GdState<D>& s, bool active,
                                              int max_iters, double* lds, const double* TC, int lane, int sub) {
    constexpr int C = LPE / 2;
    using L = WideLdsM<D, C>;
    constexpr int KP = L::KP;
    const int team = sub / C; // 0: evaluates q - g in the line search, 1: q + g
    const int r = sub % C;
    const int ebase = lane - sub;
    double* const T = lds + (lane / C) * L::STRIDE;
    const double h = p.step_size;
    const int n_tips = tip_count<D>(c);

    WideLaneM<KP> wl;
    double loc[KP], bst[KP], grd[KP];
    {
        CK<D> cl = fresh(c);
        const uint32_t bounded_mask = cl.bounded_mask;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int j = r + k * C;
            wl.valid[k] = j < D;
            const int jj = wl.valid[k] ? j : 0;
            wl.j[k] = jj;
            wl.clo[k] = cl.clo[jj];
            wl.chi[k] = cl.chi[jj];
            wl.bounded[k] = (bounded_mask >> jj) & 1u;
            const int dump = L::DUM0 + 4 * r;
            wl.sc[k] = wl.valid[k] ? L::SC0 + 6 * jj : dump;
            wl.qq[k] = wl.valid[k] ? L::QQ0 + jj : dump + 3;
            wl.gg[k] = wl.valid[k] ? L::GG0 + jj : dump;
            double v = s.local[0]; // (selects between opaque copies: see gd_wide)
#pragma unroll
            for (int m = 1; m < D; ++m) {
                double e = s.local[m];
                asm volatile("" : "+v"(e));
                v = (jj == m) ? e : v;
            }
            loc[k] = v;
            bst[k] = v;
            grd[k] = 0.0;
        }
        wl.row = r < 3 ? r : 2;
        wl.frb = r < 3 ? L::FR0 + wl.row : L::DUM0 + 4 * r;
        wl.frs = r < 3 ? 6 : 0;
        wl.fr2 = r < 3 ? 3 : 1;
        wl.rtb = r < 3 ? L::RT0 + 4 * wl.row : L::DUM0 + 4 * r;
    }
    // the problem's goals, as eval_multi derives them (make_goal), once per descent: lane r of a team takes
    // tips r, r + C, ...
    for (int tip = r; tip < n_tips; tip += C) {
        GoalK g;
        make_goal(gs.ptr + 7 * tip, g);
        T[L::GL0 + 8 * tip + 0] = g.t[0];
        T[L::GL0 + 8 * tip + 1] = g.t[1];
        T[L::GL0 + 8 * tip + 2] = g.t[2];
        T[L::GL0 + 8 * tip + 3] = g.q[0];
        T[L::GL0 + 8 * tip + 4] = g.q[1];
        T[L::GL0 + 8 * tip + 5] = g.q[2];
        T[L::GL0 + 8 * tip + 6] = g.q[3];
    }
    wave_sync();
    bool done = !active;
    bool first = true;
    int num_iterations = 0;
    double previous_cost = 0.0;
    s.steps = 0;
    s.iters = 0;
    s.found = 0;

    while (__any(!done)) {
        // ---------------- accept evaluation at `loc` (both teams, redundantly) + the probes ----------------
        EvalOut e;
        double gk[KP];
        eval_wide_multi<D, C, true>(c, p, seed, seed_gptr, loc, wl, r, T, TC, n_tips, e, gk);
        if (first) {
            first = false;
            s.local_cost = e.cost;
            s.best_cost = e.cost;
            s.best_sol = e.sol;
            if (!done) {
                if (MODE == GD_LOCAL && p.stop_on_valid && e.sol) {
                    s.found = 2; // ik_gradient early return, src/ik_gradient.cpp:102-104
                    done = true;
                } else if (max_iters <= 0) {
                    done = true;
                }
            }
        } else if (!done) {
            s.local_cost = e.cost;
            s.steps += 1;
            const bool improved = e.cost < s.best_cost;
            if (improved) {
#pragma unroll
                for (int k = 0; k < KP; ++k) bst[k] = loc[k];
                s.best_cost = e.cost;
                s.best_sol = e.sol;
            }
            if (MODE == GD_LOCAL && improved && p.stop_on_valid && e.sol) {
                s.found = 1; // src/ik_gradient.cpp:117-121
                s.iters = num_iterations + 1;
                done = true;
            } else if (fabs(e.cost - previous_cost) <= p.min_cost_delta) {
                s.iters = num_iterations;
                done = true;
            } else {
                previous_cost = e.cost;
                num_iterations += 1;
                s.iters = num_iterations;
                if (num_iterations >= max_iters) done = true;
            }
        }
        // ---------------- gradient direction ----------------
        {
#pragma unroll
            for (int k = 0; k < KP; ++k) T[wl.gg[k]] = gk[k];
            wave_sync();
            double sum = h;
#pragma unroll
            for (int j = 0; j < D; ++j) sum = sum + fabs(T[L::GG0 + j]);
            const double f = 1.0 / sum * h;
            if (!done) {
#pragma unroll
                for (int k = 0; k < KP; ++k) grd[k] = gk[k] * f;
            }
        }
        // ---------------- line search: q - g on team 0, q + g on team 1 ----------------
        double qe[KP];
        const double sg = team ? 1.0 : -1.0;
#pragma unroll
        for (int k = 0; k < KP; ++k) qe[k] = loc[k] + sg * grd[k];
        EvalOut e2;
        double unused[KP];
        eval_wide_multi<D, C, false>(c, p, seed, seed_gptr, qe, wl, r, T, TC, n_tips, e2, unused);
        const double p1 = shfl_f64(e2.cost, ebase);
        const double p3 = shfl_f64(e2.cost, ebase + C);
        const double p2 = (p1 + p3) * 0.5;
        const double cost_diff = (p3 - p1) * 0.5;
        double joint_diff = p2 / cost_diff;
        if (!isfinite(joint_diff)) joint_diff = 0.0;
        if (!done) {
#pragma unroll
            for (int k = 0; k < KP; ++k) loc[k] = clamp_lim(gd_update(loc[k], grd[k], joint_diff), wl.clo[k], wl.chi[k]);
        }
    }
    // ---- back to the replicated layout: best genes and the last normalised gradient ----
    wave_sync();
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (wl.valid[k]) {
            T[L::SC0 + wl.j[k]] = bst[k];
            T[L::SC0 + D + wl.j[k]] = grd[k];
        }
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < D; ++j) {
        s.best[j] = T[L::SC0 + j];
        s.grad[j] = T[L::SC0 + D + j];
    }
    wave_sync();
}

// ------------------------------------------------------------------------------------------
// parity-hook kernels
// ------------------------------------------------------------------------------------------
template <int D, bool MULTI = false>
__global__ __launch_bounds__(256) void fk_kernel(const ConstsK<D>* __restrict__ kc, long long n,
                                                 const double* __restrict__ q,
                                                 double* __restrict__ pos_quat) {
    PIK_CONSTS(kc);
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double qq[D];
#pragma unroll
    for (int j = 0; j < D; ++j) qq[j] = q[i * D + j];
    if constexpr (MULTI) {
        // pos_quat [n][n_tips][7]
        const int n_tips = tip_count<D>(c);
#pragma unroll 1
        for (int k = 0; k < n_tips; ++k) {
            double R[9], t[3], qt[4];
            fk<D, false, true>(tip_chain<D>(c, k), qq, R, t, nullptr, 0);
            matrix_to_quat(R, qt);
            double* o = pos_quat + 7 * (i * n_tips + k);
            o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
            o[3] = qt[0]; o[4] = qt[1]; o[5] = qt[2]; o[6] = qt[3];
        }
    } else {
        double R[9], t[3], qt[4];
        fk<D, false>(c, qq, R, t, nullptr, 0);
        matrix_to_quat(R, qt);
        double* o = pos_quat + 7 * i;
        o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
        o[3] = qt[0]; o[4] = qt[1]; o[5] = qt[2]; o[6] = qt[3];
    }
}

template <int D, bool MULTI = false>
__global__ __launch_bounds__(WAVE) void cost_kernel(const ConstsK<D>* __restrict__ kc, long long n,
                                                    const double* __restrict__ goal,
                                                    const double* __restrict__ seed,
                                                    const double* __restrict__ q,
                                                    double* __restrict__ cost,
                                                    int* __restrict__ is_solution) {
    PIK_CONSTS(kc);
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typename GoalSel<MULTI>::type g;
    load_goals<D>(c, goal, i, g);
    double qq[D], sd[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        qq[j] = q[i * D + j];
        sd[j] = seed[i * D + j];
    }
    EvalOut e;
    evaluate<D>(c, p, g, sd, qq, e);
    if (cost) cost[i] = e.cost;
    if (is_solution) is_solution[i] = e.sol ? 1 : 0;
}

template <int D, bool MULTI = false>
__global__ __launch_bounds__(WAVE) void gd_step_kernel(
    const ConstsK<D>* __restrict__ kc, long long n, const double* __restrict__ goal,
    const double* __restrict__ seed, double* __restrict__ local, double* __restrict__ best,
    double* __restrict__ local_cost, double* __restrict__ best_cost, double* __restrict__ gradient,
    int* __restrict__ improved) {
    PIK_CONSTS(kc);
    __shared__ double frames[GD_ROWS(D) * WAVE];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n;
    const long long ii = active ? i : 0;
    typename GoalSel<MULTI>::type g;
    load_goals<D>(c, goal, ii, g);
    double sd[D];
    GdState<D> s;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        sd[j] = seed[ii * D + j];
        s.local[j] = local[ii * D + j];
        s.best[j] = best[ii * D + j];
    }
    s.local_cost = local_cost[ii];
    s.best_cost = best_cost[ii];
    s.best_sol = false;
    const double bc_in = s.best_cost;
    PIK_DESCENT(GD_SINGLE, 1, g, sd, nullptr, s, active, 1, frames, (int)threadIdx.x, 0);
    if (!active) return;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        local[i * D + j] = s.local[j];
        best[i * D + j] = s.best[j];
        gradient[i * D + j] = s.grad[j];
    }
    local_cost[i] = s.local_cost;
    best_cost[i] = s.best_cost;
    if (improved) improved[i] = (s.best_cost < bc_in) ? 1 : 0;
}

// "local" mode: ik_gradient -- src/ik_gradient.cpp:96-139, one lane per problem
template <int D, bool MULTI = false>
__global__ __launch_bounds__(WAVE) void ik_gradient_kernel(const ConstsK<D>* __restrict__ kc,
                                                           SolveArgs a) {
    PIK_CONSTS(kc);
    __shared__ double frames[GD_ROWS(D) * WAVE];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < a.B;
    const BatchK* const bk = find_batch(a, active ? i : 0);
    const long long ii = active ? i - bk->start : 0; // batch-local index
    typename GoalSel<MULTI>::type g;
    load_goals<D>(c, bk->goal, ii, g);
    double sd[D], guess[D];
    GdState<D> s;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        sd[j] = bk->seed[ii * D + j];
        guess[j] = bk->guess[ii * D + j];
        s.local[j] = guess[j];
        s.best[j] = guess[j];
    }
    s.local_cost = 0.0;
    s.best_cost = 0.0;
    s.best_sol = false;
    PIK_DESCENT(GD_LOCAL, 1, g, sd, nullptr, s, active, p.local_max_iters, frames, (int)threadIdx.x, 0);
    // post-loop -- src/ik_gradient.cpp:130-138
    int status = PIKAMD_NO_IK_SOLUTION_K;
    if (s.found) {
        status = 1;
    } else if (!p.stop_on_valid && s.best_sol) {
        status = 1;
    } else if (p.approx) {
        status = 2;
    }
    double first_cost = 0.0; // cost of the initial guess, reported on failure
    if (__any(active && status < 0)) {
        EvalOut e;
        evaluate<D>(c, p, g, sd, guess, e);
        first_cost = e.cost;
    }
    if (active) {
#pragma unroll
        for (int j = 0; j < D; ++j) bk->solution[ii * D + j] = (status > 0) ? s.best[j] : sd[j];
        bk->status[ii] = status;
        if (bk->cost) bk->cost[ii] = (status > 0) ? s.best_cost : first_cost;
        if (bk->stats) {
            StatsK st;
            st.cost_evals = (s.found == 2) ? 0 : 1 + (long long)s.steps * (2 * D + 3);
            st.generations = s.iters;
            st.wipeouts = 0;
            st.pool_erasures = 0;
            st.reserved = 0;
            bk->stats[ii] = st;
        }
    }
    if (a.signal && active) signal_completed(bk);
}

#if !defined(PIK_STRICT)
// "local" mode with LPE lanes per problem (the cooperative descent, gd_wide): what a caller with a
// handful of targets wants -- ik_gradient is 100 sequential steps of three evaluations each, and one
// lane runs them in ~1.2 ms; sixteen lanes in a third of that.  Bit-identical to the one-lane kernel.
template <int D, int LPE, bool MULTI = false>
__global__ __launch_bounds__(WAVE) void ik_gradient_wide_kernel(const ConstsK<D>* __restrict__ kc,
                                                                SolveArgs a) {
    PIK_CONSTS(kc);
    constexpr int GDR = GD_ROWS(D, LPE, !MULTI);
    constexpr int TCR = MULTI ? (MAX_TIPS * (7 * D + 24) + WAVE - 1) / WAVE : 0; // (several tips: the chains' constants)
    __shared__ double lds[(GDR + TCR) * WAVE];
    constexpr int PER_WAVE = WAVE / LPE;
    const int lane = threadIdx.x;
    const int sub = lane % LPE;
    const long long i = (long long)blockIdx.x * PER_WAVE + lane / LPE;
    const bool active = i < a.B;
    const BatchK* const bk = find_batch(a, active ? i : 0);
    const long long ii = active ? i - bk->start : 0; // batch-local index
    typename GoalSel<MULTI>::type g;
    load_goals<D>(c, bk->goal, ii, g);
    double sd[D], guess[D];
    GdState<D> s;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        sd[j] = bk->seed[ii * D + j];
        guess[j] = bk->guess[ii * D + j];
        s.local[j] = guess[j];
        s.best[j] = guess[j];
        s.grad[j] = 0.0;
    }
    s.local_cost = 0.0;
    s.best_cost = 0.0;
    s.best_sol = false;
    if constexpr (MULTI) {
        stage_tip_constants<D>(c, lds + GDR * WAVE, lane);
        gd_wide_multi<D, LPE, GD_LOCAL>(c, p, g, sd, bk->seed + ii * D, s, active, p.local_max_iters, lds,
                                        lds + GDR * WAVE, lane, sub);
    } else {
        gd_wide<D, LPE, GD_LOCAL>(c, p, g, sd, bk->seed + ii * D, s, active, p.local_max_iters, lds, lane, sub);
    }
    // post-loop -- src/ik_gradient.cpp:130-138
    int status = PIKAMD_NO_IK_SOLUTION_K;
    if (s.found) {
        status = 1;
    } else if (!p.stop_on_valid && s.best_sol) {
        status = 1;
    } else if (p.approx) {
        status = 2;
    }
    double first_cost = 0.0; // cost of the initial guess, reported on failure
    if (__any(active && status < 0)) {
        EvalOut e;
        evaluate<D>(c, p, g, sd, guess, e);
        first_cost = e.cost;
    }
    if (active && sub == 0) {
#pragma unroll
        for (int j = 0; j < D; ++j) bk->solution[ii * D + j] = (status > 0) ? s.best[j] : sd[j];
        bk->status[ii] = status;
        if (bk->cost) bk->cost[ii] = (status > 0) ? s.best_cost : first_cost;
        if (bk->stats) {
            StatsK st;
            st.cost_evals = (s.found == 2) ? 0 : 1 + (long long)s.steps * (2 * D + 3);
            st.generations = s.iters;
            st.wipeouts = 0;
            st.pool_erasures = 0;
            st.reserved = 0;
            bk->stats[ii] = st;
        }
    }
    if (a.signal && active && sub == 0) signal_completed(bk); // (the lane that stored the results)
}
#endif

#if defined(PIK_STRICT)
// "local" mode of the EXACT flavours with LPE lanes per problem: the team forms of the memoised descent
// (pik_exact.hpp: the lanes share the accept evaluation, the probes are one flat pass, the two line-search points go
// to two teams) serve ik_gradient as they serve an elite.  One lane runs a local-mode query's ~27 steps of 2D + 3
// evaluations in 0.61 ms (Panda) -- slower than one CPU core (0.15 ms), and the MoveIt plugin's local mode is one such
// query per call; sixteen lanes take a quarter of that.  Bit-identical to the one-lane kernel (every lane of a problem
// holds the same state; the results are stored by the problem's first lane).  One tip frame.
template <int D, int LPE>
__global__ __launch_bounds__(WAVE) void ik_gradient_team_kernel(const ConstsK<D>* __restrict__ kc, SolveArgs a) {
    PIK_CONSTS(kc);
    __shared__ double lds[GD_ROWS(D, LPE) * WAVE];
    constexpr int PER_WAVE = WAVE / LPE;
    const int lane = threadIdx.x;
    const int sub = lane % LPE;
    const long long i = (long long)blockIdx.x * PER_WAVE + lane / LPE;
    const bool active = i < a.B;
    const BatchK* const bk = find_batch(a, active ? i : 0);
    const long long ii = active ? i - bk->start : 0; // batch-local index
    GoalK g;
    load_goals<D>(c, bk->goal, ii, g);
    double sd[D], guess[D];
    GdState<D> s;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        sd[j] = bk->seed[ii * D + j];
        guess[j] = bk->guess[ii * D + j];
        s.local[j] = guess[j];
        s.best[j] = guess[j];
        s.grad[j] = 0.0;
    }
    s.local_cost = 0.0;
    s.best_cost = 0.0;
    s.best_sol = false;
    PIK_DESCENT(GD_LOCAL, LPE, g, sd, nullptr, s, active, p.local_max_iters, lds, lane, sub);
    // post-loop -- src/ik_gradient.cpp:130-138
    int status = PIKAMD_NO_IK_SOLUTION_K;
    if (s.found) {
        status = 1;
    } else if (!p.stop_on_valid && s.best_sol) {
        status = 1;
    } else if (p.approx) {
        status = 2;
    }
    double first_cost = 0.0; // cost of the initial guess, reported on failure
    if (__any(active && status < 0)) {
        EvalOut e;
        evaluate<D>(c, p, g, sd, guess, e);
        first_cost = e.cost;
    }
    if (active && sub == 0) {
#pragma unroll
        for (int j = 0; j < D; ++j) bk->solution[ii * D + j] = (status > 0) ? s.best[j] : sd[j];
        bk->status[ii] = status;
        if (bk->cost) bk->cost[ii] = (status > 0) ? s.best_cost : first_cost;
        if (bk->stats) {
            StatsK st;
            st.cost_evals = (s.found == 2) ? 0 : 1 + (long long)s.steps * (2 * D + 3);
            st.generations = s.iters;
            st.wipeouts = 0;
            st.pool_erasures = 0;
            st.reserved = 0;
            bk->stats[ii] = st;
        }
    }
    if (a.signal && active && sub == 0) signal_completed(bk); // (the lane that stored the results)
}
#endif

// ------------------------------------------------------------------------------------------
// "global" mode: ik_memetic -- src/ik_memetic.cpp
// ------------------------------------------------------------------------------------------


// (fitness, slot) lexicographic "less": the deterministic order used instead of std::sort's
// unspecified tie order (src/ik_memetic.cpp:200-203)
__device__ __forceinline__ bool key_less(double fa, int ia, double fb, int ib) {
    return (fa < fb) || (!(fb < fa) && ia < ib);
}

// index of the k-th (0-based) set bit of m
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int k) {
    for (int i = 0; i < k; ++i) m &= m - 1;
    return __ffsll((long long)m) - 1;
}

// LDS layout per wavefront: rows of 64 doubles, row r of lane l at [r * 64 + l] (conflict-free
// for lane-contiguous access, broadcast when lanes of a group read the same parent).
//   rows 0 .. 2D+1       parent table: genes[D], gradient[D], fitness, extinction of lane's elite
//   rows 2D+2 .. 4D+1    kept table  : genes[D], gradient[D] of the candidate held in lane's slot
//   row  4D+2            rank -> lane inverse permutation (ints)
//   rows 0 .. 8D-1       scratch rows of gradient_descent() (GD_ROWS) -- ALIASED with the
//                        tables above: they live only inside gradient_descent(), the tables only
//                        between the end of it and the end of the generation
// With LPE lanes per elite a problem's group has G = GS * LPE lanes; elite e occupies the LPE
// adjacent lanes [e * LPE, (e + 1) * LPE) of the group, all holding the same elite state.
// D = 7, one lane per elite, one tip: 36 rows = 18 KB, so that EIGHT wavefronts (two per SIMD) fit the
// 160 KB of a CU; with the first joint's frame stored (42 rows, 21 KB) only seven did.
template <int D, int LPE = 2, bool MULTI = false>
struct MemeticLds {
    static constexpr int PAR_ROWS = 2 * D + 2;
    static constexpr int KEPT_ROWS = 2 * D;
    static constexpr int INV_ROW = PAR_ROWS + KEPT_ROWS;
    static constexpr int GD = GD_ROWS(D, LPE, !MULTI);
    // several tips at 8 / 16 lanes per elite: the constants of the tips' chains, copied once per kernel, behind
    // everything that is aliased (stage_tip_constants)
    static constexpr int TC_ROW = (INV_ROW + 1 > GD) ? INV_ROW + 1 : GD;
    static constexpr int TC_ROWS = (MULTI && LPE >= 8) ? (MAX_TIPS * (7 * D + 24) + WAVE - 1) / WAVE : 0;
    static constexpr int ROWS = TC_ROW + TC_ROWS;
};

// OCC = wavefronts per SIMD the kernel is compiled for: 1 -> 512 registers per lane (no scratch, the
// fastest single wavefront: what a batch that cannot fill the chip wants), 2 -> 256 registers (cold
// state spilled to scratch at phase boundaries) so that a batch with more wavefronts than SIMDs
// keeps two per SIMD and hides FP64 / scalar-load latency behind each other (+26 % at B = 65 536).
template <int D, int LPE, bool MULTI = false, int OCC = 1>
__global__ __launch_bounds__(WAVE, OCC) void memetic_kernel(const ConstsK<D>* __restrict__ kc,
                                                       SolveArgs a) {
    static_assert(LPE == 1 || LPE == 2 || LPE == 4 || LPE == 8 || LPE == 16, "LPE must be a power of two");
    PIK_CONSTS(kc);
    __shared__ double lds[MemeticLds<D, LPE, MULTI>::ROWS * WAVE];
    double* const par = lds;                                   // [PAR_ROWS][64]
    double* const kept = lds + MemeticLds<D, LPE>::PAR_ROWS * WAVE; // [KEPT_ROWS][64]
    int* const inv = reinterpret_cast<int*>(lds + MemeticLds<D, LPE, MULTI>::INV_ROW * WAVE);

    PIK_TIMING_DECL();
    const int lane = threadIdx.x;
#if !defined(PIK_STRICT)
    if constexpr (MULTI && LPE >= 8) stage_tip_constants<D>(c, lds + MemeticLds<D, LPE, MULTI>::TC_ROW * WAVE, lane);
#endif
    const int GS = (PIK_COMMON ? 4 : (1 << a.gs_log2)) * LPE; // lanes per problem (common configuration: four elites)
    const int lid = lane & (GS - 1);
    const int gbase = lane - lid;
    const int sub = lid & (LPE - 1); // sub-lane within the elite
    const int el = lid / LPE;        // elite index owned by this lane
    const unsigned long long gmask_all = (GS == 64) ? ~0ull : ((1ull << GS) - 1ull);
    const int E = PIK_COMMON ? 4 : p.elites;
    const int P = p.population;
    const bool elite_lane = el < E;
    const bool lead_lane = elite_lane && sub == 0; // one representative lane per elite
    const double inv_gene = 1.0 / (double)D;
    const double INF = __builtin_inf();
    // species: SP = pow2ceil(S) groups ("super-group") per problem
    const int S = PIK_COMMON ? 1 : a.species;
    const int SP = PIK_COMMON ? 1 : (1 << a.sp_log2);
    const int SGS = GS * SP;
    const int sp = (lane / GS) & (SP - 1);
    const int sbase = lane - (lane & (SGS - 1));
    const bool sp_ok = sp < S;
    const unsigned sp_key = (unsigned)sp << 20; // species folded into the RNG individual index

    // ---- per-group problem state (replicated in every lane of the group) ----
    bool pend = false;   // super-group: a problem is in progress (result not yet written)
    bool sp_has = false; // this species returned a value (std::optional has_value)
    bool sp_val = false; // ... and it passed solution_fn
    bool act = false;    // this species is still running generations
    bool exhausted = false; // the work queue had no more problems for this group
    bool need_init = false; // (re)build the population from `best` at the top of the loop
    bool pop_guess = true;  // stored population: slots >= E still hold copies of the guess
    long long prob = 0;     // virtual problem index of the call (all batches, see BatchK)
    unsigned long long gprob = 0; // random-stream key: the batch's problem_offset + batch-local index
    const double* seed_ptr = a.batches[0].seed; // this problem's ik_seed_state in HBM
    typename GoalSel<MULTI>::type goal; // one tip frame, or several (MULTI)
    double seed[D];
    double best[D];
    double best_fit = 0.0;
    bool best_sol = false;
    double seed_cost = 0.0;
    double prev_fit = 0.0;
    bool has_prev = false;
    int gen = 0;
    unsigned init_epoch = 0;
    int wipeouts = 0, erasures = 0;
    unsigned gd_steps = 0; // total step() calls of all elites of this problem
    // (gradientDescent() calls need no counter: E per generation, i.e. E * gen)
    // ---- this lane's elite ----
    double eg[D], egrad[D];
    double efit = 0.0, eext = 0.0;
    bool esol = false;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        seed[j] = 0.0;
        best[j] = 0.0;
        eg[j] = 0.0;
        egrad[j] = 0.0;
    }
    goal_reset(goal, a.batches[0].goal);

    // conclude(): this species stops; has/valid mirror ik_memetic_impl's std::optional result
    auto conclude = [&](bool has, bool valid) {
        sp_has = has;
        sp_val = valid;
        act = false;
    };

    // resolve(): when every species of a problem has stopped, pick the minimum-fitness value
    // (ties: lowest species, as the reference's queue drain with `cost < min_cost` does,
    // src/ik_memetic.cpp:356-370) and write the problem's result.
    auto resolve = [&]() {
        bool any_run = false;
        for (int k = 0; k < SP; ++k) any_run = any_run || (shfl_i32(act ? 1 : 0, sbase + k * GS) != 0);
        if (pend && !any_run) {
            // literal cost_fn invocation count of the reference for this species' trajectory:
            //   MemeticIk::from 1, initPopulation E + P (once + per wipeout),
            //   per gradientDescent 2 + steps * (2D + 3), per generation P - E children
            const long long my_evals = (init_epoch > 0 ? 1 : 0) + (long long)init_epoch * (E + P) +
                                       2ll * E * gen + (long long)gd_steps * (2 * D + 3) + (long long)gen * (P - E);
            int win = -1;
            double win_fit = INF;
            bool win_val = false;
            long long evals = 0;
            int gmax = 0;
            for (int k = 0; k < SP; ++k) {
                const int src = sbase + k * GS;
                const bool has_k = shfl_i32(sp_has ? 1 : 0, src) != 0;
                const bool val_k = shfl_i32(sp_val ? 1 : 0, src) != 0;
                const double fit_k = shfl_f64(best_fit, src);
                const int gen_k = shfl_i32(gen, src);
                const long long ev_k =
                    (long long)(((unsigned long long)(unsigned)shfl_i32((int)(my_evals >> 32), src) << 32) |
                                (unsigned)shfl_i32((int)(my_evals & 0xffffffffll), src));
                if (k < S) {
                    evals += ev_k;
                    gmax = gen_k > gmax ? gen_k : gmax;
                    if (has_k && fit_k < win_fit) {
                        win = k;
                        win_fit = fit_k;
                        win_val = val_k;
                    }
                }
            }
            // ONE lane stores everything of the problem: the winning species' lead lane (its genes),
            // or the problem's first lane when no species returned a value
            const int w0 = shfl_i32(wipeouts, sbase), e0 = shfl_i32(erasures, sbase);
            if (lane == sbase + (win >= 0 ? win : 0) * GS) {
                const BatchK* const bk = find_batch(a, prob);
                const long long lp = prob - bk->start;
                if (win >= 0) {
#pragma unroll
                    for (int j = 0; j < D; ++j) bk->solution[lp * D + j] = best[j];
                    bk->status[lp] = win_val ? 1 : 2;
                    if (bk->cost) bk->cost[lp] = best_fit;
                } else {
                    // solution = ik_seed_state on failure -- src/pick_ik_plugin.cpp:213-217
#pragma unroll
                    for (int j = 0; j < D; ++j) bk->solution[lp * D + j] = seed[j];
                    bk->status[lp] = PIKAMD_NO_IK_SOLUTION_K;
                    if (bk->cost) bk->cost[lp] = seed_cost; // cost of the initial guess
                }
                if (bk->stats) {
                    StatsK st;
                    st.cost_evals = evals;
                    st.generations = gmax;
                    st.wipeouts = w0;
                    st.pool_erasures = e0;
                    st.reserved = 0;
                    bk->stats[lp] = st;
                }
                if (a.signal) signal_completed(bk);
            }
            pend = false;
        }
    };

    const long long n_items = a.list_in ? (long long)(*a.n_in) : a.B;
    if (a.list_in && !((unsigned)n_items > a.sel_lo && (unsigned)n_items <= a.sel_hi)) return; // not this variant's pass

    // park(): a running problem reached this pass's generation mark.  One record per (problem, species):
    // with several species the whole problem is parked -- the species that have already stopped as well,
    // with their verdicts, because resolve() needs every species' outcome.
    auto park = [&]() {
        using SR = StateRows<D>;
        const long long DR = SR::D_ROWS(E); // one contiguous record per (problem, species)
        const long long rec = prob * S + (sp_ok ? sp : 0);
        if (sp_ok && lead_lane) {
            const int er = el * SR::ELITE;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                a.st_d[rec * DR + (er + j)] = eg[j];
                a.st_d[rec * DR + (er + D + j)] = egrad[j];
            }
            a.st_d[rec * DR + (er + 2 * D)] = efit;
            a.st_d[rec * DR + (er + 2 * D + 1)] = eext;
            a.st_d[rec * DR + (er + 2 * D + 2)] = esol ? 1.0 : 0.0;
        }
        if (sp_ok && lid == 0) {
#pragma unroll
            for (int j = 0; j < D; ++j) a.st_d[rec * DR + (SR::BEST0(E) + j)] = best[j];
            a.st_d[rec * DR + (SR::SCAL0(E) + 0)] = best_fit;
            a.st_d[rec * DR + (SR::SCAL0(E) + 1)] = best_sol ? 1.0 : 0.0;
            a.st_d[rec * DR + (SR::SCAL0(E) + 2)] = seed_cost;
            a.st_d[rec * DR + (SR::SCAL0(E) + 3)] = prev_fit;
            a.st_i[rec * SR::I_ROWS + 0] = gen;
            a.st_i[rec * SR::I_ROWS + 1] = (int)init_epoch;
            a.st_i[rec * SR::I_ROWS + 2] = wipeouts;
            a.st_i[rec * SR::I_ROWS + 3] = erasures;
            a.st_i[rec * SR::I_ROWS + 4] = has_prev ? 1 : 0;
            a.st_i[rec * SR::I_ROWS + 5] = need_init ? 1 : 0;
            a.st_i[rec * SR::I_ROWS + 6] = pop_guess ? 1 : 0;
            a.st_i[rec * SR::I_ROWS + 7] = act ? 1 : 0;
            a.st_i[rec * SR::I_ROWS + 8] = sp_has ? 1 : 0;
            a.st_i[rec * SR::I_ROWS + 9] = sp_val ? 1 : 0;
            a.st_l[rec * SR::L_ROWS + 0] = (long long)gd_steps;
        }
        if (lane == sbase) { // once per problem
            const unsigned slot = atomicAdd(a.n_out, 1u);
            a.list_out[slot] = (int)prob;
        }
        act = false;
        pend = false;
    };

    for (;;) {
        // ------------------------------------------------------------------ refill
        bool fresh_problem = false;
        if (!pend && !exhausted) {
            unsigned long long idx = 0;
            if (lane == sbase) idx = atomicAdd(a.work_counter, 1ull);
            idx = ((unsigned long long)(unsigned)shfl_i32((int)(idx & 0xffffffffu), sbase)) |
                  ((unsigned long long)(unsigned)shfl_i32((int)(idx >> 32), sbase) << 32);
            if ((long long)idx < n_items) {
                prob = a.list_in ? (long long)a.list_in[idx] : (long long)idx;
                const BatchK* const bk = find_batch(a, prob);
                const long long lp = prob - bk->start;
                gprob = (unsigned long long)(bk->problem_offset + lp);
                load_goals<D>(c, bk->goal, lp, goal);
                seed_ptr = bk->seed + lp * D;
#pragma unroll
                for (int j = 0; j < D; ++j) seed[j] = seed_ptr[j];
                pend = true;
                act = sp_ok;
                sp_has = false;
                sp_val = false;
                if (a.fresh) {
                    const double* const gp = bk->guess + lp * D;
#pragma unroll
                    for (int j = 0; j < D; ++j) best[j] = gp[j]; // MemeticIk::from: best_ = guess
                    gen = 0;
                    init_epoch = 0;
                    wipeouts = 0;
                    erasures = 0;
                    gd_steps = 0;
                    need_init = true;
                    fresh_problem = true;
                } else {
                    // resume a parked problem (this species' record)
                    using SR = StateRows<D>;
                    const long long DR = SR::D_ROWS(E);
                    const long long rec = prob * S + (sp_ok ? sp : 0);
                    const int er = (elite_lane ? el : 0) * SR::ELITE;
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        eg[j] = a.st_d[rec * DR + (er + j)];
                        egrad[j] = a.st_d[rec * DR + (er + D + j)];
                        best[j] = a.st_d[rec * DR + (SR::BEST0(E) + j)];
                    }
                    efit = a.st_d[rec * DR + (er + 2 * D)];
                    eext = a.st_d[rec * DR + (er + 2 * D + 1)];
                    esol = a.st_d[rec * DR + (er + 2 * D + 2)] != 0.0;
                    best_fit = a.st_d[rec * DR + (SR::SCAL0(E) + 0)];
                    best_sol = a.st_d[rec * DR + (SR::SCAL0(E) + 1)] != 0.0;
                    seed_cost = a.st_d[rec * DR + (SR::SCAL0(E) + 2)];
                    prev_fit = a.st_d[rec * DR + (SR::SCAL0(E) + 3)];
                    gen = a.st_i[rec * SR::I_ROWS + 0];
                    init_epoch = (unsigned)a.st_i[rec * SR::I_ROWS + 1];
                    wipeouts = a.st_i[rec * SR::I_ROWS + 2];
                    erasures = a.st_i[rec * SR::I_ROWS + 3];
                    has_prev = a.st_i[rec * SR::I_ROWS + 4] != 0;
                    need_init = a.st_i[rec * SR::I_ROWS + 5] != 0;
                    pop_guess = a.st_i[rec * SR::I_ROWS + 6] != 0;
                    act = sp_ok && a.st_i[rec * SR::I_ROWS + 7] != 0;
                    sp_has = sp_ok && a.st_i[rec * SR::I_ROWS + 8] != 0;
                    sp_val = sp_ok && a.st_i[rec * SR::I_ROWS + 9] != 0;
                    gd_steps = (unsigned)a.st_l[rec * SR::L_ROWS + 0];
                }
            } else {
                exhausted = true;
            }
        }
        if (!__any(pend)) {
            if (__all(exhausted)) break;
            continue;
        }

        // ------------------------------------------------------------------ initPopulation
        // src/ik_memetic.cpp:93-117: lane e builds elite e from `best` (elite 0 = the guess
        // itself, the others uniform random valid configurations); the P - E non-elite copies of
        // the guess all carry the guess's cost and are overwritten by the first reproduce().
        // One code path serves a fresh problem (guess = seed; elite 0's evaluation is also
        // MemeticIk::from's and the early-accept test of ik_memetic, :294-296) and a wipeout.
        if (__any(act && need_init)) {
            const bool doing = act && need_init;
            const unsigned epoch = init_epoch;
            double cand[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double v = best[j];
                if (el > 0) {
                    // Robot::set_random_valid_configuration -- src/robot.cpp:87-95, 23-30
                    const U4 w = rng_block(a.rng_seed, STREAM_INIT,
                                           gprob, epoch,
                                           (unsigned)el | sp_key, (unsigned)(j >> 1));
                    const double u = (j & 1) ? u01_from_words(w.z, w.w) : u01_from_words(w.x, w.y);
                    const bool bounded = PIK_COMMON ? true : (((c.bounded_mask >> j) & 1u) != 0);
                    v = bounded ? uniform_real(c.qmin[j], c.qmax[j], u)
                                : uniform_real(v - M_PI, v + M_PI, u);
                }
                cand[j] = v;
            }
            EvalOut e;
            evaluate<D, OCC>(c, p, goal, seed, cand, e);
            if (doing && elite_lane) {
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    eg[j] = cand[j];
                    egrad[j] = 0.0;
                }
                efit = e.cost;
                esol = e.sol;
            }
            // computeExtinctions on the unsorted population: front = elite 0, back = a copy of
            // the guess (same genes as elite 0) -- src/ik_memetic.cpp:57-64, 115
            const double f0 = shfl_f64(efit, gbase);
            const bool s0 = shfl_i32(esol ? 1 : 0, gbase) != 0;
            if (doing) {
                eext = (efit + f0 * ((double)el / (double)(P - 1) - 1.0)) / f0;
                has_prev = false;
                init_epoch = epoch + 1;
                need_init = false;
                pop_guess = true; // initPopulation: every non-elite slot is a copy of the guess
                if (fresh_problem) {
                    seed_cost = f0;
                    best_fit = f0;
                    best_sol = s0;
                    if (p.stop_on_valid && s0) {
                        init_epoch = 0; // the reference returns before constructing anything
                        conclude(true, true); // best == the initial guess here
                    } else if (gen >= p.max_generations) {
                        // loop never runs: post-loop of ik_memetic_impl, src/ik_memetic.cpp:272-282
                        if (!p.stop_on_valid && best_sol) {
                            conclude(true, true);
                        } else if (p.approx) {
                            conclude(true, false);
                        } else {
                            conclude(false, false);
                        }
                    }
                }
            }
        }
        resolve(); // problems decided without running a generation
        if (!__any(act)) continue;

        // ------------------------------------------------------------------ one generation
        PIK_T0();
        // (1) gradient descent on the elites -- src/ik_memetic.cpp:230-239, 66-91
        {
            GdState<D> s;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                s.local[j] = eg[j];
                s.best[j] = eg[j];
            }
            s.local_cost = efit;
            s.best_cost = efit;
            s.best_sol = esol;
            const bool gd_active = act && elite_lane;
#if !defined(PIK_STRICT)
            if constexpr (LPE >= 8 && !MULTI) {
                gd_wide<D, LPE>(c, p, goal, seed, seed_ptr, s, gd_active, p.gd_max_iters, lds, lane, sub);
            } else if constexpr (LPE >= 8 && MULTI) {
                gd_wide_multi<D, LPE>(c, p, goal, seed, seed_ptr, s, gd_active, p.gd_max_iters, lds,
                                      lds + MemeticLds<D, LPE, MULTI>::TC_ROW * WAVE, lane, sub);
            } else
#endif
            {
                // (strict build, LPE > 1: the literal 2D + 3 evaluations of a step dealt out to the
                //  elite's lanes -- 2 + ceil(2D / LPE) evaluations deep instead of 2D + 3)
                PIK_DESCENT_OCC(GD_ELITE, LPE, OCC, goal, seed, seed_ptr, s, gd_active, p.gd_max_iters, lds, lane, sub);
            }
            if (gd_active) {
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    eg[j] = s.best[j];
                    egrad[j] = s.grad[j];
                }
                efit = s.best_cost; // cost_fn(individual.genes): same value as best_cost
                esol = s.best_sol;
            }
            int st = (gd_active && sub == 0) ? s.steps : 0;
            for (int off = 1; off < GS; off <<= 1) st += shfl_i32(st, lane ^ off);
            if (act) {
                gd_steps += (unsigned)st;
            }
        }

        PIK_TICK(0); // gradient descent
        // publish parents + seed the kept set with the elites themselves
        wave_sync();
#pragma unroll
        for (int j = 0; j < D; ++j) {
            par[j * WAVE + lane] = eg[j];
            par[(D + j) * WAVE + lane] = egrad[j];
            kept[j * WAVE + lane] = eg[j];
            kept[(D + j) * WAVE + lane] = egrad[j];
        }
        par[(2 * D) * WAVE + lane] = efit;
        par[(2 * D + 1) * WAVE + lane] = eext;
        wave_sync();

        // stored population (chains with unbounded variables only): this generation's buffer and
        // the previous generation's (read by the empty-pool branch)
        double* const pop_cur = PIK_POP(a) ? PIK_POP(a) + (((prob * S + sp) * 2 + (gen & 1)) * a.pop_stride) : nullptr;
        const double* const pop_prev = PIK_POP(a) ? PIK_POP(a) + (((prob * S + sp) * 2 + ((gen + 1) & 1)) * a.pop_stride) : nullptr;
        if (PIK_POP(a) && act && lead_lane) {
            pop_cur[el] = efit;
#pragma unroll
            for (int j = 0; j < D; ++j) pop_cur[P + el * D + j] = eg[j];
        }

        double kfit = (act && lead_lane) ? efit : INF; // key of the candidate in this lane's slot
        int kidx = lead_lane ? el : (0x40000000 + lid); // unique keys => ranks are a permutation
        bool ksol = esol;
        double maxfit = (act && lead_lane) ? efit : -INF;
        // group-uniform worst kept key.  Only the E lead lanes are slots of the kept set: the next
        // generation needs the top E and nothing else, and with all GS * LPE lanes as slots every
        // one of the first GS * LPE - E children of a generation "qualified" and was inserted
        // serially (half of the reproduction time at 4 lanes per elite).
        double wfit;
        int widx, wlane;
        auto recompute_worst = [&]() {
            if constexpr (LPE >= 8) {
                // the cooperative variants: the E lead slots gathered one by one (E shuffles) instead of a
                // butterfly over all GS lanes (log2 GS rounds of four shuffles) -- the same maximum of the
                // same unique keys.  (Not at four lanes per elite: memetic_kernel<8, 4>, at the register cap
                // like every narrow multi-lane kernel of a long chain, came out faulting with this form.)
                double f = -INF;
                int ix = -1, ln = gbase;
                for (int m = 0; m < E; ++m) {
                    const double f2 = shfl_f64(kfit, gbase + m * LPE);
                    const int ix2 = shfl_i32(kidx, gbase + m * LPE);
                    if (key_less(f, ix, f2, ix2)) {
                        f = f2;
                        ix = ix2;
                        ln = gbase + m * LPE;
                    }
                }
                wfit = f;
                widx = ix;
                wlane = ln;
                return;
            }
            double f = lead_lane ? kfit : -INF;
            int ix = lead_lane ? kidx : -1, ln = lane;
            for (int off = 1; off < GS; off <<= 1) {
                const double f2 = shfl_f64(f, lane ^ off);
                const int ix2 = shfl_i32(ix, lane ^ off);
                const int ln2 = shfl_i32(ln, lane ^ off);
                if (key_less(f, ix, f2, ix2)) {
                    f = f2;
                    ix = ix2;
                    ln = ln2;
                }
            }
            wfit = f;
            widx = ix;
            wlane = ln;
        };
        recompute_worst();

        PIK_TICK(1); // publish + worst
        // (2) reproduce -- src/ik_memetic.cpp:119-190
        unsigned long long pool = (E == 64) ? ~0ull : ((1ull << E) - 1ull);
        int next = E;
        for (;;) {
            const int i = next + lid;
            const bool valid = act && i < P;
            if (!__any(valid)) break;

            double cg[D], cgrad[D];
            double cfit = INF;
            bool csol = false;
            unsigned long long erase_bits = 0;
            double pfitA = 0.0, pfitB = 0.0;
            int pia = -1, pib = -1;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                cg[j] = 0.0;
                cgrad[j] = 0.0;
            }
            PIK_TICK(2); // round head
            if (valid) {
                const int pool_n = __popcll(pool);
                // A lane whose mating pool has run empty makes a fresh random member instead of a
                // child (src/ik_memetic.cpp:181-188).  Both kinds use the same per-gene random block,
                // so they share ONE code path with per-lane selects: as separate branches the wave ran
                // both whenever any of its groups had an empty pool (2.3 erasures per generation).
                const bool have_pool = pool_n > 0;
                const U4 w0 = rng_block(a.rng_seed, STREAM_REPRODUCE, gprob, (unsigned)gen, (unsigned)i | sp_key, 0u);
                const int ka = (int)(((unsigned long long)w0.x * (unsigned)(have_pool ? pool_n : 1)) >> 32);
                const double mix = u01_from_words(w0.z, w0.w);
                // idxB: uniform over the OTHER pool members -- the distribution of the
                // reference's "draw until != idxA" loop (src/ik_memetic.cpp:132-135) without a
                // data-dependent rejection loop (the slowest lane of 64 set the pace: 6 % of a solve)
                int kb = ka;
                if (pool_n > 1) {
                    kb = (int)(((unsigned long long)w0.y * (unsigned)(pool_n - 1)) >> 32);
                    kb += (kb >= ka) ? 1 : 0;
                }
                const int ia = have_pool ? nth_set_bit(pool, ka) : 0, ib = have_pool ? nth_set_bit(pool, kb) : 0;
                const int la = gbase + ia * LPE, lb = gbase + ib * LPE;
                pfitA = par[(2 * D) * WAVE + la];
                pfitB = par[(2 * D) * WAVE + lb];
                pia = have_pool ? ia : -1;
                pib = have_pool ? ib : -1;
                const double extA = par[(2 * D + 1) * WAVE + la], extB = par[(2 * D + 1) * WAVE + lb];
                const double extinction = 0.5 * (extA + extB);
                const double mutation_prob = extinction * (1.0 - inv_gene) + inv_gene;
                // joint limits / half spans: reloaded per round (scalar cache) instead of being
                // hoisted out of the generation loop and parked in spilled scalar registers
                CK<D> cr = fresh_after(c, mix);
                const uint32_t bounded_mask = PIK_COMMON ? ~0u : cr.bounded_mask; // (common configuration: all bounded)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const U4 wj = rng_block(a.rng_seed, STREAM_REPRODUCE, gprob, (unsigned)gen, (unsigned)i | sp_key,
                                            (unsigned)(1 + j));
                    // child of two parents -- src/ik_memetic.cpp:136-170
                    double gene = mix * par[j * WAVE + la] + (1.0 - mix) * par[j * WAVE + lb];
                    gene += u01_from_word(wj.x) * par[(D + j) * WAVE + la] +
                            u01_from_word(wj.y) * par[(D + j) * WAVE + lb];
                    const double original_gene = gene;
                    if (u01_from_word(wj.z) < mutation_prob) {
                        gene += extinction * cr.hspan[j] * uniform_real(-1.0, 1.0, u01_from_word(wj.w));
                    }
                    gene = clamp_joint<D>(cr, j, gene);
                    // fresh random member (empty pool)
                    const double u = u01_from_words(wj.z, wj.w);
                    double v = uniform_real(cr.qmin[j], cr.qmax[j], u);
                    if (!((bounded_mask >> j) & 1u)) { // chain-uniform: continuous joints only
                        if (!have_pool) {
                            // generate_valid_value(population_[i].genes[j]): the stale content of
                            // slot i = the guess right after an initPopulation, else the previous
                            // generation's rank-i individual
                            double cur = best[j];
                            // pin the value: otherwise the two loads are merged into one load of a
                            // selected POINTER, which takes &best[j] and forces best[] into scratch
                            asm volatile("" : "+v"(cur));
                            if (!pop_guess) {
                                const int* order = reinterpret_cast<const int*>(pop_prev + P + (long long)P * D);
                                cur = pop_prev[P + (long long)order[i] * D + j];
                            }
                            v = uniform_real(cur - M_PI, cur + M_PI, u);
                        }
                    }
                    cg[j] = have_pool ? gene : v;
                    cgrad[j] = have_pool ? gene - original_gene : 0.0;
                }
            }
            PIK_TICK(3); // child genes (RNG + mixing)
            {
                EvalOut e;
                evaluate<D, OCC>(c, p, goal, seed, cg, e);
                if (valid) {
                    cfit = e.cost;
                    csol = e.sol;
                    if (pia >= 0) {
                        if (cfit < pfitA) erase_bits |= 1ull << pia;
                        if (cfit < pfitB) erase_bits |= 1ull << pib;
                    }
                }
            }

            PIK_TICK(4); // child evaluation
            // sequential mating-pool semantics: accept up to and including the first eraser
            const unsigned long long er = __ballot(erase_bits != 0ull);
            const unsigned long long ger = (er >> gbase) & gmask_all;
            const int first = ger ? (__ffsll((long long)ger) - 1) : GS;
            const bool accepted = valid && lid <= first;
            {
                const int src = gbase + (first < GS ? first : 0);
                const unsigned lo = (unsigned)shfl_i32((int)(erase_bits & 0xffffffffu), src);
                const unsigned hi = (unsigned)shfl_i32((int)(erase_bits >> 32), src);
                if (first < GS) {
                    const unsigned long long eb = ((unsigned long long)hi << 32) | lo;
                    pool &= ~eb;
                    if (act) erasures += __popcll(eb);
                    next += first + 1;
                } else {
                    next += GS;
                }
            }
            if (accepted) maxfit = fmax(maxfit, cfit);
            if (PIK_POP(a) && accepted) {
                pop_cur[i] = cfit;
#pragma unroll
                for (int j = 0; j < D; ++j) pop_cur[P + (long long)i * D + j] = cg[j];
            }

            PIK_TICK(5); // accept / erase
            // running top-GS: insert accepted children that beat the current worst kept key
            bool qual = accepted && key_less(cfit, i, wfit, widx);
            while (__any(qual)) {
                const unsigned long long qb = __ballot(qual);
                const unsigned long long gq = (qb >> gbase) & gmask_all;
                const bool has = gq != 0ull;
                const int srcl = gbase + (has ? (__ffsll((long long)gq) - 1) : 0);
                const double cf_s = shfl_f64(cfit, srcl);
                const int ci_s = shfl_i32(i, srcl);
                const bool cs_s = shfl_i32(csol ? 1 : 0, srcl) != 0;
                const bool ins = has && key_less(cf_s, ci_s, wfit, widx);
                wave_sync();
                if (ins && lane == srcl) {
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        kept[j * WAVE + wlane] = cg[j];
                        kept[(D + j) * WAVE + wlane] = cgrad[j];
                    }
                }
                wave_sync();
                if (ins && lane == wlane) {
                    kfit = cf_s;
                    kidx = ci_s;
                    ksol = cs_s;
                }
                if (lane == srcl) qual = false;
                recompute_worst();
                qual = qual && key_less(cfit, i, wfit, widx);
            }
            PIK_TICK(6); // insertion into the kept set
        }

        PIK_TICK(7); // after the reproduce loop
        // stored population: full order of this generation (rank -> slot), the sorted population
        // the NEXT generation's empty-pool branch indexes
        if (PIK_POP(a)) {
            wave_sync();
            if (act) {
                int* order = reinterpret_cast<int*>(pop_cur + P + (long long)P * D);
                for (int i = lid; i < P; i += GS) {
                    const double fi = pop_cur[i];
                    int r = 0;
                    for (int m = 0; m < P; ++m) r += key_less(pop_cur[m], m, fi, i) ? 1 : 0;
                    order[r] = i;
                }
            }
            wave_sync();
            pop_guess = false;
        }

        // (3) sortPopulation -- src/ik_memetic.cpp:200-209: only the top E and the extremes matter
        // Only the E lead lanes are slots of the kept set (every other lane's key is +inf with an index above
        // every real one), so a slot's rank is the number of LEAD slots with a smaller key: E shuffles per
        // generation instead of GS (64 at sixteen lanes per elite -- 4 % of such a generation).
        int rank = 0;
        for (int m = 0; m < E; ++m) {
            const double f2 = shfl_f64(kfit, gbase + m * LPE);
            const int i2 = shfl_i32(kidx, gbase + m * LPE);
            rank += key_less(f2, i2, kfit, kidx) ? 1 : 0;
        }
        wave_sync();
        inv[lane] = lane; // keeps every entry a valid lane even if NaN fitness breaks the order
        wave_sync();
        if (lead_lane) inv[gbase + rank] = lane;
        wave_sync();
        const int srcl = inv[gbase + el]; // lane holding the candidate of rank `el`
        efit = shfl_f64(kfit, srcl);
        esol = shfl_i32(ksol ? 1 : 0, srcl) != 0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            eg[j] = kept[j * WAVE + srcl];
            egrad[j] = kept[(D + j) * WAVE + srcl];
        }
        double fmax_all = maxfit;
        for (int off = 1; off < GS; off <<= 1) fmax_all = fmax(fmax_all, shfl_f64(fmax_all, lane ^ off));
        const double fmin = shfl_f64(efit, gbase);
        const bool smin = shfl_i32(esol ? 1 : 0, gbase) != 0;
        // computeExtinctions -- src/ik_memetic.cpp:57-64
        eext = (efit + fmin * ((double)el / (double)(P - 1) - 1.0)) / fmax_all;
        // best_curr_ = population_[0]; best_ = running minimum
        const double curr_fit = fmin;
        if (act && curr_fit < best_fit) {
            const int l0 = inv[gbase];
#pragma unroll
            for (int j = 0; j < D; ++j) best[j] = kept[j * WAVE + l0];
            best_fit = curr_fit;
            best_sol = smin;
        }

        PIK_TICK(8); // sort / rank / extinctions
        // (4) termination / wipeout -- src/ik_memetic.cpp:252-268
        bool just_solved = false;
        bool wipe_pending = false; // wipeout decided this generation, initPopulation not yet run
        if (act) {
            if (p.stop_on_valid && best_sol) {
                gen += 1; // generations completed
                just_solved = true;
                conclude(true, true);
            } else {
                // checkWipeout -- src/ik_memetic.cpp:43-55
                bool wipe = false;
                if (has_prev) {
                    const bool improved = curr_fit < prev_fit - p.wipeout_tol;
                    if (!improved) wipe = true;
                }
                if (!wipe) {
                    prev_fit = curr_fit;
                    has_prev = true;
                }
                gen += 1;
                if (wipe) {
                    wipeouts += 1;
                    need_init = true; // ik.initPopulation(robot, cost_fn, ik.best().genes)
                    wipe_pending = true;
                }
            }
        }
        // species: `terminate` -- the first generation in which any species returns a solution
        // ends the race for all of them (src/ik_memetic.cpp:263-266, 337-348)
        bool terminate = false;
        if (S > 1 && p.stop_on_first) {
            for (int k = 0; k < SP; ++k) terminate = terminate || (shfl_i32(just_solved ? 1 : 0, sbase + k * GS) != 0);
        }
        if (act && (gen >= p.max_generations || terminate)) {
            // leaving the loop: the reference has already run initPopulation for a wipeout decided
            // in this last generation (E + P evaluations) -- count it, nothing will use it
            if (wipe_pending) {
                init_epoch += 1;
                need_init = false;
            }
            // post-loop of ik_memetic_impl -- src/ik_memetic.cpp:272-282
            if (!p.stop_on_valid && best_sol) {
                conclude(true, true);
            } else if (p.approx) {
                conclude(true, false);
            } else {
                conclude(false, false);
            }
        }
        resolve();
        // compaction: still running at this pass's generation mark -> park for the next pass (several
        // species: the running ones are in lock-step; the problem parks when they reach the mark)
        {
            bool want = act && gen >= a.pause_gen;
            if (S > 1) {
                bool any = false;
                for (int k = 0; k < SP; ++k) any = any || (shfl_i32(want ? 1 : 0, sbase + k * GS) != 0);
                want = pend && any;
            }
            if (want) park();
        }
        PIK_TICK(9); // termination / resolve / park
    }
    PIK_TIMING_FLUSH();
    // The last wavefront out re-arms this launch's counters for the next batch on the slot, so the
    // host enqueues no memset between the passes: with tens of streams in flight a dependent
    // dispatch costs ~1 ms of queue latency, 14 of them per batch were a third of its latency.
    // (every wavefront has read *n_in before it gets here; the stores are visible to the next
    // kernel of the stream at the kernel boundary)
    if (lane == 0) {
        __threadfence();
        const unsigned d = atomicAdd(a.done, 1u);
        if (d + 1u == gridDim.x) {
            *a.work_counter = 0ull;
            if (a.n_in) *const_cast<unsigned*>(a.n_in) = 0u;
            *a.done = 0u;
        }
    }
}

} // namespace pik
