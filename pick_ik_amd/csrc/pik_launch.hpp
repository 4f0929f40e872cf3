// pik_launch.hpp -- kernel launches for one chain length D (templates; instantiated by pik_inst.hip
// once per D so that the per-DOF kernels compile in parallel translation units).
#pragma once

#include <new>

#include "pik_kernels.hpp"
#include "pik_solver.hpp"

namespace pik {

static_assert(sizeof(StatsK) == sizeof(pikamd_stats), "stats layout");
static_assert(sizeof(BatchRecord) == sizeof(BatchK), "batch record layout");
static_assert(offsetof(BatchRecord, completed) == offsetof(BatchK, completed), "batch record layout");

inline int pow2ceil_log2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// Makes the slot's device constants buffer hold this call's chain + params.  The upload is
// skipped when the slot already holds the same bytes (the common case: same robot, same params
// every call), so steady-state launches cost one table copy + the kernels.  When the contents
// change, the slot's previous stream is drained first so no in-flight kernel can observe the rewrite.
template <int D>
int upload_consts(pikamd_solver* s, const ParamsK* pk, int slot, hipStream_t st, const ConstsK<D>** out) {
    static_assert(sizeof(ConstsK<D>) <= CONSTS_STRIDE, "constants slot too small");
    static_assert(MAX_TIPS == PIKAMD_MAX_TIPS, "tip limit");
    // staging copy: per handle (handles are used from one thread at a time, different handles may
    // be used concurrently)
    ConstsK<D>& want = *reinterpret_cast<ConstsK<D>*>(s->consts_tmp);
    // only the chains in use are compared / uploaded
    const size_t used = offsetof(ConstsK<D>, more) + sizeof(ChainK<D>) * (size_t)(s->n_tips - 1);
    std::memset(&want, 0, used);
    want.chain = make_chain_k<D>(s->chain);
    for (int k = 1; k < s->n_tips; ++k) want.more[k - 1] = make_chain_k<D>(s->more[k - 1]);
    want.n_tips = s->n_tips;
    if (pk) want.params = *pk;
    char* host = s->consts_host + (size_t)slot * CONSTS_STRIDE;
    char* dev = s->consts_dev + (size_t)slot * CONSTS_STRIDE;
    if (!s->consts_valid[slot] || std::memcmp(host, &want, used) != 0) {
        if (s->consts_valid[slot]) HIP_TRY(hipStreamSynchronize(s->consts_stream[slot]));
        s->consts_valid[slot] = false;
        std::memcpy(host, &want, used);
        HIP_TRY(hipMemcpyAsync(dev, host, used, hipMemcpyHostToDevice, st));
        // later calls on OTHER streams may reuse these bytes without copying: make them visible
        HIP_TRY(hipStreamSynchronize(st));
        s->consts_valid[slot] = true;
    }
    s->consts_stream[slot] = st;
    *out = reinterpret_cast<const ConstsK<D>*>(dev);
    return 0;
}

template <int D>
int launch_fk(pikamd_solver* s, long long n, const double* d_q, double* d_out, hipStream_t st) {
    if (n == 0) return 0;
    const ConstsK<D>* kc = nullptr;
    if (int rc = upload_consts<D>(s, nullptr, SLOT_HOOKS, st, &kc)) return rc;
    const int block = 256;
    const long long grid = (n + block - 1) / block;
    if (s->n_tips > 1)
        hipLaunchKernelGGL((fk_kernel<D, true>), dim3((unsigned)grid), dim3(block), 0, st, kc, n, d_q, d_out);
    else
        hipLaunchKernelGGL(fk_kernel<D>, dim3((unsigned)grid), dim3(block), 0, st, kc, n, d_q, d_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int D>
int launch_cost(pikamd_solver* s, const ParamsK& pk, long long n, const double* d_goal, const double* d_seed,
                const double* d_q, double* d_cost, int* d_sol, hipStream_t st) {
    if (n == 0) return 0;
    const ConstsK<D>* kc = nullptr;
    if (int rc = upload_consts<D>(s, &pk, SLOT_HOOKS, st, &kc)) return rc;
    const int block = 64;
    const long long grid = (n + block - 1) / block;
    if (s->n_tips > 1)
        hipLaunchKernelGGL((cost_kernel<D, true>), dim3((unsigned)grid), dim3(block), 0, st, kc, n, d_goal,
                           d_seed, d_q, d_cost, d_sol);
    else
        hipLaunchKernelGGL(cost_kernel<D>, dim3((unsigned)grid), dim3(block), 0, st, kc, n, d_goal, d_seed, d_q,
                           d_cost, d_sol);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int D>
int launch_step(pikamd_solver* s, const ParamsK& pk, long long n, const double* d_goal, const double* d_seed,
                double* d_local, double* d_best, double* d_lc, double* d_bc, double* d_grad, int* d_imp,
                hipStream_t st) {
    if (n == 0) return 0;
    const ConstsK<D>* kc = nullptr;
    if (int rc = upload_consts<D>(s, &pk, SLOT_HOOKS, st, &kc)) return rc;
    const int block = 64;
    const long long grid = (n + block - 1) / block;
    if (s->n_tips > 1)
        hipLaunchKernelGGL((gd_step_kernel<D, true>), dim3((unsigned)grid), dim3(block), 0, st, kc, n, d_goal,
                           d_seed, d_local, d_best, d_lc, d_bc, d_grad, d_imp);
    else
        hipLaunchKernelGGL(gd_step_kernel<D>, dim3((unsigned)grid), dim3(block), 0, st, kc, n, d_goal, d_seed,
                           d_local, d_best, d_lc, d_bc, d_grad, d_imp);
    HIP_TRY(hipGetLastError());
    return 0;
}

// The call's batch table goes to the device through a ring of table slots.  The kernels of the call
// read their table whenever a problem starts, resumes or finishes (find_batch), so a slot may only be
// rewritten once the LAST kernel of the call that used it has run: TableSlot records the slot's event
// when launch_solve returns -- behind every launch of the call, on whichever path it leaves -- and the
// ring waits for that event when it comes round (practically never blocks: 256 calls in flight).
struct TableSlot {
    pikamd_solver* s = nullptr;
    int slot = -1;
    hipStream_t st = nullptr;
    TableSlot() = default;
    TableSlot(const TableSlot&) = delete;
    TableSlot& operator=(const TableSlot&) = delete;
    ~TableSlot() {
        if (slot < 0) return;
        // (if the record fails the slot stays marked unused-but-unsafe: drain the stream instead)
        if (hipEventRecord(s->table_event[slot], st) != hipSuccess) (void)hipStreamSynchronize(st);
        s->table_used[slot] = true;
    }
};

inline int upload_batch_table(pikamd_solver* s, BatchRecord* batches, int n, hipStream_t st, const BatchK** out,
                              long long* total, TableSlot& guard) {
    long long start = 0;
    for (int k = 0; k < n; ++k) {
        batches[k].start = start;
        start += batches[k].B;
    }
    *total = start;
    const int slot = s->table_next;
    s->table_next = (slot + 1) % TABLE_RING;
    if (s->table_used[slot]) HIP_TRY(hipEventSynchronize(s->table_event[slot]));
    BatchRecord* host = s->tables_host + (size_t)slot * PIKAMD_MAX_BATCHES;
    BatchRecord* dev = s->tables_dev + (size_t)slot * PIKAMD_MAX_BATCHES;
    std::memcpy(host, batches, sizeof(BatchRecord) * (size_t)n);
    s->table_used[slot] = false;
    guard.s = s;
    guard.slot = slot;
    guard.st = st;
    HIP_TRY(hipMemcpyAsync(dev, host, sizeof(BatchRecord) * (size_t)n, hipMemcpyHostToDevice, st));
    *out = reinterpret_cast<const BatchK*>(dev);
    return 0;
}

// launch schedule of a memetic call
struct Schedule {
    // forced lanes per elite (experiments / tests): passes starting at generation >= lpe_from[i] run
    // with lpe_of[i]; n_sched == 0: adaptive (the default, see launch_solve)
    int lpe_from[4] = {0, 0, 0, 0}, lpe_of[4] = {1, 1, 1, 1}, n_sched = 0;
    int marks[16], n_marks = 0;
    bool occ2_ok = true;
    long long occ2_from = 0;
};

// the handle's scheduling options (pikamd_set_option) -> the launch schedule of one call
inline void make_schedule(const pikamd_solver* s, const ParamsK& pk, int gs, int S, Schedule& sc) {
    const bool multi = s->n_tips > 1;
    const SolverOptions& o = s->opt;
    auto ok = [&](int v) { return lpe_allowed(s, v, gs, S, multi); };
    if (o.lpe > 0 && ok(o.lpe)) {
        sc.lpe_of[0] = o.lpe;
        sc.n_sched = 1;
    }
    if (o.n_sched > 0) {
        bool good = true;
        for (int i = 0; i < o.n_sched; ++i) good = good && ok(o.sched_of[i]);
        if (good) {
            sc.n_sched = o.n_sched;
            for (int i = 0; i < o.n_sched; ++i) {
                sc.lpe_from[i] = o.sched_from[i];
                sc.lpe_of[i] = o.sched_of[i];
            }
        }
    }
    if (S != 1 && sc.n_sched > 1) sc.n_sched = 1; // (species: one choice for all passes when forced)
    if (multi && sc.n_sched > 0 && !ok(sc.lpe_of[0])) sc.n_sched = 0; // (a request several tips cannot serve: adaptive)
    if (multi && sc.n_sched > 1) sc.n_sched = 1;
    (void)ok;
    // Compaction passes: generation marks at which still-running problems are parked in HBM and
    // re-packed densely for the next launch (results do not depend on the marks).  Dense in the
    // tail: the survivor count halves every ~10 generations, and every pass re-chooses the lanes
    // per elite for the survivors it gets.
    {
        static const int default_marks[] = {2, 4, 8, 12, 16, 24, 32, 40, 48, 64, 80};
        const int* m = o.passes_set ? o.marks : default_marks;
        const int n = o.passes_set ? o.n_marks : (int)(sizeof default_marks / sizeof default_marks[0]);
        (void)S; // (several species park and resume like one: a record per (problem, species))
        for (int i = 0; i < n && sc.n_marks < 15; ++i)
            if (m[i] > 0 && m[i] < pk.max_generations && (sc.n_marks == 0 || m[i] > sc.marks[sc.n_marks - 1]))
                sc.marks[sc.n_marks++] = m[i];
    }
    sc.occ2_ok = S == 1;
    // first-pass wavefronts from which the two-per-SIMD variant pays (measured crossover with
    // overlapped batches on 1024 SIMDs: slower at 512, +3 % at 640, +5 % at 768, +26 % at 1024)
    sc.occ2_from = (long long)s->num_cu * 4 * 5 / 8;
    if (o.two_per_simd >= 0) {
        sc.occ2_ok = sc.occ2_ok && o.two_per_simd != 0;
        if (o.two_per_simd > 1) sc.occ2_from = o.two_per_simd; // (explicit threshold)
    }
    if (o.force_occ2) sc.occ2_from = 0;
}

template <int D>
int launch_solve(pikamd_solver* s, const pikamd_params* p, const ParamsK& pk, BatchRecord* batches, int n_batches,
                 unsigned long long rng_seed, hipStream_t st, int slot, bool reserve_only) {
    long long B = 0;
    for (int k = 0; k < n_batches; ++k) B += batches[k].B;
    if (B == 0) return 0;
    const ConstsK<D>* kc = nullptr;
    if (int rc = upload_consts<D>(s, &pk, slot, st, &kc)) return rc;
    SolveArgs a;
    TableSlot table; // (declared before any launch: its destructor records the event behind the last one)
    std::memset(&a, 0, sizeof a);
    a.n_batches = n_batches;
    for (int k = 0; k < n_batches; ++k) a.signal |= batches[k].completed != nullptr;
    a.rng_seed = rng_seed;
    a.B = B;
    if (p->mode == 1) {
        if (reserve_only) return 0;
        if (int rc = upload_batch_table(s, batches, n_batches, st, &a.batches, &a.B, table)) return rc;
        const int block = 64;
        const long long grid = (a.B + block - 1) / block;
#if !defined(PIK_STRICT)
        // few targets: the cooperative descent with 16 (or 8) lanes per problem, as long as every
        // problem gets its wavefront share in one round -- a third of the one-lane latency (the
        // plugin's local mode is called with one target at a time); option lanes_per_elite=1 forces one lane
        {
            int lpe = 0;
            const long long simds = (long long)s->num_cu * 4;
            const bool multi = s->n_tips > 1;
            if (lpe_allowed(s, 16, 1, 1, multi) && a.B <= simds * (WAVE / 16)) lpe = 16;
            else if (lpe_allowed(s, 8, 1, 1, multi) && a.B <= simds * (WAVE / 8)) lpe = 8;
            if (s->opt.lpe > 0) { // forced lanes per problem (1 = the one-lane kernel)
                const int v = s->opt.lpe;
                lpe = (v == 16 && lpe_allowed(s, 16, 1, 1, multi)) ? 16 : (v == 8 && lpe_allowed(s, 8, 1, 1, multi)) ? 8 : 0;
            }
            if (lpe) {
                const long long per_wave = WAVE / lpe;
                const dim3 g((unsigned)((a.B + per_wave - 1) / per_wave));
                if (multi) {
                    if (lpe == 16) hipLaunchKernelGGL((ik_gradient_wide_kernel<D, 16, true>), g, dim3(block), 0, st, kc, a);
                    else hipLaunchKernelGGL((ik_gradient_wide_kernel<D, 8, true>), g, dim3(block), 0, st, kc, a);
                } else if (lpe == 16) hipLaunchKernelGGL((ik_gradient_wide_kernel<D, 16>), g, dim3(block), 0, st, kc, a);
                else hipLaunchKernelGGL((ik_gradient_wide_kernel<D, 8>), g, dim3(block), 0, st, kc, a);
                HIP_TRY(hipGetLastError());
                return 0;
            }
        }
#else
        // exact flavours, one tip frame: the team forms of the memoised descent with 16 (or 4) lanes per problem, as long
        // as every problem gets its wavefront share in one round (pik_kernels.hpp ik_gradient_team_kernel); option
        // lanes_per_elite = 1 forces one lane, 16 / 4 one of the team kernels whatever the size of the call
        if (s->n_tips == 1) {
            int lpe = 0;
            const long long simds = (long long)s->num_cu * 4;
            if (lpe_allowed(s, 16, 1, 1, false) && a.B <= simds * (WAVE / 16)) lpe = 16;
            else if (lpe_allowed(s, 4, 1, 1, false) && a.B <= simds * (WAVE / 4)) lpe = 4;
            if (s->opt.lpe > 0) {
                const int v = s->opt.lpe;
                lpe = ((v == 16 || v == 4) && lpe_allowed(s, v, 1, 1, false)) ? v : 0;
            }
            if (lpe) {
                const long long per_wave = WAVE / lpe;
                const dim3 g((unsigned)((a.B + per_wave - 1) / per_wave));
                if (lpe == 16) hipLaunchKernelGGL((ik_gradient_team_kernel<D, 16>), g, dim3(block), 0, st, kc, a);
                else hipLaunchKernelGGL((ik_gradient_team_kernel<D, 4>), g, dim3(block), 0, st, kc, a);
                HIP_TRY(hipGetLastError());
                return 0;
            }
        }
#endif
        if (s->n_tips > 1)
            hipLaunchKernelGGL((ik_gradient_kernel<D, true>), dim3((unsigned)grid), dim3(block), 0, st, kc, a);
        else
            hipLaunchKernelGGL(ik_gradient_kernel<D>, dim3((unsigned)grid), dim3(block), 0, st, kc, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // memetic: groups of GS * LPE lanes per problem, one wavefront per workgroup, persistent waves
    a.gs_log2 = pow2ceil_log2(pk.elites);
    const int gs = 1 << a.gs_log2;
    // species: pow2ceil(S) groups per problem share a wavefront and park / resume together
    const int S = p->memetic_num_threads > 1 ? p->memetic_num_threads : 1;
    a.species = S;
    a.sp_log2 = pow2ceil_log2(S);
    Schedule sc;
    make_schedule(s, pk, gs, S, sc);
    // Regime.  More lanes per elite buy latency with throughput (a problem-generation costs 31 us of
    // SIMD time at one lane per elite, 68 / 95 / 158 us at 4 / 8 / 16): right when the chip would
    // otherwise idle behind this call's long-running problems, wrong when other calls are queued up
    // to use it.  Measured (512 x 4096-problem steps in pools of 64): 2 calls in flight 4.14 (adaptive)
    // vs 3.84 M solves/s (one lane per elite everywhere), 4 calls in flight 4.32 vs 4.69.  So: with
    // three or more OTHER calls of this handle still in flight, every pass uses one lane per elite.
    bool throughput_regime = false;
    {
        int others = 0;
        for (int k = 0; k < N_DEVICE_SLOTS + N_HOST_JOBS; ++k)
            if (k != slot && s->slot_event_used[k] && hipEventQuery(s->slot_event[k]) == hipErrorNotReady) ++others;
        (void)hipGetLastError(); // (hipErrorNotReady is an answer, not a failure)
        throughput_regime = others >= 3;
        if (s->opt.regime != 0) throughput_regime = s->opt.regime == 2; // forced (option "regime")
    }
    // ... and a call with the chip to itself takes the two-per-SIMD variant of the one-lane kernel only when
    // its wavefronts outnumber the SIMDs (measured on the driver's 20-batch pool: threshold 5/8 -> 9/8 of
    // the SIMD count: 2.85 -> 2.90 M solves/s; with other calls queued up the lower threshold stays)
    if (!throughput_regime && s->opt.two_per_simd < 2 && !s->opt.force_occ2) sc.occ2_from = (long long)s->num_cu * 4 * 9 / 8;
    int n_marks = sc.n_marks;
    // A call whose problems each get a wavefront of the widest variant in one round gains nothing from
    // compaction (there is nothing to re-pack into): one launch, no passes -- 3-6 % off the latency of
    // the plugin-style calls (B = 1 .. 256).  Not in the throughput regime, where the one-lane
    // wavefronts hold 16 problems each and re-packing is what keeps them full.
    if (!reserve_only && !throughput_regime && sc.n_sched == 0 && !s->opt.passes_set) {
        int widest = 1;
        for (int l : {16, 8, 4, 2})
            if (widest == 1 && lpe_allowed(s, l, gs, S, s->n_tips > 1)) widest = l;
        if (widest > 1 && B <= (long long)s->num_cu * 4 * (WAVE / (gs * widest * (1 << a.sp_log2)))) n_marks = 0;
    }
    // per-slot scratch: parked state (one record per problem and species), two survivor lists
    const long long cap = B;
    const size_t recs = (size_t)B * (size_t)S;
    const size_t d_rows = (size_t)StateRows<D>::D_ROWS(pk.elites);
    const size_t off_d = 0;
    const size_t off_l = off_d + sizeof(double) * d_rows * recs;
    const size_t off_i = off_l + sizeof(long long) * StateRows<D>::L_ROWS * recs;
    const size_t off_list = off_i + sizeof(int) * StateRows<D>::I_ROWS * recs;
    const size_t off_cnt = off_list + sizeof(int) * 2 * (size_t)cap;
    // stored population for chains with unbounded variables: 2 parities x (P fitness + P*D genes +
    // P ints order) per problem
    const bool has_unbounded = s->chain.bounded_mask != ((1u << s->chain.dof) - 1u);
    const size_t pop_stride = (size_t)pk.population * (1 + D) + ((size_t)pk.population + 1) / 2;
    const size_t off_pop = (off_cnt + 64 + 63) / 64 * 64;
    const size_t total = off_pop + (has_unbounded ? sizeof(double) * 2 * pop_stride * (size_t)cap * (size_t)S : 0);
    if (n_marks > 0 || has_unbounded) {
        if (int rc = s->slot_state[slot].ensure(total)) return rc;
    }
    if (reserve_only) return 0;
    if (int rc = upload_batch_table(s, batches, n_batches, st, &a.batches, &a.B, table)) return rc;
    char* base = (char*)s->slot_state[slot].p;
    a.pop = has_unbounded ? (double*)(base + off_pop) : nullptr;
    a.pop_stride = (long long)pop_stride;
    a.cap = cap;
    a.st_d = n_marks ? (double*)(base + off_d) : nullptr;
    a.st_l = n_marks ? (long long*)(base + off_l) : nullptr;
    a.st_i = n_marks ? (int*)(base + off_i) : nullptr;
    int* lists[2] = {n_marks ? (int*)(base + off_list) : nullptr, n_marks ? (int*)(base + off_list) + cap : nullptr};
    unsigned char* cblk = s->counters + COUNTER_BLOCK * (size_t)slot;
    unsigned long long* c_work = (unsigned long long*)cblk;
    unsigned* c_nlist = (unsigned*)(cblk + 128);
    unsigned* c_done = (unsigned*)(cblk + 256);
    if (s->counters_dirty[slot]) HIP_TRY(hipMemsetAsync(cblk, 0, COUNTER_BLOCK, st)); // after a failed launch
    s->counters_dirty[slot] = true;

    // waves per CU of each kernel variant: asked once per handle and FLAVOUR (the general and the exact kernels are
    // different code with different register counts: a handle switched between them by the option "arithmetic"
    // must not size one flavour's grids by the other's figure)
#define PIK_OCC_ROW (PIK_COMMON ? (PIK_NO_GOALS ? 1 : 2) : (PIK_XF ? 3 : 0))
    auto capacity_of = [&](auto kernel, int variant, long long* cap_out) -> int {
        int per_cu = s->occupancy_cache[PIK_OCC_ROW][variant];
        if (per_cu == 0) {
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, WAVE, 0));
            if (per_cu < 1) per_cu = 1;
            s->occupancy_cache[PIK_OCC_ROW][variant] = per_cu;
        }
        *cap_out = (long long)s->num_cu * per_cu;
        return 0;
    };
    // one kernel variant: lanes per elite, wavefronts per SIMD it is compiled for
    struct Variant {
        int lpe, id;
        long long capacity;  // wavefronts the chip holds of it
        long long hi;        // largest problem count it is chosen for (adaptive schedule)
    };
    Variant var[8];
    int n_var = 0;
    auto add_variant = [&](auto kernel, int lpe_, int id) -> int {
        Variant v{lpe_, id, 0, 0};
        if (int rc = capacity_of(kernel, id, &v.capacity)) return rc;
        var[n_var++] = v;
        return 0;
    };
    auto launch_variant = [&](const Variant& v, unsigned lo, unsigned hi) -> int {
        const long long groups_per_wave = WAVE / (gs * v.lpe * (1 << a.sp_log2));
        const long long waves_needed = (a.B + groups_per_wave - 1) / groups_per_wave;
        const long long grid = waves_needed < v.capacity ? waves_needed : v.capacity;
        a.sel_lo = lo;
        a.sel_hi = hi;
        const dim3 g((unsigned)grid), b(WAVE);
        switch (v.id) {
            case 5: hipLaunchKernelGGL((memetic_kernel<D, 16>), g, b, 0, st, kc, a); break;
            case 4: hipLaunchKernelGGL((memetic_kernel<D, 8>), g, b, 0, st, kc, a); break;
            case 3: hipLaunchKernelGGL((memetic_kernel<D, 4>), g, b, 0, st, kc, a); break;
            case 2: hipLaunchKernelGGL((memetic_kernel<D, 2>), g, b, 0, st, kc, a); break;
            case 7:
                if constexpr (D <= 9) hipLaunchKernelGGL((memetic_kernel<D, 1, false, 2>), g, b, 0, st, kc, a);
                break;
#if !defined(PIK_STRICT)
            case 10: hipLaunchKernelGGL((memetic_kernel<D, 16, true>), g, b, 0, st, kc, a); break;
            case 9: hipLaunchKernelGGL((memetic_kernel<D, 8, true>), g, b, 0, st, kc, a); break;
#endif
            case 8: hipLaunchKernelGGL((memetic_kernel<D, 2, true>), g, b, 0, st, kc, a); break;
            case 6: hipLaunchKernelGGL((memetic_kernel<D, 1, true>), g, b, 0, st, kc, a); break;
            default: hipLaunchKernelGGL((memetic_kernel<D, 1>), g, b, 0, st, kc, a); break;
        }
        HIP_TRY(hipGetLastError());
        return 0;
    };
    // candidate variants, widest first.  Adaptive rule: a pass runs with the MOST lanes per elite
    // whose wavefronts still fit the chip in one round for the problems it has (fewest generations'
    // latency without queueing); the one-lane variant takes everything larger, compiled for two
    // wavefronts per SIMD from the measured crossover on.
    const bool multi = s->n_tips > 1;
    const long long occ2_from_problems = sc.occ2_from * WAVE / gs; // first-pass wavefronts -> problems
    if (multi) {
#if !defined(PIK_STRICT)
        if (!throughput_regime || sc.n_sched > 0) {
            if (lpe_allowed(s, 16, gs, S, multi))
                if (int rc = add_variant(memetic_kernel<D, 16, true>, 16, 10)) return rc;
            if (lpe_allowed(s, 8, gs, S, multi))
                if (int rc = add_variant(memetic_kernel<D, 8, true>, 8, 9)) return rc;
        }
#endif
        if (!throughput_regime || sc.n_sched > 0)
            if (lpe_allowed(s, 2, gs, S, multi))
                if (int rc = add_variant(memetic_kernel<D, 2, true>, 2, 8)) return rc;
        if (int rc = add_variant(memetic_kernel<D, 1, true>, 1, 6)) return rc;
    } else {
        const bool wide_ok = !throughput_regime || sc.n_sched > 0; // (a forced schedule may ask for any)
        if (wide_ok && lpe_allowed(s, 16, gs, S, multi))
            if (int rc = add_variant(memetic_kernel<D, 16>, 16, 5)) return rc;
        if (wide_ok && lpe_allowed(s, 8, gs, S, multi))
            if (int rc = add_variant(memetic_kernel<D, 8>, 8, 4)) return rc;
        if (wide_ok && lpe_allowed(s, 4, gs, S, multi))
            if (int rc = add_variant(memetic_kernel<D, 4>, 4, 3)) return rc;
        if (wide_ok && lpe_allowed(s, 2, gs, S, multi))
            if (int rc = add_variant(memetic_kernel<D, 2>, 2, 2)) return rc;
        if (int rc = add_variant(memetic_kernel<D, 1>, 1, 1)) return rc;
        if constexpr (D <= 9) {
            // (its LDS footprint, 6 D rows, lets 5..8 wavefronts share a CU up to D = 9; beyond that
            //  the register cap would cost scratch traffic for nothing.  The exact flavours have it too: their
            //  descent and evaluations are calls that need fewer than 256 registers, and a second wavefront per
            //  SIMD hides the latencies a lone one waits out)
            bool occ2 = sc.occ2_ok && !(disabled_lanes_of(s, EXACT_FLAVOUR) & 1u);
#if defined(PIK_STRICT)
            occ2 = occ2 && s->chain.float_mask == 0u && s->chain.n_mimic == 0; // (a floating or a mimic joint: the literal descent, one per SIMD only)
#endif
            if (occ2)
                if (int rc = add_variant(memetic_kernel<D, 1, false, 2>, 1, 7)) return rc;
        }
    }
    for (int i = 0; i < n_var; ++i) {
        const long long per_wave = WAVE / (gs * var[i].lpe * (1 << a.sp_log2));
        var[i].hi = (var[i].lpe > 1) ? (long long)s->num_cu * 4 * per_wave // one wavefront per SIMD
                    : (var[i].id == 7 || i == n_var - 1) ? 0xffffffffll
                                                         : occ2_from_problems - 1;
    }
    for (int k = 0; k <= n_marks; ++k) {
        a.fresh = (k == 0);
        a.pause_gen = (k < n_marks) ? sc.marks[k] : 0x7fffffff;
        a.list_in = (k == 0) ? nullptr : lists[(k - 1) & 1];
        a.n_in = (k == 0) ? nullptr : c_nlist + k;
        a.list_out = n_marks ? lists[k & 1] : nullptr;
        a.n_out = n_marks ? c_nlist + (k + 1) : nullptr;
        a.work_counter = c_work + k;
        a.done = c_done + k;
        const int start_gen = (k == 0) ? 0 : sc.marks[k - 1];
        if (sc.n_sched > 0) {
            // forced lanes per elite
            int lpe_k = sc.lpe_of[0];
            for (int i = 1; i < sc.n_sched; ++i)
                if (start_gen >= sc.lpe_from[i]) lpe_k = sc.lpe_of[i];
            int pick = -1;
            for (int i = 0; i < n_var && pick < 0; ++i)
                if (var[i].lpe == lpe_k) pick = i; // (one lane per elite: the one-per-SIMD build comes first)
            if (pick < 0) pick = n_var - 1;
            // one lane per elite and a call that fills the chip: the two-per-SIMD build
            if (var[pick].lpe == 1 && var[n_var - 1].id == 7 && a.B >= occ2_from_problems) pick = n_var - 1;
            if (int rc = launch_variant(var[pick], 0u, 0xffffffffu)) return rc;
            continue;
        }
        // adaptive: variant i serves problem counts in (hi of the next wider one, its own hi]
        long long lo = 0;
        for (int i = 0; i < n_var; ++i) {
            const long long hi = var[i].hi < lo ? lo : var[i].hi;
            if (k == 0) {
                // the first pass knows its size: one launch
                if (a.B > lo && a.B <= hi) {
                    if (int rc = launch_variant(var[i], 0u, 0xffffffffu)) return rc;
                    break;
                }
            } else if (a.B > lo && hi > lo) {
                // (a pass cannot have more survivors than the call has problems: narrower variants
                //  whose range starts beyond that are not enqueued)
                if (int rc = launch_variant(var[i], (unsigned)lo, (unsigned)(hi > 0xffffffffll ? 0xffffffffll : hi))) return rc;
            }
            lo = hi;
        }
    }
    s->counters_dirty[slot] = false;
    if (!s->slot_event[slot]) HIP_TRY(hipEventCreateWithFlags(&s->slot_event[slot], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(s->slot_event[slot], st));
    s->slot_event_used[slot] = true;
    return 0;
}

template <int D>
const LaunchOps* make_ops() {
    static const LaunchOps ops = {&launch_fk<D>, &launch_cost<D>, &launch_step<D>, &launch_solve<D>};
    return &ops;
}

} // namespace pik
