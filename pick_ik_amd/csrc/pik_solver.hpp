// pik_solver.hpp -- host-side state of a solver handle, shared by the C ABI translation unit
// (pik_amd.hip) and the per-DOF kernel translation units (pik_inst.hip, one per chain length, so
// that the library builds in parallel).  No device code here.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/pick_ik_amd.h"
#include "pik_host.hpp"

namespace pik {

// thread-local message of the last failure (pikamd_last_error); defined in pik_amd.hip
char* error_buffer();
constexpr size_t ERROR_BUFFER_SIZE = 512;

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), ERROR_BUFFER_SIZE, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return ::pik::fail(PIKAMD_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

// device (or pinned host) scratch that outlives a call
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool pinned_host = false;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        release();
        const size_t want = bytes < 4096 ? 4096 : bytes;
        const hipError_t e = pinned_host ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(PIKAMD_EHIP, "%s(%zu) failed: %s", pinned_host ? "hipHostMalloc" : "hipMalloc", want,
                        hipGetErrorString(e));
        }
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)(pinned_host ? hipHostFree(p) : hipFree(p));
        p = nullptr;
        cap = 0;
    }
};

constexpr size_t CONSTS_STRIDE = 57344; // >= ConstsK<16> with PIKAMD_MAX_TIPS chains
constexpr size_t COUNTER_BLOCK = 512;
// Internal scratch slots: the caller's device-entry-point slots, then the slots of the host-pointer
// jobs (pikamd_solve_batches_async: their own streams, staging and scratch, so that they never
// collide with a caller's in-flight device batches), then one for the parity hooks' constants.
constexpr int N_DEVICE_SLOTS = PIKAMD_MAX_SLOTS;
constexpr int N_HOST_JOBS = PIKAMD_MAX_HOST_JOBS;
// ... one more host job (and its slot) for the library's own self test, so that it never touches the staging
// memory of a caller's job
constexpr int JOB_SELF_TEST = N_HOST_JOBS;
constexpr int SLOT_HOOKS = N_DEVICE_SLOTS + N_HOST_JOBS + 1;
constexpr int N_SLOTS = SLOT_HOOKS + 1;
// ring of batch tables (one per call): a call's table must stay intact until its kernels have run
constexpr int TABLE_RING = 256;

// Scheduling options of a handle (pikamd_set_option).  None of them changes a result -- the kernel
// variants and pass schedules re-order the same arithmetic (asserted bit for bit by the tests) -- they
// exist for experiments, tests and callers who know their load better than the adaptive rule does.
struct SolverOptions {
    int lpe = 0;                       // "lanes_per_elite": 0 = adaptive, else 1 / 2 / 4 / 8 / 16
    int n_sched = 0;                   // "lanes_per_elite_schedule": passes from generation from[i] on run of[i]
    int sched_from[4] = {0, 0, 0, 0}, sched_of[4] = {1, 1, 1, 1};
    bool passes_set = false;           // "passes": generation marks of the compaction passes
    int marks[16] = {}, n_marks = 0;
    int two_per_simd = -1;             // "two_per_simd": -1 adaptive, 0 never, 1 from the default threshold,
                                       //                 > 1 from that many first-pass wavefronts
    int regime = 0;                    // "regime": 0 adaptive, 1 latency, 2 throughput
    // kernel variants pikamd_self_test found disagreeing with the one-lane kernel on this handle's chain:
    // bit v (2, 4, 8, 16) = v lanes per elite, bit 1 = the two-per-SIMD build of the one-lane kernel
    unsigned disabled_lanes = 0;       // ... of the product flavours' kernels (general, common-configuration)
    unsigned disabled_lanes_exact = 0; // ... of the exact kernels: a width switched off for one flavour stays on for the other
    bool specialised = true;           // "specialised": use the common-configuration kernels when a call qualifies
    int shard_chunks = 0;              // "shard_chunks": host jobs per device of pikamd_solve_batch_sharded (0 = default)
    bool soa = false;                  // "joint_layout": the joint-vector arrays of the solve entry points are [dof][B]
    bool exact = true;                 // "arithmetic" = "exact" (the DEFAULT): every call runs the exact kernels
                                       // (pik_exact: joint vectors = the reference algorithm's); "fast" = opt-in
    bool auto_self_test = true;        // "self_test" = "auto": pikamd_self_test once per parameter set served by the
                                       // general or the exact kernels, in front of the first solve ("off": never)
    bool force_occ2 = false;           // (pikamd_self_test only: the two-per-SIMD kernel whatever the call's size)
    // pikamd_solve_batch_host: wall-clock limits in seconds as the reference's loops have them (0 = none) --
    // "host_max_time": the whole call (MemeticIkParams::max_time / GradientIkParams::max_time of a one-problem call,
    // tested in front of every generation / step, src/ik_memetic.cpp:226-228, src/ik_gradient.cpp:112-115);
    // "host_gd_max_time": one elite's descent (memetic_gd_max_time, src/ik_memetic.cpp:75-78)
    double host_max_time = 0.0, host_gd_max_time = 0.0;
};

// mirror of the kernels' BatchK (pik_kernels.hpp), kept here so that this header needs no device code
struct BatchRecord {
    long long start, B;
    const double* goal;
    const double* seed;
    const double* guess;
    long long problem_offset;
    double* solution;
    int* status;
    double* cost;
    void* stats;
    unsigned* completed;
};

// one host-pointer job in flight (pikamd_solve_batches_async .. pikamd_wait)
struct HostJob {
    bool pending = false;
    hipStream_t stream = nullptr;
    DevBuf dev;                 // inputs | outputs of all batches of the job
    DevBuf host;                // pinned mirror
    size_t in_bytes = 0, out_bytes = 0;
    int n_batches = 0;
    int dof = 0;
    struct Out {
        int64_t B;
        double* solution;
        int32_t* status;
        double* final_cost;
        pikamd_stats* stats;
        size_t off_solution, off_status, off_cost, off_stats; // in the staging buffers
    } outs[PIKAMD_MAX_BATCHES];
};

} // namespace pik

struct pikamd_solver {
    int device = -1;
    int num_cu = 0;
    pik::ChainHost chain;                      // tip 0 (and the variables' limits)
    pik::ChainHost more[PIKAMD_MAX_TIPS - 1];  // tips 1.. of a multi-tip chain (padded, see pik_host.hpp)
    int n_tips = 1;
    // per slot one 512-byte block of counters, zero whenever no batch is in flight on the slot (the
    // kernels re-arm what they used): u64 work[16] | u32 n_list[17] @128 | u32 done[16] @256
    unsigned char* counters = nullptr;
    bool counters_dirty[pik::N_SLOTS] = {};
    char* consts_dev = nullptr;             // [N_SLOTS][CONSTS_STRIDE] ConstsK<D> per slot
    char* consts_host = nullptr;            // pinned mirror
    alignas(16) char consts_tmp[pik::CONSTS_STRIDE]; // staging copy of one ConstsK<D> (upload_consts)
    bool consts_valid[pik::N_SLOTS] = {};
    hipStream_t consts_stream[pik::N_SLOTS] = {};
    pik::DevBuf stage[8];                   // staging for the parity hooks (fk / cost / step)
    pik::DevBuf slot_state[pik::N_SLOTS];   // parked solver state + survivor lists of each slot
    pik::DevBuf slot_soa[pik::N_SLOTS];     // option joint_layout = soa: the [B][dof] copies the kernels work on
    // batch tables: ring of TABLE_RING entries of PIKAMD_MAX_BATCHES records
    pik::BatchRecord* tables_dev = nullptr;
    pik::BatchRecord* tables_host = nullptr; // pinned
    hipEvent_t table_event[pik::TABLE_RING] = {};
    bool table_used[pik::TABLE_RING] = {};
    int table_next = 0;
    pik::HostJob jobs[pik::N_HOST_JOBS + 1];  // (+ JOB_SELF_TEST)
    int occupancy_cache[4][16] = {};        // waves per CU of the memetic kernel variants (0 = not asked yet),
                                            // general [0], common-configuration [1], common + joint goals [2], exact [3] kernels
    // an event behind the last launch of every slot: how many OTHER calls are still in flight decides
    // between the latency-greedy and the efficiency-greedy choice of kernel variants (launch_solve)
    hipEvent_t slot_event[pik::N_SLOTS] = {};
    bool slot_event_used[pik::N_SLOTS] = {};
    char kernel_name[64];
    pik::SolverOptions opt;
    // parameter sets whose kernel variants the self test has compared on this handle (option self_test = auto)
    std::vector<unsigned long long> self_tested; // kernel sets the automatic self test has passed (maybe_self_test)
    bool in_self_test = false;
    int self_test_runs = 0;     // automatic self tests run on this handle, and what they cost (pikamd_self_test_cost)
    double self_test_ms = 0.0;
};

namespace pik {

// what a per-DOF translation unit exports: the launches of every kernel for chains of that length
struct LaunchOps {
    int (*fk)(pikamd_solver*, long long n, const double* d_q, double* d_out, hipStream_t);
    int (*cost)(pikamd_solver*, const ParamsK&, long long n, const double* d_goal, const double* d_seed,
                const double* d_q, double* d_cost, int* d_sol, hipStream_t);
    int (*step)(pikamd_solver*, const ParamsK&, long long n, const double* d_goal, const double* d_seed,
                double* d_local, double* d_best, double* d_lc, double* d_bc, double* d_grad, int* d_imp,
                hipStream_t);
    // batches: host array of n_batches records with device pointers (start is filled in here)
    int (*solve)(pikamd_solver*, const pikamd_params*, const ParamsK&, BatchRecord* batches, int n_batches,
                 unsigned long long rng_seed, hipStream_t, int slot, bool reserve_only);
};

#define PIK_DECLARE_OPS(N) const LaunchOps* launch_ops_d##N();
PIK_DECLARE_OPS(1) PIK_DECLARE_OPS(2) PIK_DECLARE_OPS(3) PIK_DECLARE_OPS(4) PIK_DECLARE_OPS(5) PIK_DECLARE_OPS(6)
PIK_DECLARE_OPS(7) PIK_DECLARE_OPS(8) PIK_DECLARE_OPS(9) PIK_DECLARE_OPS(10) PIK_DECLARE_OPS(11) PIK_DECLARE_OPS(12)
PIK_DECLARE_OPS(13) PIK_DECLARE_OPS(14) PIK_DECLARE_OPS(15) PIK_DECLARE_OPS(16)
#undef PIK_DECLARE_OPS

inline const LaunchOps* launch_ops(int dof) {
    switch (dof) {
        case 1: return launch_ops_d1();
        case 2: return launch_ops_d2();
        case 3: return launch_ops_d3();
        case 4: return launch_ops_d4();
        case 5: return launch_ops_d5();
        case 6: return launch_ops_d6();
        case 7: return launch_ops_d7();
        case 8: return launch_ops_d8();
        case 9: return launch_ops_d9();
        case 10: return launch_ops_d10();
        case 11: return launch_ops_d11();
        case 12: return launch_ops_d12();
        case 13: return launch_ops_d13();
        case 14: return launch_ops_d14();
        case 15: return launch_ops_d15();
        case 16: return launch_ops_d16();
        default: return nullptr;
    }
}

// Does the launcher serve v lanes per elite (gs = pow2ceil(elites), S species, several tips?) for this handle?
// exact: the rules of the exact flavours' kernels (the cooperative descent of the product flavours -- 8 / 16
// lanes, plain Denavit-Hartenberg chains only -- does not exist there; every width is the dealt-out literal one).
#if defined(PIK_STRICT)
constexpr bool EXACT_FLAVOUR = true;
#else
constexpr bool EXACT_FLAVOUR = false;
#endif
inline unsigned disabled_lanes_of(const pikamd_solver* s, bool exact) {
    return exact ? s->opt.disabled_lanes_exact : s->opt.disabled_lanes;
}
inline bool lpe_allowed(const pikamd_solver* s, int v, int gs, int S, bool multi, bool exact = EXACT_FLAVOUR) {
    constexpr int WAVE_LANES = 64;
    if (v > 1 && (disabled_lanes_of(s, exact) & (unsigned)v)) return false; // switched off by pikamd_self_test
    // species: pow2ceil(S) groups of a wavefront per problem -- the lanes of all of them have to fit; one tip
    if (S != 1) {
        int sp = 1;
        while (sp < S) sp <<= 1;
        if (v == 1) return true;
        if (gs * v * sp > WAVE_LANES) return false;
        if (!multi) {
            if (!exact && v >= 8 && s->chain.dh_general_mask != 0u) return false;
            return v == 2 || v == 4 || v == 8 || v == 16;
        }
        // (several tips: the choices below, with the species' lanes counted in)
        gs *= sp;
    }
    // several tips: one lane per elite, or two -- the pair that evaluates the two line-search points
    // of a gradient step side by side (the gradient comes with the accept evaluation there)
    if (multi) {
        if (v == 1 || (v == 2 && gs * v <= WAVE_LANES)) return true;
        // ... or the cooperative routine for several tips (gd_wide_multi): 8 / 16, plain DH chains on every tip
        if (!exact && (v == 8 || v == 16) && gs * v <= WAVE_LANES) {
            bool plain = s->chain.dh_general_mask == 0u;
            for (int k = 1; k < s->n_tips; ++k) plain = plain && s->more[k - 1].dh_general_mask == 0u;
            return plain;
        }
        return false;
    }
    // 8 / 16 lanes per elite: the cooperative routine (gd_wide), plain DH chains only.  (A copy of its chain
    // loop with the general step of an ill-conditioned pair of axes was built and taken out again: with it the
    // general-flavour kernels for 16 variables -- 512 registers + scratch -- came out wrong at 8 and 16 lanes,
    // found by the fuzz of the common-configuration kernels, which runs the general ones as its reference.)
    if (!exact && v >= 8 && s->chain.dh_general_mask != 0u) return false;
    return (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) && gs * v <= WAVE_LANES;
}

} // namespace pik
