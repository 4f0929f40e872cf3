// pik_amd.hip -- C ABI (include/pick_ik_amd.h) over the gfx950 kernels.
//
// Host side of the drop-in boundary: model extraction (pik_host.hpp), launch geometry, staging
// for the host-pointer entry points.  No CPU compute path exists here: without a HIP device every
// entry point fails with PIKAMD_ENODEVICE.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/pick_ik_amd.h"
#include "pik_host.hpp"
#include "pik_kernels.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(PIKAMD_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));           \
    } while (0)

static_assert(sizeof(pik::StatsK) == sizeof(pikamd_stats), "stats layout");

// device scratch that outlives a call
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes < 4096 ? 4096 : bytes;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(PIKAMD_EHIP, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

} // namespace

constexpr size_t CONSTS_STRIDE = 20480; // ConstsK<12> with PIKAMD_MAX_TIPS chains
constexpr size_t COUNTER_BLOCK = 512;

struct pikamd_solver {
    int device = -1;
    int num_cu = 0;
    pik::ChainHost chain;                      // tip 0 (and the variables' limits)
    pik::ChainHost more[PIKAMD_MAX_TIPS - 1];  // tips 1.. of a multi-tip chain (padded, see pik_host.hpp)
    int n_tips = 1;
    // per slot one 512-byte block of counters, zero whenever no batch is in flight on the slot (the
    // kernels re-arm what they used): u64 work[16] | u32 n_list[17] @128 | u32 done[16] @256
    unsigned char* counters = nullptr;
    bool counters_dirty[PIKAMD_MAX_SLOTS + 1] = {};
    char* consts_dev = nullptr;             // [PIKAMD_MAX_SLOTS + 1][CONSTS_STRIDE] ConstsK<D> per slot
    char* consts_host = nullptr;            // pinned mirror
    alignas(16) char consts_tmp[CONSTS_STRIDE]; // staging copy of one ConstsK<D> (upload_consts)
    bool consts_valid[PIKAMD_MAX_SLOTS + 1] = {};
    hipStream_t consts_stream[PIKAMD_MAX_SLOTS + 1] = {};
    DevBuf stage[8];                        // staging for the host-pointer entry points
    DevBuf slot_state[PIKAMD_MAX_SLOTS];    // parked solver state + survivor lists of each slot
    char kernel_name[64];
    bool latency_mode = false; // set by the synchronous host-pointer entry point
};

namespace {

int pow2ceil_log2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// ---- dispatch on the compile-time DOF ------------------------------------------------------
#define PIK_DISPATCH_D(dof, CALL)                                                   \
    switch (dof) {                                                                  \
        case 1: { constexpr int D = 1; CALL; } break;                               \
        case 2: { constexpr int D = 2; CALL; } break;                               \
        case 3: { constexpr int D = 3; CALL; } break;                               \
        case 4: { constexpr int D = 4; CALL; } break;                               \
        case 5: { constexpr int D = 5; CALL; } break;                               \
        case 6: { constexpr int D = 6; CALL; } break;                               \
        case 7: { constexpr int D = 7; CALL; } break;                               \
        case 8: { constexpr int D = 8; CALL; } break;                               \
        case 9: { constexpr int D = 9; CALL; } break;                              \
        case 10: { constexpr int D = 10; CALL; } break;                              \
        case 11: { constexpr int D = 11; CALL; } break;                              \
        case 12: { constexpr int D = 12; CALL; } break;                              \
        default:                                                                    \
            return fail(PIKAMD_EUNSUPPORTED, "dof %d: kernels are instantiated for 1..12", dof); \
    }

// Makes the slot's device constants buffer hold this call's chain + params.  The upload is
// skipped when the slot already holds the same bytes (the common case: same robot, same params
// every call), so steady-state launches cost one memset + one kernel.  When the contents change,
// the slot's previous stream is drained first so no in-flight kernel can observe the rewrite.
template <int D>
int upload_consts(pikamd_solver* s, const pik::ParamsK* pk, int slot, hipStream_t st,
                  const pik::ConstsK<D>** out) {
    static_assert(sizeof(pik::ConstsK<D>) <= CONSTS_STRIDE, "constants slot too small");
    static_assert(pik::MAX_TIPS == PIKAMD_MAX_TIPS, "tip limit");
    // staging copy: per handle (handles are used from one thread at a time, different handles may
    // be used concurrently), on the heap (14 KB)
    pik::ConstsK<D>& want = *reinterpret_cast<pik::ConstsK<D>*>(s->consts_tmp);
    // only the chains in use are compared / uploaded
    const size_t used = offsetof(pik::ConstsK<D>, more) + sizeof(pik::ChainK<D>) * (size_t)(s->n_tips - 1);
    std::memset(&want, 0, used);
    want.chain = pik::make_chain_k<D>(s->chain);
    for (int k = 1; k < s->n_tips; ++k) want.more[k - 1] = pik::make_chain_k<D>(s->more[k - 1]);
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
    *out = reinterpret_cast<const pik::ConstsK<D>*>(dev);
    return 0;
}

template <int D>
int launch_fk(pikamd_solver* s, long long n, const double* d_q, double* d_out, hipStream_t st) {
    if (n == 0) return 0;
    const pik::ConstsK<D>* kc = nullptr;
    if (int rc = upload_consts<D>(s, nullptr, PIKAMD_MAX_SLOTS, st, &kc)) return rc;
    const int block = 256;
    const long long grid = (n + block - 1) / block;
    if (s->n_tips > 1)
        hipLaunchKernelGGL((pik::fk_kernel<D, true>), dim3((unsigned)grid), dim3(block), 0, st, kc, n, d_q, d_out);
    else
        hipLaunchKernelGGL(pik::fk_kernel<D>, dim3((unsigned)grid), dim3(block), 0, st, kc, n, d_q, d_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int D>
int launch_cost(pikamd_solver* s, const pik::ParamsK& pk, long long n, const double* d_goal,
                const double* d_seed, const double* d_q, double* d_cost, int* d_sol, hipStream_t st) {
    if (n == 0) return 0;
    const pik::ConstsK<D>* kc = nullptr;
    if (int rc = upload_consts<D>(s, &pk, PIKAMD_MAX_SLOTS, st, &kc)) return rc;
    const int block = 64;
    const long long grid = (n + block - 1) / block;
    if (s->n_tips > 1)
        hipLaunchKernelGGL((pik::cost_kernel<D, true>), dim3((unsigned)grid), dim3(block), 0, st, kc, n,
                           d_goal, d_seed, d_q, d_cost, d_sol);
    else
        hipLaunchKernelGGL(pik::cost_kernel<D>, dim3((unsigned)grid), dim3(block), 0, st, kc, n, d_goal,
                           d_seed, d_q, d_cost, d_sol);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int D>
int launch_step(pikamd_solver* s, const pik::ParamsK& pk, long long n, const double* d_goal,
                const double* d_seed, double* d_local, double* d_best, double* d_lc, double* d_bc,
                double* d_grad, int* d_imp, hipStream_t st) {
    if (n == 0) return 0;
    const pik::ConstsK<D>* kc = nullptr;
    if (int rc = upload_consts<D>(s, &pk, PIKAMD_MAX_SLOTS, st, &kc)) return rc;
    const int block = 64;
    const long long grid = (n + block - 1) / block;
    if (s->n_tips > 1)
        hipLaunchKernelGGL((pik::gd_step_kernel<D, true>), dim3((unsigned)grid), dim3(block), 0, st, kc, n,
                           d_goal, d_seed, d_local, d_best, d_lc, d_bc, d_grad, d_imp);
    else
        hipLaunchKernelGGL(pik::gd_step_kernel<D>, dim3((unsigned)grid), dim3(block), 0, st, kc, n, d_goal,
                           d_seed, d_local, d_best, d_lc, d_bc, d_grad, d_imp);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int D>
int launch_solve(pikamd_solver* s, const pikamd_params* p, const pik::ParamsK& pk, pik::SolveArgs a,
                 hipStream_t st, int slot, bool latency_mode, bool reserve_only = false) {
    if (a.B == 0) return 0;
    const pik::ConstsK<D>* kc = nullptr;
    if (int rc = upload_consts<D>(s, &pk, slot, st, &kc)) return rc;
    if (p->mode == 1) {
        if (reserve_only) return 0;
        const int block = 64;
        const long long grid = (a.B + block - 1) / block;
        if (s->n_tips > 1)
            hipLaunchKernelGGL((pik::ik_gradient_kernel<D, true>), dim3((unsigned)grid), dim3(block), 0, st, kc, a);
        else
            hipLaunchKernelGGL(pik::ik_gradient_kernel<D>, dim3((unsigned)grid), dim3(block), 0, st, kc, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // memetic: groups of GS * LPE lanes per problem, one wavefront per workgroup, persistent waves
    a.gs_log2 = pow2ceil_log2(pk.elites);
    const int gs = 1 << a.gs_log2;
    // Lanes per elite: a small batch cannot fill the chip at one lane per elite (4096 problems x 4
    // elites = 256 wavefronts for 1024 SIMDs); spreading each elite over LPE lanes shortens every
    // generation (probes and line-search probes run side by side) and fills the idle SIMDs.
    // Results do not depend on LPE.
    // Two regimes: a caller that waits for one batch (pikamd_solve_batch) wants the shortest
    // critical path -> LPE 4 from the start when the batch is too small to fill the chip; a caller
    // that keeps many batches in flight (pikamd_solve_batch_device on several streams) is bound by
    // wave slots -> LPE 1 while most problems are alive, LPE 4 only for the late passes, where a
    // few survivors run long and their latency bounds the batch.
    // species: pow2ceil(S) groups per problem share a wavefront; no passes / extra lanes then
    const int S = p->memetic_num_threads > 1 ? p->memetic_num_threads : 1;
    a.species = S;
    a.sp_log2 = pow2ceil_log2(S);
    // schedule: passes starting at generation >= lpe_from[i] run with lpe_of[i] lanes per elite
    int lpe_from[4] = {0, 0, 0, 0}, lpe_of[4] = {1, 1, 1, 1}, n_sched = 1;
#if !defined(PIK_STRICT)
    {
        const long long waves1 = (a.B * gs + pik::WAVE - 1) / pik::WAVE;
        const long long simds = (long long)s->num_cu * 4;
        const bool multi = s->n_tips > 1; // several tips: one lane per elite
        const bool small = S == 1 && !multi && gs * 4 <= pik::WAVE && waves1 * 4 <= simds;
        auto ok = [&](int v) { return S == 1 && !multi && (v == 1 || v == 2 || v == 4) && gs * v <= pik::WAVE; };
        if (small) {
            lpe_of[0] = latency_mode ? 4 : 1;
            lpe_from[1] = 32;
            lpe_of[1] = 4;
            n_sched = 2;
        }
        if (const char* ev = std::getenv("PIK_LPE")) {
            const int v = std::atoi(ev);
            if (ok(v)) {
                lpe_of[0] = v;
                n_sched = 1;
            }
        }
        if (const char* ev = std::getenv("PIK_LPE_TAIL")) {
            const int v = std::atoi(ev);
            if (ok(v)) {
                lpe_from[1] = 32;
                lpe_of[1] = v;
                n_sched = 2;
            }
        }
        if (const char* ev = std::getenv("PIK_TAIL_FROM")) {
            if (n_sched >= 2) lpe_from[1] = std::atoi(ev);
        }
        // PIK_LPE_SCHED="g0:l0,g1:l1,..." (ascending generations, first must be 0), e.g. "0:1,16:4" (8 lanes per elite was measured: no gain over 4)
        if (const char* ev = std::getenv("PIK_LPE_SCHED")) {
            int n = 0, from[4], of[4];
            const char* q = ev;
            bool good = true;
            while (*q && n < 4) {
                from[n] = std::atoi(q);
                while (*q && *q != ':') ++q;
                if (*q != ':') { good = false; break; }
                of[n] = std::atoi(++q);
                good = good && ok(of[n]) && (n == 0 ? from[0] == 0 : from[n] > from[n - 1]);
                ++n;
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
            if (good && n > 0) {
                n_sched = n;
                for (int i = 0; i < n; ++i) {
                    lpe_from[i] = from[i];
                    lpe_of[i] = of[i];
                }
            }
        }
    }
#endif
    (void)latency_mode;
    // Compaction passes: generation marks at which still-running problems are parked in HBM and
    // re-packed densely for the next launch (results do not depend on the marks).
    int marks[16];
    int n_marks = 0;
    {
        const char* ev = std::getenv("PIK_PASSES");
        const char* spec = S > 1 ? "none" : (ev ? ev : "2,4,8,16,32,64");
        const char* q = spec;
        while (*q && n_marks < 15) {
            const int v = std::atoi(q);
            if (v > 0 && v < pk.max_generations && (n_marks == 0 || v > marks[n_marks - 1])) marks[n_marks++] = v;
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    }
    // per-slot scratch: parked state (SoA over problems), two survivor lists, counters
    const long long cap = a.B;
    const size_t d_rows = (size_t)pik::StateRows<D>::D_ROWS(pk.elites);
    const size_t off_d = 0;
    const size_t off_l = off_d + sizeof(double) * d_rows * (size_t)cap;
    const size_t off_i = off_l + sizeof(long long) * pik::StateRows<D>::L_ROWS * (size_t)cap;
    const size_t off_list = off_i + sizeof(int) * pik::StateRows<D>::I_ROWS * (size_t)cap;
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
    if (n_marks > 15) return fail(PIKAMD_EINVAL, "too many compaction passes");
    if (s->counters_dirty[slot]) HIP_TRY(hipMemsetAsync(cblk, 0, COUNTER_BLOCK, st)); // after a failed launch
    s->counters_dirty[slot] = true;

    bool occ2_ok = S == 1;
    // first-pass wavefronts from which the two-per-SIMD variant pays (measured crossover with
    // overlapped batches on 1024 SIMDs: slower at 512, +3 % at 640, +5 % at 768, +26 % at 1024)
    long long occ2_from = (long long)s->num_cu * 4 * 5 / 8;
    if (const char* ev = std::getenv("PIK_OCC2")) {
        occ2_ok = occ2_ok && std::atoi(ev) != 0;
        if (std::atoi(ev) > 1) occ2_from = std::atoi(ev); // (experiments: explicit threshold)
    }
    (void)occ2_ok;
    auto launch = [&](auto kernel, int lpe_) -> int {
        const long long groups_per_wave = pik::WAVE / (gs * lpe_ * (1 << a.sp_log2));
        const long long waves_needed = (a.B + groups_per_wave - 1) / groups_per_wave;
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, pik::WAVE, 0));
        if (per_cu < 1) per_cu = 1;
        const long long capacity = (long long)s->num_cu * per_cu;
        const long long grid = waves_needed < capacity ? waves_needed : capacity;
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(pik::WAVE), 0, st, kc, a);
        HIP_TRY(hipGetLastError());
        return 0;
    };
    for (int k = 0; k <= n_marks; ++k) {
        a.fresh = (k == 0);
        a.pause_gen = (k < n_marks) ? marks[k] : 0x7fffffff;
        a.list_in = (k == 0) ? nullptr : lists[(k - 1) & 1];
        a.n_in = (k == 0) ? nullptr : c_nlist + k;
        a.list_out = n_marks ? lists[k & 1] : nullptr;
        a.n_out = n_marks ? c_nlist + (k + 1) : nullptr;
        a.work_counter = c_work + k;
        a.done = c_done + k;
        int rc;
        const int start_gen = (k == 0) ? 0 : marks[k - 1];
        int lpe_k = lpe_of[0];
        for (int i = 1; i < n_sched; ++i)
            if (start_gen >= lpe_from[i]) lpe_k = lpe_of[i];
#if !defined(PIK_STRICT)
        if (lpe_k == 4)
            rc = launch(pik::memetic_kernel<D, 4>, 4);
        else if (lpe_k == 2)
            rc = launch(pik::memetic_kernel<D, 2>, 2);
        else
#endif
        if (s->n_tips > 1) {
            rc = launch(pik::memetic_kernel<D, 1, true>, 1);
        } else {
#if !defined(PIK_STRICT)
            // a batch whose first pass (nearly) fills the chip by itself: the two-per-SIMD build
            // (its LDS footprint, 6 D rows, lets 5..8 wavefronts share a CU up to D = 9; beyond
            //  that the register cap would cost scratch traffic for nothing)
            const long long waves1 = (a.B * gs + pik::WAVE - 1) / pik::WAVE;
            if constexpr (D <= 9) {
                if (occ2_ok && waves1 >= occ2_from)
                    rc = launch(pik::memetic_kernel<D, 1, false, 2>, 1);
                else
                    rc = launch(pik::memetic_kernel<D, 1>, 1);
            } else
#endif
                rc = launch(pik::memetic_kernel<D, 1>, 1);
            (void)occ2_ok;
        }
        (void)lpe_k;
        (void)lpe_from;
        if (rc) return rc;
    }
    s->counters_dirty[slot] = false;
    return 0;
}

int check_solver(const pikamd_solver* s) {
    if (!s) return fail(PIKAMD_EINVAL, "solver handle is NULL");
    return 0;
}

} // namespace

extern "C" {

void pikamd_default_params(pikamd_params* p) {
    if (!p) return;
    // src/pick_ik_parameters.yaml defaults
    p->mode = 0;
    p->gd_step_size = 0.0001;
    p->gd_max_iters = 100;
    p->gd_min_cost_delta = 1.0e-12;
    p->position_threshold = 0.001;
    p->orientation_threshold = 0.001;
    p->cost_threshold = 0.001;
    p->position_scale = 1.0;
    p->rotation_scale = 0.5;
    p->center_joints_weight = 0.0;
    p->avoid_joint_limits_weight = 0.0;
    p->minimal_displacement_weight = 0.0;
    p->stop_optimization_on_valid_solution = 1;
    p->memetic_num_threads = 1;
    p->memetic_stop_on_first_solution = 1;
    p->memetic_population_size = 16;
    p->memetic_elite_size = 4;
    p->memetic_wipeout_fitness_tol = 0.00001;
    p->memetic_max_generations = 100;
    p->memetic_gd_max_iters = 25;
    p->return_approximate_solution = 0;
}

const char* pikamd_last_error(void) { return g_err; }

const char* pikamd_version(void) { return "pick_ik_amd 0.1.0 (gfx950)"; }

static int32_t create_solver(const pik::ChainHost* chains, int n_tips, int32_t device_ordinal,
                             pikamd_solver** out) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
        return fail(PIKAMD_ENODEVICE, "no HIP device available (this library has no CPU path)");
    if (device_ordinal < 0 || device_ordinal >= count)
        return fail(PIKAMD_EINVAL, "device_ordinal %d out of range [0, %d)", device_ordinal, count);
    HIP_TRY(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
    pikamd_solver* s = new (std::nothrow) pikamd_solver();
    if (!s) return fail(PIKAMD_EHIP, "out of host memory");
    s->device = device_ordinal;
    s->num_cu = prop.multiProcessorCount;
    s->chain = chains[0];
    s->n_tips = n_tips;
    for (int k = 1; k < n_tips; ++k) s->more[k - 1] = chains[k];
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->counters), COUNTER_BLOCK * (PIKAMD_MAX_SLOTS + 1));
    if (e == hipSuccess) e = hipMemset(s->counters, 0, COUNTER_BLOCK * (PIKAMD_MAX_SLOTS + 1));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->consts_dev), CONSTS_STRIDE * (PIKAMD_MAX_SLOTS + 1));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&s->consts_host), CONSTS_STRIDE * (PIKAMD_MAX_SLOTS + 1), hipHostMallocDefault);
    if (e != hipSuccess) {
        pikamd_destroy(s);
        return fail(PIKAMD_EHIP, "device allocation failed: %s", hipGetErrorString(e));
    }
    *out = s;
    return 0;
}

int32_t pikamd_create(const pikamd_chain* chain, int32_t device_ordinal, pikamd_solver** out) {
    if (!out) return fail(PIKAMD_EINVAL, "out is NULL");
    *out = nullptr;
    pik::ChainHost ch;
    if (const char* msg = pik::build_chain(chain, ch)) return fail(PIKAMD_EINVAL, "%s", msg);
    return create_solver(&ch, 1, device_ordinal, out);
}

int32_t pikamd_create_multi(const pikamd_multi_chain* chain, int32_t device_ordinal,
                            pikamd_solver** out) {
    if (!out) return fail(PIKAMD_EINVAL, "out is NULL");
    *out = nullptr;
    if (!chain || !chain->tips) return fail(PIKAMD_EINVAL, "multi-tip chain is NULL");
    if (chain->n_tips < 1 || chain->n_tips > PIKAMD_MAX_TIPS)
        return fail(PIKAMD_EINVAL, "n_tips %d out of range [1, %d]", chain->n_tips, PIKAMD_MAX_TIPS);
    if (!chain->qmin || !chain->qmax) return fail(PIKAMD_EINVAL, "chain has NULL arrays");
    std::vector<pik::ChainHost> ch((size_t)chain->n_tips); // (fresh: build_chain accumulates flag bits)
    uint32_t used = 0;
    for (int k = 0; k < chain->n_tips; ++k) {
        if (const char* msg = pik::build_tip_chain(chain, k, ch[k])) return fail(PIKAMD_EINVAL, "tip %d: %s", k, msg);
        used |= ch[k].active_mask;
    }
    // every variable must move some tip (get_active_variable_indices: the union over the tips)
    if (used != ((chain->dof >= 32) ? ~0u : ((1u << chain->dof) - 1u)))
        return fail(PIKAMD_EINVAL, "a variable is on no tip's path");
    return create_solver(ch.data(), chain->n_tips, device_ordinal, out);
}

int32_t pikamd_n_tips(const pikamd_solver* s) { return s ? s->n_tips : 0; }

void pikamd_destroy(pikamd_solver* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->counters) (void)hipFree(s->counters);
    if (s->consts_dev) (void)hipFree(s->consts_dev);
    if (s->consts_host) (void)hipHostFree(s->consts_host);
    for (auto& b : s->stage) b.release();
    for (auto& b : s->slot_state) b.release();
    delete s;
}

int32_t pikamd_variables(const pikamd_solver* s, double* out) {
    if (int rc = check_solver(s)) return rc;
    if (!out) return fail(PIKAMD_EINVAL, "out is NULL");
    for (int j = 0; j < s->chain.dof; ++j) {
        out[7 * j + 0] = s->chain.qmin[j];
        out[7 * j + 1] = s->chain.qmax[j];
        out[7 * j + 2] = s->chain.mid[j];
        out[7 * j + 3] = s->chain.hspan[j];
        out[7 * j + 4] = s->chain.vrcp[j];
        out[7 * j + 5] = s->chain.mdf[j];
        out[7 * j + 6] = ((s->chain.bounded_mask >> j) & 1u) ? 1.0 : 0.0;
    }
    return 0;
}

int32_t pikamd_fk_batch_device(pikamd_solver* s, int64_t n, const double* d_q, double* d_pos_quat,
                               void* stream) {
    if (int rc = check_solver(s)) return rc;
    if (n < 0 || (n > 0 && (!d_q || !d_pos_quat))) return fail(PIKAMD_EINVAL, "bad arguments");
    HIP_TRY(hipSetDevice(s->device));
    PIK_DISPATCH_D(s->chain.dof, return launch_fk<D>(s, n, d_q, d_pos_quat, (hipStream_t)stream));
    return 0;
}

int32_t pikamd_fk_batch(pikamd_solver* s, int64_t n, const double* q, double* pos_quat) {
    if (int rc = check_solver(s)) return rc;
    if (n < 0 || (n > 0 && (!q || !pos_quat))) return fail(PIKAMD_EINVAL, "bad arguments");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(s->device));
    const size_t d = (size_t)s->chain.dof;
    if (int rc = s->stage[0].ensure(sizeof(double) * d * (size_t)n)) return rc;
    if (int rc = s->stage[1].ensure(sizeof(double) * 7 * (size_t)s->n_tips * (size_t)n)) return rc;
    HIP_TRY(hipMemcpy(s->stage[0].p, q, sizeof(double) * d * (size_t)n, hipMemcpyHostToDevice));
    if (int rc = pikamd_fk_batch_device(s, n, (const double*)s->stage[0].p, (double*)s->stage[1].p, nullptr))
        return rc;
    HIP_TRY(hipMemcpy(pos_quat, s->stage[1].p, sizeof(double) * 7 * (size_t)s->n_tips * (size_t)n, hipMemcpyDeviceToHost));
    return 0;
}

int32_t pikamd_cost_batch(pikamd_solver* s, const pikamd_params* p, int64_t n,
                          const double* goal_pos_quat, const double* seed, const double* q,
                          double* cost, int32_t* is_solution) {
    if (int rc = check_solver(s)) return rc;
    pik::ParamsK pk;
    if (!p) return fail(PIKAMD_EINVAL, "params is NULL");
    if (const char* msg = pik::make_params_k(p, pk)) {
        // the memetic-only constraints do not apply to the cost hooks
        pikamd_params q2 = *p;
        q2.mode = 1;
        if (const char* m2 = pik::make_params_k(&q2, pk)) return fail(PIKAMD_EINVAL, "%s", m2);
        (void)msg;
    }
    if (n < 0 || (n > 0 && (!goal_pos_quat || !seed || !q))) return fail(PIKAMD_EINVAL, "bad arguments");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(s->device));
    const size_t d = (size_t)s->chain.dof, N = (size_t)n;
    if (int rc = s->stage[0].ensure(sizeof(double) * 7 * (size_t)s->n_tips * N)) return rc;
    if (int rc = s->stage[1].ensure(sizeof(double) * d * N)) return rc;
    if (int rc = s->stage[2].ensure(sizeof(double) * d * N)) return rc;
    if (int rc = s->stage[3].ensure(sizeof(double) * N)) return rc;
    if (int rc = s->stage[4].ensure(sizeof(int) * N)) return rc;
    HIP_TRY(hipMemcpy(s->stage[0].p, goal_pos_quat, sizeof(double) * 7 * (size_t)s->n_tips * N, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->stage[1].p, seed, sizeof(double) * d * N, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->stage[2].p, q, sizeof(double) * d * N, hipMemcpyHostToDevice));
    PIK_DISPATCH_D(s->chain.dof, {
        if (int rc = launch_cost<D>(s, pk, n, (const double*)s->stage[0].p, (const double*)s->stage[1].p,
                                    (const double*)s->stage[2].p, cost ? (double*)s->stage[3].p : nullptr,
                                    is_solution ? (int*)s->stage[4].p : nullptr, nullptr))
            return rc;
    });
    if (cost) HIP_TRY(hipMemcpy(cost, s->stage[3].p, sizeof(double) * N, hipMemcpyDeviceToHost));
    if (is_solution) HIP_TRY(hipMemcpy(is_solution, s->stage[4].p, sizeof(int) * N, hipMemcpyDeviceToHost));
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

int32_t pikamd_gd_step_batch(pikamd_solver* s, const pikamd_params* p, int64_t n,
                             const double* goal_pos_quat, const double* seed, double* local,
                             double* best, double* local_cost, double* best_cost,
                             double* gradient, int32_t* improved) {
    if (int rc = check_solver(s)) return rc;
    pik::ParamsK pk;
    pikamd_params q2;
    if (!p) return fail(PIKAMD_EINVAL, "params is NULL");
    q2 = *p;
    q2.mode = 1;
    if (const char* msg = pik::make_params_k(&q2, pk)) return fail(PIKAMD_EINVAL, "%s", msg);
    if (n < 0 || (n > 0 && (!goal_pos_quat || !seed || !local || !best || !local_cost || !best_cost || !gradient)))
        return fail(PIKAMD_EINVAL, "bad arguments");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(s->device));
    const size_t d = (size_t)s->chain.dof, N = (size_t)n;
    const size_t sz[8] = {sizeof(double) * 7 * (size_t)s->n_tips * N, sizeof(double) * d * N, sizeof(double) * d * N,
                          sizeof(double) * d * N, sizeof(double) * N,     sizeof(double) * N,
                          sizeof(double) * d * N, sizeof(int) * N};
    for (int i = 0; i < 8; ++i)
        if (int rc = s->stage[i].ensure(sz[i])) return rc;
    HIP_TRY(hipMemcpy(s->stage[0].p, goal_pos_quat, sz[0], hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->stage[1].p, seed, sz[1], hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->stage[2].p, local, sz[2], hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->stage[3].p, best, sz[3], hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->stage[4].p, local_cost, sz[4], hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->stage[5].p, best_cost, sz[5], hipMemcpyHostToDevice));
    PIK_DISPATCH_D(s->chain.dof, {
        if (int rc = launch_step<D>(s, pk, n, (const double*)s->stage[0].p, (const double*)s->stage[1].p,
                                    (double*)s->stage[2].p, (double*)s->stage[3].p, (double*)s->stage[4].p,
                                    (double*)s->stage[5].p, (double*)s->stage[6].p, (int*)s->stage[7].p, nullptr))
            return rc;
    });
    HIP_TRY(hipMemcpy(local, s->stage[2].p, sz[2], hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(best, s->stage[3].p, sz[3], hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(local_cost, s->stage[4].p, sz[4], hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(best_cost, s->stage[5].p, sz[5], hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(gradient, s->stage[6].p, sz[6], hipMemcpyDeviceToHost));
    if (improved) HIP_TRY(hipMemcpy(improved, s->stage[7].p, sz[7], hipMemcpyDeviceToHost));
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

int32_t pikamd_solve_batch_device(pikamd_solver* s, const pikamd_params* p, int64_t B,
                                  const double* d_goal_pos_quat, const double* d_seed,
                                  uint64_t rng_seed, int64_t problem_offset, double* d_solution,
                                  int32_t* d_status, double* d_final_cost, pikamd_stats* d_stats,
                                  void* stream, int32_t slot) {
    if (int rc = check_solver(s)) return rc;
    pik::ParamsK pk;
    if (const char* msg = pik::make_params_k(p, pk)) return fail(PIKAMD_EINVAL, "%s", msg);
    if (B < 0 || (B > 0 && (!d_goal_pos_quat || !d_seed || !d_solution || !d_status)))
        return fail(PIKAMD_EINVAL, "bad arguments");
    if (slot < 0 || slot >= PIKAMD_MAX_SLOTS) return fail(PIKAMD_EINVAL, "slot out of range");
    HIP_TRY(hipSetDevice(s->device));
    pik::SolveArgs a;
    std::memset(&a, 0, sizeof a);
    a.B = B;
    a.goal = d_goal_pos_quat;
    a.seed = d_seed;
    a.rng_seed = rng_seed;
    a.problem_offset = problem_offset;
    a.solution = d_solution;
    a.status = d_status;
    a.cost = d_final_cost;
    a.stats = reinterpret_cast<pik::StatsK*>(d_stats);
    PIK_DISPATCH_D(s->chain.dof, return launch_solve<D>(s, p, pk, a, (hipStream_t)stream, slot, s->latency_mode));
    return 0;
}

int32_t pikamd_reserve(pikamd_solver* s, const pikamd_params* p, int64_t B, int32_t slot, void* stream) {
    if (int rc = check_solver(s)) return rc;
    pik::ParamsK pk;
    if (const char* msg = pik::make_params_k(p, pk)) return fail(PIKAMD_EINVAL, "%s", msg);
    if (B < 0) return fail(PIKAMD_EINVAL, "bad arguments");
    if (slot < 0 || slot >= PIKAMD_MAX_SLOTS) return fail(PIKAMD_EINVAL, "slot out of range");
    if (B == 0) return 0;
    HIP_TRY(hipSetDevice(s->device));
    pik::SolveArgs a;
    std::memset(&a, 0, sizeof a);
    a.B = B;
    PIK_DISPATCH_D(s->chain.dof, return launch_solve<D>(s, p, pk, a, (hipStream_t)stream, slot, false, true));
    return 0;
}

int32_t pikamd_solve_batch(pikamd_solver* s, const pikamd_params* p, int64_t B,
                           const double* goal_pos_quat, const double* seed, uint64_t rng_seed,
                           int64_t problem_offset, double* solution, int32_t* status,
                           double* final_cost, pikamd_stats* stats) {
    if (int rc = check_solver(s)) return rc;
    if (B < 0 || (B > 0 && (!goal_pos_quat || !seed || !solution || !status)))
        return fail(PIKAMD_EINVAL, "bad arguments");
    if (B == 0) {
        pik::ParamsK pk;
        if (const char* msg = pik::make_params_k(p, pk)) return fail(PIKAMD_EINVAL, "%s", msg);
        return 0;
    }
    HIP_TRY(hipSetDevice(s->device));
    const size_t d = (size_t)s->chain.dof, N = (size_t)B;
    const size_t sz[6] = {sizeof(double) * 7 * (size_t)s->n_tips * N, sizeof(double) * d * N, sizeof(double) * d * N,
                          sizeof(int) * N,        sizeof(double) * N,     sizeof(pikamd_stats) * N};
    for (int i = 0; i < 6; ++i)
        if (int rc = s->stage[i].ensure(sz[i])) return rc;
    HIP_TRY(hipMemcpy(s->stage[0].p, goal_pos_quat, sz[0], hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->stage[1].p, seed, sz[1], hipMemcpyHostToDevice));
    s->latency_mode = true;
    const int rc_solve = pikamd_solve_batch_device(s, p, B, (const double*)s->stage[0].p, (const double*)s->stage[1].p,
                                                   rng_seed, problem_offset, (double*)s->stage[2].p,
                                                   (int32_t*)s->stage[3].p, (double*)s->stage[4].p,
                                                   (pikamd_stats*)s->stage[5].p, nullptr, 0);
    s->latency_mode = false;
    if (rc_solve) return rc_solve;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(solution, s->stage[2].p, sz[2], hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(status, s->stage[3].p, sz[3], hipMemcpyDeviceToHost));
    if (final_cost) HIP_TRY(hipMemcpy(final_cost, s->stage[4].p, sz[4], hipMemcpyDeviceToHost));
    if (stats) HIP_TRY(hipMemcpy(stats, s->stage[5].p, sz[5], hipMemcpyDeviceToHost));
    return 0;
}

const char* pikamd_kernel_name(const pikamd_solver* s, const pikamd_params* p) {
    if (!s || !p) return "";
    pikamd_solver* m = const_cast<pikamd_solver*>(s);
    snprintf(m->kernel_name, sizeof m->kernel_name, "%s<%d>",
             p->mode == 1 ? "ik_gradient_kernel" : "memetic_kernel", s->chain.dof);
    return m->kernel_name;
}

} // extern "C"
