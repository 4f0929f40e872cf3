// pik_amd.hip -- C ABI (include/pick_ik_amd.h) of the gfx950 solver library.
//
// Host side of the drop-in boundary: model extraction (pik_host.hpp), handle life cycle, staging
// for the host-pointer entry points.  The kernels and their launches live in one translation unit
// per chain length (pik_inst.hip -> pik_launch.hpp), reached through pik::launch_ops(dof).
// No CPU compute path exists here: without a HIP device every entry point fails with
// PIKAMD_ENODEVICE.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "pik_solver.hpp"
#include "pik_urdf.hpp"

namespace pik {
char* error_buffer() {
    static thread_local char buf[ERROR_BUFFER_SIZE] = "";
    return buf;
}
} // namespace pik

using pik::fail;

#if !defined(PIK_STRICT)
// The EXACT kernels inside the product library: the per-length objects of the exact flavour with fused
// multiply-adds (-DPIK_STRICT -DPIK_EXACT_FMA, namespace pik_exact: MoveIt's chain product, the literal 2 dof + 3
// cost evaluations per gradient step with the accept evaluation's work re-used, IEEE square roots and divisions
// -- bit-identical to the oracle's math mode "fma") are linked in as well.  They solve the chains the
// Denavit-Hartenberg kernels cannot express (a floating joint) and every call of a handle whose option
// `arithmetic` is `exact`.  All flavours are compiled from the same headers, so the handle, parameter and
// batch-record layouts are the same types under several namespace names; the launch tables are reached through
// their mangled names.
namespace pik_exact {
char* error_buffer() { return ::pik::error_buffer(); }
#define PIK_LITERAL_OPS(N) const void* launch_ops_d##N();
PIK_LITERAL_OPS(1) PIK_LITERAL_OPS(2) PIK_LITERAL_OPS(3) PIK_LITERAL_OPS(4) PIK_LITERAL_OPS(5) PIK_LITERAL_OPS(6)
PIK_LITERAL_OPS(7) PIK_LITERAL_OPS(8) PIK_LITERAL_OPS(9) PIK_LITERAL_OPS(10) PIK_LITERAL_OPS(11) PIK_LITERAL_OPS(12)
PIK_LITERAL_OPS(13) PIK_LITERAL_OPS(14) PIK_LITERAL_OPS(15) PIK_LITERAL_OPS(16)
#undef PIK_LITERAL_OPS
} // namespace pik_exact
// ... and the kernels specialised for the common configuration (flavour -DPIK_COMMON=1, namespace pik_common;
// pik_math.hpp says what that is and what it buys)
namespace pik_common {
char* error_buffer() { return ::pik::error_buffer(); }
#define PIK_COMMON_OPS(N) const void* launch_ops_d##N();
PIK_COMMON_OPS(1) PIK_COMMON_OPS(2) PIK_COMMON_OPS(3) PIK_COMMON_OPS(4) PIK_COMMON_OPS(5) PIK_COMMON_OPS(6)
PIK_COMMON_OPS(7) PIK_COMMON_OPS(8) PIK_COMMON_OPS(9) PIK_COMMON_OPS(10) PIK_COMMON_OPS(11) PIK_COMMON_OPS(12)
PIK_COMMON_OPS(13) PIK_COMMON_OPS(14) PIK_COMMON_OPS(15) PIK_COMMON_OPS(16)
#undef PIK_COMMON_OPS
} // namespace pik_common
// ... and the same with the joint goals left in (-DPIK_NO_GOALS=0): BASELINE config 3's kind of call
namespace pik_common_goals {
char* error_buffer() { return ::pik::error_buffer(); }
#define PIK_COMMON_OPS(N) const void* launch_ops_d##N();
PIK_COMMON_OPS(1) PIK_COMMON_OPS(2) PIK_COMMON_OPS(3) PIK_COMMON_OPS(4) PIK_COMMON_OPS(5) PIK_COMMON_OPS(6)
PIK_COMMON_OPS(7) PIK_COMMON_OPS(8) PIK_COMMON_OPS(9) PIK_COMMON_OPS(10) PIK_COMMON_OPS(11) PIK_COMMON_OPS(12)
PIK_COMMON_OPS(13) PIK_COMMON_OPS(14) PIK_COMMON_OPS(15) PIK_COMMON_OPS(16)
#undef PIK_COMMON_OPS
} // namespace pik_common_goals
#endif

namespace {

#if !defined(PIK_STRICT)
const pik::LaunchOps* literal_ops(int dof) {
    const void* p = nullptr;
    switch (dof) {
#define PIK_LITERAL_CASE(N) case N: p = pik_exact::launch_ops_d##N(); break;
        PIK_LITERAL_CASE(1) PIK_LITERAL_CASE(2) PIK_LITERAL_CASE(3) PIK_LITERAL_CASE(4) PIK_LITERAL_CASE(5)
        PIK_LITERAL_CASE(6) PIK_LITERAL_CASE(7) PIK_LITERAL_CASE(8) PIK_LITERAL_CASE(9) PIK_LITERAL_CASE(10)
        PIK_LITERAL_CASE(11) PIK_LITERAL_CASE(12) PIK_LITERAL_CASE(13) PIK_LITERAL_CASE(14) PIK_LITERAL_CASE(15)
        PIK_LITERAL_CASE(16)
#undef PIK_LITERAL_CASE
        default: break;
    }
    return static_cast<const pik::LaunchOps*>(p);
}
#endif

#if !defined(PIK_STRICT)
const pik::LaunchOps* common_ops(int dof, bool goals = false) {
    const void* p = nullptr;
    switch (dof) {
#define PIK_COMMON_CASE(N) case N: p = goals ? pik_common_goals::launch_ops_d##N() : pik_common::launch_ops_d##N(); break;
        PIK_COMMON_CASE(1) PIK_COMMON_CASE(2) PIK_COMMON_CASE(3) PIK_COMMON_CASE(4) PIK_COMMON_CASE(5)
        PIK_COMMON_CASE(6) PIK_COMMON_CASE(7) PIK_COMMON_CASE(8) PIK_COMMON_CASE(9) PIK_COMMON_CASE(10)
        PIK_COMMON_CASE(11) PIK_COMMON_CASE(12) PIK_COMMON_CASE(13) PIK_COMMON_CASE(14) PIK_COMMON_CASE(15)
        PIK_COMMON_CASE(16)
#undef PIK_COMMON_CASE
        default: break;
    }
    return static_cast<const pik::LaunchOps*>(p);
}
#endif

int check_solver(const pikamd_solver* s) {
    if (!s) return fail(PIKAMD_EINVAL, "solver handle is NULL");
    return 0;
}

// the kernels of a handle: Denavit-Hartenberg (product arithmetic) unless the chain needs the literal ones
[[maybe_unused]] bool needs_literal(const pikamd_solver* s) {
    bool f = s->opt.exact || s->chain.float_mask != 0u || s->chain.n_mimic != 0; // (option arithmetic = exact, a floating or a mimic joint)
    for (int k = 1; k < s->n_tips; ++k) f = f || s->more[k - 1].float_mask != 0u || s->more[k - 1].n_mimic != 0;
    return f;
}
const pik::LaunchOps* ops_of(const pikamd_solver* s) {
#if !defined(PIK_STRICT)
    if (needs_literal(s)) return literal_ops(s->chain.dof);
#endif
    return pik::launch_ops(s->chain.dof);
}

// Does this call have the common configuration (pik_math.hpp PIK_COMMON)?  Chain: every variable a bounded
// revolute joint, no general Denavit-Hartenberg step -- on every tip's path; parameters: both pose-cost terms
// on, the line-search angle addition applicable, and for the memetic solver four elites and one species.  (Two
// flavours of it: joint goals compiled out -- the default parameters -- and left in.)
[[maybe_unused]] bool common_eligible(const pikamd_solver* s, const pikamd_params* p, const pik::ParamsK& pk) {
    if (!s->opt.specialised || needs_literal(s)) return false;
    const uint32_t all = (s->chain.dof >= 32) ? ~0u : ((1u << s->chain.dof) - 1u);
    auto chain_ok = [&](const pik::ChainHost& c) {
        return c.prismatic_mask == 0u && c.dh_general_mask == 0u && c.bounded_mask == all;
    };
    if (!chain_ok(s->chain)) return false;
    for (int k = 1; k < s->n_tips; ++k)
        if (!chain_ok(s->more[k - 1])) return false;
    // (a joint goal enabled: the flavour with the goals left in, see solve_ops_of)
    if (pk.line_delta == 0 || !(pk.pos_scale > 0.0) || !(pk.rot_scale > 0.0)) return false;
    if (p->mode == 0 && (pk.elites != 4 || p->memetic_num_threads > 1)) return false;
    return true;
}
// the kernels of one solve call
const pik::LaunchOps* solve_ops_of(const pikamd_solver* s, const pikamd_params* p, const pik::ParamsK& pk) {
#if !defined(PIK_STRICT)
    if (common_eligible(s, p, pk))
        if (const pik::LaunchOps* o = common_ops(s->chain.dof, pk.goal_mask != 0)) return o;
#else
    (void)p;
    (void)pk;
#endif
    return ops_of(s);
}

int no_kernels(int dof) {
    return fail(PIKAMD_EUNSUPPORTED, "dof %d: kernels are instantiated for 1..16", dof);
}

size_t align8(size_t v) { return (v + 7) & ~(size_t)7; }

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

const char* pikamd_last_error(void) { return pik::error_buffer(); }

const char* pikamd_version(void) { return "pick_ik_amd 0.3.0 (gfx950)"; }

static int32_t create_solver(const pik::ChainHost* chains, int n_tips, int32_t device_ordinal,
                             pikamd_solver** out) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
        return fail(PIKAMD_ENODEVICE, "no HIP device available (this library has no CPU path)");
    if (device_ordinal < 0 || device_ordinal >= count)
        return fail(PIKAMD_EINVAL, "device_ordinal %d out of range [0, %d)", device_ordinal, count);
    if (!pik::launch_ops(chains[0].dof)) return no_kernels(chains[0].dof);
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
    const size_t table_bytes = sizeof(pik::BatchRecord) * PIKAMD_MAX_BATCHES * pik::TABLE_RING;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->counters), pik::COUNTER_BLOCK * pik::N_SLOTS);
    if (e == hipSuccess) e = hipMemset(s->counters, 0, pik::COUNTER_BLOCK * pik::N_SLOTS);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->consts_dev), pik::CONSTS_STRIDE * pik::N_SLOTS);
    if (e == hipSuccess)
        e = hipHostMalloc(reinterpret_cast<void**>(&s->consts_host), pik::CONSTS_STRIDE * pik::N_SLOTS, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->tables_dev), table_bytes);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&s->tables_host), table_bytes, hipHostMallocDefault);
    for (int i = 0; i < pik::TABLE_RING && e == hipSuccess; ++i)
        e = hipEventCreateWithFlags(&s->table_event[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        pikamd_destroy(s);
        return fail(PIKAMD_EHIP, "device allocation failed: %s", hipGetErrorString(e));
    }
    for (auto& j : s->jobs) j.host.pinned_host = true;
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

int32_t pikamd_urdf_extract(const char* urdf_xml, const char* base_link, const char* const* tip_links,
                            int32_t n_tips, pikamd_urdf_model* out) {
    if (!out) return fail(PIKAMD_EINVAL, "out is NULL");
    const std::string err = pik::urdf::extract(urdf_xml, base_link, tip_links, n_tips, *out);
    if (!err.empty()) return fail(PIKAMD_EINVAL, "%s", err.c_str());
    return 0;
}

int32_t pikamd_create_from_urdf(const char* urdf_xml, const char* base_link, const char* const* tip_links,
                                int32_t n_tips, int32_t device_ordinal, pikamd_solver** out) {
    if (!out) return fail(PIKAMD_EINVAL, "out is NULL");
    *out = nullptr;
    auto m = std::make_unique<pikamd_urdf_model>();
    if (int rc = pikamd_urdf_extract(urdf_xml, base_link, tip_links, n_tips, m.get())) return rc;
    if (m->n_tips == 1) {
        const auto& t = m->tips[0];
        pikamd_chain c{m->dof, &t.origin_xyz_rpy[0][0], &t.axis[0][0], t.joint_type, t.tip_xyz_rpy,
                       m->qmin,  m->qmax,               m->vmax,       m->bounded};
        if (int rc = pikamd_create(&c, device_ordinal, out)) return rc;
        if (m->n_mimic)
            if (int rc = pikamd_set_mimic_joints(*out, m->n_mimic, m->mimic)) {
                pikamd_destroy(*out);
                *out = nullptr;
                return rc;
            }
        return 0;
    }
    pikamd_tip tips[PIKAMD_MAX_TIPS];
    for (int k = 0; k < m->n_tips; ++k) {
        const auto& t = m->tips[k];
        tips[k] = pikamd_tip{t.n_joints, t.variable, &t.origin_xyz_rpy[0][0], &t.axis[0][0], t.joint_type, t.tip_xyz_rpy};
    }
    pikamd_multi_chain c{m->dof, m->n_tips, tips, m->qmin, m->qmax, m->vmax, m->bounded};
    if (int rc = pikamd_create_multi(&c, device_ordinal, out)) return rc;
    if (m->n_mimic)
        if (int rc = pikamd_set_mimic_joints(*out, m->n_mimic, m->mimic)) {
            pikamd_destroy(*out);
            *out = nullptr;
            return rc;
        }
    return 0;
}

int32_t pikamd_n_tips(const pikamd_solver* s) { return s ? s->n_tips : 0; }

int32_t pikamd_set_mimic_joints(pikamd_solver* s, int32_t n, const pikamd_mimic_joint* joints) {
    if (int rc = check_solver(s)) return rc;
    if (n < 0 || (n > 0 && !joints)) return fail(PIKAMD_EINVAL, "bad arguments");
    pik::ChainHost saved[PIKAMD_MAX_TIPS];
    for (int k = 0; k < s->n_tips; ++k) saved[k] = k == 0 ? s->chain : s->more[k - 1];
    auto path = [&](int k) -> pik::ChainHost& { return k == 0 ? s->chain : s->more[k - 1]; };
    for (int k = 0; k < s->n_tips; ++k) pik::clear_mimic_joints(path(k));
    for (int i = 0; i < n; ++i) {
        const pikamd_mimic_joint& m = joints[i];
        const char* msg = (m.tip < 0 || m.tip >= s->n_tips) ? "mimic joint: tip out of range" : pik::add_mimic_joint(path(m.tip), m);
        if (!msg && s->n_tips > 1) { // the variables it names must lie on that tip's path
            const uint32_t act = path(m.tip).active_mask;
            if (!((act >> m.master_variable) & 1u) || (m.after_variable >= 0 && !((act >> m.after_variable) & 1u)))
                msg = "mimic joint: master_variable / after_variable is not on that tip's path";
        }
        if (msg) {
            for (int k = 0; k < s->n_tips; ++k) path(k) = saved[k];
            return fail(PIKAMD_EINVAL, "%s", msg);
        }
    }
    // a different chain: what the self test found out about the old one no longer holds
    s->self_tested.clear();
    s->opt.disabled_lanes = 0;
    s->opt.disabled_lanes_exact = 0;
    return 0;
}

void pikamd_destroy(pikamd_solver* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    if (s->counters) (void)hipFree(s->counters);
    if (s->consts_dev) (void)hipFree(s->consts_dev);
    if (s->consts_host) (void)hipHostFree(s->consts_host);
    if (s->tables_dev) (void)hipFree(s->tables_dev);
    if (s->tables_host) (void)hipHostFree(s->tables_host);
    for (auto& ev : s->table_event)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : s->slot_event)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& b : s->stage) b.release();
    for (auto& b : s->slot_state) b.release();
    for (auto& b : s->slot_soa) b.release();
    for (auto& j : s->jobs) {
        j.dev.release();
        j.host.release();
        if (j.stream) (void)hipStreamDestroy(j.stream);
    }
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
    return ops_of(s)->fk(s, n, d_q, d_pos_quat, (hipStream_t)stream);
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
    if (int rc = ops_of(s)->cost(s, pk, n, (const double*)s->stage[0].p, (const double*)s->stage[1].p,
                                 (const double*)s->stage[2].p, cost ? (double*)s->stage[3].p : nullptr,
                                 is_solution ? (int*)s->stage[4].p : nullptr, nullptr))
        return rc;
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
    if (int rc = ops_of(s)->step(s, pk, n, (const double*)s->stage[0].p, (const double*)s->stage[1].p,
                                 (double*)s->stage[2].p, (double*)s->stage[3].p, (double*)s->stage[4].p,
                                 (double*)s->stage[5].p, (double*)s->stage[6].p, (int*)s->stage[7].p, nullptr))
        return rc;
    HIP_TRY(hipMemcpy(local, s->stage[2].p, sz[2], hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(best, s->stage[3].p, sz[3], hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(local_cost, s->stage[4].p, sz[4], hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(best_cost, s->stage[5].p, sz[5], hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(gradient, s->stage[6].p, sz[6], hipMemcpyDeviceToHost));
    if (improved) HIP_TRY(hipMemcpy(improved, s->stage[7].p, sz[7], hipMemcpyDeviceToHost));
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}


// ---- solvers ---------------------------------------------------------------------------------

// validates the records of a call and converts them (device pointers) for the launch
static int make_records(const pikamd_solver* s, int32_t n_batches, const pikamd_batch* batches,
                        pik::BatchRecord* rec, long long* total) {
    if (n_batches < 0 || n_batches > PIKAMD_MAX_BATCHES)
        return fail(PIKAMD_EINVAL, "n_batches %d out of range [0, %d]", n_batches, PIKAMD_MAX_BATCHES);
    if (n_batches > 0 && !batches) return fail(PIKAMD_EINVAL, "batches is NULL");
    long long sum = 0;
    int n = 0;
    for (int k = 0; k < n_batches; ++k) {
        const pikamd_batch& b = batches[k];
        if (b.B < 0 || (b.B > 0 && (!b.goal_pos_quat || !b.seed || !b.solution || !b.status)))
            return fail(PIKAMD_EINVAL, "batch %d: bad arguments", k);
        if (b.B == 0) continue; // (an empty batch takes no part in the call)
        pik::BatchRecord& r = rec[n++];
        r.start = 0;
        r.B = b.B;
        r.goal = b.goal_pos_quat;
        r.seed = b.seed;
        r.guess = b.initial_guess ? b.initial_guess : b.seed;
        r.problem_offset = b.problem_offset;
        r.solution = b.solution;
        r.status = b.status;
        r.cost = b.final_cost;
        r.stats = b.stats;
        r.completed = b.completed;
        sum += b.B;
    }
    (void)s;
    *total = sum;
    return n;
}

// option joint_layout = soa: [dof][B] <-> [B][dof] (180 bytes per problem against ~5 Mflop of solving: the
// kernels keep the one layout they gather a problem's vector from with a single cache line)
__global__ void pik_joint_layout_kernel(const double* __restrict__ in, double* __restrict__ out, long long B, int D,
                                        int to_aos) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    if (to_aos) { // out [B][D] <- in [D][B]
        const long long b = i / D;
        const int j = (int)(i - b * D);
        out[i] = in[(long long)j * B + b];
    } else { // out [D][B] <- in [B][D]
        const long long j = i / B, b = i - j * B;
        out[i] = in[b * D + j];
    }
}

static int maybe_self_test(pikamd_solver* s, const pikamd_params* p);

static int32_t solve_records(pikamd_solver* s, const pikamd_params* p, pik::BatchRecord* rec, int n,
                             uint64_t rng_seed, hipStream_t stream, int slot) {
    pik::ParamsK pk;
    if (const char* msg = pik::make_params_k(p, pk)) return fail(PIKAMD_EINVAL, "%s", msg);
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(s->device));
    if (!s->opt.soa) return solve_ops_of(s, p, pk)->solve(s, p, pk, rec, n, rng_seed, stream, slot, false);
    // joint vectors structure-of-arrays: seed / initial guess are transposed into scratch in front of the
    // kernels, the solutions out of scratch behind them, all on the call's stream
    const int D = s->chain.dof;
    size_t doubles = 0;
    for (int k = 0; k < n; ++k) {
        if (rec[k].completed)
            return fail(PIKAMD_EINVAL, "joint_layout soa: completion counters are not available (the solutions of a "
                                       "batch are in place when the stream has passed the call)");
        doubles += (size_t)rec[k].B * (size_t)D * (rec[k].guess != rec[k].seed ? 3u : 2u);
    }
    if (int rc = s->slot_soa[slot].ensure(sizeof(double) * doubles)) return rc;
    double* w = (double*)s->slot_soa[slot].p;
    double* user_solution[PIKAMD_MAX_BATCHES];
    auto move = [&](const double* in, double* out, long long B, int to_aos) -> int {
        const long long items = B * D;
        hipLaunchKernelGGL(pik_joint_layout_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, in, out, B,
                           D, to_aos);
        HIP_TRY(hipGetLastError());
        return 0;
    };
    for (int k = 0; k < n; ++k) {
        const long long B = rec[k].B;
        const bool own_guess = rec[k].guess != rec[k].seed;
        double* sd = w;
        w += (size_t)B * D;
        if (int rc = move(rec[k].seed, sd, B, 1)) return rc;
        double* gs = sd;
        if (own_guess) {
            gs = w;
            w += (size_t)B * D;
            if (int rc = move(rec[k].guess, gs, B, 1)) return rc;
        }
        user_solution[k] = rec[k].solution;
        rec[k].seed = sd;
        rec[k].guess = gs;
        rec[k].solution = w;
        w += (size_t)B * D;
    }
    if (int rc = solve_ops_of(s, p, pk)->solve(s, p, pk, rec, n, rng_seed, stream, slot, false)) return rc;
    for (int k = 0; k < n; ++k)
        if (int rc = move(rec[k].solution, user_solution[k], rec[k].B, 0)) return rc;
    return 0;
}

int32_t pikamd_solve_batches_device(pikamd_solver* s, const pikamd_params* p, int32_t n_batches,
                                    const pikamd_batch* batches, uint64_t rng_seed, void* stream,
                                    int32_t slot) {
    if (int rc = check_solver(s)) return rc;
    if (slot < 0 || slot >= PIKAMD_MAX_SLOTS) return fail(PIKAMD_EINVAL, "slot out of range");
    // (no automatic self test here: this entry point is stream-ordered and must not block or synchronise -- it may
    //  be under stream capture.  pikamd_reserve, which a caller of this entry point runs first, carries it.)
    pik::BatchRecord rec[PIKAMD_MAX_BATCHES];
    long long total = 0;
    const int n = make_records(s, n_batches, batches, rec, &total);
    if (n < 0) return n;
    return solve_records(s, p, rec, n, rng_seed, (hipStream_t)stream, slot);
}

int32_t pikamd_solve_batch_device(pikamd_solver* s, const pikamd_params* p, int64_t B,
                                  const double* d_goal_pos_quat, const double* d_seed,
                                  uint64_t rng_seed, int64_t problem_offset, double* d_solution,
                                  int32_t* d_status, double* d_final_cost, pikamd_stats* d_stats,
                                  void* stream, int32_t slot) {
    const pikamd_batch b = {B,          d_goal_pos_quat, d_seed,       nullptr, problem_offset,
                            d_solution, d_status,        d_final_cost, d_stats, nullptr};
    return pikamd_solve_batches_device(s, p, 1, &b, rng_seed, stream, slot);
}

int32_t pikamd_reserve(pikamd_solver* s, const pikamd_params* p, int64_t B, int32_t slot, void* stream) {
    if (int rc = check_solver(s)) return rc;
    pik::ParamsK pk;
    if (const char* msg = pik::make_params_k(p, pk)) return fail(PIKAMD_EINVAL, "%s", msg);
    if (B < 0) return fail(PIKAMD_EINVAL, "bad arguments");
    if (slot < 0 || slot >= PIKAMD_MAX_SLOTS) return fail(PIKAMD_EINVAL, "slot out of range");
    if (B == 0) return 0;
    if (int rc = maybe_self_test(s, p)) return rc; // (here rather than inside the first timed call)
    HIP_TRY(hipSetDevice(s->device));
    // option joint_layout = soa: the [B][dof] copies of seed, initial guess and solution the kernels work on
    // (sized here so that no enqueue path allocates -- a grow frees, and hipFree synchronises the device)
    if (s->opt.soa)
        if (int rc = s->slot_soa[slot].ensure(sizeof(double) * (size_t)B * (size_t)s->chain.dof * 3u)) return rc;
    pik::BatchRecord rec;
    std::memset(&rec, 0, sizeof rec);
    rec.B = B;
    return solve_ops_of(s, p, pk)->solve(s, p, pk, &rec, 1, 0, (hipStream_t)stream, slot, true);
}

// Host-pointer jobs: inputs -> pinned staging -> one H2D copy, kernels, one D2H copy into pinned
// staging, all on the job's own stream; pikamd_wait copies the results out.  Jobs on different
// slots overlap their transfers and kernels.
static int32_t start_job(pikamd_solver* s, const pikamd_params* p, int32_t n_batches,
                         const pikamd_batch* batches, uint64_t rng_seed, int job) {
    if (int rc = check_solver(s)) return rc;
    if (job < 0 || job > pik::JOB_SELF_TEST) return fail(PIKAMD_EINVAL, "job out of range");
    if (job != pik::JOB_SELF_TEST)
        if (int rc = maybe_self_test(s, p)) return rc;
    pik::HostJob& J = s->jobs[job];
    if (J.pending) return fail(PIKAMD_EINVAL, "job %d is still in flight: call pikamd_wait first", job);
    pik::BatchRecord rec[PIKAMD_MAX_BATCHES];
    long long total = 0;
    const int n = make_records(s, n_batches, batches, rec, &total);
    if (n < 0) return n;
    {
        pik::ParamsK pk;
        if (const char* msg = pik::make_params_k(p, pk)) return fail(PIKAMD_EINVAL, "%s", msg);
    }
    J.n_batches = 0;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(s->device));
    if (!J.stream) HIP_TRY(hipStreamCreateWithFlags(&J.stream, hipStreamNonBlocking));
    const size_t d = (size_t)s->chain.dof, g7 = 7 * (size_t)s->n_tips;
    // layout: [inputs of every batch][outputs of every batch], 8-byte aligned pieces
    size_t off = 0;
    size_t in_goal[PIKAMD_MAX_BATCHES], in_seed[PIKAMD_MAX_BATCHES], in_guess[PIKAMD_MAX_BATCHES];
    for (int k = 0; k < n; ++k) {
        const size_t N = (size_t)rec[k].B;
        in_goal[k] = off;
        off += sizeof(double) * g7 * N;
        in_seed[k] = off;
        off += sizeof(double) * d * N;
        in_guess[k] = (rec[k].guess != rec[k].seed) ? off : in_seed[k];
        if (rec[k].guess != rec[k].seed) off += sizeof(double) * d * N;
    }
    J.in_bytes = off;
    for (int k = 0; k < n; ++k) {
        const size_t N = (size_t)rec[k].B;
        pik::HostJob::Out& o = J.outs[k];
        o.B = rec[k].B;
        o.solution = rec[k].solution;
        o.status = rec[k].status;
        o.final_cost = rec[k].cost;
        o.stats = (pikamd_stats*)rec[k].stats;
        o.off_solution = off;
        off += sizeof(double) * d * N;
        o.off_cost = off;
        off += sizeof(double) * N;
        o.off_stats = off;
        off += sizeof(pikamd_stats) * N;
        o.off_status = off;
        off += align8(sizeof(int32_t) * N);
    }
    J.out_bytes = off - J.in_bytes;
    if (int rc = J.dev.ensure(off)) return rc;
    if (int rc = J.host.ensure(off)) return rc;
    char* hb = (char*)J.host.p;
    char* db = (char*)J.dev.p;
    for (int k = 0; k < n; ++k) {
        const size_t N = (size_t)rec[k].B;
        std::memcpy(hb + in_goal[k], rec[k].goal, sizeof(double) * g7 * N);
        std::memcpy(hb + in_seed[k], rec[k].seed, sizeof(double) * d * N);
        if (in_guess[k] != in_seed[k]) std::memcpy(hb + in_guess[k], rec[k].guess, sizeof(double) * d * N);
        rec[k].goal = (const double*)(db + in_goal[k]);
        rec[k].seed = (const double*)(db + in_seed[k]);
        rec[k].guess = (const double*)(db + in_guess[k]);
        rec[k].solution = (double*)(db + J.outs[k].off_solution);
        rec[k].status = (int*)(db + J.outs[k].off_status);
        rec[k].cost = (double*)(db + J.outs[k].off_cost);
        rec[k].stats = (void*)(db + J.outs[k].off_stats);
        rec[k].completed = nullptr;
    }
    HIP_TRY(hipMemcpyAsync(db, hb, J.in_bytes, hipMemcpyHostToDevice, J.stream));
    // From here on work of this job is (or may be) in flight on its stream.  If a later enqueue fails, the
    // stream is drained before the error is returned: the job stays not-pending, and the next job must not
    // write into staging memory a copy is still reading, nor find kernels of this one still running.
    if (int rc = solve_records(s, p, rec, n, rng_seed, J.stream, pik::N_DEVICE_SLOTS + job)) {
        (void)hipStreamSynchronize(J.stream);
        return rc;
    }
    {
        const hipError_t e = hipMemcpyAsync(hb + J.in_bytes, db + J.in_bytes, J.out_bytes, hipMemcpyDeviceToHost, J.stream);
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(J.stream);
            return fail(PIKAMD_EHIP, "result copy could not be enqueued: %s", hipGetErrorString(e));
        }
    }
    J.n_batches = n;
    J.dof = (int)d;
    J.pending = true;
    return 0;
}

int32_t pikamd_solve_batches_async(pikamd_solver* s, const pikamd_params* p, int32_t n_batches,
                                   const pikamd_batch* batches, uint64_t rng_seed, int32_t job) {
    if (job < 0 || job >= PIKAMD_MAX_HOST_JOBS - 1) // (the last job is the synchronous entry points')
        return fail(PIKAMD_EINVAL, "job out of range [0, %d)", PIKAMD_MAX_HOST_JOBS - 1);
    return start_job(s, p, n_batches, batches, rng_seed, job);
}

static int32_t wait_job(pikamd_solver* s, int job) {
    pik::HostJob& J = s->jobs[job];
    if (!J.pending) return 0;
    HIP_TRY(hipSetDevice(s->device));
    // A failed synchronise ends the job: its results are undefined and are NOT copied out, the error is
    // returned once, and the job index is free again (left pending, every later call on it failed for the life
    // of the handle).
    {
        const hipError_t e = hipStreamSynchronize(J.stream);
        J.pending = false;
        if (e != hipSuccess) return fail(PIKAMD_EHIP, "job %d: %s (its results are lost)", job, hipGetErrorString(e));
    }
    const char* hb = (const char*)J.host.p;
    const size_t d = (size_t)J.dof;
    for (int k = 0; k < J.n_batches; ++k) {
        const pik::HostJob::Out& o = J.outs[k];
        const size_t N = (size_t)o.B;
        std::memcpy(o.solution, hb + o.off_solution, sizeof(double) * d * N);
        std::memcpy(o.status, hb + o.off_status, sizeof(int32_t) * N);
        if (o.final_cost) std::memcpy(o.final_cost, hb + o.off_cost, sizeof(double) * N);
        if (o.stats) std::memcpy(o.stats, hb + o.off_stats, sizeof(pikamd_stats) * N);
    }
    return 0;
}

int32_t pikamd_wait(pikamd_solver* s, int32_t job) {
    if (int rc = check_solver(s)) return rc;
    if (job < 0 || job >= pik::N_HOST_JOBS) return fail(PIKAMD_EINVAL, "job out of range");
    return wait_job(s, job);
}

int32_t pikamd_solve_batches(pikamd_solver* s, const pikamd_params* p, int32_t n_batches,
                             const pikamd_batch* batches, uint64_t rng_seed) {
    const int job = PIKAMD_MAX_HOST_JOBS - 1;
    if (int rc = start_job(s, p, n_batches, batches, rng_seed, job)) return rc;
    return pikamd_wait(s, job);
}

int32_t pikamd_solve_batch(pikamd_solver* s, const pikamd_params* p, int64_t B,
                           const double* goal_pos_quat, const double* seed, uint64_t rng_seed,
                           int64_t problem_offset, double* solution, int32_t* status,
                           double* final_cost, pikamd_stats* stats) {
    const pikamd_batch b = {B,        goal_pos_quat, seed,       nullptr, problem_offset,
                            solution, status,        final_cost, stats,   nullptr};
    const int job = PIKAMD_MAX_HOST_JOBS - 1;
    // (a call with nothing else in flight takes the latency-greedy kernel variants: launch_solve)
    if (int rc = start_job(s, p, 1, &b, rng_seed, job)) return rc;
    return pikamd_wait(s, job);
}

int32_t pikamd_set_option(pikamd_solver* s, const char* name, const char* value) {
    if (int rc = check_solver(s)) return rc;
    if (!name) return fail(PIKAMD_EINVAL, "option name is NULL");
    const std::string n = name, v = value ? value : "";
    pik::SolverOptions& o = s->opt;
    // "a,b,c" -> ints; false on anything that is not a number
    auto ints = [](const std::string& text, char sep, std::vector<int>& out) {
        size_t i = 0;
        while (i < text.size()) {
            char* end = nullptr;
            const long x = std::strtol(text.c_str() + i, &end, 10);
            if (end == text.c_str() + i) return false;
            out.push_back((int)x);
            i = (size_t)(end - text.c_str());
            if (i < text.size()) {
                if (text[i] != sep) return false;
                ++i;
            }
        }
        return true;
    };
    if (n == "lanes_per_elite") {
        if (v.empty()) { o.lpe = 0; return 0; }
        std::vector<int> x;
        if (!ints(v, ',', x) || x.size() != 1 || !(x[0] == 0 || x[0] == 1 || x[0] == 2 || x[0] == 4 || x[0] == 8 || x[0] == 16))
            return fail(PIKAMD_EINVAL, "lanes_per_elite: expected 0 (adaptive), 1, 2, 4, 8 or 16, got '%s'", v.c_str());
        o.lpe = x[0];
        return 0;
    }
    if (n == "lanes_per_elite_schedule") { // "g0:l0,g1:l1,..." ascending generations, the first one 0
        if (v.empty()) { o.n_sched = 0; return 0; }
        int cnt = 0, from[4], of[4];
        size_t i = 0;
        while (i < v.size()) {
            const size_t comma = v.find(',', i), colon = v.find(':', i);
            const size_t end = comma == std::string::npos ? v.size() : comma;
            std::vector<int> a, b;
            if (cnt >= 4 || colon == std::string::npos || colon > end || !ints(v.substr(i, colon - i), ',', a) ||
                !ints(v.substr(colon + 1, end - colon - 1), ',', b) || a.size() != 1 || b.size() != 1 ||
                !(b[0] == 1 || b[0] == 2 || b[0] == 4 || b[0] == 8 || b[0] == 16) ||
                (cnt == 0 ? a[0] != 0 : a[0] <= from[cnt - 1]))
                return fail(PIKAMD_EINVAL, "lanes_per_elite_schedule: expected 'g0:l0,g1:l1,...' with g0 = 0, ascending "
                                           "generations and lanes in {1,2,4,8,16} (at most 4 entries), got '%s'", v.c_str());
            from[cnt] = a[0];
            of[cnt] = b[0];
            ++cnt;
            i = end + (end < v.size() ? 1 : 0);
        }
        o.n_sched = cnt;
        for (int k = 0; k < cnt; ++k) o.sched_from[k] = from[k], o.sched_of[k] = of[k];
        return 0;
    }
    if (n == "passes") { // "2,4,8" generation marks, "none", or "" = the default marks
        if (v.empty()) { o.passes_set = false; return 0; }
        std::vector<int> x;
        if (v != "none" && (!ints(v, ',', x) || x.size() > 15))
            return fail(PIKAMD_EINVAL, "passes: expected 'none' or up to 15 ascending generation marks, got '%s'", v.c_str());
        for (size_t k = 0; k < x.size(); ++k)
            if (x[k] <= 0 || (k > 0 && x[k] <= x[k - 1]))
                return fail(PIKAMD_EINVAL, "passes: marks must be positive and ascending, got '%s'", v.c_str());
        o.passes_set = true;
        o.n_marks = (int)x.size();
        for (size_t k = 0; k < x.size(); ++k) o.marks[k] = x[k];
        return 0;
    }
    if (n == "two_per_simd") {
        if (v.empty()) { o.two_per_simd = -1; return 0; }
        std::vector<int> x;
        if (!ints(v, ',', x) || x.size() != 1 || x[0] < 0)
            return fail(PIKAMD_EINVAL, "two_per_simd: expected 0, 1 or a wavefront threshold, got '%s'", v.c_str());
        o.two_per_simd = x[0];
        return 0;
    }
    if (n == "arithmetic") { // "exact" (default): the exact kernels (oracle math mode "fma"); "fast": the Denavit-Hartenberg kernels (opt-in)
#if defined(PIK_STRICT)
        if (v.empty() || v == "exact") return 0; // (the verification library is exact anyway: oracle math mode "portable")
        return fail(PIKAMD_EINVAL, "arithmetic: the verification library only has 'exact', got '%s'", v.c_str());
#else
        if (v.empty() || v == "exact") { o.exact = true; return 0; }
        if (v == "fast") { o.exact = false; return 0; }
        return fail(PIKAMD_EINVAL, "arithmetic: expected 'fast' or 'exact', got '%s'", v.c_str());
#endif
    }
    if (n == "self_test") { // "auto" (default): see maybe_self_test; "off": only when pikamd_self_test is called
        if (v.empty() || v == "auto") { o.auto_self_test = true; return 0; }
        if (v == "off") { o.auto_self_test = false; return 0; }
        return fail(PIKAMD_EINVAL, "self_test: expected 'auto' or 'off', got '%s'", v.c_str());
    }
    if (n == "specialised") { // "1" (default): the common-configuration kernels for calls that qualify; "0": never
        if (v.empty() || v == "1") { o.specialised = true; return 0; }
        if (v == "0") { o.specialised = false; return 0; }
        return fail(PIKAMD_EINVAL, "specialised: expected '0' or '1', got '%s'", v.c_str());
    }
    if (n == "shard_chunks") { // pikamd_solve_batch_sharded: host jobs per device ("" = default)
        if (v.empty()) { o.shard_chunks = 0; return 0; }
        std::vector<int> x;
        if (!ints(v, ',', x) || x.size() != 1 || x[0] < 1 || x[0] > 8)
            return fail(PIKAMD_EINVAL, "shard_chunks: expected 1..8, got '%s'", v.c_str());
        o.shard_chunks = x[0];
        return 0;
    }
    if (n == "joint_layout") { // "aos" (default): seed / initial_guess / solution are [B][dof]; "soa": [dof][B]
        if (v.empty() || v == "aos") { o.soa = false; return 0; }
        if (v == "soa") { o.soa = true; return 0; }
        return fail(PIKAMD_EINVAL, "joint_layout: expected 'aos' or 'soa', got '%s'", v.c_str());
    }
    if (n == "host_max_time" || n == "host_gd_max_time") { // seconds; "" or "0" = no limit
        double x = 0.0;
        if (!v.empty()) {
            char* end = nullptr;
            x = std::strtod(v.c_str(), &end);
            if (end == v.c_str() || *end != 0 || !(x >= 0.0) || !std::isfinite(x))
                return fail(PIKAMD_EINVAL, "%s: expected a time in seconds >= 0, got '%s'", n.c_str(), v.c_str());
        }
        (n == "host_max_time" ? o.host_max_time : o.host_gd_max_time) = x;
        return 0;
    }
    if (n == "regime") {
        if (v.empty() || v == "adaptive") { o.regime = 0; return 0; }
        if (v == "latency") { o.regime = 1; return 0; }
        if (v == "throughput") { o.regime = 2; return 0; }
        return fail(PIKAMD_EINVAL, "regime: expected 'adaptive', 'latency' or 'throughput', got '%s'", v.c_str());
    }
    return fail(PIKAMD_EINVAL, "unknown option '%s'", n.c_str());
}

// ---- several devices ----------------------------------------------------------------------------

void pikamd_shard_bounds(int64_t total, int32_t rank, int32_t world, int64_t* lo, int64_t* hi) {
    if (world < 1) world = 1;
    if (total < 0) total = 0;
    const int64_t base = total / world, rem = total % world;
    const int64_t l = (int64_t)rank * base + (rank < rem ? rank : rem);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (rank < rem ? 1 : 0);
}

int32_t pikamd_solve_batch_sharded(pikamd_solver* const* solvers, int32_t n_devices, const pikamd_params* p,
                                   int64_t B, const double* goal_pos_quat, const double* seed,
                                   const double* initial_guess, uint64_t rng_seed, int64_t problem_offset,
                                   double* solution, int32_t* status, double* final_cost, pikamd_stats* stats) {
    if (!solvers || n_devices < 1) return fail(PIKAMD_EINVAL, "no solver handles");
    if (B < 0 || (B > 0 && (!goal_pos_quat || !seed || !solution || !status))) return fail(PIKAMD_EINVAL, "bad arguments");
    for (int r = 0; r < n_devices; ++r) {
        if (!solvers[r]) return fail(PIKAMD_EINVAL, "solver handle %d is NULL", r);
        if (solvers[r]->chain.dof != solvers[0]->chain.dof || solvers[r]->n_tips != solvers[0]->n_tips)
            return fail(PIKAMD_EINVAL, "solver handle %d describes a different chain", r);
        for (int q = 0; q < r; ++q)
            if (solvers[q] == solvers[r]) return fail(PIKAMD_EINVAL, "solver handle %d is given twice", r);
    }
    if (B == 0) return 0;
    for (int r = 0; r < n_devices; ++r)
        if (solvers[r]->opt.soa)
            return fail(PIKAMD_EINVAL, "joint_layout soa: not with pikamd_solve_batch_sharded (a shard of a [dof][B] "
                                       "array is not contiguous)");
    const size_t d = (size_t)solvers[0]->chain.dof, g7 = 7 * (size_t)solvers[0]->n_tips;
    // host jobs per device: the second one's PCIe copies overlap the first one's kernels.  Measured (Panda, 1 M
    // targets, host arrays in and out): population 128: 166 / 163 / 168 / 193 / 225 ms with 1 / 2 / 3 / 4 / 8 jobs,
    // population 512: 348 / 354 / 361 / 414 / 485 ms -- every job is a call with its own long-running tail, and
    // the tails do not hide each other; option "shard_chunks" overrides
    constexpr int MAX_CHUNKS = 2;
    constexpr int64_t CHUNK_MIN = 32768;     // (no point cutting a shard finer than this)
    std::vector<int> rc((size_t)n_devices, 0);
    std::vector<std::string> msg((size_t)n_devices);
    // one host thread per device: staging copy, enqueue and wait of a shard all happen on it
    auto work = [&](int r) {
        int64_t lo = 0, hi = 0;
        pikamd_shard_bounds(B, r, n_devices, &lo, &hi);
        const int64_t n = hi - lo;
        if (n == 0) return;
        int chunks = (int)((n + CHUNK_MIN - 1) / CHUNK_MIN);
        const int max_chunks = solvers[r]->opt.shard_chunks > 0 ? solvers[r]->opt.shard_chunks : MAX_CHUNKS;
        chunks = chunks < 1 ? 1 : chunks > max_chunks ? max_chunks : chunks;
        int started = 0;
        for (int c = 0; c < chunks && rc[r] == 0; ++c) {
            int64_t clo = 0, chi = 0;
            pikamd_shard_bounds(n, c, chunks, &clo, &chi);
            const int64_t a = lo + clo;
            const pikamd_batch b = {chi - clo,
                                    goal_pos_quat + (size_t)a * g7,
                                    seed + (size_t)a * d,
                                    initial_guess ? initial_guess + (size_t)a * d : nullptr,
                                    problem_offset + a, // random streams keyed by the GLOBAL problem index
                                    solution + (size_t)a * d,
                                    status + a,
                                    final_cost ? final_cost + a : nullptr,
                                    stats ? stats + a : nullptr,
                                    nullptr};
            rc[r] = pikamd_solve_batches_async(solvers[r], p, 1, &b, rng_seed, c);
            if (rc[r] == 0) ++started;
        }
        for (int c = 0; c < started; ++c) {
            const int w = pikamd_wait(solvers[r], c);
            if (rc[r] == 0) rc[r] = w;
        }
        if (rc[r] != 0) msg[r] = pik::error_buffer(); // (thread-local: carried back to the caller below)
    };
    if (n_devices == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        th.reserve((size_t)n_devices);
        for (int r = 0; r < n_devices; ++r) th.emplace_back(work, r);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < n_devices; ++r)
        if (rc[r] != 0) return fail(rc[r], "device shard %d: %s", r, msg[r].c_str());
    return 0;
}

// ---- host cost functions (pik_host_solve.hpp) -------------------------------------------------
#if defined(PIK_STRICT)
#define PIK_HOST_SOLVE pik_strict_host_solve
#else
#define PIK_HOST_SOLVE pik_exact_host_solve
#endif
} // extern "C"
extern "C" int PIK_HOST_SOLVE(const pikamd_solver* s, const pikamd_params* p, long long B, const double* goal,
                              const double* seed, const double* guess, unsigned long long rng_seed,
                              long long problem_offset, pikamd_cost_fn cb, void* user, double* solution, int32_t* status,
                              double* final_cost, pikamd_stats* stats);
extern "C" {

int32_t pikamd_solve_batch_host(pikamd_solver* s, const pikamd_params* p, int64_t B, const double* goal_pos_quat,
                                const double* seed, const double* initial_guess, uint64_t rng_seed,
                                int64_t problem_offset, pikamd_cost_fn cost_fn, void* user, double* solution,
                                int32_t* status, double* final_cost, pikamd_stats* stats) {
    if (int rc = check_solver(s)) return rc;
    if (!p) return fail(PIKAMD_EINVAL, "params is NULL");
    if (!cost_fn)
        return fail(PIKAMD_EINVAL, "pikamd_solve_batch_host is for queries with a host cost function; without one "
                                   "use pikamd_solve_batch (the GPU)");
    if (B < 0 || (B > 0 && (!goal_pos_quat || !seed || !solution || !status))) return fail(PIKAMD_EINVAL, "bad arguments");
    if (p->mode == 0 && (p->memetic_elite_size < 1 || p->memetic_population_size <= p->memetic_elite_size))
        return fail(PIKAMD_EINVAL, "memetic_population_size must exceed memetic_elite_size >= 1");
    if (s->opt.soa)
        return fail(PIKAMD_EINVAL, "joint_layout soa: not with pikamd_solve_batch_host (its arrays are [B][dof])");
    return PIK_HOST_SOLVE(s, p, B, goal_pos_quat, seed, initial_guess, rng_seed, problem_offset, cost_fn, user, solution,
                          status, final_cost, stats);
}

// ---- self test -------------------------------------------------------------------------------
int32_t pikamd_self_test(pikamd_solver* s, const pikamd_params* p, int32_t n, uint32_t* disabled_out) {
    if (int rc = check_solver(s)) return rc;
    if (!p) return fail(PIKAMD_EINVAL, "params is NULL");
    if (n < 1 || n > 4096) return fail(PIKAMD_EINVAL, "n out of range [1, 4096]");
    // the flavour under test: the exact kernels (this library is the verification build, the option "arithmetic" =
    // exact, or a chain only they serve) or the product flavours' -- each keeps its own mask of switched-off widths
    const bool exact_now = pik::EXACT_FLAVOUR || s->opt.exact || needs_literal(s);
    auto mask_of = [exact_now](pik::SolverOptions& o) -> unsigned& { return exact_now ? o.disabled_lanes_exact : o.disabled_lanes; };
    if (disabled_out) *disabled_out = mask_of(s->opt);
    const int d = s->chain.dof, tips = s->n_tips;
    // reachable targets: joint vectors drawn inside the limits (a plain 64-bit LCG: nothing here needs to
    // be reproducible across machines), their forward kinematics as goals, the range midpoints as seeds
    std::vector<double> q((size_t)n * d), goal((size_t)n * 7 * tips), seed((size_t)n * d);
    unsigned long long x = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < d; ++j) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            const double u = (double)(x >> 11) * (1.0 / 9007199254740992.0);
            const bool bounded = (s->chain.bounded_mask >> j) & 1u;
            const double lo = bounded ? s->chain.qmin[j] : -3.0, hi = bounded ? s->chain.qmax[j] : 3.0;
            q[(size_t)i * d + j] = lo + (hi - lo) * u;
            seed[(size_t)i * d + j] = bounded ? 0.5 * (lo + hi) : 0.0;
        }
    if (int rc = pikamd_fk_batch(s, n, q.data(), goal.data())) return rc;
    struct Out {
        std::vector<double> sol, cost;
        std::vector<int32_t> st;
        std::vector<pikamd_stats> stats;
    };
    pik::SolverOptions saved = s->opt;
    auto run = [&](int lanes, bool passes, bool occ2, Out& o) -> int {
        s->opt = saved;
        s->opt.soa = false; // (the test's own arrays are [n][dof])
        s->opt.lpe = lanes;
        s->opt.n_sched = 0;
        s->opt.passes_set = true;
        s->opt.n_marks = 0;
        if (passes) {
            const int m[] = {1, 2, 3, 4, 7, 11, 16, 24, 40, 64};
            for (int v : m) s->opt.marks[s->opt.n_marks++] = v;
        }
        s->opt.two_per_simd = occ2 ? 1 : 0;
        s->opt.force_occ2 = occ2; // (whatever the call's size)
        s->opt.regime = 1;
        o.sol.assign((size_t)n * d, 0.0);
        o.cost.assign((size_t)n, 0.0);
        o.st.assign((size_t)n, 0);
        o.stats.assign((size_t)n, pikamd_stats{});
        const pikamd_batch b = {n,            goal.data(), seed.data(),   nullptr,        0,
                                o.sol.data(), o.st.data(), o.cost.data(), o.stats.data(), nullptr};
        int rc = start_job(s, p, 1, &b, 12345, pik::JOB_SELF_TEST); // (a job of its own: no caller's staging touched)
        if (!rc) rc = wait_job(s, pik::JOB_SELF_TEST);
        s->opt = saved;
        return rc;
    };
    auto same = [&](const Out& a, const Out& b) {
        return std::memcmp(a.sol.data(), b.sol.data(), sizeof(double) * a.sol.size()) == 0 &&
               std::memcmp(a.cost.data(), b.cost.data(), sizeof(double) * a.cost.size()) == 0 &&
               std::memcmp(a.st.data(), b.st.data(), sizeof(int32_t) * a.st.size()) == 0 &&
               std::memcmp(a.stats.data(), b.stats.data(), sizeof(pikamd_stats) * a.stats.size()) == 0;
    };
    // would the launcher serve `lanes` lanes per elite (per problem in local mode) for this chain and these
    // parameters?  The widths it would not serve fall back to the adaptive choice, which is not what is tested.
    const bool multi = tips > 1;
    const int S = (p->mode == 0 && p->memetic_num_threads > 1) ? p->memetic_num_threads : 1;
    int gs = 1;
    while (p->mode == 0 && gs < p->memetic_elite_size) gs <<= 1;
    auto served = [&](int lanes) -> bool {
        if (mask_of(saved) & (unsigned)lanes) return false;
        return pik::lpe_allowed(s, lanes, gs, S, multi, exact_now);
    };
    Out ref, got;
    unsigned disabled = 0;
    if (p->mode != 0) {
        // local mode: the cooperative descent with 8 / 16 lanes per problem against the one-lane kernel
        if (int rc = run(1, false, false, ref)) return rc;
#if !defined(PIK_STRICT)
        if (!needs_literal(s) && !exact_now)
            for (int lanes : {8, 16}) {
                if (!served(lanes)) continue;
                if (int rc = run(lanes, false, false, got)) return rc;
                if (!same(ref, got)) disabled |= (unsigned)lanes;
            }
#endif
        // exact flavours, one tip frame: the team kernels of local mode (4 / 16 lanes per problem, pik_launch.hpp)
        if (exact_now && !multi)
            for (int lanes : {4, 16}) {
                if (!served(lanes)) continue;
                if (int rc = run(lanes, false, false, got)) return rc;
                if (!same(ref, got)) disabled |= (unsigned)lanes;
            }
        mask_of(s->opt) = mask_of(saved) | disabled;
        if (disabled_out) *disabled_out = mask_of(s->opt);
        return 0;
    }
    if (int rc = run(1, false, false, ref)) return rc; // the reference: one lane per elite, one launch, one per SIMD
    for (int lanes : {2, 4, 8, 16}) {
        if (!served(lanes)) continue;
        for (int passes = 0; passes < 2; ++passes) {
            if (int rc = run(lanes, passes != 0, false, got)) return rc;
            if (!same(ref, got)) disabled |= (unsigned)lanes;
        }
    }
#if !defined(PIK_STRICT)
    // the kernels specialised for the common configuration against the general ones (when this call has it)
    {
        pik::ParamsK pk;
        if (!pik::make_params_k(p, pk) && common_eligible(s, p, pk)) {
            const bool was = s->opt.specialised;
            s->opt.specialised = false;
            Out general;
            saved.specialised = false; // (run() works on a copy of `saved`)
            const int rc = run(1, false, false, general);
            saved.specialised = was;
            s->opt.specialised = was;
            if (rc) return rc;
            if (!same(ref, general)) disabled |= 32u;
        }
    }
#endif
    // the two-per-SIMD build of the one-lane kernel, forced whatever the size of the call (one species, one tip)
    if (S == 1 && !multi) {
        if (int rc = run(1, false, true, got)) return rc;
        if (!same(ref, got)) disabled |= 1u;
    }
    if (int rc = run(1, true, false, got)) return rc;
    if (!same(ref, got)) return fail(PIKAMD_EHIP, "self test: the one-lane kernel disagrees with itself under compaction passes");
    if (disabled & 32u) s->opt.specialised = false; // (the two flavours disagree: keep to the general kernels)
    mask_of(s->opt) = mask_of(saved) | (disabled & ~32u);
    disabled = (disabled & 32u);
    if (disabled_out) *disabled_out = mask_of(s->opt) | disabled;
    return 0;
}

int32_t pikamd_self_test_cost(const pikamd_solver* s, int32_t* runs, double* total_ms) {
    if (!s) return fail(PIKAMD_EINVAL, "solver is NULL");
    if (runs) *runs = s->self_test_runs;
    if (total_ms) *total_ms = s->self_test_ms;
    return 0;
}

} // extern "C"

// Option self_test = auto: the first solve (or reserve) of a parameter set that the GENERAL or the EXACT kernels
// serve runs pikamd_self_test on 32 generated targets of the handle's own chain first -- every kernel variant
// against the one-lane kernel, bit for bit; a variant that disagrees is switched off for the handle.  Those
// kernels sit at the register cap for long chains and have come out of the compiler wrong before (DESIGN.md
// section 3); the kernels of the common configuration are fuzzed at every chain length by the test suite and
// are not re-checked here.  Costs a dozen small solves, once per parameter set and handle.
static int maybe_self_test(pikamd_solver* s, const pikamd_params* p) {
    if (!s->opt.auto_self_test || s->in_self_test || !p) return 0;
    pik::ParamsK pk;
    if (pik::make_params_k(p, pk)) return 0; // (the solve itself reports the bad parameter)
#if !defined(PIK_STRICT)
    if (common_eligible(s, p, pk) && common_ops(s->chain.dof, pk.goal_mask != 0)) return 0;
#endif
    // One run per KERNEL SET, not per parameter set: what selects the kernels and the paths inside them is the
    // flavour, the mode, species and elites (how a wavefront is dealt out), which joint goals and cost terms are on,
    // the line-search form and the approximate-solution return -- thresholds, weights, population and budgets do not,
    // so a plugin whose parameters change at run time does not pay again.
    unsigned long long h = 1469598103934665603ull;
    const long long key[] = {s->opt.exact ? 1 : 0, p->mode, p->mode == 0 ? p->memetic_num_threads : 1,
                             p->mode == 0 ? p->memetic_elite_size : 0, pk.goal_mask, pk.line_delta, pk.approx,
                             pk.has_pos_thr, pk.has_ori_thr, pk.stop_on_valid, pk.stop_on_first,
                             p->position_scale > 0.0 ? 1 : 0, p->rotation_scale > 0.0 ? 1 : 0}; // (the pose cost's two terms)
    const unsigned char* b = reinterpret_cast<const unsigned char*>(key);
    for (size_t i = 0; i < sizeof key; ++i) h = (h ^ b[i]) * 1099511628211ull;
    for (unsigned long long k : s->self_tested)
        if (k == h) return 0;
    // ... and a SHORT run: the variants re-schedule one arithmetic, a disagreement shows in the first descent step of
    // the first generation; four generations of six steps on 32 targets cross every phase of a generation, the
    // compaction marks 1, 2 and 3 of pikamd_self_test's list (park, re-pack and resume, three times; its later marks
    // are only reached by a full-length call of pikamd_self_test) and both line-search forms (~10 ms instead of the
    // ~230 ms of full-length solves on the exact kernels).  pikamd_self_test itself runs whatever parameters it is
    // given.  The stream-ordered entry points never come here (they must not block): reserve first.
    pikamd_params q = *p;
    if (q.mode == 0) {
        q.memetic_max_generations = q.memetic_max_generations < 4 ? q.memetic_max_generations : 4;
        q.memetic_gd_max_iters = q.memetic_gd_max_iters < 6 ? q.memetic_gd_max_iters : 6;
    } else {
        q.gd_max_iters = q.gd_max_iters < 24 ? q.gd_max_iters : 24;
    }
    const auto t0 = std::chrono::steady_clock::now();
    s->in_self_test = true;
    const int rc = pikamd_self_test(s, &q, 32, nullptr);
    s->in_self_test = false;
    if (rc) return rc; // (a run that failed to run is not counted and will be tried again)
    s->self_test_runs += 1;
    s->self_test_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    s->self_tested.push_back(h);
    return 0;
}

extern "C" {

const char* pikamd_kernel_name(const pikamd_solver* s, const pikamd_params* p) {
    if (!s || !p) return "";
    pikamd_solver* m = const_cast<pikamd_solver*>(s);
    // the flavour that serves a solve call with these parameters (see solve_ops_of)
    const char* ns = "pik";
#if defined(PIK_STRICT)
    ns = "pik_strict";
#else
    pik::ParamsK pk;
    if (needs_literal(s)) ns = "pik_exact";
    else if (!pik::make_params_k(p, pk) && common_eligible(s, p, pk) && common_ops(s->chain.dof, pk.goal_mask != 0))
        ns = pk.goal_mask != 0 ? "pik_common_goals" : "pik_common";
#endif
    snprintf(m->kernel_name, sizeof m->kernel_name, "%s::%s<%d>", ns,
             p->mode == 1 ? "ik_gradient_kernel" : "memetic_kernel", s->chain.dof);
    return m->kernel_name;
}

} // extern "C"
