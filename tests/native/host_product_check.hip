// The PRODUCT kernels' arithmetic as a sequential host loop (pik_host_solve.hpp compiled as the fast flavour): reads
// a chain, parameters and problems, solves them one after the other on this thread and prints solution / status /
// cost / counters with all digits -- tests/test_gpu_product_arithmetic.py compares them, bit for bit, with what the
// product library's kernels return for the same problems.  Host only: hipcc --cuda-host-only -ffp-contract=on -mfma
// (the contraction rule the device code is compiled with; fused multiply-adds as one instruction).
#if defined(PIK_STRICT)
#error "the fast flavour"
#endif
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "../../pick_ik_amd/csrc/pik_host_solve.hpp"

namespace pik {
char* error_buffer() {
    static char buf[ERROR_BUFFER_SIZE];
    return buf;
}
} // namespace pik

template <int D>
static int run(const pikamd_solver* s, const pikamd_params& pp, long long B, const double* goal, const double* seed,
               unsigned long long rng_seed, long long offset) {
    pik::ParamsK pk;
    if (pp.mode != 2)
        if (const char* m = pik::make_params_k(&pp, pk)) {
            std::fprintf(stderr, "%s\n", m);
            return 1;
        }
    if (pp.mode == 2) { // cost + verdict of the joint vectors given as "seed" (primitive check)
        std::vector<char> mem(sizeof(pik::ConstsK<D>));
        pik::ConstsK<D>& kc = *new (mem.data()) pik::ConstsK<D>;
        std::memset(&kc, 0, sizeof kc);
        kc.chain = pik::make_chain_k<D>(s->chain);
        kc.n_tips = 1;
        pikamd_params p1 = pp;
        p1.mode = 0;
        if (pik::make_params_k(&p1, pk)) return 1;
        kc.params = pk;
        for (long long b = 0; b < B; ++b) {
            pik::HostProblem<D> pb;
            pb.kc = &kc;
            pik::make_goal(goal + 7 * b, pb.goal);
            double q[D];
            for (int j = 0; j < D; ++j) pb.seed[j] = q[j] = seed[b * D + j];
            pik::EvalOut e;
            pik::host_evaluate<D>(pb, q, e);
            std::printf("%d %.17g 0 0 0 0", e.sol ? 1 : 0, e.cost);
            for (int j = 0; j < D; ++j) std::printf(" %.17g", q[j]);
            std::printf("\n");
        }
        return 0;
    }
    std::vector<double> sol((size_t)B * D), cost((size_t)B);
    std::vector<int32_t> st((size_t)B);
    std::vector<pikamd_stats> stats((size_t)B);
    pik::host_solve_batch<D>(s, &pp, pk, B, goal, seed, nullptr, rng_seed, offset, nullptr, nullptr, sol.data(), st.data(),
                             cost.data(), stats.data());
    for (long long b = 0; b < B; ++b) {
        std::printf("%d %.17g %lld %d %d %d", st[(size_t)b], cost[(size_t)b], (long long)stats[(size_t)b].cost_evals,
                    stats[(size_t)b].generations, stats[(size_t)b].wipeouts, stats[(size_t)b].pool_erasures);
        for (int j = 0; j < D; ++j) std::printf(" %.17g", sol[(size_t)b * D + j]);
        std::printf("\n");
    }
    return 0;
}

int main() {
    // stdin: dof B rng_seed offset ; chain arrays ; tip paths ; species ; the parameters ; B x (goal[7 n_tips] seed[dof])
    int dof;
    long long B, offset;
    unsigned long long rng_seed;
    if (std::scanf("%d %lld %llu %lld", &dof, &B, &rng_seed, &offset) != 4) return 2;
    std::vector<double> o(6 * dof), ax(3 * dof), tip(6), lo(dof), hi(dof), vm(dof);
    std::vector<int32_t> jt(dof);
    std::vector<uint8_t> bd(dof);
    auto rd = [](std::vector<double>& v) { for (double& x : v) if (std::scanf("%lf", &x) != 1) std::exit(2); };
    rd(o); rd(ax); rd(tip); rd(lo); rd(hi); rd(vm);
    for (int j = 0; j < dof; ++j) { int a, b; if (std::scanf("%d %d", &a, &b) != 2) return 2; jt[j] = a; bd[j] = (uint8_t)b; }
    auto s = std::make_unique<pikamd_solver>();
    // tip paths: "n_tips" (0 = the arrays above are ONE serial chain), then per tip: n_joints, the variable of each
    // joint, origins [n][6], axes [n][3], joint types [n], tip transform [6]
    int n_tips = 0;
    if (std::scanf("%d", &n_tips) != 1) return 2;
    std::vector<std::vector<int32_t>> tv((size_t)n_tips), tjt((size_t)n_tips);
    std::vector<std::vector<double>> to((size_t)n_tips), tax((size_t)n_tips), ttip((size_t)n_tips);
    if (n_tips == 0) {
        pikamd_chain ch{dof, o.data(), ax.data(), jt.data(), tip.data(), lo.data(), hi.data(), vm.data(), bd.data()};
        if (const char* m = pik::build_chain(&ch, s->chain)) { std::fprintf(stderr, "%s\n", m); return 1; }
        s->n_tips = 1;
    } else {
        std::vector<pikamd_tip> tips((size_t)n_tips);
        for (int k = 0; k < n_tips; ++k) {
            int n;
            if (std::scanf("%d", &n) != 1) return 2;
            tv[(size_t)k].resize((size_t)n); tjt[(size_t)k].resize((size_t)n);
            to[(size_t)k].resize((size_t)n * 6); tax[(size_t)k].resize((size_t)n * 3); ttip[(size_t)k].resize(6);
            for (int j = 0; j < n; ++j) if (std::scanf("%d", &tv[(size_t)k][(size_t)j]) != 1) return 2;
            rd(to[(size_t)k]); rd(tax[(size_t)k]);
            for (int j = 0; j < n; ++j) if (std::scanf("%d", &tjt[(size_t)k][(size_t)j]) != 1) return 2;
            rd(ttip[(size_t)k]);
            tips[(size_t)k] = pikamd_tip{n, tv[(size_t)k].data(), to[(size_t)k].data(), tax[(size_t)k].data(), tjt[(size_t)k].data(), ttip[(size_t)k].data()};
        }
        pikamd_multi_chain mc{dof, n_tips, tips.data(), lo.data(), hi.data(), vm.data(), bd.data()};
        for (int k = 0; k < n_tips; ++k)
            if (const char* m = pik::build_tip_chain(&mc, k, k == 0 ? s->chain : s->more[k - 1])) { std::fprintf(stderr, "%s\n", m); return 1; }
        s->n_tips = n_tips;
    }
    pikamd_params pp;
    if (std::scanf("%d %d", &pp.memetic_num_threads, &pp.memetic_stop_on_first_solution) != 2) return 2;
    int stop, approx;
    if (std::scanf("%d %lf %d %lf %lf %lf %lf %lf %lf %lf %lf %lf %d %d %d %lf %d %d %d", &pp.mode, &pp.gd_step_size, &pp.gd_max_iters,
                   &pp.gd_min_cost_delta, &pp.position_threshold, &pp.orientation_threshold, &pp.cost_threshold, &pp.position_scale,
                   &pp.rotation_scale, &pp.center_joints_weight, &pp.avoid_joint_limits_weight, &pp.minimal_displacement_weight,
                   &stop, &pp.memetic_population_size, &pp.memetic_elite_size, &pp.memetic_wipeout_fitness_tol,
                   &pp.memetic_max_generations, &pp.memetic_gd_max_iters, &approx) != 19) return 2;
    pp.stop_optimization_on_valid_solution = stop;
    pp.return_approximate_solution = approx;
    const int g7 = 7 * s->n_tips;
    std::vector<double> goal((size_t)B * g7), seed((size_t)B * dof);
    for (long long b = 0; b < B; ++b) {
        for (int j = 0; j < g7; ++j) if (std::scanf("%lf", &goal[(size_t)b * g7 + j]) != 1) return 2;
        for (int j = 0; j < dof; ++j) if (std::scanf("%lf", &seed[(size_t)b * dof + j]) != 1) return 2;
    }
    switch (dof) {
#define PIK_CASE(N) case N: return run<N>(s.get(), pp, B, goal.data(), seed.data(), rng_seed, offset);
        PIK_CASE(1) PIK_CASE(2) PIK_CASE(3) PIK_CASE(4) PIK_CASE(5) PIK_CASE(6) PIK_CASE(7) PIK_CASE(8)
        PIK_CASE(9) PIK_CASE(10) PIK_CASE(11) PIK_CASE(12) PIK_CASE(13) PIK_CASE(14) PIK_CASE(15) PIK_CASE(16)
#undef PIK_CASE
        default: return 3;
    }
}
