// Compiles the kernels' arithmetic (pick_ik_amd/csrc/pik_math.hpp, PIK_HD = host) and the host-side
// model extraction (pik_host.hpp) with plain g++ and dumps results for tests/test_host_math_cpu.py
// to compare with the CPU oracle: FK (Denavit-Hartenberg form of the fast build), cost + solution verdict, the
// frame-based gradient probes against literal central differences, sincos/atan2, Philox.
// Build flavour: -DPIK_STRICT selects the strict-arithmetic code paths.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../pick_ik_amd/csrc/pik_host.hpp"

using namespace pik;

template <int D>
int run(const ChainHost& h, const pikamd_params& pp, int n, const double* q, const double* goal,
        const double* seed) {
    static ChainK<D> c;
    c = make_chain_k<D>(h);
    ParamsK p;
    if (const char* m = make_params_k(&pp, p)) {
        std::fprintf(stderr, "%s\n", m);
        return 1;
    }
#if PIK_XF
    // the chain class the exact flavour's kernels pick their form by, and the kinds of the fixed transforms
    std::printf("class %u %x %u\n", c.uniform_z, c.origin_kinds, c.tip_kind);
#endif
    for (int i = 0; i < n; ++i) {
        double qq[D], sd[D];
        for (int j = 0; j < D; ++j) {
            qq[j] = q[i * D + j];
            sd[j] = seed[i * D + j];
        }
        GoalK g;
        g.t[0] = goal[7 * i];
        g.t[1] = goal[7 * i + 1];
        g.t[2] = goal[7 * i + 2];
        const double gq[4] = {goal[7 * i + 3], goal[7 * i + 4], goal[7 * i + 5], goal[7 * i + 6]};
        double GR[9];
        quat_to_matrix(gq, GR);
        matrix_to_quat(GR, g.q);
        // FK + quaternion
        double R[9], t[3], qt[4];
        fk<D, false>(c, qq, R, t, nullptr, 0);
        matrix_to_quat(R, qt);
        std::printf("fk %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", t[0], t[1], t[2], qt[0], qt[1], qt[2], qt[3]);
#if PIK_XF
        // the forward kinematics of the class forms (fk_uz: the products by the exact ones and zeros of the fixed
        // transforms left out, x_iso_mul) -- the same bits as the literal chain product above
        if constexpr (PIK_XUZ_D(D)) {
            if (c.uniform_z != 0u) {
                double Ru[9], tu[3], qu[4];
                if (c.uniform_z == 1u) fk_uz<D, 1>(c, qq, Ru, tu);
                else fk_uz<D, 2>(c, qq, Ru, tu);
                matrix_to_quat(Ru, qu);
                std::printf("fkuz %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", tu[0], tu[1], tu[2], qu[0], qu[1], qu[2], qu[3]);
            }
        }
#endif
        // cost + verdict (+ frames)
        EvalOut e;
        double tipt[3], d0[4];
        std::vector<double> fr(6 * D);
        eval_pose<D, true>(c, p, g, sd, qq, e, tipt, d0, fr.data(), 1);
        std::printf("cost %.17g %d\n", e.cost, e.sol ? 1 : 0);
#if !defined(PIK_STRICT)
        // frame-based probes vs literal central differences through the same eval
        double grad[D];
        probe_gradient<D>(c, p, g, sd, qq, e, tipt, d0, fr.data(), 1, grad);
        std::printf("grad");
        for (int j = 0; j < D; ++j) {
            double a[D], b[D];
            for (int k = 0; k < D; ++k) a[k] = b[k] = qq[k];
            a[j] -= p.step_size;
            b[j] += p.step_size;
            EvalOut ea, eb;
            double tt[3], dd[4];
            eval_pose<D, false>(c, p, g, sd, a, ea, tt, dd, nullptr, 0);
            eval_pose<D, false>(c, p, g, sd, b, eb, tt, dd, nullptr, 0);
            std::printf(" %.17g %.17g", grad[j], eb.cost - ea.cost);
        }
        std::printf("\n");
#endif
    }
    return 0;
}

int main(int argc, char** argv) {
    // stdin: dof n ; chain arrays ; params subset ; then n x (q[dof] goal[7] seed[dof])
    int dof, n;
    if (std::scanf("%d %d", &dof, &n) != 2) return 2;
    std::vector<double> o(6 * dof), ax(3 * dof), tip(6), lo(dof), hi(dof), vm(dof);
    std::vector<int32_t> jt(dof);
    std::vector<uint8_t> bd(dof);
    auto rd = [](std::vector<double>& v) { for (double& x : v) if (std::scanf("%lf", &x) != 1) std::exit(2); };
    rd(o); rd(ax); rd(tip); rd(lo); rd(hi); rd(vm);
    for (int j = 0; j < dof; ++j) { int a, b; if (std::scanf("%d %d", &a, &b) != 2) return 2; jt[j] = a; bd[j] = (uint8_t)b; }
    pikamd_chain ch{dof, o.data(), ax.data(), jt.data(), tip.data(), lo.data(), hi.data(), vm.data(), bd.data()};
    ChainHost h;
    if (const char* m = build_chain(&ch, h)) { std::fprintf(stderr, "%s\n", m); return 1; }
    pikamd_params pp;
    // defaults (src/pick_ik_parameters.yaml) + the three goal weights from stdin
    pp.mode = 1; pp.gd_step_size = 1e-4; pp.gd_max_iters = 100; pp.gd_min_cost_delta = 1e-12;
    pp.position_threshold = 1e-3; pp.orientation_threshold = 1e-3; pp.cost_threshold = 1e-3;
    pp.position_scale = 1.0; pp.rotation_scale = 0.5;
    pp.stop_optimization_on_valid_solution = 1; pp.memetic_num_threads = 1; pp.memetic_stop_on_first_solution = 1;
    pp.memetic_population_size = 16; pp.memetic_elite_size = 4; pp.memetic_wipeout_fitness_tol = 1e-5;
    pp.memetic_max_generations = 100; pp.memetic_gd_max_iters = 25; pp.return_approximate_solution = 0;
    if (std::scanf("%lf %lf %lf", &pp.center_joints_weight, &pp.avoid_joint_limits_weight, &pp.minimal_displacement_weight) != 3) return 2;
    std::vector<double> q(n * dof), goal(n * 7), seed(n * dof);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < dof; ++j) if (std::scanf("%lf", &q[i * dof + j]) != 1) return 2;
        for (int j = 0; j < 7; ++j) if (std::scanf("%lf", &goal[i * 7 + j]) != 1) return 2;
        for (int j = 0; j < dof; ++j) if (std::scanf("%lf", &seed[i * dof + j]) != 1) return 2;
    }
    // sincos / atan2 / philox spot values
    MathTab mt;
    fill_math_tab(mt);
    for (double x : {0.0, 0.5, -2.0, 3.0, 100.25, -7e4}) {
        double s, c;
        sincos_f64(mt, x, s, c);
        std::printf("sincos %.17g %.17g %.17g\n", x, s, c);
    }
    for (double y : {0.0, 1e-9, 0.3, 1.0, 5.0})
        for (double x : {0.0, 1e-9, 0.7, 1.0}) std::printf("atan2 %.17g %.17g %.17g\n", y, x, atan2_pos(mt, y, x));
#if !defined(PIK_STRICT)
    // line-search angle addition (product build only): sin / cos of theta + d from those of theta
    for (double th : {0.0, 0.7, -2.9, 3.1, 1.5707963267948966})
        for (double d : {0.0, 1e-9, -1e-4, 1e-4, 1e-3, -1e-3}) {
            double s2, c2;
            sincos_delta(std::sin(th), std::cos(th), d, s2, c2);
            std::printf("sincosdelta %.17g %.17g %.17g %.17g\n", th, d, s2, c2);
        }
#endif
    const U4 r = philox4x32_10(0x243f6a88u, 0x85a308d3u, 0x13198a2eu, 0x03707344u, 0xa4093822u, 0x299f31d0u);
    std::printf("philox %08x %08x %08x %08x\n", r.x, r.y, r.z, r.w);
    (void)argc; (void)argv;
    switch (dof) {
        case 1: return run<1>(h, pp, n, q.data(), goal.data(), seed.data());
        case 2: return run<2>(h, pp, n, q.data(), goal.data(), seed.data());
        case 3: return run<3>(h, pp, n, q.data(), goal.data(), seed.data());
        case 4: return run<4>(h, pp, n, q.data(), goal.data(), seed.data());
        case 5: return run<5>(h, pp, n, q.data(), goal.data(), seed.data());
        case 6: return run<6>(h, pp, n, q.data(), goal.data(), seed.data());
        case 7: return run<7>(h, pp, n, q.data(), goal.data(), seed.data());
        case 8: return run<8>(h, pp, n, q.data(), goal.data(), seed.data());
        case 9: return run<9>(h, pp, n, q.data(), goal.data(), seed.data());
        case 10: return run<10>(h, pp, n, q.data(), goal.data(), seed.data());
        case 11: return run<11>(h, pp, n, q.data(), goal.data(), seed.data());
        case 12: return run<12>(h, pp, n, q.data(), goal.data(), seed.data());
        case 13: return run<13>(h, pp, n, q.data(), goal.data(), seed.data());
        case 14: return run<14>(h, pp, n, q.data(), goal.data(), seed.data());
        case 15: return run<15>(h, pp, n, q.data(), goal.data(), seed.data());
        case 16: return run<16>(h, pp, n, q.data(), goal.data(), seed.data());
        default: return 3;
    }
}
