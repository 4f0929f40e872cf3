// Exercises the C++ host mirror (pick_ik_amd/host/pick_ik_amd.hpp) against the reference's own
// ik_tests.cpp cases.  argv[1] = "nogpu": only checks that construction fails loudly without a
// device (no CPU fallback); "gpu": runs the RR and Panda gradient cases and a memetic solve.
#include <cmath>
#include <cstdio>
#include <thread>
#include <cstring>

#include "../../pick_ik_amd/host/pick_ik_amd.hpp"

using namespace pick_ik_amd;

static Chain rr_chain() {
    Chain c;
    Joint a, b;
    b.origin_xyz = {2.0, 0.0, 0.0};
    a.max_velocity = b.max_velocity = 1.0;
    c.joints = {a, b};
    c.tip_xyz = {1.0, 0.0, 0.0};
    return c;
}

static Chain panda_chain() {
    const double PI = M_PI;
    const double o[7][6] = {{0, 0, 0.333, 0, 0, 0},        {0, 0, 0, -PI / 2, 0, 0},
                            {0, -0.316, 0, PI / 2, 0, 0},  {0.0825, 0, 0, PI / 2, 0, 0},
                            {-0.0825, 0.384, 0, -PI / 2, 0, 0}, {0, 0, 0, PI / 2, 0, 0},
                            {0.088, 0, 0, PI / 2, 0, 0}};
    const double lo[7] = {-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973};
    const double hi[7] = {2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973};
    const double vm[7] = {2.175, 2.175, 2.175, 2.175, 2.61, 2.61, 2.61};
    Chain c;
    for (int j = 0; j < 7; ++j) {
        Joint J;
        J.origin_xyz = {o[j][0], o[j][1], o[j][2]};
        J.origin_rpy = {o[j][3], o[j][4], o[j][5]};
        J.min = lo[j];
        J.max = hi[j];
        J.max_velocity = vm[j];
        c.joints.push_back(J);
    }
    c.tip_xyz = {0, 0, 0.107};
    c.tip_rpy = {0, 0, -PI / 4};
    return c;
}

#define CHECK(cond)                                                      \
    do {                                                                 \
        if (!(cond)) {                                                   \
            std::printf("CHECK FAILED line %d: %s\n", __LINE__, #cond);  \
            return 1;                                                    \
        }                                                                \
    } while (0)

int main(int argc, char** argv) {
    const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
    if (!gpu) {
        try {
            Solver s(rr_chain());
            std::printf("unexpected: solver constructed without a GPU\n");
            return 1;
        } catch (const std::runtime_error& e) {
            std::printf("expected failure: %s\n", e.what());
            return std::strstr(e.what(), "no HIP device") ? 0 : 1;
        }
    }
    // ---- RR arm, tests/ik_tests.cpp:137-238 ----
    Solver rr(rr_chain());
    CostSpec c;
    c.position_threshold = 1e-4;
    c.orientation_threshold = 1e-3;
    c.cost_threshold = 1e-4;
    c.rotation_scale = 1.0;
    GradientIkParams gd;
    Pose p0 = rr.fk({0.0, 0.0});
    CHECK(std::fabs(p0.x - 3.0) < 1e-12 && std::fabs(p0.y) < 1e-12);
    auto r = rr.ik_gradient({0.1, -0.1}, Pose{3, 0, 0, 1, 0, 0, 0}, c, gd);
    CHECK(r && std::fabs((*r)[0]) < 0.01 && std::fabs((*r)[1]) < 0.01);
    r = rr.ik_gradient({0.0, 0.0}, Pose{0, 0, 0, 1, 0, 0, 0}, c, gd); // unreachable position
    CHECK(!r);
    // ---- Panda, tests/ik_tests.cpp:240-293 and tests/ik_memetic_tests.cpp:110-164 ----
    Solver pa(panda_chain());
    CHECK(pa.robot().variables.size() == 7);
    const std::vector<double> home = {0.0, -M_PI / 4, 0.0, -3.0 * M_PI / 4, 0.0, M_PI / 2, M_PI / 4};
    std::vector<double> actual = home;
    const double d[7] = {0.1, -0.1, 0.1, -0.1, 0.1, -0.1, 0.1};
    for (int i = 0; i < 7; ++i) actual[i] += d[i];
    c.rotation_scale = 0.5;
    r = pa.ik_gradient(home, pa.fk(actual), c, gd);
    CHECK(r);
    for (int i = 0; i < 7; ++i) CHECK(std::fabs((*r)[i] - actual[i]) < 0.025);
    CostSpec cm;
    cm.orientation_threshold = 0.01;
    MemeticIkParams mp;
    const Pose goal = pa.fk(home);
    r = pa.ik_memetic(std::vector<double>(7, 0.0), goal, cm, mp, false, 1);
    CHECK(r);
    const Pose got = pa.fk(*r);
    CHECK(std::fabs(got.x - goal.x) < 1e-3 && std::fabs(got.y - goal.y) < 1e-3 && std::fabs(got.z - goal.z) < 1e-3);
    // cost_fn / solution_fn of a joint vector (the approximate-solution gate of the plugin uses it)
    {
        auto ev = pa.evaluate(*r, {goal}, home, cm);
        CHECK(ev.is_solution && ev.cost < 1e-5);
        std::vector<double> off = *r;
        off[1] += 0.2;
        ev = pa.evaluate(off, {goal}, home, cm);
        CHECK(!ev.is_solution && ev.cost > 1e-3);
    }
    // batch form + error behaviour
    std::vector<double> seeds;
    std::vector<Pose> goals;
    for (int b = 0; b < 8; ++b) {
        seeds.insert(seeds.end(), home.begin(), home.end());
        std::vector<double> q = home;
        q[0] += 0.05 * b;
        goals.push_back(pa.fk(q));
    }
    auto br = pa.ik_memetic_batch(seeds, goals, cm, mp, false, 5);
    int ok = 0;
    for (int s : br.status) ok += s == PIKAMD_SUCCESS;
    CHECK(ok == 8);
    // ---- several tip frames: a torso yaw shared by two 2-joint arms ----
    {
        MultiChain mc;
        mc.variables.resize(5);
        for (auto& v : mc.variables) {
            v.min = -2.0;
            v.max = 2.0;
            v.max_velocity = 1.0;
        }
        Joint torso;
        torso.origin_xyz = {0, 0, 0.4};
        for (int side = 0; side < 2; ++side) {
            const double y = side ? -0.2 : 0.2;
            Joint sh, el;
            sh.origin_xyz = {0, y, 0.3};
            sh.axis = {0, 1, 0};
            el.origin_xyz = {0.3, 0, 0};
            el.axis = {0, 1, 0};
            TipPath t;
            t.variable = {0, side ? 3 : 1, side ? 4 : 2};
            t.joints = {torso, sh, el};
            t.tip_xyz = {0.25, 0, 0};
            mc.tips.push_back(t);
        }
        Solver tree(mc);
        CHECK(tree.n_tips() == 2 && tree.dof() == 5);
        const std::vector<double> qt = {0.3, 0.4, -0.5, -0.2, 0.6};
        const std::vector<Pose> tips = tree.fk_tips(qt);
        CHECK(tips.size() == 2 && std::fabs(tips[0].z - tips[1].z) > 1e-3);
        CostSpec ct;
        ct.orientation_threshold = 0.01;
        MemeticIkParams mt;
        mt.population_size = 32;
        auto rt = tree.ik_memetic(std::vector<double>(5, 0.0), tips, ct, mt, false, 3);
        CHECK(rt);
        const std::vector<Pose> got2 = tree.fk_tips(*rt);
        for (int k = 0; k < 2; ++k)
            CHECK(std::fabs(got2[k].x - tips[k].x) < 1e-3 && std::fabs(got2[k].y - tips[k].y) < 1e-3 &&
                  std::fabs(got2[k].z - tips[k].z) < 1e-3);
        bool threw2 = false;
        try {
            tree.fk(qt); // one-tip accessor on a two-tip solver
        } catch (const std::invalid_argument&) {
            threw2 = true;
        }
        CHECK(threw2);
    }
    // ---- one handle per host thread, used concurrently (the multi-GPU front-end pattern) ----
    {
        mp.population_size = 16;
        const auto serial = pa.ik_memetic_batch(seeds, goals, cm, mp, false, 5);
        std::vector<BatchResult> got_t(3);
        std::vector<std::thread> th;
        for (int t = 0; t < 3; ++t)
            th.emplace_back([&, t] {
                Solver mine(panda_chain());
                for (int rep = 0; rep < 4; ++rep) got_t[t] = mine.ik_memetic_batch(seeds, goals, cm, mp, false, 5);
            });
        for (auto& x : th) x.join();
        for (int t = 0; t < 3; ++t) CHECK(got_t[t].solution == serial.solution && got_t[t].status == serial.status);
    }
    bool threw = false;
    try {
        mp.population_size = 4; // == elite_size: invalid
        pa.ik_memetic(home, goal, cm, mp);
    } catch (const std::runtime_error&) {
        threw = true;
    }
    CHECK(threw);
    std::printf("host C++ checks OK\n");
    return 0;
}
