// Drives the MoveIt plugin shim (pick_ik_amd/host/pick_ik_plugin_shim.cpp) through the declaration
// stubs of tests/native/ros2_stubs/ onto the real libpick_ik_amd.so: initialize (model walk, names,
// error behaviour of src/pick_ik_plugin.cpp:22-71), searchPositionIK (parameter mapping :165-196,
// restarts :276-291, solution callback :269-273, approximate-solution gate :219-267, seed returned
// on failure :213-217), every overload (:314-401).  argv[1] = "gpu" runs it; anything else only
// checks the parts that need no device.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../oracle/pik_oracle.h" // (the checker: replays the plugin's exact-arithmetic queries)
#include "../../pick_ik_amd/host/pick_ik_plugin_shim.cpp"

#define CHECK(cond)                                                      \
    do {                                                                 \
        if (!(cond)) {                                                   \
            std::printf("CHECK FAILED line %d: %s\n", __LINE__, #cond);  \
            return 1;                                                    \
        }                                                                \
    } while (0)

namespace mc = moveit::core;

static Eigen::Isometry3d origin(double x, double y, double z, double roll, double pitch, double yaw) {
    // urdf::Rotation::setFromRPY -> quaternion -> matrix
    double const phi = roll / 2, the = pitch / 2, psi = yaw / 2;
    Eigen::Quaterniond q(std::cos(phi) * std::cos(the) * std::cos(psi) + std::sin(phi) * std::sin(the) * std::sin(psi),
                         std::sin(phi) * std::cos(the) * std::cos(psi) - std::cos(phi) * std::sin(the) * std::sin(psi),
                         std::cos(phi) * std::sin(the) * std::cos(psi) + std::sin(phi) * std::cos(the) * std::sin(psi),
                         std::cos(phi) * std::cos(the) * std::sin(psi) - std::sin(phi) * std::sin(the) * std::cos(psi));
    Eigen::Isometry3d T;
    T.R = q.toRotationMatrix();
    T.t = Eigen::Vector3d(x, y, z);
    return T;
}

// the Panda of moveit_resources, world -> panda_link0 -> ... -> panda_hand
static void build_panda(mc::RobotModel& m) {
    double const PI = M_PI;
    double const o[7][6] = {{0, 0, 0.333, 0, 0, 0},          {0, 0, 0, -PI / 2, 0, 0},
                            {0, -0.316, 0, PI / 2, 0, 0},    {0.0825, 0, 0, PI / 2, 0, 0},
                            {-0.0825, 0.384, 0, -PI / 2, 0, 0}, {0, 0, 0, PI / 2, 0, 0},
                            {0.088, 0, 0, PI / 2, 0, 0}};
    double const lo[7] = {-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973};
    double const hi[7] = {2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973};
    double const vm[7] = {2.175, 2.175, 2.175, 2.175, 2.61, 2.61, 2.61};
    m.add_root("world");
    m.add_link("panda_link0", "world", "virtual_joint", mc::JointModel::FIXED, origin(0.2, -0.1, 0.05, 0, 0, 0.3),
               Eigen::Vector3d(0, 0, 1), {});
    std::vector<std::string> joints;
    for (int j = 0; j < 7; ++j) {
        mc::VariableBounds b;
        b.position_bounded_ = true;
        b.min_position_ = lo[j];
        b.max_position_ = hi[j];
        b.max_velocity_ = vm[j];
        b.velocity_bounded_ = true;
        std::string const link = "panda_link" + std::to_string(j + 1), parent = "panda_link" + std::to_string(j);
        joints.push_back("panda_joint" + std::to_string(j + 1));
        m.add_link(link, parent, joints.back(), mc::JointModel::REVOLUTE,
                   origin(o[j][0], o[j][1], o[j][2], o[j][3], o[j][4], o[j][5]), Eigen::Vector3d(0, 0, 1), b);
    }
    m.add_link("panda_link8", "panda_link7", "panda_joint8", mc::JointModel::FIXED, origin(0, 0, 0.107, 0, 0, 0),
               Eigen::Vector3d(0, 0, 1), {});
    m.add_link("panda_hand", "panda_link8", "panda_hand_joint", mc::JointModel::FIXED, origin(0, 0, 0, 0, 0, -PI / 4),
               Eigen::Vector3d(0, 0, 1), {});
    joints.push_back("panda_joint8");
    m.add_group("panda_arm", joints);
}

static geometry_msgs::msg::Pose pose_of(Eigen::Isometry3d const& T) {
    Eigen::Quaterniond const q(T.rotation());
    geometry_msgs::msg::Pose p;
    p.position.x = T.translation().x();
    p.position.y = T.translation().y();
    p.position.z = T.translation().z();
    p.orientation.w = q.w();
    p.orientation.x = q.x();
    p.orientation.y = q.y();
    p.orientation.z = q.z();
    return p;
}

int main(int argc, char** argv) {
    bool const gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
    // pluginlib identity (pick_ik_kinematics_description.xml:1-4, src/pick_ik_plugin.cpp:405)
    CHECK(pluginlib_stub::exported().size() == 1);
    CHECK(pluginlib_stub::exported()[0].first == "pick_ik::PickIKPlugin");
    CHECK(pluginlib_stub::exported()[0].second == "kinematics::KinematicsBase");

    mc::RobotModel model;
    build_panda(model);
    auto node = std::make_shared<rclcpp::Node>();
    std::string const ns = "robot_description_kinematics.panda_arm.";
    pick_ik::PickIKPlugin plugin;
    kinematics::KinematicsBase& base = plugin; // MoveIt only ever sees the base class

    if (!gpu) {
        // unknown group -> false, before any device is needed (src/pick_ik_plugin.cpp:36-40)
        CHECK(!base.initialize(node, model, "no_such_group", "panda_link0", {"panda_hand"}, 0.1));
        // unknown tip link -> std::invalid_argument (:65-67)
        bool threw = false;
        try {
            base.initialize(node, model, "panda_arm", "panda_link0", {"no_such_link"}, 0.1);
        } catch (std::invalid_argument const&) {
            threw = true;
        }
        CHECK(threw);
        // no GPU: initialize fails loudly (no CPU fallback), logs the reason
        pick_ik::PickIKPlugin p2;
        bool const ok = p2.initialize(node, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1);
        if (!ok) {
            bool logged = false;
            for (auto const& l : rclcpp::stub_log()) logged = logged || l.find("no HIP device") != std::string::npos;
            CHECK(logged);
            std::printf("shim checks without a device OK (initialize refused: no HIP device)\n");
        } else {
            std::printf("shim checks without a device OK (a device is present)\n");
        }
        return 0;
    }

    CHECK(base.initialize(node, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1));
    CHECK(base.getJointNames().size() == 7 && base.getJointNames()[0] == "panda_joint1");
    CHECK(base.getLinkNames() == std::vector<std::string>{"panda_hand"});
    CHECK(base.getBaseFrame() == "panda_link0");
    {   // the stubs of the reference: getPositionFK / getPositionIK return false (:300-312)
        std::vector<geometry_msgs::msg::Pose> poses;
        std::vector<double> sol;
        moveit_msgs::msg::MoveItErrorCodes ec;
        CHECK(!base.getPositionFK({"panda_hand"}, std::vector<double>(7, 0.0), poses));
        CHECK(!base.getPositionIK(geometry_msgs::msg::Pose(), std::vector<double>(7, 0.0), sol, ec));
    }

    // target = tip pose at a known joint vector, expressed in the group's base frame
    std::vector<double> const home = {0, -M_PI / 4, 0, -3 * M_PI / 4, 0, M_PI / 2, M_PI / 4};
    std::vector<double> const actual = {0.1, -M_PI / 4 - 0.1, 0.1, -3 * M_PI / 4 - 0.1, 0.1, M_PI / 2 - 0.1, M_PI / 4 + 0.1};
    auto const jmg = model.getJointModelGroup("panda_arm");
    auto tip_in_base = [&](std::vector<double> const& q) {
        mc::RobotState st(mc::RobotModelConstPtr(&model, [](mc::RobotModel const*) {}));
        st.setJointGroupPositions(jmg, q);
        return st.getGlobalLinkTransform("panda_link0").inverse() * st.getGlobalLinkTransform("panda_hand");
    };
    geometry_msgs::msg::Pose const target = pose_of(tip_in_base(actual));
    auto reached = [&](std::vector<double> const& q, double pos_tol) {
        auto const T = tip_in_base(q);
        double const dx = T.translation().x() - target.position.x, dy = T.translation().y() - target.position.y,
                     dz = T.translation().z() - target.position.z;
        return std::sqrt(dx * dx + dy * dy + dz * dz) <= pos_tol;
    };
    moveit_msgs::msg::MoveItErrorCodes ec;
    std::vector<double> sol;

    // ---- global mode, defaults; integer parameters set the way a yaml sets them (int64) ----
    node->set_parameter(ns + "memetic_population_size", int64_t{32});
    CHECK(base.searchPositionIK(target, home, 1.0, sol, ec));
    CHECK(ec.val == ec.SUCCESS && sol.size() == 7 && reached(sol, 1.1e-3));
    for (size_t i = 0; i < 7; ++i) CHECK(sol[i] >= model.getJointModelGroup("panda_arm")->getActiveJointModels()[i]->getVariableBounds()[0].min_position_ - 1e-12);

    // ---- solution callback: called on success with ik_poses.front(); a callback that rejects
    //      (collision checker) makes the plugin restart from a random state (:269-291) ----
    int calls = 0;
    auto rejecting = [&](geometry_msgs::msg::Pose const& p, std::vector<double> const& q, moveit_msgs::msg::MoveItErrorCodes& e) {
        ++calls;
        if (p.position.x != target.position.x || q.size() != 7) e.val = -1000;
        else if (calls < 3) e.val = e.FAILURE; // first two candidates "in collision"
    };
    CHECK(base.searchPositionIK(target, home, 30.0, sol, rejecting, ec));
    CHECK(calls == 3 && ec.val == ec.SUCCESS && reached(sol, 1.1e-3));

    // ---- failure: unreachable target -> false, NO_IK_SOLUTION, solution = ik_seed_state, several attempts ----
    geometry_msgs::msg::Pose far = target;
    far.position.x = 2.5;
    far.position.z = 2.0;
    node->set_parameter(ns + "memetic_max_generations", int64_t{3});
    {
        pick_ik::PickIKPlugin quick; // (parameters are declared per node: a fresh node for the new values)
        auto node2 = std::make_shared<rclcpp::Node>();
        node2->set_parameter(ns + "memetic_max_generations", int64_t{3});
        node2->set_parameter(ns + "memetic_num_threads", int64_t{2});           // species on the GPU
        node2->set_parameter(ns + "memetic_stop_on_first_solution", false);
        CHECK(quick.initialize(node2, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1));
        CHECK(!quick.searchPositionIK({far}, home, 0.05, {}, sol, kinematics::KinematicsBase::IKCallbackFn(), ec));
        CHECK(ec.val == ec.NO_IK_SOLUTION && sol == home);
        // approximate mode on the unreachable target: the solver returns its best, the plugin's gate
        // (regular frame tests, :243-248) rejects it
        kinematics::KinematicsQueryOptions approx;
        approx.return_approximate_solution = true;
        CHECK(!quick.searchPositionIK({far}, home, 0.05, {}, sol, kinematics::KinematicsBase::IKCallbackFn(), ec, approx));
        CHECK(ec.val == ec.NO_IK_SOLUTION && sol == home);
        // reachable target in approximate mode passes the gate ...
        auto node3 = std::make_shared<rclcpp::Node>();
        pick_ik::PickIKPlugin ap;
        CHECK(ap.initialize(node3, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1));
        CHECK(ap.searchPositionIK({target}, home, 5.0, {}, sol, kinematics::KinematicsBase::IKCallbackFn(), ec, approx));
        CHECK(ec.val == ec.SUCCESS && reached(sol, 1.1e-3));
        // ... unless a joint moved further than approximate_solution_joint_threshold (:251-259)
        auto node4 = std::make_shared<rclcpp::Node>();
        node4->set_parameter(ns + "approximate_solution_joint_threshold", 0.01);
        pick_ik::PickIKPlugin ap2;
        CHECK(ap2.initialize(node4, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1));
        CHECK(!ap2.searchPositionIK({target}, home, 0.05, {}, sol, kinematics::KinematicsBase::IKCallbackFn(), ec, approx));
        CHECK(ec.val == ec.NO_IK_SOLUTION && sol == home);
    }

    // ---- a host IKCostFn inside the search (the default, cost_fn_mode = search): one more goal of weight 1 per
    //      pose in cost_fn and under cost_threshold^2 in solution_fn, as in the reference
    //      (src/pick_ik_plugin.cpp:130-135) -- solved on the host (pikamd_solve_batch_host) ----
    {
        using CostFn = kinematics::KinematicsBase::IKCostFn;
        long n_calls = 0;
        CostFn zero = [&](geometry_msgs::msg::Pose const&, mc::RobotState const&, mc::JointModelGroup const*,
                          std::vector<double> const&) { ++n_calls; return 0.0; };
        CHECK(base.searchPositionIK({target}, home, 30.0, {}, sol, kinematics::KinematicsBase::IKCallbackFn(), zero, ec));
        CHECK(ec.val == ec.SUCCESS && reached(sol, 1.1e-3));
        CHECK(n_calls > 100); // every cost evaluation of the search calls it
        CostFn big = [](geometry_msgs::msg::Pose const&, mc::RobotState const&, mc::JointModelGroup const*,
                        std::vector<double> const&) { return 1.0; };
        CHECK(!base.searchPositionIK({target}, home, 0.05, {}, sol, kinematics::KinematicsBase::IKCallbackFn(), big, ec));
        CHECK(ec.val == ec.NO_IK_SOLUTION && sol == home);
        // a cost only a GUIDED search gets under the threshold: joint 2 within 1.4e-3 rad of 0.5 (0.5 d^2 < 1e-6).
        // The Panda is redundant, so such solutions exist; a finished candidate of a search that does not know the
        // cost lands in that window with probability ~1e-3 (cost_fn_mode = rank needs hundreds of candidates).
        double const want2 = 0.5;
        bool state_ok = true;
        CostFn narrow = [&](geometry_msgs::msg::Pose const&, mc::RobotState const& st, mc::JointModelGroup const* jmg,
                            std::vector<double> const& seed_state) {
            std::vector<double> q;
            st.copyJointGroupPositions(jmg, q);
            state_ok = state_ok && q.size() == 7 && seed_state == home;
            return 0.5 * (q[2] - want2) * (q[2] - want2);
        };
        CHECK(base.searchPositionIK({target}, home, 60.0, {}, sol, kinematics::KinematicsBase::IKCallbackFn(), narrow, ec));
        CHECK(ec.val == ec.SUCCESS && reached(sol, 1.1e-3) && std::abs(sol[2] - want2) < 1.5e-3 && state_ok);
        std::printf("IKCostFn inside the search: q2 = %.5f (asked %.1f), %ld callback evaluations for the zero cost\n", sol[2], want2, n_calls);
    }

    // ---- cost_fn_mode = rank: the GPU proposes `cost_fn_candidates` solutions per attempt, the callback
    //      re-scores them (ranks, and gates at cost_threshold^2 like every goal of the reference) ----
    {
        auto noder = std::make_shared<rclcpp::Node>();
        noder->set_parameter(ns + "cost_fn_mode", std::string("rank"));
        pick_ik::PickIKPlugin ranked;
        CHECK(ranked.initialize(noder, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1));
        auto& base = ranked; // (the checks below were written for the plugin they now configure)
        using CostFn = kinematics::KinematicsBase::IKCostFn;
        int n_calls = 0;
        // (1) a callback that is always 0: every candidate passes, the result is a plain solution;
        //     called once per pose for at most cost_fn_candidates candidates per attempt
        CostFn zero = [&](geometry_msgs::msg::Pose const&, mc::RobotState const&, mc::JointModelGroup const*,
                          std::vector<double> const&) { ++n_calls; return 0.0; };
        CHECK(base.searchPositionIK({target}, home, 5.0, {}, sol, kinematics::KinematicsBase::IKCallbackFn(), zero, ec));
        CHECK(ec.val == ec.SUCCESS && reached(sol, 1.1e-3) && n_calls >= 1 && n_calls <= 32);
        // (2) a callback above cost_threshold^2 (1e-6) rejects every candidate: no solution
        CostFn big = [](geometry_msgs::msg::Pose const&, mc::RobotState const&, mc::JointModelGroup const*,
                        std::vector<double> const&) { return 1.0; };
        CHECK(!base.searchPositionIK({target}, home, 0.05, {}, sol, kinematics::KinematicsBase::IKCallbackFn(), big, ec));
        CHECK(ec.val == ec.NO_IK_SOLUTION && sol == home);
        // (3) a preference (joint 0 near 1.2 rad, small enough to stay under the threshold): the
        //     returned solutions sit closer to it than those of the same queries without the callback,
        //     and the callback sees the candidate through the RobotState it is handed
        double const want = 1.2;
        bool state_ok = true;
        CostFn prefer = [&](geometry_msgs::msg::Pose const&, mc::RobotState const& st, mc::JointModelGroup const* jmg,
                            std::vector<double> const& seed_state) {
            std::vector<double> q;
            st.copyJointGroupPositions(jmg, q);
            state_ok = state_ok && q.size() == 7 && seed_state == home;
            return 1.0e-8 * (q[0] - want) * (q[0] - want);
        };
        double with = 0.0, without = 0.0;
        for (int rep = 0; rep < 6; ++rep) {
            CHECK(base.searchPositionIK({target}, home, 5.0, {}, sol, kinematics::KinematicsBase::IKCallbackFn(), prefer, ec));
            CHECK(reached(sol, 1.1e-3));
            with += std::abs(sol[0] - want);
            CHECK(base.searchPositionIK(target, home, 5.0, sol, ec));
            without += std::abs(sol[0] - want);
        }
        CHECK(state_ok);
        CHECK(with < without);
        std::printf("IKCostFn preference: mean |q0 - %.1f| %.3f with the callback, %.3f without\n", want, with / 6, without / 6);
    }

    // ---- seed outside the joint limits: warning + random valid start (:156-163); the "zero seed"
    //      of the reference tests violates joint 4's limits ----
    {
        size_t const before = rclcpp::stub_log().size();
        CHECK(base.searchPositionIK(target, std::vector<double>(7, 0.0), 30.0, sol, ec));
        CHECK(ec.val == ec.SUCCESS && reached(sol, 1.1e-3));
        bool warned = false;
        for (size_t i = before; i < rclcpp::stub_log().size(); ++i)
            warned = warned || rclcpp::stub_log()[i].find("exceeds joint limits") != std::string::npos;
        CHECK(warned);
    }

    // ---- the remaining overloads forward (:314-401) ----
    CHECK(base.searchPositionIK(target, home, 5.0, std::vector<double>(7, 0.1), sol, ec) && reached(sol, 1.1e-3));
    calls = 10;
    CHECK(base.searchPositionIK(target, home, 5.0, std::vector<double>(7, 0.1), sol, rejecting, ec) && calls == 11);
    calls = 10;
    CHECK(base.searchPositionIK(std::vector<geometry_msgs::msg::Pose>{target}, home, 5.0, std::vector<double>{}, sol,
                                rejecting, ec) && calls == 11);
    CHECK(!base.searchPositionIK(std::vector<geometry_msgs::msg::Pose>{target, target}, home, 1.0, std::vector<double>{},
                                 sol, kinematics::KinematicsBase::IKCallbackFn(), ec)); // 2 poses, 1 tip frame

    // ---- local mode (mode = "local"), the reference's perturbed-home case (tests/ik_tests.cpp:272-292) ----
    {
        auto node5 = std::make_shared<rclcpp::Node>();
        node5->set_parameter(ns + "mode", std::string("local"));
        node5->set_parameter(ns + "position_threshold", 1e-4);
        node5->set_parameter(ns + "gd_max_iters", int64_t{100});
        pick_ik::PickIKPlugin local;
        CHECK(local.initialize(node5, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1));
        CHECK(local.searchPositionIK(target, home, 1.0, sol, ec));
        for (size_t i = 0; i < 7; ++i) CHECK(std::fabs(sol[i] - actual[i]) < 0.025);
        // an unknown mode is an error (:204-207)
        auto node6 = std::make_shared<rclcpp::Node>();
        node6->set_parameter(ns + "mode", std::string("sideways"));
        pick_ik::PickIKPlugin bad;
        CHECK(bad.initialize(node6, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1));
        CHECK(!bad.searchPositionIK(target, home, 1.0, sol, ec));
        // a wrongly typed parameter (a yaml that says memetic_population_size: 32.0) surfaces as
        // rclcpp's type error, it is not coerced
        auto node7 = std::make_shared<rclcpp::Node>();
        node7->set_parameter(ns + "memetic_population_size", 32.0);
        pick_ik::PickIKPlugin typed;
        CHECK(typed.initialize(node7, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1));
        // ... but it never leaves searchPositionIK as an exception into MoveIt's planning thread: the
        // query fails, the seed is returned (the reference reports through the return value as well,
        // src/pick_ik_plugin.cpp:204-217)
        bool threw = false, ok = true;
        try {
            ok = typed.searchPositionIK(target, home, 1.0, sol, ec);
        } catch (std::exception const&) {
            threw = true;
        }
        CHECK(!threw && !ok && ec.val == ec.NO_IK_SOLUTION && sol == home);
        // the same for a parameter combination the library refuses (population <= elites)
        auto node8 = std::make_shared<rclcpp::Node>();
        node8->set_parameter(ns + "memetic_population_size", int64_t{4});
        node8->set_parameter(ns + "memetic_elite_size", int64_t{4});
        pick_ik::PickIKPlugin refused;
        CHECK(refused.initialize(node8, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1));
        threw = false;
        try {
            ok = refused.searchPositionIK(target, home, 1.0, sol, ec);
        } catch (std::exception const&) {
            threw = true;
        }
        CHECK(!threw && !ok && ec.val == ec.NO_IK_SOLUTION && sol == home);
        // a CPU-style thread count (32 threads, 4 elites: more species than a wavefront holds) is clamped
        auto node9 = std::make_shared<rclcpp::Node>();
        node9->set_parameter(ns + "memetic_num_threads", int64_t{32});
        node9->set_parameter(ns + "memetic_population_size", int64_t{32});
        pick_ik::PickIKPlugin many;
        CHECK(many.initialize(node9, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1));
        CHECK(many.searchPositionIK(target, home, 5.0, sol, ec));
        CHECK(ec.val == ec.SUCCESS && reached(sol, 1.1e-3));
    }

    // ---- goal frames: base_frame != model frame, and a base frame that is not the chain's root ----
    {
        pick_ik::PickIKPlugin w;
        auto node8 = std::make_shared<rclcpp::Node>();
        CHECK(w.initialize(node8, model, "panda_arm", "world", {"panda_hand"}, 0.1));
        mc::RobotState st(mc::RobotModelConstPtr(&model, [](mc::RobotModel const*) {}));
        st.setJointGroupPositions(jmg, actual);
        geometry_msgs::msg::Pose const in_world = pose_of(st.getGlobalLinkTransform("panda_hand"));
        CHECK(w.searchPositionIK(in_world, home, 5.0, sol, ec));
        CHECK(reached(sol, 1.1e-3)); // transform_poses_to_frames (src/robot.cpp:169-181)
    }
    // ---- a mobile base: planar virtual joint (x, y, theta) + two arm joints = 5 variables ----
    {
        mc::RobotModel mobile;
        mobile.add_root("odom");
        mc::VariableBounds xy;
        xy.position_bounded_ = true;
        xy.min_position_ = -1.0;
        xy.max_position_ = 1.0;
        xy.max_velocity_ = 0.5;
        mobile.add_link("base", "odom", "virtual", mc::JointModel::PLANAR, origin(0.1, 0.2, 0.05, 0, 0, 0.3),
                        Eigen::Vector3d(0, 0, 1), xy);
        mc::VariableBounds rb;
        rb.position_bounded_ = true;
        rb.min_position_ = -2.5;
        rb.max_position_ = 2.5;
        rb.max_velocity_ = 1.0;
        mobile.add_link("l1", "base", "shoulder", mc::JointModel::REVOLUTE, origin(0.0, 0, 0.4, 0, 0, 0),
                        Eigen::Vector3d(0, 1, 0), rb);
        mobile.add_link("l2", "l1", "elbow", mc::JointModel::REVOLUTE, origin(0.35, 0, 0, 0, 0, 0),
                        Eigen::Vector3d(0, 1, 0), rb);
        mobile.add_link("tool", "l2", "tool_fixed", mc::JointModel::FIXED, origin(0.3, 0, 0, 0, 0, 0),
                        Eigen::Vector3d(0, 0, 1), {});
        mobile.add_group("mobile_arm", {"virtual", "shoulder", "elbow"});
        auto nodem = std::make_shared<rclcpp::Node>();
        std::string const nsm = "robot_description_kinematics.mobile_arm.";
        nodem->set_parameter(nsm + "memetic_population_size", int64_t{32});
        nodem->set_parameter(nsm + "orientation_threshold", 0.01);
        pick_ik::PickIKPlugin mp;
        CHECK(mp.initialize(nodem, mobile, "mobile_arm", "odom", {"tool"}, 0.1));
        CHECK(mp.getJointNames().size() == 3); // joints; the joint vector has 5 variables
        std::vector<double> const qm = {0.3, -0.2, 0.5, 0.4, -0.8};
        auto const jm = mobile.getJointModelGroup("mobile_arm");
        mc::RobotState stm(mc::RobotModelConstPtr(&mobile, [](mc::RobotModel const*) {}));
        stm.setJointGroupPositions(jm, qm);
        geometry_msgs::msg::Pose const goal_m = pose_of(stm.getGlobalLinkTransform("tool"));
        std::vector<double> solm;
        CHECK(mp.searchPositionIK(goal_m, std::vector<double>(5, 0.0), 30.0, solm, ec));
        CHECK(ec.val == ec.SUCCESS && solm.size() == 5);
        mc::RobotState chk(mc::RobotModelConstPtr(&mobile, [](mc::RobotModel const*) {}));
        chk.setJointGroupPositions(jm, solm);
        auto const Tm = chk.getGlobalLinkTransform("tool");
        double const ex = Tm.translation().x() - goal_m.position.x, ey = Tm.translation().y() - goal_m.position.y,
                     ez = Tm.translation().z() - goal_m.position.z;
        CHECK(std::sqrt(ex * ex + ey * ey + ez * ez) <= 1.1e-3);
    }
    // ---- a moving joint of the path that is not in the group sits at its DEFAULT position (the reference's FK
    //      state is setToDefaultValues() + the group's positions, src/fk_moveit.cpp:15-22): here a lift whose
    //      range excludes zero, default = the middle of the range ----
    {
        mc::RobotModel lifted;
        lifted.add_root("world");
        mc::VariableBounds lb;
        lb.position_bounded_ = true;
        lb.min_position_ = 0.4;
        lb.max_position_ = 0.8; // default 0.6
        lb.max_velocity_ = 0.2;
        lifted.add_link("column", "world", "lift", mc::JointModel::PRISMATIC, origin(0, 0, 0.1, 0, 0, 0),
                        Eigen::Vector3d(0, 0, 1), lb);
        mc::VariableBounds rb;
        rb.position_bounded_ = true;
        rb.min_position_ = -2.5;
        rb.max_position_ = 2.5;
        rb.max_velocity_ = 1.0;
        lifted.add_link("a1", "column", "j1", mc::JointModel::REVOLUTE, origin(0, 0, 0.2, 0, 0, 0), Eigen::Vector3d(0, 0, 1), rb);
        lifted.add_link("a2", "a1", "j2", mc::JointModel::REVOLUTE, origin(0.3, 0, 0, 0, 0, 0), Eigen::Vector3d(0, 1, 0), rb);
        lifted.add_link("a3", "a2", "j3", mc::JointModel::REVOLUTE, origin(0.3, 0, 0, 0, 0, 0), Eigen::Vector3d(0, 1, 0), rb);
        lifted.add_link("tool", "a3", "tool_fixed", mc::JointModel::FIXED, origin(0.2, 0, 0, 0, 0, 0), Eigen::Vector3d(0, 0, 1), {});
        lifted.add_group("arm", {"j1", "j2", "j3"}); // the lift is NOT in the group
        auto nodel = std::make_shared<rclcpp::Node>();
        std::string const nsl = "robot_description_kinematics.arm.";
        nodel->set_parameter(nsl + "memetic_population_size", int64_t{32});
        nodel->set_parameter(nsl + "rotation_scale", 0.0); // three joints: position only
        pick_ik::PickIKPlugin lp;
        CHECK(lp.initialize(nodel, lifted, "arm", "world", {"tool"}, 0.1));
        auto const jl = lifted.getJointModelGroup("arm");
        mc::RobotState stl(mc::RobotModelConstPtr(&lifted, [](mc::RobotModel const*) {}));
        stl.setToDefaultValues();
        stl.setJointGroupPositions(jl, {0.4, -0.5, 0.9});
        auto const Tl = stl.getGlobalLinkTransform("tool");
        // (the lift contributes its default 0.6 m: a tool at rest would sit at z = 0.1 + 0.6 + 0.2)
        CHECK(Tl.translation().z() > 0.5);
        std::vector<double> soll;
        CHECK(lp.searchPositionIK(pose_of(Tl), {0.0, 0.0, 0.0}, 30.0, soll, ec));
        mc::RobotState chk(mc::RobotModelConstPtr(&lifted, [](mc::RobotModel const*) {}));
        chk.setToDefaultValues();
        chk.setJointGroupPositions(jl, soll);
        auto const Tc = chk.getGlobalLinkTransform("tool");
        double const lx = Tc.translation().x() - Tl.translation().x(), ly = Tc.translation().y() - Tl.translation().y(),
                     lz = Tc.translation().z() - Tl.translation().z();
        CHECK(std::sqrt(lx * lx + ly * ly + lz * lz) <= 1.1e-3);
    }
    // ---- a mimic joint on the path: one that FOLLOWS its master (the reference's FK moves it: setJointGroupPositions
    //      -> updateMimicJoints, src/fk_moveit.cpp:22) becomes a mimic step of the chain -- the solutions hold with
    //      the mimic joint moved by RobotState --; a constant one (multiplier 0) is folded at its offset ----
    for (double factor : {-0.8, 0.0}) {
        mc::RobotModel mim;
        mim.add_root("world");
        mc::VariableBounds rb;
        rb.position_bounded_ = true;
        rb.min_position_ = -2.5;
        rb.max_position_ = 2.5;
        rb.max_velocity_ = 1.0;
        mim.add_link("a1", "world", "j1", mc::JointModel::REVOLUTE, origin(0, 0, 0.2, 0, 0, 0), Eigen::Vector3d(0, 0, 1), rb);
        auto* follower = mim.add_link("a2", "a1", "j2_mimic", mc::JointModel::REVOLUTE, origin(0.3, 0, 0, 0, 0, 0),
                                      Eigen::Vector3d(0, 1, 0), rb);
        mim.add_link("a3", "a2", "j3", mc::JointModel::REVOLUTE, origin(0.3, 0, 0, 0, 0, 0), Eigen::Vector3d(0, 1, 0), rb);
        mim.add_link("tool", "a3", "tool_fixed", mc::JointModel::FIXED, origin(0.2, 0, 0, 0, 0, 0), Eigen::Vector3d(0, 0, 1), {});
        auto* fj = const_cast<mc::JointModel*>(follower->getParentJointModel());
        fj->mimic_ = mim.getLinkModel("a1")->getParentJointModel();
        fj->mimic_factor_ = factor;
        fj->mimic_offset_ = 0.25;
        mim.add_group("arm", {"j1", "j3"});
        auto nodem2 = std::make_shared<rclcpp::Node>();
        nodem2->set_parameter(std::string("robot_description_kinematics.arm.") + "memetic_population_size", int64_t{32});
        nodem2->set_parameter(std::string("robot_description_kinematics.arm.") + "rotation_scale", 0.0);
        pick_ik::PickIKPlugin mp2;
        CHECK(mp2.initialize(nodem2, mim, "arm", "world", {"tool"}, 0.1));
        auto const jg = mim.getJointModelGroup("arm");
        mc::RobotState want(mc::RobotModelConstPtr(&mim, [](mc::RobotModel const*) {}));
        want.setToDefaultValues();
        want.setJointGroupPositions(jg, {0.7, -0.4}); // (the state moves j2_mimic to factor * j1 + 0.25)
        auto const Tw = want.getGlobalLinkTransform("tool");
        std::vector<double> solm;
        CHECK(mp2.searchPositionIK(pose_of(Tw), {0.0, 0.0}, 30.0, solm, ec));
        mc::RobotState got(mc::RobotModelConstPtr(&mim, [](mc::RobotModel const*) {}));
        got.setToDefaultValues();
        got.setJointGroupPositions(jg, solm);
        auto const Tg = got.getGlobalLinkTransform("tool");
        double const mx = Tg.translation().x() - Tw.translation().x(), my = Tg.translation().y() - Tw.translation().y(),
                     mz = Tg.translation().z() - Tw.translation().z();
        CHECK(std::sqrt(mx * mx + my * my + mz * mz) <= 1.1e-3);
    }
    // ---- the caller's timeout bounds an attempt: a generation budget far beyond it is cut to what fits ----
    {
        auto nodet = std::make_shared<rclcpp::Node>();
        nodet->set_parameter(ns + "memetic_max_generations", int64_t{200000}); // ~20 s of generations
        nodet->set_parameter(ns + "memetic_population_size", int64_t{32});
        pick_ik::PickIKPlugin timed;
        CHECK(timed.initialize(nodet, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1));
        geometry_msgs::msg::Pose out_of_reach = target;
        out_of_reach.position.x = 2.5;
        out_of_reach.position.z = 2.0;
        CHECK(!timed.searchPositionIK(out_of_reach, home, 0.02, sol, ec)); // (the first query also calibrates)
        auto const t0 = std::chrono::steady_clock::now();
        CHECK(!timed.searchPositionIK(out_of_reach, home, 0.02, sol, ec));
        double const took = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        CHECK(ec.val == ec.NO_IK_SOLUTION && sol == home);
        CHECK(took < 0.2); // 20 ms asked for; one attempt may end a little late, not 20 s late
        // and a reachable target is still found under a generous timeout
        CHECK(timed.searchPositionIK(target, home, 1.0, sol, ec));
        CHECK(ec.val == ec.SUCCESS && reached(sol, 1.1e-3));
    }
    // ---- a free-flying base: floating virtual joint (7 variables) + two arm joints = 9 variables ----
    {
        mc::RobotModel flyer;
        flyer.add_root("world");
        mc::VariableBounds tb;
        tb.position_bounded_ = true;
        tb.min_position_ = -0.5;
        tb.max_position_ = 0.5;
        tb.max_velocity_ = 0.5;
        flyer.add_link("body", "world", "free", mc::JointModel::FLOATING, origin(0.1, -0.1, 0.2, 0.1, 0.2, -0.3),
                       Eigen::Vector3d(0, 0, 1), tb);
        mc::VariableBounds rb;
        rb.position_bounded_ = true;
        rb.min_position_ = -2.5;
        rb.max_position_ = 2.5;
        rb.max_velocity_ = 1.0;
        flyer.add_link("l1", "body", "shoulder", mc::JointModel::REVOLUTE, origin(0.0, 0, 0.2, 0, 0, 0),
                       Eigen::Vector3d(0, 1, 0), rb);
        flyer.add_link("l2", "l1", "elbow", mc::JointModel::REVOLUTE, origin(0.3, 0, 0, 0, 0, 0),
                       Eigen::Vector3d(0, 0, 1), rb);
        flyer.add_link("tool", "l2", "tool_fixed", mc::JointModel::FIXED, origin(0.25, 0, 0, 0, 0, 0),
                       Eigen::Vector3d(0, 0, 1), {});
        flyer.add_group("flying_arm", {"free", "shoulder", "elbow"});
        auto nodef = std::make_shared<rclcpp::Node>();
        std::string const nsf = "robot_description_kinematics.flying_arm.";
        nodef->set_parameter(nsf + "memetic_population_size", int64_t{32});
        nodef->set_parameter(nsf + "orientation_threshold", 0.01);
        pick_ik::PickIKPlugin fp;
        CHECK(fp.initialize(nodef, flyer, "flying_arm", "world", {"tool"}, 0.1));
        CHECK(fp.getJointNames().size() == 3); // joints; the joint vector has 9 variables
        double const n = std::sqrt(0.1 * 0.1 + 0.2 * 0.2 + 0.3 * 0.3 + 0.9 * 0.9);
        std::vector<double> const qf = {0.2, -0.1, 0.3, 0.1 / n, -0.2 / n, 0.3 / n, 0.9 / n, 0.4, -0.8};
        auto const jf = flyer.getJointModelGroup("flying_arm");
        mc::RobotState stf(mc::RobotModelConstPtr(&flyer, [](mc::RobotModel const*) {}));
        stf.setJointGroupPositions(jf, qf);
        geometry_msgs::msg::Pose const goal_f = pose_of(stf.getGlobalLinkTransform("tool"));
        std::vector<double> const seed_f = {0, 0, 0, 0, 0, 0, 1, 0, 0};
        std::vector<double> solf;
        CHECK(fp.searchPositionIK(goal_f, seed_f, 30.0, solf, ec));
        CHECK(ec.val == ec.SUCCESS && solf.size() == 9);
        mc::RobotState chk(mc::RobotModelConstPtr(&flyer, [](mc::RobotModel const*) {}));
        chk.setJointGroupPositions(jf, solf);
        auto const Tf = chk.getGlobalLinkTransform("tool");
        double const ex = Tf.translation().x() - goal_f.position.x, ey = Tf.translation().y() - goal_f.position.y,
                     ez = Tf.translation().z() - goal_f.position.z;
        CHECK(std::sqrt(ex * ex + ey * ey + ez * ez) <= 1.1e-3);
    }
    // ---- the DEFAULT arithmetic (exact; no parameter set): what the plugin returns IS the reference algorithm's joint
    //      vector (src/pick_ik_plugin.cpp:182-188 hands back ik_memetic's / ik_gradient's).  The last solve the
    //      plugin handed to the library (Solver::last_call) is replayed through the CPU oracle (oracle/pik_oracle.c,
    //      math mode "fma" = the product library's exact kernels) on the chain the plugin extracted from the robot
    //      model: the reference's perturbed-home case in local mode (tests/ik_tests.cpp:272-292) and one memetic
    //      query with a fixed rng_seed, every bit of the seven joint values equal ----
    {
        auto const replay = [&](pick_ik::PickIKPlugin const& pl, std::vector<double>& out, int& status) -> int {
            auto const& ch = pl.chain_description();
            CHECK(ch.tips.size() == 1 && ch.mimics.empty());
            auto const& tp = ch.tips[0];
            int const dof = static_cast<int>(ch.variables.size());
            CHECK(static_cast<int>(tp.joints.size()) == dof);
            std::vector<double> o(6 * dof), ax(3 * dof), lo(dof), hi(dof), vm(dof), tip(6);
            std::vector<int32_t> jt(dof, 0);
            std::vector<uint8_t> bd(dof);
            for (int j = 0; j < dof; ++j) {
                for (int k = 0; k < 3; ++k) {
                    o[6 * j + k] = tp.joints[j].origin_xyz[k];
                    o[6 * j + 3 + k] = tp.joints[j].origin_rpy[k];
                    ax[3 * j + k] = tp.joints[j].axis[k];
                }
                CHECK(!tp.joints[j].prismatic && tp.joints[j].planar == 0 && tp.joints[j].floating == 0);
                lo[j] = ch.variables[j].min;
                hi[j] = ch.variables[j].max;
                vm[j] = ch.variables[j].max_velocity;
                bd[j] = ch.variables[j].bounded ? 1 : 0;
            }
            for (int k = 0; k < 3; ++k) tip[k] = tp.tip_xyz[k], tip[3 + k] = tp.tip_rpy[k];
            pko_chain* oc = pko_chain_create(dof, o.data(), ax.data(), jt.data(), tip.data(), lo.data(), hi.data(), vm.data(), bd.data());
            CHECK(oc != nullptr);
            auto const& rec = pl.solver().last_call();
            pikamd_params const& a = rec.params;
            pko_params b;
            pko_default_params(&b);
            b.mode = a.mode;
            b.gd_step_size = a.gd_step_size;
            b.gd_max_iters = a.gd_max_iters;
            b.gd_min_cost_delta = a.gd_min_cost_delta;
            b.position_threshold = a.position_threshold;
            b.orientation_threshold = a.orientation_threshold;
            b.cost_threshold = a.cost_threshold;
            b.position_scale = a.position_scale;
            b.rotation_scale = a.rotation_scale;
            b.center_joints_weight = a.center_joints_weight;
            b.avoid_joint_limits_weight = a.avoid_joint_limits_weight;
            b.minimal_displacement_weight = a.minimal_displacement_weight;
            b.stop_optimization_on_valid_solution = a.stop_optimization_on_valid_solution;
            b.memetic_num_threads = a.memetic_num_threads;
            b.memetic_stop_on_first_solution = a.memetic_stop_on_first_solution;
            b.memetic_population_size = a.memetic_population_size;
            b.memetic_elite_size = a.memetic_elite_size;
            b.memetic_wipeout_fitness_tol = a.memetic_wipeout_fitness_tol;
            b.memetic_max_generations = a.memetic_max_generations;
            b.memetic_gd_max_iters = a.memetic_gd_max_iters;
            b.return_approximate_solution = a.return_approximate_solution;
            out.assign(static_cast<size_t>(dof), 0.0);
            int32_t st = 0;
            pko_set_math_mode(2); // "fma": the arithmetic of the product library's exact kernels
            CHECK(pko_solve_batch_guess(oc, &b, 1, rec.goal_pos_quat.data(), rec.seed.data(), rec.initial_guess.data(),
                                        rec.rng_seed, rec.problem_offset, out.data(), &st, nullptr, nullptr, 1) == 0);
            status = st;
            pko_chain_destroy(oc);
            return 0;
        };
        auto const same_bits = [](std::vector<double> const& a, std::vector<double> const& b) {
            return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(double)) == 0;
        };
        std::vector<double> want;
        int st = 0;
        // local mode, perturbed home
        auto nodex = std::make_shared<rclcpp::Node>();
        nodex->set_parameter(ns + "mode", std::string("local")); // (no "arithmetic" parameter: the default)
        nodex->set_parameter(ns + "position_threshold", 1e-4);
        nodex->set_parameter(ns + "gd_max_iters", int64_t{100});
        pick_ik::PickIKPlugin xl;
        CHECK(xl.initialize(nodex, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1));
        xl.solver().set_record_last_call(true); // (diagnostics: what the plugin hands to the library, for the replay)
        CHECK(xl.searchPositionIK(target, home, 5.0, sol, ec));
        for (size_t i = 0; i < 7; ++i) CHECK(std::fabs(sol[i] - actual[i]) < 0.025);
        CHECK(replay(xl, want, st) == 0);
        CHECK(st == 1 && same_bits(sol, want));
        // global mode, a fixed random stream
        auto nodeg = std::make_shared<rclcpp::Node>();
        nodeg->set_parameter(ns + "memetic_population_size", int64_t{32}); // (the default arithmetic again)
        nodeg->set_parameter(ns + "rng_seed", int64_t{20260929});
        pick_ik::PickIKPlugin xg;
        CHECK(xg.initialize(nodeg, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1));
        xg.solver().set_record_last_call(true);
        std::vector<double> first;
        CHECK(xg.searchPositionIK(target, std::vector<double>(7, 0.3), 30.0, first, ec));
        CHECK(ec.val == ec.SUCCESS && reached(first, 1.1e-3));
        CHECK(replay(xg, want, st) == 0);
        CHECK(st == 1 && same_bits(first, want));
        CHECK(xg.searchPositionIK(target, std::vector<double>(7, 0.3), 30.0, sol, ec) && same_bits(sol, first)); // reproducible
        // ... "exact" named explicitly is the same query, "fast" is the opt-in flavour (a valid solution, other bits)
        nodeg->set_parameter(ns + "arithmetic", std::string("exact"));
        CHECK(xg.searchPositionIK(target, std::vector<double>(7, 0.3), 30.0, sol, ec) && same_bits(sol, first));
        nodeg->set_parameter(ns + "arithmetic", std::string("fast"));
        CHECK(xg.searchPositionIK(target, std::vector<double>(7, 0.3), 30.0, sol, ec));
        CHECK(ec.val == ec.SUCCESS && reached(sol, 1.1e-3));
        // ... and an unknown arithmetic is refused
        auto nodeb = std::make_shared<rclcpp::Node>();
        nodeb->set_parameter(ns + "arithmetic", std::string("sloppy"));
        pick_ik::PickIKPlugin xb;
        CHECK(xb.initialize(nodeb, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1));
        CHECK(!xb.searchPositionIK(target, home, 1.0, sol, ec));
        std::printf("default arithmetic (exact): local and memetic query identical to the oracle\n");
    }
    // ---- a host cost function and a short timeout: the search runs on the host, and the host loops read the clock
    //      as the reference's do (in front of every generation / descent step) -- a 50 ms budget is 50 ms ----
    {
        using CostFn = kinematics::KinematicsBase::IKCostFn;
        auto nodec = std::make_shared<rclcpp::Node>();
        nodec->set_parameter(ns + "memetic_max_generations", int64_t{100000});
        nodec->set_parameter(ns + "memetic_population_size", int64_t{32});
        pick_ik::PickIKPlugin tc;
        CHECK(tc.initialize(nodec, model, "panda_arm", "panda_link0", {"panda_hand"}, 0.1));
        long n_calls = 0;
        CostFn never = [&](geometry_msgs::msg::Pose const&, mc::RobotState const&, mc::JointModelGroup const*,
                           std::vector<double> const&) { ++n_calls; return 1.0; }; // never under cost_threshold^2
        for (std::string const& m : {std::string("global"), std::string("local")}) {
            nodec->set_parameter(ns + "mode", m);
            nodec->set_parameter(ns + "gd_max_iters", int64_t{10000000});
            n_calls = 0;
            auto const t0 = std::chrono::steady_clock::now();
            CHECK(!tc.searchPositionIK({target}, home, 0.05, {}, sol, kinematics::KinematicsBase::IKCallbackFn(), never, ec));
            double const took = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            CHECK(ec.val == ec.NO_IK_SOLUTION && sol == home && n_calls > 0);
            std::printf("IKCostFn + 50 ms timeout (%s): returned after %.1f ms, %ld callback evaluations\n", m.c_str(), took * 1e3, n_calls);
            CHECK(took < 0.5); // (ten times the budget: a loaded box must not fail this; the printed figure is the information)
        }
    }
    std::printf("plugin shim checks OK\n");
    return 0;
}
