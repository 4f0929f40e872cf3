// pick_ik_plugin_shim.cpp -- drop-in `pick_ik/PickIkPlugin` for MoveIt 2 on top of libpick_ik_amd.so.
//
// Same pluginlib identity as the reference (pick_ik_kinematics_description.xml:1-4: library
// `pick_ik_plugin`, class name `pick_ik/PickIkPlugin`, type `pick_ik::PickIKPlugin`, base
// `kinematics::KinematicsBase`; export macro src/pick_ik_plugin.cpp:405), same parameters
// (src/pick_ik_parameters.yaml, namespace robot_description_kinematics.<group>), same return
// conventions (src/pick_ik_plugin.cpp:209-217, 264-273).  The solver call of the reference
// (:182-203) becomes one pikamd_solve_batch with B = 1.
//
// This translation unit needs ROS 2 + MoveIt 2 headers; it is compiled only where they exist.
// They do not in the build container of this repository, where tests/test_plugin_shim.py compiles
// it against the declaration stubs of tests/native/ros2_stubs/ and drives it on the GPU
// (tests/native/shim_check.cpp).  Deliberate differences from the reference, all documented in
// INTEGRATION.md:
//   * wall-clock limits become iteration budgets; `timeout` only bounds the number of restarts
//   * restarts really start from the re-randomised state (the reference re-randomises
//     `init_state` but keeps passing `ik_seed_state`); ik_seed_state stays the
//     minimal-displacement reference and the vector returned on failure, exactly as there
//   * a host IKCostFn cannot run inside the GPU search.  "GPU proposes, CPU re-scores" (SURVEY.md
//     section 8(f)3): each attempt solves `cost_fn_candidates` (parameter, default 32) independent
//     copies of the query in one batch -- the first from the given start, the others from random valid
//     states --, the callback is evaluated on every candidate that passed the solver's own
//     tests, candidates whose callback cost reaches cost_threshold^2 are discarded (the reference
//     applies that threshold to every goal, src/goal.cpp:175-182) and the one with the lowest total
//     cost is returned.  The callback RANKS AND GATES solutions; unlike in the reference
//     (src/pick_ik_plugin.cpp:130-135) it does not steer the search.
#if __has_include(<moveit/kinematics_base/kinematics_base.h>) && __has_include(<rclcpp/rclcpp.hpp>)

#include <moveit/kinematics_base/kinematics_base.h>
#include <moveit/robot_model/robot_model.h>
#include <moveit/robot_state/robot_state.h>
#include <pluginlib/class_list_macros.hpp>
#include <rclcpp/rclcpp.hpp>
#include <tf2_eigen/tf2_eigen.hpp>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <set>
#include <type_traits>

#include "pick_ik_amd.hpp"

namespace pick_ik {
namespace {
auto const LOGGER = rclcpp::get_logger("pick_ik");

// ROS parameters, declared on first use with the yaml's default (src/pick_ik_parameters.yaml).
// rclcpp is strictly typed: an integer parameter is int64_t there, so every integer is declared and
// read as int64_t (never int / size_t), doubles as double, flags as bool.
template <typename T>
T param(rclcpp::Node::SharedPtr const& node, std::string const& ns, std::string const& name, T def) {
    static_assert(std::is_same_v<T, int64_t> || std::is_same_v<T, double> || std::is_same_v<T, bool> ||
                      std::is_same_v<T, std::string>,
                  "ROS parameter types: int64_t, double, bool, std::string");
    auto const full = ns + "." + name;
    if (!node->has_parameter(full)) node->declare_parameter<T>(full, def);
    T v = def;
    node->get_parameter(full, v);
    return v;
}

std::array<double, 3> rpy_of(Eigen::Matrix3d const& R) {
    // URDF convention R = Rz(yaw) Ry(pitch) Rx(roll)
    double const pitch = std::atan2(-R(2, 0), std::hypot(R(0, 0), R(1, 0)));
    double const yaw = std::atan2(R(1, 0), R(0, 0));
    double const roll = std::atan2(R(2, 1), R(2, 2));
    return {roll, pitch, yaw};
}
} // namespace

class PickIKPlugin : public kinematics::KinematicsBase {
    rclcpp::Node::SharedPtr node_;
    moveit::core::JointModelGroup const* jmg_ = nullptr;
    std::string param_ns_;
    std::vector<std::string> joint_names_, link_names_;
    std::unique_ptr<pick_ik_amd::Solver> solver_;
    std::vector<std::string> chain_roots_; // per tip: the link its joint path starts from
    pick_ik_amd::MultiChain chain_;        // what initialize() extracted from the robot model

  public:
    // diagnostics: the chain description handed to the library and the library object (its last_call() is what the
    // last attempt of the last query solved)
    pick_ik_amd::MultiChain const& chain_description() const { return chain_; }
    pick_ik_amd::Solver const& solver() const { return *solver_; }

    bool initialize(rclcpp::Node::SharedPtr const& node, moveit::core::RobotModel const& robot_model,
                    std::string const& group_name, std::string const& base_frame,
                    std::vector<std::string> const& tip_frames, double search_discretization) override {
        node_ = node;
        param_ns_ = "robot_description_kinematics." + group_name;
        storeValues(robot_model, group_name, base_frame, tip_frames, search_discretization);
        jmg_ = robot_model_->getJointModelGroup(group_name);
        if (!jmg_) {
            RCLCPP_ERROR(LOGGER, "failed to get joint model group %s", group_name.c_str());
            return false;
        }
        if (tip_frames.empty() || tip_frames.size() > PIKAMD_MAX_TIPS) {
            RCLCPP_ERROR(LOGGER, "pick_ik_amd supports 1..%d tip frames per group", PIKAMD_MAX_TIPS);
            return false;
        }
        // the active variables: the group's active single-variable joints that lie on the way to
        // some tip, in the group's order (get_active_variable_indices, reference src/robot.cpp:130-160)
        std::vector<moveit::core::JointModel const*> variables;
        // revolute / prismatic joints (one variable), planar joints (x, y, theta: three variables,
        // PIKAMD_JOINT_PLANAR_*) and floating joints (seven variables, PIKAMD_JOINT_FLOATING_*)
        auto const usable = [&](moveit::core::JointModel const* joint) {
            return joint && jmg_->hasJointModel(joint->getName()) && !joint->getMimic() &&
                   ((joint->getVariableCount() == 1 && (joint->getType() == moveit::core::JointModel::REVOLUTE ||
                                                        joint->getType() == moveit::core::JointModel::PRISMATIC)) ||
                    (joint->getVariableCount() == 3 && joint->getType() == moveit::core::JointModel::PLANAR) ||
                    (joint->getVariableCount() == 7 && joint->getType() == moveit::core::JointModel::FLOATING));
        };
        {
            std::set<moveit::core::JointModel const*> on_path;
            for (auto const& name : tip_frames) {
                auto const* tip = robot_model_->getLinkModel(name);
                if (!tip) throw std::invalid_argument("link not found: " + name);
                for (auto const* l = tip; l; l = l->getParentLinkModel())
                    if (usable(l->getParentJointModel())) on_path.insert(l->getParentJointModel());
            }
            for (auto const* joint : jmg_->getActiveJointModels())
                if (on_path.count(joint)) variables.push_back(joint);
        }
        pick_ik_amd::MultiChain mc;
        std::vector<int32_t> first_variable; // of each joint in `variables`
        for (auto const* joint : variables) {
            first_variable.push_back(static_cast<int32_t>(mc.variables.size()));
            for (auto const& b : joint->getVariableBounds()) { // one per variable (src/robot.cpp:56-75)
                pick_ik_amd::Joint v;
                v.bounded = b.position_bounded_;
                v.min = b.min_position_;
                v.max = b.max_position_;
                v.max_velocity = b.max_velocity_;
                mc.variables.push_back(v);
            }
            joint_names_.push_back(joint->getName());
        }
        // one path per tip: walk tip -> root, fold everything fixed (or foreign) into the next
        // origin; a path starts at the parent link of its first variable (chain_roots_[k]), the
        // goal of tip k is expressed in that frame at query time
        for (auto const& name : tip_frames) {
            auto const* tip = robot_model_->getLinkModel(name);
            std::vector<moveit::core::LinkModel const*> up;
            for (auto const* l = tip; l; l = l->getParentLinkModel()) up.push_back(l);
            pick_ik_amd::TipPath path;
            std::string root = robot_model_->getModelFrame();
            Eigen::Isometry3d pending = Eigen::Isometry3d::Identity();
            std::set<moveit::core::JointModel const*> on_this_path;
            for (auto const* l : up)
                if (l->getParentJointModel()) on_this_path.insert(l->getParentJointModel());
            int path_mimics = 0; // mimic steps of this path so far
            for (auto it = up.rbegin(); it != up.rend(); ++it) {
                auto const* link = *it;
                auto const* joint = link->getParentJointModel();
                pending = pending * link->getJointOriginTransform();
                if (!usable(joint)) {
                    // A mimic joint on the path FOLLOWS its master in the reference: setJointGroupPositions
                    // (src/fk_moveit.cpp:22) ends in updateMimicJoints, i.e. the joint sits at multiplier * master +
                    // offset.  It becomes a mimic step of the chain (pikamd_mimic_joint, as the robot-description
                    // readers make it) when its master is a variable of this path and the joint has one axis;
                    // otherwise the query would silently solve a different robot, and the plugin refuses.
                    // (Multiplier 0: a constant joint at `offset`, folded below like any joint outside the group.)
                    if (joint && joint->getMimic() && joint->getMimicFactor() != 0.0) {
                        auto const* master = joint->getMimic();
                        auto const mit = std::find(variables.begin(), variables.end(), master);
                        bool const one_axis = joint->getType() == moveit::core::JointModel::REVOLUTE ||
                                              joint->getType() == moveit::core::JointModel::PRISMATIC;
                        bool const master_on_path = mit != variables.end() && on_this_path.count(master) &&
                                                    master->getVariableCount() == 1;
                        if (!one_axis || !master_on_path) {
                            RCLCPP_ERROR(LOGGER, "pick_ik_amd: joint %s mimics %s and lies on the path to %s: only a revolute / "
                                         "prismatic joint that follows a single-variable joint of the same path is supported",
                                         joint->getName().c_str(), master->getName().c_str(), name.c_str());
                            return false;
                        }
                        pick_ik_amd::MimicJoint mj;
                        mj.tip = static_cast<int>(mc.tips.size());
                        mj.after_variable = path.variable.empty() ? -1 : path.variable.back();
                        mj.master_variable = first_variable[static_cast<size_t>(mit - variables.begin())];
                        if (path.joints.empty() && path_mimics == 0) {
                            root = link->getParentLinkModel() ? link->getParentLinkModel()->getName() : robot_model_->getModelFrame();
                            pending = link->getJointOriginTransform(); // the path starts at `root`
                        }
                        mj.joint.origin_xyz = {pending.translation().x(), pending.translation().y(), pending.translation().z()};
                        mj.joint.origin_rpy = rpy_of(pending.rotation());
                        Eigen::Vector3d axis;
                        if (auto const* r = dynamic_cast<moveit::core::RevoluteJointModel const*>(joint)) {
                            axis = r->getAxis();
                        } else {
                            axis = static_cast<moveit::core::PrismaticJointModel const*>(joint)->getAxis();
                            mj.joint.prismatic = true;
                        }
                        mj.joint.axis = {axis.x(), axis.y(), axis.z()};
                        mj.multiplier = joint->getMimicFactor();
                        mj.offset = joint->getMimicOffset();
                        mc.mimics.push_back(mj);
                        ++path_mimics;
                        pending = Eigen::Isometry3d::Identity();
                        continue;
                    }
                    // a moving joint of the path that is no variable of the group (not in the group, or a
                    // constant mimic): the reference's FK state is made by setToDefaultValues() and only ever
                    // receives the group's positions (src/fk_moveit.cpp:15-22), so such a joint sits at its
                    // DEFAULT position -- zero, or the middle of its range when zero is out of bounds (a constant
                    // mimic: at its offset)
                    if (joint && joint->getVariableCount() > 0) {
                        std::vector<double> v(joint->getVariableCount());
                        joint->getVariableDefaultPositions(v.data());
                        if (joint->getMimic()) v[0] = joint->getMimicOffset(); // (multiplier 0)
                        Eigen::Isometry3d J;
                        joint->computeTransform(v.data(), J);
                        pending = pending * J;
                    }
                    continue;
                }
                if (path.joints.empty() && path_mimics == 0) {
                    root = link->getParentLinkModel() ? link->getParentLinkModel()->getName() : robot_model_->getModelFrame();
                    pending = link->getJointOriginTransform(); // the path starts at `root`
                }
                pick_ik_amd::Joint j;
                j.origin_xyz = {pending.translation().x(), pending.translation().y(), pending.translation().z()};
                j.origin_rpy = rpy_of(pending.rotation());
                auto const pos = first_variable[static_cast<size_t>(
                    std::find(variables.begin(), variables.end(), joint) - variables.begin())];
                if (!path.variable.empty() && pos <= path.variable.back()) {
                    RCLCPP_ERROR(LOGGER, "pick_ik_amd: group joint order is not root-to-tip along %s", name.c_str());
                    return false;
                }
                if (joint->getType() == moveit::core::JointModel::PLANAR) {
                    // x, y, theta: three consecutive variables, the first one carries the origin
                    for (int k = 0; k < 3; ++k) {
                        pick_ik_amd::Joint p = k == 0 ? j : pick_ik_amd::Joint{};
                        p.planar = k + 1;
                        path.joints.push_back(p);
                        path.variable.push_back(pos + k);
                    }
                } else if (joint->getType() == moveit::core::JointModel::FLOATING) {
                    // trans_x .. rot_w: seven consecutive variables, the first one carries the origin
                    for (int k = 0; k < 7; ++k) {
                        pick_ik_amd::Joint p = k == 0 ? j : pick_ik_amd::Joint{};
                        p.floating = k + 1;
                        path.joints.push_back(p);
                        path.variable.push_back(pos + k);
                    }
                } else {
                    Eigen::Vector3d axis;
                    if (auto const* r = dynamic_cast<moveit::core::RevoluteJointModel const*>(joint)) {
                        axis = r->getAxis();
                    } else {
                        axis = static_cast<moveit::core::PrismaticJointModel const*>(joint)->getAxis();
                        j.prismatic = true;
                    }
                    j.axis = {axis.x(), axis.y(), axis.z()};
                    path.joints.push_back(j);
                    path.variable.push_back(pos);
                }
                pending = Eigen::Isometry3d::Identity();
            }
            path.tip_xyz = {pending.translation().x(), pending.translation().y(), pending.translation().z()};
            path.tip_rpy = rpy_of(pending.rotation());
            mc.tips.push_back(path);
            chain_roots_.push_back(root);
        }
        link_names_ = tip_frames;
        chain_ = mc;
        try {
            solver_ = std::make_unique<pick_ik_amd::Solver>(
                mc, static_cast<int>(param<int64_t>(node_, param_ns_, "gpu_device", int64_t{0})));
        } catch (std::exception const& e) {
            RCLCPP_ERROR(LOGGER, "pick_ik_amd: %s", e.what());
            return false;
        }
        // The first query of a parameter set runs the library's self test and measures what a generation costs
        // (generation_cost below).  Done HERE with the declared parameters -- a query for the pose the robot is
        // in at its default positions, answered by its own seed -- so that no caller's timeout pays for it.  (A
        // parameter changed later is measured inside the query that first meets it, on that query's clock.)
        try {
            moveit::core::RobotState st(robot_model_);
            st.setToDefaultValues();
            st.update();
            std::vector<double> seed;
            st.copyJointGroupPositions(jmg_, seed);
            std::vector<geometry_msgs::msg::Pose> poses;
            Eigen::Isometry3d const base = st.getGlobalLinkTransform(getBaseFrame()).inverse();
            for (auto const& name : tip_frames) poses.push_back(tf2::toMsg(base * st.getGlobalLinkTransform(name)));
            std::vector<double> sol;
            moveit_msgs::msg::MoveItErrorCodes ec;
            std::lock_guard<std::mutex> lock(solver_mutex_);
            (void)search(poses, seed, 1.0, {}, sol, IKCallbackFn(), IKCostFn(), ec, kinematics::KinematicsQueryOptions(),
                         nullptr);
        } catch (std::exception const& e) {
            RCLCPP_WARN(LOGGER, "pick_ik_amd: warm-up query failed (%s); the first query will measure instead", e.what());
        }
        return true;
    }

    // The reference's searchPositionIK is re-entrant (its FK closure takes fk_mutex_,
    // src/fk_moveit.cpp:21) and never throws past MoveIt.  A library handle is used from one thread at a
    // time (its staging buffers, stream and counters belong to the call in flight), so calls on one
    // plugin instance are serialised here; and whatever the library or the C++ mirror refuses (a
    // parameter combination, a HIP error) ends the query as NO_IK_SOLUTION with the seed returned
    // instead of an exception inside the caller's planning thread.
    bool searchPositionIK(std::vector<geometry_msgs::msg::Pose> const& ik_poses,
                          std::vector<double> const& ik_seed_state, double timeout,
                          std::vector<double> const& consistency_limits, std::vector<double>& solution,
                          IKCallbackFn const& solution_callback, IKCostFn const& cost_function,
                          moveit_msgs::msg::MoveItErrorCodes& error_code,
                          kinematics::KinematicsQueryOptions const& options = kinematics::KinematicsQueryOptions(),
                          moveit::core::RobotState const* context_state = nullptr) const override {
        std::lock_guard<std::mutex> lock(solver_mutex_);
        try {
            return search(ik_poses, ik_seed_state, timeout, consistency_limits, solution, solution_callback,
                          cost_function, error_code, options, context_state);
        } catch (std::exception const& e) {
            RCLCPP_ERROR(LOGGER, "pick_ik_amd: query failed: %s", e.what());
        } catch (...) {
            RCLCPP_ERROR(LOGGER, "pick_ik_amd: query failed");
        }
        solution = ik_seed_state;
        error_code.val = error_code.NO_IK_SOLUTION;
        return false;
    }

  private:
    mutable std::mutex solver_mutex_;
    // A MoveIt caller passes a wall-clock `timeout` (kinematics_solver_timeout); the library's budgets are
    // generation counts.  The first query with a given (population, elite size, species) measures what a
    // generation of ONE problem costs on this device -- a target out of reach, so that nothing stops early
    // -- and every attempt's generation budget is then cut to what fits the time left
    // (src/pick_ik_plugin.cpp:147-149, 276-290: the reference checks the clock inside its loops).
    struct GenerationCost {
        double fixed_ms = 0.0, per_generation_ms = 0.0;
    };
    mutable std::map<std::array<int64_t, 7>, GenerationCost> generation_cost_;
    mutable std::string arithmetic_ = "exact"; // the handle's current "arithmetic" option

    GenerationCost const& generation_cost(pick_ik_amd::MemeticIkParams m, std::vector<pick_ik_amd::Pose> far,
                                          pick_ik_amd::CostSpec const& costs, std::vector<double> const& start) const {
        // everything that changes what a generation costs or which kernel flavour serves it: population, elites,
        // species, the descent's iteration budget, which joint goals are on, the gradient step (the line-search
        // form)
        int64_t const goal_mask = (costs.center_joints_weight > 0.0 ? 1 : 0) | (costs.avoid_joint_limits_weight > 0.0 ? 2 : 0) |
                                  (costs.minimal_displacement_weight > 0.0 ? 4 : 0);
        int64_t step_bits = 0;
        std::memcpy(&step_bits, &m.gd_params.step_size, sizeof step_bits);
        std::array<int64_t, 7> const key{static_cast<int64_t>(m.population_size), static_cast<int64_t>(m.elite_size),
                                         static_cast<int64_t>(m.num_threads), static_cast<int64_t>(m.gd_params.max_iterations), goal_mask,
                                         step_bits, arithmetic_ == "exact" ? 1 : 0};
        auto it = generation_cost_.find(key);
        if (it != generation_cost_.end()) return it->second;
        // a new parameter set: every kernel variant the library may choose for it must agree with the one-lane
        // kernel on this robot (pikamd_self_test); one that does not is switched off for the handle
        if (uint32_t const off = solver_->self_test(costs, m))
            RCLCPP_WARN(LOGGER, "pick_ik_amd: kernel variants switched off by the self test, mask 0x%x", off);
        for (auto& p : far) p.x += 100.0; // nothing reaches this: every generation of the budget is run
        auto const run = [&](int generations) {
            m.max_generations = generations;
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) { // (the first repetition pays allocations / constant uploads)
                auto const t0 = std::chrono::steady_clock::now();
                (void)solver_->ik_memetic(start, far, costs, m, false, 1, &start);
                best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }
            return best;
        };
        double const t1 = run(1), t25 = run(25);
        GenerationCost c;
        c.per_generation_ms = std::max(1e-3, (t25 - t1) / 24.0);
        c.fixed_ms = std::max(0.0, t1 - c.per_generation_ms);
        return generation_cost_.emplace(key, c).first->second;
    }

    bool search(std::vector<geometry_msgs::msg::Pose> const& ik_poses,
                std::vector<double> const& ik_seed_state, double timeout,
                std::vector<double> const&, std::vector<double>& solution,
                IKCallbackFn const& solution_callback, IKCostFn const& cost_function,
                moveit_msgs::msg::MoveItErrorCodes& error_code,
                kinematics::KinematicsQueryOptions const& options,
                moveit::core::RobotState const* context_state) const {
        (void)context_state; // not used (neither does the reference, src/pick_ik_plugin.cpp:83)
        auto const P = [&](auto name, auto def) { return param(node_, param_ns_, std::string(name), def); };
        solution = ik_seed_state;
        error_code.val = error_code.NO_IK_SOLUTION;
        // goal in the chain's base frame (transform_poses_to_frames, src/robot.cpp:169-181, then
        // into the frame of the chain's first joint parent)
        moveit::core::RobotState state(robot_model_);
        state.setToDefaultValues();
        state.setJointGroupPositions(jmg_, ik_seed_state);
        state.update();
        if (ik_poses.size() != chain_roots_.size()) {
            RCLCPP_ERROR(LOGGER, "pick_ik_amd: %zu poses for %zu tip frames", ik_poses.size(), chain_roots_.size());
            return false;
        }
        std::vector<pick_ik_amd::Pose> g;
        for (size_t k = 0; k < ik_poses.size(); ++k) {
            Eigen::Isometry3d p;
            tf2::fromMsg(ik_poses[k], p);
            Eigen::Isometry3d const goal_model = state.getGlobalLinkTransform(getBaseFrame()) * p;
            Eigen::Isometry3d const goal = state.getGlobalLinkTransform(chain_roots_[k]).inverse() * goal_model;
            Eigen::Quaterniond const q(goal.rotation());
            g.push_back(pick_ik_amd::Pose{goal.translation().x(), goal.translation().y(), goal.translation().z(),
                                          q.w(), q.x(), q.y(), q.z()});
        }

        pick_ik_amd::CostSpec costs;
        costs.position_scale = P("position_scale", 1.0);
        costs.rotation_scale = P("rotation_scale", 0.5);
        costs.position_threshold = P("position_threshold", 0.001);
        costs.orientation_threshold = P("orientation_threshold", 0.001);
        costs.cost_threshold = P("cost_threshold", 0.001);
        costs.center_joints_weight = P("center_joints_weight", 0.0);
        costs.avoid_joint_limits_weight = P("avoid_joint_limits_weight", 0.0);
        costs.minimal_displacement_weight = P("minimal_displacement_weight", 0.0);
        std::string const mode = P("mode", std::string("global"));
        // "exact" (default): the kernels whose joint vectors are the reference algorithm's bit for bit -- what
        // src/pick_ik_plugin.cpp:182-188 hands back is ik_memetic's vector, so that is what a caller who changes
        // nothing gets here; "fast" (opt-in): the Denavit-Hartenberg kernels (about twice the throughput, whole
        // solves agree with the CPU reference statistically; pikamd_set_option "arithmetic")
        std::string const arithmetic = P("arithmetic", std::string("exact"));
        if (arithmetic != "fast" && arithmetic != "exact") {
            RCLCPP_ERROR(LOGGER, "Invalid arithmetic: %s (fast | exact)", arithmetic.c_str());
            return false;
        }
        if (arithmetic != arithmetic_) {
            solver_->set_option("arithmetic", arithmetic.c_str());
            arithmetic_ = arithmetic;
        }
        // (read every call like the reference's parameter_listener_->get_params())
        int64_t const num_threads = P("memetic_num_threads", int64_t{1});
        bool const stop_on_first = P("memetic_stop_on_first_solution", true);
        bool const approx = options.return_approximate_solution;

        auto const& robot = solver_->robot();
        std::vector<double> init = ik_seed_state;
        // (rng_seed: 0 = a fresh random stream per query, like the reference's unseeded generators; any other
        //  value makes a query reproducible -- with arithmetic = exact down to the last bit of the joint vector)
        int64_t const fixed_seed = P("rng_seed", int64_t{0});
        std::mt19937_64 rng{fixed_seed != 0 ? static_cast<uint64_t>(fixed_seed) : std::random_device{}()};
        auto randomise = [&] {
            for (size_t i = 0; i < init.size(); ++i) {
                auto const& v = robot.variables[i];
                std::uniform_real_distribution<double> d(v.bounded ? v.min : init[i] - M_PI,
                                                         v.bounded ? v.max : init[i] + M_PI);
                init[i] = d(rng);
            }
        };
        if (!robot.is_valid_configuration(init)) {
            RCLCPP_WARN(LOGGER, "Initial guess exceeds joint limits. Regenerating a random valid configuration.");
            randomise();
        }

        // parameter mapping of the two solvers (src/pick_ik_plugin.cpp:165-196), read per attempt
        auto const memetic_params = [&] {
            pick_ik_amd::MemeticIkParams m;
            m.population_size = static_cast<size_t>(P("memetic_population_size", int64_t{16}));
            m.elite_size = static_cast<size_t>(P("memetic_elite_size", int64_t{4}));
            m.wipeout_fitness_tol = P("memetic_wipeout_fitness_tol", 0.00001);
            m.max_generations = static_cast<int>(P("memetic_max_generations", int64_t{100}));
            m.stop_optimization_on_valid_solution = P("stop_optimization_on_valid_solution", true);
            m.gd_params.step_size = P("gd_step_size", 0.0001);
            m.gd_params.min_cost_delta = P("gd_min_cost_delta", 1.0e-12);
            m.gd_params.max_iterations = static_cast<int>(P("memetic_gd_max_iters", int64_t{25}));
            // memetic_num_threads is a thread count in the reference (src/pick_ik_plugin.cpp:171), here the
            // number of species that share a wavefront with the elites: pow2ceil(species) * pow2ceil(elites)
            // <= 64 lanes.  A CPU-style setting beyond that is clamped, not refused.
            size_t gs = 1;
            while (gs < m.elite_size) gs <<= 1;
            size_t max_species = gs <= 64 ? 64 / gs : 1, fit = 1;
            while (fit * 2 <= max_species) fit *= 2;
            size_t species = static_cast<size_t>(std::max<int64_t>(1, num_threads));
            if (species > fit) {
                RCLCPP_WARN(LOGGER, "memetic_num_threads %zu exceeds the %zu species a wavefront holds with "
                                    "elite size %zu; using %zu", species, fit, m.elite_size, fit);
                species = fit;
            }
            m.num_threads = species;
            m.stop_on_first_soln = stop_on_first;             // :172
            return m;
        };
        auto const gradient_params = [&] {
            pick_ik_amd::GradientIkParams gd;
            gd.step_size = P("gd_step_size", 0.0001);
            gd.min_cost_delta = P("gd_min_cost_delta", 1.0e-12);
            gd.max_iterations = static_cast<int>(P("gd_max_iters", int64_t{100}));
            gd.stop_optimization_on_valid_solution = P("stop_optimization_on_valid_solution", true);
            return gd;
        };
        // A host cost function (IKCostFn).  cost_fn_mode = "search" (default): the callback is a goal inside the
        // search, as in the reference -- solved on the host.  "rank": the GPU solves cost_fn_candidates copies of the
        // query without the callback and the callback ranks / gates the finished candidates (fast; a cost that
        // only a guided search gets under the threshold fails there).
        bool const cost_fn_in_search = P("cost_fn_mode", std::string("search")) != "rank";
        // "rank": candidates per attempt, each re-scored with the callback
        size_t const n_cand = cost_function ? static_cast<size_t>(std::max<int64_t>(1, P("cost_fn_candidates", int64_t{32}))) : 1;
        // sum over the poses of the callback's cost for one joint vector (one Goal of weight 1 per pose,
        // src/pick_ik_plugin.cpp:130-135); `worst` = the largest single term, what cost_threshold tests
        auto callback_cost = [&, st = state](std::vector<double> const& q, double& worst) mutable {
            st.setJointGroupPositions(jmg_, q);
            st.update();
            double sum = 0.0;
            worst = 0.0;
            for (auto const& pose : ik_poses) {
                double const c = cost_function(pose, st, jmg_, ik_seed_state);
                sum += c;
                worst = std::max(worst, c);
            }
            return sum;
        };
        // (a parameter set not met before -- initialize() has measured the declared one -- is self-tested and its
        //  generation cost measured now, ON the caller's clock)
        auto const t0 = std::chrono::steady_clock::now();
        if (mode == "global" && !(cost_function && cost_fn_in_search)) // (the host path carries its own clock)
            (void)generation_cost(memetic_params(), g, costs, ik_seed_state);
        // an attempt's generation budget: what fits the time that is left (at least one generation)
        auto const budgeted_memetic_params = [&] {
            auto m = memetic_params();
            auto const& gc = generation_cost(m, g, costs, ik_seed_state);
            double const left_ms = (timeout - std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()) * 1e3;
            double const fit = (left_ms - gc.fixed_ms) / gc.per_generation_ms;
            if (fit < static_cast<double>(m.max_generations)) m.max_generations = std::max(1, static_cast<int>(fit));
            return m;
        };
        bool found = false;
        while (true) {
            std::optional<std::vector<double>> r;
            if (cost_function && cost_fn_in_search) {
                // The reference's semantics: the callback is a goal INSIDE the search (src/pick_ik_plugin.cpp:130-135)
                // -- on the host, with the exact kernels' arithmetic (pikamd_solve_batch_host).  Every evaluation
                // of the search calls it (13 000 per default solve), as in the reference.
                // The robot state is built once per query and its group positions overwritten per evaluation, as
                // make_ik_cost_fn's captured copy is (src/goal.cpp:151-159).  The caller's wall-clock budget is
                // enforced the reference's way on this path, by the clock inside the loops: what is left of
                // `timeout` bounds the attempt (tested in front of every generation / step,
                // src/pick_ik_plugin.cpp:172, 186), memetic_gd_max_time one elite's descent (:177).
                pick_ik_amd::Solver::HostCostFn const hc = [&, st = state](std::vector<double> const& q, int pose) mutable {
                    st.setJointGroupPositions(jmg_, q);
                    st.update();
                    return cost_function(ik_poses[static_cast<size_t>(pose)], st, jmg_, ik_seed_state);
                };
                double const left = std::max(1.0e-6, timeout - std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
                if (mode == "global") {
                    auto m = memetic_params();
                    m.max_time = left;
                    m.gd_params.max_time = P("memetic_gd_max_time", 0.005);
                    r = solver_->ik_memetic(init, g, costs, m, hc, approx, rng(), &ik_seed_state);
                } else if (mode == "local") {
                    auto gd = gradient_params();
                    gd.max_time = left;
                    r = solver_->ik_gradient(init, g, costs, gd, hc, approx, &ik_seed_state);
                } else {
                    RCLCPP_ERROR(LOGGER, "Invalid solver mode: %s", mode.c_str());
                    return false;
                }
            } else if (cost_function) {
                // cost_fn_mode: rank -- GPU proposes ...
                std::vector<double> starts, refs;
                std::vector<pick_ik_amd::Pose> goals;
                std::vector<double> start = init;
                for (size_t k = 0; k < n_cand; ++k) {
                    if (k > 0) { // the first candidate starts where a plain query would, the others at
                        randomise(); // random valid states: from one start the descent on the first elite
                        start = init; // finds the same solution whatever the random stream
                    }
                    starts.insert(starts.end(), start.begin(), start.end());
                    refs.insert(refs.end(), ik_seed_state.begin(), ik_seed_state.end());
                    goals.insert(goals.end(), g.begin(), g.end());
                }
                pick_ik_amd::BatchResult br;
                if (mode == "global") {
                    auto const m = budgeted_memetic_params();
                    br = solver_->ik_memetic_batch(starts, goals, costs, m, approx, rng(), 0, &refs);
                } else if (mode == "local") {
                    auto const gd = gradient_params();
                    br = solver_->ik_gradient_batch(starts, goals, costs, gd, approx, &refs);
                } else {
                    RCLCPP_ERROR(LOGGER, "Invalid solver mode: %s", mode.c_str());
                    return false;
                }
                // ... CPU re-scores: lowest total cost among the candidates the callback lets through
                double const thr_sq = costs.cost_threshold * costs.cost_threshold;
                double best_total = 0.0;
                size_t const dof = ik_seed_state.size();
                for (size_t k = 0; k < n_cand; ++k) {
                    if (br.status[k] <= 0) continue;
                    std::vector<double> q(br.solution.begin() + static_cast<long>(k * dof),
                                          br.solution.begin() + static_cast<long>((k + 1) * dof));
                    double worst = 0.0;
                    double const total = br.cost[k] + callback_cost(q, worst);
                    // (an approximate solution is not held to the threshold here: the gate below decides)
                    if (!approx && !(worst < thr_sq)) continue;
                    if (!r || total < best_total) {
                        r = q;
                        best_total = total;
                    }
                }
            } else if (mode == "global") {
                auto const m = budgeted_memetic_params();
                // start at `init` (ik_seed_state, or a random valid state on restarts), measure the
                // minimal-displacement cost against ik_seed_state, return ik_seed_state on failure
                r = solver_->ik_memetic(init, g, costs, m, approx, rng(), &ik_seed_state);
            } else if (mode == "local") {
                auto const gd = gradient_params();
                r = solver_->ik_gradient(init, g, costs, gd, approx, &ik_seed_state);
            } else {
                RCLCPP_ERROR(LOGGER, "Invalid solver mode: %s", mode.c_str());
                return false;
            }
            if (r) {
                solution = *r;
                error_code.val = error_code.SUCCESS;
            } else {
                solution = ik_seed_state;
                error_code.val = error_code.NO_IK_SOLUTION;
            }
            if (approx) {
                // approximate-solution gate, as the reference applies it (src/pick_ik_plugin.cpp:
                // 219-267): the returned vector must pass make_is_solution_test_fn built from the
                // REGULAR frame tests (the approximate pose thresholds are computed there but the
                // test is made from `frame_tests`) and from the goals under
                // approximate_solution_cost_threshold (no goals when that is <= 0), then the
                // joint-jump limit
                pick_ik_amd::CostSpec gate = costs;
                double const act = P("approximate_solution_cost_threshold", 0.0);
                if (act <= 0.0) {
                    gate.center_joints_weight = gate.avoid_joint_limits_weight = gate.minimal_displacement_weight = 0.0;
                } else {
                    gate.cost_threshold = act;
                }
                bool valid = solver_->evaluate(solution, g, ik_seed_state, gate).is_solution;
                if (valid && cost_function && act > 0.0) { // the callback's goals are goals too
                    double worst = 0.0;
                    callback_cost(solution, worst);
                    valid = worst < act * act;
                }
                double const jt = P("approximate_solution_joint_threshold", 0.0);
                if (valid && jt > 0.0)
                    for (size_t i = 0; i < solution.size(); ++i)
                        if (std::abs(solution[i] - ik_seed_state[i]) > jt) {
                            valid = false;
                            break;
                        }
                if (!valid) {
                    error_code.val = error_code.NO_IK_SOLUTION;
                    solution = ik_seed_state;
                }
            }
            found = error_code.val == error_code.SUCCESS;
            if (found && solution_callback) solution_callback(ik_poses.front(), solution, error_code);
            found = error_code.val == error_code.SUCCESS;
            double const spent = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (found || spent >= timeout) break;
            randomise();
        }
        return found;
    }

  public:
    std::vector<std::string> const& getJointNames() const override { return joint_names_; }
    std::vector<std::string> const& getLinkNames() const override { return link_names_; }
    bool getPositionFK(std::vector<std::string> const&, std::vector<double> const&,
                       std::vector<geometry_msgs::msg::Pose>&) const override {
        return false;
    }
    bool getPositionIK(geometry_msgs::msg::Pose const&, std::vector<double> const&, std::vector<double>&,
                       moveit_msgs::msg::MoveItErrorCodes&,
                       kinematics::KinematicsQueryOptions const&) const override {
        return false;
    }
    // pose vector + consistency limits + callback, no cost function
    // (include/pick_ik/pick_ik_plugin.hpp:92-102, src/pick_ik_plugin.cpp:387-401)
    bool searchPositionIK(std::vector<geometry_msgs::msg::Pose> const& ik_poses, std::vector<double> const& seed,
                          double timeout, std::vector<double> const& limits, std::vector<double>& solution,
                          IKCallbackFn const& cb, moveit_msgs::msg::MoveItErrorCodes& error_code,
                          kinematics::KinematicsQueryOptions const& options = kinematics::KinematicsQueryOptions(),
                          moveit::core::RobotState const* context_state = nullptr) const override {
        return searchPositionIK(ik_poses, seed, timeout, limits, solution, cb, IKCostFn(), error_code, options,
                                context_state);
    }
    // the single-pose overloads forward to the pose-vector form (src/pick_ik_plugin.cpp:314-385)
    bool searchPositionIK(geometry_msgs::msg::Pose const& ik_pose, std::vector<double> const& seed,
                          double timeout, std::vector<double>& solution,
                          moveit_msgs::msg::MoveItErrorCodes& error_code,
                          kinematics::KinematicsQueryOptions const& options = kinematics::KinematicsQueryOptions()) const override {
        return searchPositionIK({ik_pose}, seed, timeout, {}, solution, IKCallbackFn(), IKCostFn(), error_code, options);
    }
    bool searchPositionIK(geometry_msgs::msg::Pose const& ik_pose, std::vector<double> const& seed,
                          double timeout, std::vector<double> const& limits, std::vector<double>& solution,
                          moveit_msgs::msg::MoveItErrorCodes& error_code,
                          kinematics::KinematicsQueryOptions const& options = kinematics::KinematicsQueryOptions()) const override {
        return searchPositionIK({ik_pose}, seed, timeout, limits, solution, IKCallbackFn(), IKCostFn(), error_code, options);
    }
    bool searchPositionIK(geometry_msgs::msg::Pose const& ik_pose, std::vector<double> const& seed,
                          double timeout, std::vector<double>& solution, IKCallbackFn const& cb,
                          moveit_msgs::msg::MoveItErrorCodes& error_code,
                          kinematics::KinematicsQueryOptions const& options = kinematics::KinematicsQueryOptions()) const override {
        return searchPositionIK({ik_pose}, seed, timeout, {}, solution, cb, IKCostFn(), error_code, options);
    }
    bool searchPositionIK(geometry_msgs::msg::Pose const& ik_pose, std::vector<double> const& seed,
                          double timeout, std::vector<double> const& limits, std::vector<double>& solution,
                          IKCallbackFn const& cb, moveit_msgs::msg::MoveItErrorCodes& error_code,
                          kinematics::KinematicsQueryOptions const& options = kinematics::KinematicsQueryOptions()) const override {
        return searchPositionIK({ik_pose}, seed, timeout, limits, solution, cb, IKCostFn(), error_code, options);
    }
};

} // namespace pick_ik

PLUGINLIB_EXPORT_CLASS(pick_ik::PickIKPlugin, kinematics::KinematicsBase);

#endif // MoveIt available
