// pick_ik_amd.hpp -- C++17 host-side mirror of pick_ik's solver interface over the C ABI
// (include/pick_ik_amd.h).  Header-only; needs no ROS / MoveIt / Eigen.
//
// Names and argument meaning follow the reference so that code written against
//   pick_ik::ik_memetic   (include/pick_ik/ik_memetic.hpp:97-104)
//   pick_ik::ik_gradient  (include/pick_ik/ik_gradient.hpp:43-49)
//   pick_ik::Robot        (include/pick_ik/robot.hpp:14-50)
//   pick_ik::MemeticIkParams / GradientIkParams (ik_memetic.hpp:26-45, ik_gradient.hpp:15-23)
// ports by changing the namespace: the closures (CostFn, SolutionTestFn, FkFn) of the reference are
// replaced by plain data (goal pose + CostSpec), because arbitrary host closures cannot run on
// the GPU; everything those closures can express in pick_ik's own plugin is representable.
//
// Errors: like the reference, failures to FIND a solution are std::nullopt; misuse (bad chain, no
// GPU, bad parameters) throws std::runtime_error / std::invalid_argument with the C ABI's message.
#pragma once

#include <array>
#include <cstdint>
#include <functional>
#include <cstdio>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pick_ik_amd.h"

namespace pick_ik_amd {

// include/pick_ik/ik_gradient.hpp:15-23 (max_time: honoured by the host solver -- queries with a host cost function
// --, accepted and ignored by the GPU paths, where iteration budgets bind)
struct GradientIkParams {
    double step_size = 0.0001;
    double min_cost_delta = 1.0e-12;
    double max_time = 0.05;
    int max_iterations = 100;
    bool stop_optimization_on_valid_solution = true;
};

// include/pick_ik/ik_memetic.hpp:26-45
struct MemeticIkParams {
    size_t elite_size = 4;
    size_t population_size = 16;
    double wipeout_fitness_tol = 0.00001;
    int max_generations = 100;
    double max_time = 1.0;
    size_t num_threads = 1;
    bool stop_optimization_on_valid_solution = true;
    bool stop_on_first_soln = true;
    GradientIkParams gd_params{0.0001, 1.0e-12, 0.005, 25, true};
};

// What pick_ik's plugin bakes into cost_fn / solution_fn (src/pick_ik_plugin.cpp:97-142)
struct CostSpec {
    double position_scale = 1.0;
    double rotation_scale = 0.5;
    double position_threshold = 0.001;
    double orientation_threshold = 0.001;
    double cost_threshold = 0.001;
    double center_joints_weight = 0.0;
    double avoid_joint_limits_weight = 0.0;
    double minimal_displacement_weight = 0.0;
};

// Goal frame in the chain's base frame: position + quaternion (geometry_msgs::Pose order w last in
// ROS; here explicit fields)
struct Pose {
    double x = 0, y = 0, z = 0;
    double qw = 1, qx = 0, qy = 0, qz = 0;
};

// Serial chain description (what Robot::from + make_fk_fn extract from a MoveIt RobotModel)
struct Joint {
    std::array<double, 3> origin_xyz{0, 0, 0};
    std::array<double, 3> origin_rpy{0, 0, 0};
    std::array<double, 3> axis{0, 0, 1};
    bool prismatic = false;
    // 1 / 2 / 3: the x / y / theta variable of a PLANAR joint (three consecutive entries; the x entry
    // carries the joint's origin) -- PIKAMD_JOINT_PLANAR_* of the C ABI
    int planar = 0;
    // 1 .. 7: the trans_x trans_y trans_z rot_x rot_y rot_z rot_w variable of a FLOATING joint (seven
    // consecutive entries; the first carries the joint's origin) -- PIKAMD_JOINT_FLOATING_* of the C ABI
    int floating = 0;
    double min = -3.14159265358979323846, max = 3.14159265358979323846;
    double max_velocity = 0.0;
    bool bounded = true;
};

struct Chain {
    std::vector<Joint> joints;
    std::array<double, 3> tip_xyz{0, 0, 0};
    std::array<double, 3> tip_rpy{0, 0, 0};
};

// Several tip links (the plugin's tip_frames): `variables` gives limits / velocities / bounded flags
// of the active variables (their geometry fields are unused); every tip lists the joints on ITS
// path from the base -- geometry in `joints`, and in `variable` the index of each joint's variable
// (strictly increasing along the path; a shared joint has the same index in every path).
struct TipPath {
    std::vector<int32_t> variable;
    std::vector<Joint> joints;
    std::array<double, 3> tip_xyz{0, 0, 0};
    std::array<double, 3> tip_rpy{0, 0, 0};
};
// A mimic joint on a tip path (pikamd_mimic_joint): no variable (src/robot.cpp:144-150), moved with its master by
// the reference's forward kinematics (src/fk_moveit.cpp:22 -> updateMimicJoints).  One more step of the chain
// behind the joint of variable `after_variable` (-1: in front of the first) at multiplier * q[master_variable] +
// offset; `joint` carries its origin (the fixed transform from the previous moving joint), axis and type.
struct MimicJoint {
    int tip = 0;
    int after_variable = -1, master_variable = 0;
    Joint joint;
    double multiplier = 1.0, offset = 0.0;
};
struct MultiChain {
    std::vector<Joint> variables;
    std::vector<TipPath> tips;
    std::vector<MimicJoint> mimics;
};

// pick_ik::Robot (include/pick_ik/robot.hpp:14-50): the variable table
struct Robot {
    struct Variable {
        double min, max, mid;
        bool bounded;
        double half_span;
        double max_velocity_rcp;
        double minimal_displacement_factor;
        bool is_valid(double v) const { return !bounded || (v <= max && v >= min); }
        double clamp_to_limits(double v) const {
            const double lo = bounded ? min : v - half_span, hi = bounded ? max : v + half_span;
            return v < lo ? lo : (hi < v ? hi : v);
        }
    };
    std::vector<Variable> variables;
    bool is_valid_configuration(const std::vector<double>& q) const {
        for (size_t i = 0; i < variables.size(); ++i)
            if (!variables[i].is_valid(q[i])) return false;
        return true;
    }
};

struct BatchResult {
    std::vector<double> solution; // [B][dof], seed on failure
    std::vector<int32_t> status;  // PIKAMD_SUCCESS / PIKAMD_APPROXIMATE / PIKAMD_NO_IK_SOLUTION
    std::vector<double> cost;
    std::vector<pikamd_stats> stats;
};

inline int32_t joint_type_of(const Joint& J) {
    if (J.planar >= 1 && J.planar <= 3) return PIKAMD_JOINT_PLANAR_X + (J.planar - 1);
    if (J.floating >= 1 && J.floating <= 7) return PIKAMD_JOINT_FLOATING_TX + (J.floating - 1);
    return J.prismatic ? PIKAMD_JOINT_PRISMATIC : PIKAMD_JOINT_REVOLUTE;
}

class Solver {
  public:
    Solver(const Chain& chain, int device = 0) : dof_(static_cast<int>(chain.joints.size())) {
        if (dof_ < 1) throw std::invalid_argument("pick_ik_amd: empty chain");
        std::vector<double> o(6 * dof_), ax(3 * dof_), lo(dof_), hi(dof_), vm(dof_);
        std::vector<int32_t> jt(dof_);
        std::vector<uint8_t> bd(dof_);
        for (int j = 0; j < dof_; ++j) {
            const Joint& J = chain.joints[j];
            for (int k = 0; k < 3; ++k) {
                o[6 * j + k] = J.origin_xyz[k];
                o[6 * j + 3 + k] = J.origin_rpy[k];
                ax[3 * j + k] = J.axis[k];
            }
            jt[j] = joint_type_of(J);
            lo[j] = J.min;
            hi[j] = J.max;
            vm[j] = J.max_velocity;
            bd[j] = J.bounded ? 1 : 0;
        }
        const double tip[6] = {chain.tip_xyz[0], chain.tip_xyz[1], chain.tip_xyz[2],
                               chain.tip_rpy[0], chain.tip_rpy[1], chain.tip_rpy[2]};
        pikamd_chain c{dof_, o.data(), ax.data(), jt.data(), tip, lo.data(), hi.data(), vm.data(), bd.data()};
        if (pikamd_create(&c, device, &h_) != 0) throw std::runtime_error(pikamd_last_error());
        load_variables();
    }
    // several tip frames: goals are n_tips() poses per problem, in tip order
    Solver(const MultiChain& mc, int device = 0) : dof_(static_cast<int>(mc.variables.size())) {
        if (dof_ < 1 || mc.tips.empty()) throw std::invalid_argument("pick_ik_amd: empty chain");
        n_tips_ = static_cast<int>(mc.tips.size());
        std::vector<double> lo(dof_), hi(dof_), vm(dof_);
        std::vector<uint8_t> bd(dof_);
        for (int j = 0; j < dof_; ++j) {
            lo[j] = mc.variables[j].min;
            hi[j] = mc.variables[j].max;
            vm[j] = mc.variables[j].max_velocity;
            bd[j] = mc.variables[j].bounded ? 1 : 0;
        }
        struct Arrays {
            std::vector<double> o, ax, tip;
            std::vector<int32_t> jt;
        };
        std::vector<Arrays> keep(mc.tips.size());
        std::vector<pikamd_tip> tips(mc.tips.size());
        for (size_t k = 0; k < mc.tips.size(); ++k) {
            const TipPath& t = mc.tips[k];
            if (t.variable.size() != t.joints.size()) throw std::invalid_argument("pick_ik_amd: tip path sizes differ");
            Arrays& a = keep[k];
            for (const Joint& J : t.joints) {
                for (int i = 0; i < 3; ++i) a.o.push_back(J.origin_xyz[i]);
                for (int i = 0; i < 3; ++i) a.o.push_back(J.origin_rpy[i]);
                for (int i = 0; i < 3; ++i) a.ax.push_back(J.axis[i]);
                a.jt.push_back(joint_type_of(J));
            }
            a.tip = {t.tip_xyz[0], t.tip_xyz[1], t.tip_xyz[2], t.tip_rpy[0], t.tip_rpy[1], t.tip_rpy[2]};
            tips[k] = pikamd_tip{static_cast<int32_t>(t.joints.size()), t.variable.data(), a.o.data(),
                                 a.ax.data(), a.jt.data(), a.tip.data()};
        }
        pikamd_multi_chain c{dof_, n_tips_, tips.data(), lo.data(), hi.data(), vm.data(), bd.data()};
        if (pikamd_create_multi(&c, device, &h_) != 0) throw std::runtime_error(pikamd_last_error());
        if (!mc.mimics.empty()) {
            std::vector<pikamd_mimic_joint> mj;
            for (const MimicJoint& m : mc.mimics) {
                pikamd_mimic_joint x{};
                x.tip = m.tip;
                x.after_variable = m.after_variable;
                x.master_variable = m.master_variable;
                x.joint_type = m.joint.prismatic ? PIKAMD_JOINT_PRISMATIC : PIKAMD_JOINT_REVOLUTE;
                for (int i = 0; i < 3; ++i) {
                    x.origin_xyz_rpy[i] = m.joint.origin_xyz[i];
                    x.origin_xyz_rpy[3 + i] = m.joint.origin_rpy[i];
                    x.axis[i] = m.joint.axis[i];
                }
                x.multiplier = m.multiplier;
                x.offset = m.offset;
                mj.push_back(x);
            }
            if (pikamd_set_mimic_joints(h_, static_cast<int32_t>(mj.size()), mj.data()) != 0) {
                const std::string msg = pikamd_last_error();
                pikamd_destroy(h_);
                h_ = nullptr;
                throw std::runtime_error(msg);
            }
        }
        load_variables();
    }
    ~Solver() { pikamd_destroy(h_); }
    Solver(const Solver&) = delete;
    Solver& operator=(const Solver&) = delete;

    int dof() const { return dof_; }
    int n_tips() const { return n_tips_; }
    const Robot& robot() const { return robot_; }
    // What the last GPU solve of this object handed to the C ABI (diagnostics; tests/native/shim_check.cpp replays
    // it through the CPU oracle): parameters, goals [B][n_tips][7], ik_seed_state and initial guess [B][dof].
    struct CallRecord {
        pikamd_params params{};
        std::vector<double> goal_pos_quat, seed, initial_guess;
        uint64_t rng_seed = 0;
        int64_t problem_offset = 0;
    };
    // OFF by default (a record is a deep copy of the call's arrays -- ~170 MB for a million 7-variable problems -- and
    // writing it from the const solve methods would race between concurrent callers of one object): a diagnostic
    // session switches it on, from one thread, and reads last_call() after the solve it wants to replay.
    void set_record_last_call(bool on) const { record_last_call_ = on; }
    const CallRecord& last_call() const { return last_call_; }

    // make_fk_fn: one frame per tip link, in tip order (src/fk_moveit.cpp:11-35)
    std::vector<Pose> fk_tips(const std::vector<double>& q) const {
        check_size(q);
        std::vector<double> out(7 * static_cast<size_t>(n_tips_));
        if (pikamd_fk_batch(h_, 1, q.data(), out.data()) != 0) throw std::runtime_error(pikamd_last_error());
        std::vector<Pose> r;
        for (int k = 0; k < n_tips_; ++k)
            r.push_back(Pose{out[7 * k], out[7 * k + 1], out[7 * k + 2], out[7 * k + 3], out[7 * k + 4],
                             out[7 * k + 5], out[7 * k + 6]});
        return r;
    }
    // cost_fn and solution_fn of one joint vector (make_cost_fn / make_is_solution_test_fn,
    // src/goal.cpp:163-203) for the goals/weights/thresholds in `costs`; `seed` is the
    // minimal-displacement reference.  goals: one pose per tip.
    struct Evaluation {
        double cost;
        bool is_solution;
    };
    Evaluation evaluate(const std::vector<double>& q, const std::vector<Pose>& goals,
                        const std::vector<double>& seed, const CostSpec& costs) const {
        check_size(q);
        check_size(seed);
        if (static_cast<int>(goals.size()) != n_tips_) throw std::invalid_argument("pick_ik_amd: one goal per tip is required");
        const pikamd_params p = to_params(costs, nullptr, nullptr, false);
        std::vector<double> g7;
        for (const Pose& g : goals) {
            const double v[7] = {g.x, g.y, g.z, g.qw, g.qx, g.qy, g.qz};
            g7.insert(g7.end(), v, v + 7);
        }
        double cost = 0.0;
        int32_t sol = 0;
        if (pikamd_cost_batch(h_, &p, 1, g7.data(), seed.data(), q.data(), &cost, &sol) != 0)
            throw std::runtime_error(pikamd_last_error());
        return Evaluation{cost, sol != 0};
    }

    // ik_memetic / ik_gradient with one goal per tip (searchPositionIK's ik_poses).
    // ik_seed_state (optional): the plugin's second joint vector (src/pick_ik_plugin.cpp:199-245) --
    // the minimal-displacement reference, while `initial_guess` is where the search starts (the
    // plugin re-randomises it on restarts).  nullptr = the initial guess is also the reference.
    std::optional<std::vector<double>> ik_memetic(const std::vector<double>& initial_guess,
                                                  const std::vector<Pose>& goals, const CostSpec& costs,
                                                  const MemeticIkParams& params,
                                                  bool approx_solution = false, uint64_t rng_seed = 0,
                                                  const std::vector<double>* ik_seed_state = nullptr) const {
        check_size(initial_guess);
        auto r = batch(to_params(costs, &params, nullptr, approx_solution), initial_guess, goals, rng_seed, 0,
                       ik_seed_state);
        if (r.status[0] > 0) return r.solution;
        return std::nullopt;
    }
    std::optional<std::vector<double>> ik_gradient(const std::vector<double>& initial_guess,
                                                   const std::vector<Pose>& goals, const CostSpec& costs,
                                                   const GradientIkParams& params,
                                                   bool approx_solution = false,
                                                   const std::vector<double>* ik_seed_state = nullptr) const {
        check_size(initial_guess);
        auto r = batch(to_params(costs, nullptr, &params, approx_solution), initial_guess, goals, 0, 0,
                       ik_seed_state);
        if (r.status[0] > 0) return r.solution;
        return std::nullopt;
    }

    // ... with a host cost function that takes part in the search, as kinematics::KinematicsBase::IKCostFn does
    // in the reference: one more Goal of weight 1 per tip pose inside cost_fn and under cost_threshold^2 in
    // solution_fn (src/pick_ik_plugin.cpp:130-135, src/goal.cpp:146-161, 175-182, 188-203).  Solved on the HOST
    // (pikamd_solve_batch_host: the reference's algorithm with the exact kernels' arithmetic -- an opaque host
    // closure cannot run inside a kernel); cost(q, pose_index) must be a pure function.
    using HostCostFn = std::function<double(const std::vector<double>& q, int pose_index)>;
    std::optional<std::vector<double>> ik_memetic(const std::vector<double>& initial_guess,
                                                  const std::vector<Pose>& goals, const CostSpec& costs,
                                                  const MemeticIkParams& params, const HostCostFn& cost,
                                                  bool approx_solution = false, uint64_t rng_seed = 0,
                                                  const std::vector<double>* ik_seed_state = nullptr) const {
        return host_single(to_params(costs, &params, nullptr, approx_solution), initial_guess, goals, cost, rng_seed,
                           ik_seed_state, params.max_time, params.gd_params.max_time);
    }
    std::optional<std::vector<double>> ik_gradient(const std::vector<double>& initial_guess,
                                                   const std::vector<Pose>& goals, const CostSpec& costs,
                                                   const GradientIkParams& params, const HostCostFn& cost,
                                                   bool approx_solution = false,
                                                   const std::vector<double>* ik_seed_state = nullptr) const {
        return host_single(to_params(costs, nullptr, &params, approx_solution), initial_guess, goals, cost, 0,
                           ik_seed_state, params.max_time, 0.0);
    }

    // make_fk_fn: tip pose of one joint vector
    Pose fk(const std::vector<double>& q) const {
        check_size(q);
        if (n_tips_ != 1) throw std::invalid_argument("pick_ik_amd: fk() is for one tip; use fk_tips()");
        double out[7];
        if (pikamd_fk_batch(h_, 1, q.data(), out) != 0) throw std::runtime_error(pikamd_last_error());
        return Pose{out[0], out[1], out[2], out[3], out[4], out[5], out[6]};
    }

    // pick_ik::ik_memetic (src/ik_memetic.cpp:285-373)
    std::optional<std::vector<double>> ik_memetic(const std::vector<double>& initial_guess,
                                                  const Pose& goal, const CostSpec& costs,
                                                  const MemeticIkParams& params,
                                                  bool approx_solution = false,
                                                  uint64_t rng_seed = 0) const {
        check_size(initial_guess);
        const pikamd_params p = to_params(costs, &params, nullptr, approx_solution);
        return single(p, initial_guess, goal, rng_seed);
    }

    // pick_ik::ik_gradient (src/ik_gradient.cpp:96-139)
    std::optional<std::vector<double>> ik_gradient(const std::vector<double>& initial_guess,
                                                   const Pose& goal, const CostSpec& costs,
                                                   const GradientIkParams& params,
                                                   bool approx_solution = false) const {
        check_size(initial_guess);
        const pikamd_params p = to_params(costs, nullptr, &params, approx_solution);
        return single(p, initial_guess, goal, 0);
    }

    // batch forms: goals [B] ([B][n_tips] for several tips), seeds [B][dof] row-major
    // ik_seed_states (optional, same shape as seeds): the minimal-displacement reference / the vector
    // returned on failure when it is not where the search starts (see ik_memetic above)
    BatchResult ik_memetic_batch(const std::vector<double>& seeds, const std::vector<Pose>& goals,
                                 const CostSpec& costs, const MemeticIkParams& params,
                                 bool approx_solution = false, uint64_t rng_seed = 0,
                                 int64_t problem_offset = 0,
                                 const std::vector<double>* ik_seed_states = nullptr) const {
        return batch(to_params(costs, &params, nullptr, approx_solution), seeds, goals, rng_seed,
                     problem_offset, ik_seed_states);
    }
    BatchResult ik_gradient_batch(const std::vector<double>& seeds, const std::vector<Pose>& goals,
                                  const CostSpec& costs, const GradientIkParams& params,
                                  bool approx_solution = false,
                                  const std::vector<double>* ik_seed_states = nullptr) const {
        return batch(to_params(costs, nullptr, &params, approx_solution), seeds, goals, 0, 0, ik_seed_states);
    }

    pikamd_solver* handle() const { return h_; }

    // pikamd_self_test: every kernel variant against the one-lane kernel on n generated targets of this
    // chain under these parameters; returns the mask of the variants that disagreed (now switched off for
    // this handle).  pikamd_set_option: pin a scheduling choice of the handle.
    uint32_t self_test(const CostSpec& costs, const MemeticIkParams& params, int n = 64) const {
        const pikamd_params p = to_params(costs, &params, nullptr, false);
        uint32_t mask = 0;
        if (pikamd_self_test(h_, &p, n, &mask) != 0) throw std::runtime_error(pikamd_last_error());
        return mask;
    }
    void set_option(const char* name, const char* value) const {
        if (pikamd_set_option(h_, name, value) != 0) throw std::runtime_error(pikamd_last_error());
    }

    // parameter mapping of the plugin (src/pick_ik_plugin.cpp:165-196)
    static pikamd_params to_params(const CostSpec& c, const MemeticIkParams* m,
                                   const GradientIkParams* g, bool approx) {
        pikamd_params p;
        pikamd_default_params(&p);
        p.position_scale = c.position_scale;
        p.rotation_scale = c.rotation_scale;
        p.position_threshold = c.position_threshold;
        p.orientation_threshold = c.orientation_threshold;
        p.cost_threshold = c.cost_threshold;
        p.center_joints_weight = c.center_joints_weight;
        p.avoid_joint_limits_weight = c.avoid_joint_limits_weight;
        p.minimal_displacement_weight = c.minimal_displacement_weight;
        p.return_approximate_solution = approx ? 1 : 0;
        if (m) {
            p.mode = 0;
            p.memetic_population_size = static_cast<int32_t>(m->population_size);
            p.memetic_elite_size = static_cast<int32_t>(m->elite_size);
            p.memetic_wipeout_fitness_tol = m->wipeout_fitness_tol;
            p.memetic_max_generations = m->max_generations;
            p.memetic_num_threads = static_cast<int32_t>(m->num_threads);
            p.memetic_stop_on_first_solution = m->stop_on_first_soln ? 1 : 0;
            p.stop_optimization_on_valid_solution = m->stop_optimization_on_valid_solution ? 1 : 0;
            p.gd_step_size = m->gd_params.step_size;
            p.gd_min_cost_delta = m->gd_params.min_cost_delta;
            p.memetic_gd_max_iters = m->gd_params.max_iterations;
        } else if (g) {
            p.mode = 1;
            p.gd_step_size = g->step_size;
            p.gd_min_cost_delta = g->min_cost_delta;
            p.gd_max_iters = g->max_iterations;
            p.stop_optimization_on_valid_solution = g->stop_optimization_on_valid_solution ? 1 : 0;
        }
        return p;
    }

  private:
    void check_size(const std::vector<double>& q) const {
        if (static_cast<int>(q.size()) != dof_) throw std::invalid_argument("pick_ik_amd: joint vector size != dof");
    }
    std::optional<std::vector<double>> single(const pikamd_params& p, const std::vector<double>& guess,
                                              const Pose& goal, uint64_t rng_seed) const {
        if (n_tips_ != 1) throw std::invalid_argument("pick_ik_amd: one goal per tip is required");
        const double g7[7] = {goal.x, goal.y, goal.z, goal.qw, goal.qx, goal.qy, goal.qz};
        std::vector<double> sol(dof_);
        int32_t status = 0;
        if (pikamd_solve_batch(h_, &p, 1, g7, guess.data(), rng_seed, 0, sol.data(), &status, nullptr,
                               nullptr) != 0)
            throw std::runtime_error(pikamd_last_error());
        if (status > 0) return sol;
        return std::nullopt;
    }
    // seeds: the start of the search of every problem; ik_seed_states (optional, same shape): the
    // minimal-displacement reference / the vector returned on failure when it differs from the start
    BatchResult batch(const pikamd_params& p, const std::vector<double>& seeds,
                      const std::vector<Pose>& goals, uint64_t rng_seed, int64_t offset,
                      const std::vector<double>* ik_seed_states = nullptr) const {
        if (goals.size() % static_cast<size_t>(n_tips_) != 0) throw std::invalid_argument("pick_ik_amd: goals size is not a multiple of n_tips");
        const size_t B = goals.size() / static_cast<size_t>(n_tips_);
        if (seeds.size() != B * static_cast<size_t>(dof_)) throw std::invalid_argument("pick_ik_amd: seeds size != B * dof");
        if (ik_seed_states && ik_seed_states->size() != seeds.size()) throw std::invalid_argument("pick_ik_amd: ik_seed_states size != B * dof");
        std::vector<double> g7(7 * goals.size());
        for (size_t b = 0; b < goals.size(); ++b) {
            const Pose& g = goals[b];
            const double v[7] = {g.x, g.y, g.z, g.qw, g.qx, g.qy, g.qz};
            for (int k = 0; k < 7; ++k) g7[7 * b + k] = v[k];
        }
        if (record_last_call_) last_call_ = CallRecord{p, g7, ik_seed_states ? *ik_seed_states : seeds, seeds, rng_seed, offset};
        BatchResult r;
        r.solution.resize(B * dof_);
        r.status.resize(B);
        r.cost.resize(B);
        r.stats.resize(B);
        pikamd_batch rec{};
        rec.B = static_cast<int64_t>(B);
        rec.goal_pos_quat = g7.data();
        rec.seed = ik_seed_states ? ik_seed_states->data() : seeds.data();
        rec.initial_guess = ik_seed_states ? seeds.data() : nullptr;
        rec.problem_offset = offset;
        rec.solution = r.solution.data();
        rec.status = r.status.data();
        rec.final_cost = r.cost.data();
        rec.stats = r.stats.data();
        const int rc = ik_seed_states ? pikamd_solve_batches(h_, &p, 1, &rec, rng_seed)
                                      : pikamd_solve_batch(h_, &p, rec.B, rec.goal_pos_quat, rec.seed, rng_seed, offset,
                                                           rec.solution, rec.status, rec.final_cost, rec.stats);
        if (rc != 0) throw std::runtime_error(pikamd_last_error());
        return r;
    }

    struct CostTrampoline {
        const HostCostFn* fn;
        std::vector<double> q;
    };
    static double cost_trampoline(const double* q, int32_t dof, int32_t pose, void* user) {
        auto* t = static_cast<CostTrampoline*>(user);
        t->q.assign(q, q + dof);
        return (*t->fn)(t->q, static_cast<int>(pose));
    }
    std::optional<std::vector<double>> host_single(const pikamd_params& p, const std::vector<double>& initial_guess,
                                                   const std::vector<Pose>& goals, const HostCostFn& cost, uint64_t rng_seed,
                                                   const std::vector<double>* ik_seed_state, double max_time,
                                                   double gd_max_time) const {
        // On the host the reference's wall-clock limits apply as in the reference (src/ik_memetic.cpp:75-78,
        // 226-228, src/ik_gradient.cpp:112-115): max_time of the query, max_time of one elite's descent.
        const auto seconds = [](double t) {
            char buf[40];
            std::snprintf(buf, sizeof buf, "%.9g", (t > 0.0 && t < 1.0e9) ? t : 0.0);
            return std::string(buf);
        };
        set_option("host_max_time", seconds(max_time).c_str());
        set_option("host_gd_max_time", seconds(gd_max_time).c_str());
        check_size(initial_guess);
        if (ik_seed_state) check_size(*ik_seed_state);
        if (!cost) throw std::invalid_argument("pick_ik_amd: the host solver is for queries with a cost function");
        if (static_cast<int>(goals.size()) != n_tips_) throw std::invalid_argument("pick_ik_amd: one goal per tip is required");
        std::vector<double> g7;
        for (const Pose& g : goals) {
            const double v[7] = {g.x, g.y, g.z, g.qw, g.qx, g.qy, g.qz};
            g7.insert(g7.end(), v, v + 7);
        }
        std::vector<double> sol(static_cast<size_t>(dof_));
        int32_t status = 0;
        CostTrampoline t{&cost, {}};
        const std::vector<double>& seed = ik_seed_state ? *ik_seed_state : initial_guess;
        if (pikamd_solve_batch_host(h_, &p, 1, g7.data(), seed.data(), initial_guess.data(), rng_seed, 0, &cost_trampoline,
                                    &t, sol.data(), &status, nullptr, nullptr) != 0)
            throw std::runtime_error(pikamd_last_error());
        if (status > 0) return sol;
        return std::nullopt;
    }

    void load_variables() {
        std::vector<double> v(7 * dof_);
        pikamd_variables(h_, v.data());
        for (int j = 0; j < dof_; ++j)
            robot_.variables.push_back({v[7 * j], v[7 * j + 1], v[7 * j + 2], v[7 * j + 6] != 0.0,
                                        v[7 * j + 3], v[7 * j + 4], v[7 * j + 5]});
    }

    int dof_;
    int n_tips_ = 1;
    pikamd_solver* h_ = nullptr;
    Robot robot_;
    mutable CallRecord last_call_;
    mutable bool record_last_call_ = false;
};

} // namespace pick_ik_amd
