/*
 * pik_oracle.h -- CPU ORACLE for the pick_ik hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm (PickNikRobotics/pick_ik v1.1.2):
 *   src/ik_memetic.cpp, src/ik_gradient.cpp, src/goal.cpp:17-144,163-203, src/robot.cpp:23-105,
 *   and the third-party arithmetic those files call (MoveIt RobotState FK, Eigen 3.4 quaternion
 *   conversion / angularDistance, urdfdom rpy->quaternion), each function citing what it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (pick_ik_amd/) never includes, links or calls anything in oracle/.
 *
 * Two deliberate, documented departures from the reference:
 *   1. RNG: the reference draws from rsl's thread_local std::mt19937 seeded by std::random_device
 *      (never seeded by pick_ik => non-deterministic).  The oracle draws from a counter-based
 *      Philox4x32-10 keyed by (seed, problem, epoch, individual, draw slot) so that a sequential
 *      CPU and a parallel GPU consume identical numbers.
 *   2. Wall-clock limits (max_time, memetic_gd_max_time) are disabled; the iteration budgets
 *      (max_generations, gd max_iterations) bind instead.
 *
 * PARITY STATUS: deterministic pieces (FK, pose cost, frame tests, step(), ik_gradient) are pinned
 * against the reference's own known-answer tests (tests/goal_tests.cpp, tests/ik_tests.cpp) in
 * tests/test_oracle_golden.py.  Memetic *joint vectors* are unpinned by the reference itself
 * (unseeded RNG, pose-space assertions only, tests/ik_memetic_tests.cpp:124) -- "parity unpinned"
 * for those; the oracle is pinned there only through the reference's pose-space acceptance tests.
 */
#ifndef PIK_ORACLE_H
#define PIK_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PKO_MAX_DOF 16
#define PKO_MAX_TIPS 8

/* Status codes (moveit_msgs::msg::MoveItErrorCodes values used by src/pick_ik_plugin.cpp:209-217). */
#define PKO_SUCCESS 1
#define PKO_APPROXIMATE 2 /* ik_* returned best-so-far because approx_solution was set */
#define PKO_NO_IK_SOLUTION (-31)

#define PKO_JOINT_REVOLUTE 0
#define PKO_JOINT_PRISMATIC 1
/* A PLANAR joint (moveit::core::PlanarJointModel, holonomic): three variables x, y, theta and
 * computeTransform = Translation(x, y, 0) * AngleAxis(theta, UnitZ) in the joint frame -- the product
 * of a prismatic joint along x, one along y and a revolute one about z with identity origins in
 * between.  The three variables occupy three consecutive slots of the chain arrays: _X carries the
 * joint's origin, _Y and _THETA follow it immediately (their origin / axis entries are ignored).
 * pick_ik treats every variable of a multi-variable joint independently (src/robot.cpp:144-150). */
#define PKO_JOINT_PLANAR_X 2
#define PKO_JOINT_PLANAR_Y 3
#define PKO_JOINT_PLANAR_THETA 4
/* A FLOATING joint (moveit::core::FloatingJointModel; the reference's own statement of its frame is
 * src/forward_kinematics.cpp:64-70, the live path reaches it through RobotState, src/fk_moveit.cpp:22-31):
 * seven variables trans_x trans_y trans_z rot_x rot_y rot_z rot_w in consecutive slots, ONE transform
 *   Translation3d(v[0], v[1], v[2]) * Quaterniond(w = v[6], x = v[3], y = v[4], z = v[5])
 * -- the quaternion is used as it is (Eigen toRotationMatrix, no normalisation), exactly what
 * RobotState::updateLinkTransforms multiplies in.  pick_ik treats the seven variables independently
 * (src/robot.cpp:144-150).  _TX carries the joint's origin; axis entries are ignored. */
#define PKO_JOINT_FLOATING_TX 5
#define PKO_JOINT_FLOATING_TY 6
#define PKO_JOINT_FLOATING_TZ 7
#define PKO_JOINT_FLOATING_RX 8
#define PKO_JOINT_FLOATING_RY 9
#define PKO_JOINT_FLOATING_RZ 10
#define PKO_JOINT_FLOATING_RW 11

/* Mirrors src/pick_ik_parameters.yaml (names and defaults), minus wall-clock limits. */
typedef struct pko_params {
    int32_t mode; /* 0 = "global" (memetic), 1 = "local" (gradient descent) */
    double gd_step_size;
    int32_t gd_max_iters;
    double gd_min_cost_delta;
    double position_threshold;
    double orientation_threshold;
    double cost_threshold;
    double position_scale;
    double rotation_scale;
    double center_joints_weight;
    double avoid_joint_limits_weight;
    double minimal_displacement_weight;
    int32_t stop_optimization_on_valid_solution;
    int32_t memetic_num_threads;
    int32_t memetic_stop_on_first_solution;
    int32_t memetic_population_size;
    int32_t memetic_elite_size;
    double memetic_wipeout_fitness_tol;
    int32_t memetic_max_generations;
    int32_t memetic_gd_max_iters;
    int32_t return_approximate_solution;
} pko_params;

typedef struct pko_stats {
    int64_t cost_evals;  /* literal count of cost_fn invocations the reference would make */
    int32_t generations; /* memetic generations run (species 0) / gd iterations in local mode */
    int32_t wipeouts;
    int32_t pool_erasures; /* mating-pool erase events (src/ik_memetic.cpp:172-179) */
    int32_t reserved;
} pko_stats;

typedef struct pko_chain pko_chain;

void pko_default_params(pko_params* p);

/* Serial chain base->tip.  origin_xyz_rpy [dof][6], axis [dof][3], joint_type [dof],
 * tip_xyz_rpy [6] (fixed transform after the last joint), limits [dof]. */
pko_chain* pko_chain_create(int32_t dof, const double* origin_xyz_rpy, const double* axis,
                            const int32_t* joint_type, const double* tip_xyz_rpy,
                            const double* qmin, const double* qmax, const double* vmax,
                            const uint8_t* bounded);
/* Several tip links (the plugin's tip_frames; reference src/pick_ik_plugin.cpp:57-69,
 * src/robot.cpp:105-160, src/goal.cpp:27-49, 80-89).  `dof` active variables; tip k hangs off
 * tip_n_joints[k] joints given like a chain of their own -- origin_xyz_rpy / axis / joint_type /
 * variable (index of each joint's variable, strictly increasing along a path) are the per-tip
 * arrays concatenated in tip order; tip_xyz_rpy [n_tips][6].  With several tips every goal /
 * pose array of this API holds n_tips consecutive poses per problem: goal_pos_quat [B][n_tips][7],
 * pko_fk_batch -> [n][n_tips][7], pko_fk_matrix -> [n_tips][12]. */
pko_chain* pko_chain_create_multi(int32_t dof, int32_t n_tips, const int32_t* tip_n_joints,
                                  const int32_t* variable, const double* origin_xyz_rpy,
                                  const double* axis, const int32_t* joint_type,
                                  const double* tip_xyz_rpy, const double* qmin, const double* qmax,
                                  const double* vmax, const uint8_t* bounded);
int32_t pko_chain_n_tips(const pko_chain* c);
void pko_chain_destroy(pko_chain* c);
/* out [dof][7]: min max mid half_span max_velocity_rcp minimal_displacement_factor bounded */
void pko_chain_variables(const pko_chain* c, double* out);

/* ---- primitives (parity hooks) ---- */
/* pose as row-major R[9] followed by t[3] */
void pko_fk_matrix(const pko_chain* c, const double* q, double* pose12);
/* pos_quat: x y z qw qx qy qz (Eigen matrix->quaternion of the tip rotation) */
void pko_fk_batch(const pko_chain* c, int64_t n, const double* q, double* pos_quat);
void pko_pose_from_pos_quat(const double* pos_quat7, double* pose12);
double pko_linear_distance(const double* pose12_a, const double* pose12_b);
double pko_angular_distance(const double* pose12_a, const double* pose12_b);
double pko_pose_cost(const double* goal12, const double* frame12, double position_scale,
                     double rotation_scale);
/* has_pos/has_ori: whether the optional threshold is set */
int32_t pko_frame_test(const double* goal12, const double* frame12, int32_t has_pos,
                       double pos_thr, int32_t has_ori, double ori_thr);
double pko_center_joints_cost(const pko_chain* c, const double* q);
double pko_avoid_joint_limits_cost(const pko_chain* c, const double* q);
double pko_minimal_displacement_cost(const pko_chain* c, const double* q, const double* seed);

/* cost_fn / solution_fn of one problem, batched over n candidate joint vectors q [n][dof]
 * (goal and seed shared). */
void pko_cost_batch(const pko_chain* c, const pko_params* p, const double* goal_pos_quat,
                    const double* seed, int64_t n, const double* q, double* cost,
                    int32_t* is_solution);

/* One step() of src/ik_gradient.cpp:24-94 for n independent (goal, seed, state) triples.
 * state in/out: local [n][dof], best [n][dof], local_cost [n], best_cost [n];
 * out: gradient [n][dof], improved [n]. */
void pko_gd_step_batch(const pko_chain* c, const pko_params* p, int64_t n,
                       const double* goal_pos_quat, const double* seed, double* local,
                       double* best, double* local_cost, double* best_cost, double* gradient,
                       int32_t* improved);

/* Counter-based RNG exposed for tests: Philox4x32-10 block and the [0,1) double draw. */
void pko_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double pko_rng_u01(uint64_t seed, uint32_t stream, uint64_t problem, uint32_t epoch,
                   uint32_t individual, uint32_t slot);

/* ---- solvers ---- */
/* Batch of B independent problems: goal [B][7] (x y z qw qx qy qz, chain base frame),
 * seed [B][dof]; problem b uses RNG problem index problem_offset + b.
 * solution [B][dof] (seed on failure, src/pick_ik_plugin.cpp:215-216), status [B],
 * final_cost [B] (may be NULL), stats [B] (may be NULL).  num_threads: OpenMP threads (>=1). */
int32_t pko_solve_batch(const pko_chain* c, const pko_params* p, int64_t B,
                        const double* goal_pos_quat, const double* seed, uint64_t rng_seed,
                        int64_t problem_offset, double* solution, int32_t* status,
                        double* final_cost, pko_stats* stats, int32_t num_threads);

/* Same with a separate start of the search: `seed` stays the minimal-displacement reference and the
 * vector returned on failure (ik_seed_state), `initial_guess` [B][dof] is where the search starts
 * (init_state; re-randomised by the plugin on restarts, src/pick_ik_plugin.cpp:199-245).
 * NULL = seed.  final_cost on failure is the cost of the initial guess. */
int32_t pko_solve_batch_guess(const pko_chain* c, const pko_params* p, int64_t B,
                              const double* goal_pos_quat, const double* seed,
                              const double* initial_guess, uint64_t rng_seed,
                              int64_t problem_offset, double* solution, int32_t* status,
                              double* final_cost, pko_stats* stats, int32_t num_threads);

/* ... with a host cost function: kinematics::KinematicsBase::IKCostFn as pick_ik uses it -- one more Goal of
 * weight 1 per tip pose behind the joint goals, summed into cost_fn and held below cost_threshold^2 by
 * solution_fn (src/pick_ik_plugin.cpp:130-135, src/goal.cpp:146-161, 175-182, 188-203).
 * cost_function(q, dof, pose_index, user) must be a pure function; NULL = none.  One thread when given. */
typedef double (*pko_cost_fn)(const double* q, int32_t dof, int32_t pose_index, void* user);
int32_t pko_solve_batch_cost_fn(const pko_chain* c, const pko_params* p, int64_t B,
                                const double* goal_pos_quat, const double* seed,
                                const double* initial_guess, uint64_t rng_seed,
                                int64_t problem_offset, pko_cost_fn cost_function, void* user,
                                double* solution, int32_t* status,
                                double* final_cost, pko_stats* stats, int32_t num_threads);

/* Mimic joints on a tip path: no variables (src/robot.cpp:144-150), moved with their master by the reference's
 * forward kinematics (RobotState::setJointGroupPositions -> updateMimicJoints, src/fk_moveit.cpp:22): one more step
 * of the chain product, behind the joint of variable `after_variable` (-1: in front of the first), at
 * multiplier * q[master_variable] + offset.  Same struct as include/pick_ik_amd.h pikamd_mimic_joint. */
#define PKO_MAX_MIMIC 4
typedef struct pko_mimic_joint {
    int32_t tip, after_variable, master_variable, joint_type;
    double origin_xyz_rpy[6];
    double axis[3];
    double multiplier, offset;
} pko_mimic_joint;
int32_t pko_chain_set_mimic(pko_chain* c, int32_t n, const pko_mimic_joint* joints);

int32_t pko_max_threads(void);

/* 0 = libm (reference semantics, default), 1 = portable (bit-compatible with the strict GPU build) */
void pko_set_math_mode(int32_t mode);
int32_t pko_get_math_mode(void);
void pko_sincos(double x, double* s, double* c);
double pko_atan2(double y, double x);

#ifdef __cplusplus
}
#endif
#endif
