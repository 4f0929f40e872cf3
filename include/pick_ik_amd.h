/*
 * pick_ik_amd.h -- C ABI of the MI355X-native batched IK solver (libpick_ik_amd.so).
 *
 * This is the boundary a MoveIt-side maintainer binds: pick_ik's plugin
 * (reference src/pick_ik_plugin.cpp:73-294) builds cost_fn/solution_fn closures and calls
 * ik_memetic()/ik_gradient() once per pose; this library takes the same inputs as plain arrays
 * (B = 1 from the plugin shim, B = thousands..millions from batch callers) and runs the whole
 * solve on the GPU.  No torch / Eigen / MoveIt types appear in any signature.
 *
 * Conventions
 *   - every function returns 0 on success or a negative PIKAMD_E* code and never throws;
 *     pikamd_last_error() returns a thread-local message for the last failure on this thread.
 *   - *_batch entry points take HOST pointers and do their own H2D/D2H;
 *     *_device entry points take DEVICE pointers (HBM-resident inputs/outputs) plus a hipStream_t
 *     passed as void* and only enqueue work -- nothing is synchronised.
 *   - all floating point is IEEE-754 binary64, exactly like the reference (double / Eigen::Isometry3d).
 *   - a solver handle is bound to one GPU and is not thread-safe (one handle per host thread/GPU).
 *   - there is NO CPU fallback: every entry point fails with PIKAMD_ENODEVICE when no gfx950
 *     device is available.
 */
#ifndef PICK_IK_AMD_H
#define PICK_IK_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIKAMD_MAX_DOF 16 /* variables of a chain; the kernels are instantiated for 1..16 */

/* status[] values: moveit_msgs::msg::MoveItErrorCodes as used in src/pick_ik_plugin.cpp:209-217 */
#define PIKAMD_SUCCESS 1
#define PIKAMD_APPROXIMATE 2 /* best-so-far returned because return_approximate_solution was set
                                (src/ik_memetic.cpp:277-280, src/ik_gradient.cpp:135-137) */
#define PIKAMD_NO_IK_SOLUTION (-31)

/* error codes */
#define PIKAMD_EINVAL (-1)
#define PIKAMD_ENODEVICE (-2)
#define PIKAMD_EHIP (-3)
#define PIKAMD_EUNSUPPORTED (-4)

#define PIKAMD_JOINT_REVOLUTE 0
#define PIKAMD_JOINT_PRISMATIC 1
/* A PLANAR joint (moveit::core::PlanarJointModel, the usual virtual joint of a mobile base; the
 * reference reaches it through RobotState, src/forward_kinematics.cpp:72-79 / src/fk_moveit.cpp): three
 * variables x, y, theta with computeTransform = Translation(x, y, 0) * AngleAxis(theta, UnitZ) in the
 * joint frame, i.e. exactly a prismatic joint along x, one along y and a revolute one about z with
 * identity origins in between.  Its variables occupy three consecutive slots of the chain arrays:
 * _X carries the joint's origin, _Y and _THETA follow immediately (their origin / axis entries are
 * ignored); qmin / qmax / vmax / bounded are per variable as for every joint (pick_ik treats the
 * variables of a multi-variable joint independently, src/robot.cpp:144-150). */
#define PIKAMD_JOINT_PLANAR_X 2
#define PIKAMD_JOINT_PLANAR_Y 3
#define PIKAMD_JOINT_PLANAR_THETA 4
/* A FLOATING joint (moveit::core::FloatingJointModel, the virtual joint of a free-flying base; the
 * reference states its frame in src/forward_kinematics.cpp:64-70 and reaches it live through
 * RobotState, src/fk_moveit.cpp:22-31): seven variables trans_x trans_y trans_z rot_x rot_y rot_z rot_w
 * in seven consecutive slots and ONE transform,
 *     Translation3d(v[0], v[1], v[2]) * Quaterniond(w = v[6], x = v[3], y = v[4], z = v[5]),
 * the quaternion taken as it is (Eigen toRotationMatrix of the unnormalised value -- what MoveIt
 * multiplies in; pick_ik treats the seven variables independently, src/robot.cpp:144-150, so the
 * search moves them one by one between their bounds).  _TX carries the joint's origin; axis entries
 * are ignored.  These variables are not single-axis motions: a chain with a floating joint runs the
 * literal kernels (MoveIt's chain product and 2 dof + 3 evaluations per gradient step, the arithmetic
 * the verification build checks bit for bit against the oracle), not the Denavit-Hartenberg ones. */
#define PIKAMD_JOINT_FLOATING_TX 5
#define PIKAMD_JOINT_FLOATING_TY 6
#define PIKAMD_JOINT_FLOATING_TZ 7
#define PIKAMD_JOINT_FLOATING_RX 8
#define PIKAMD_JOINT_FLOATING_RY 9
#define PIKAMD_JOINT_FLOATING_RZ 10
#define PIKAMD_JOINT_FLOATING_RW 11

/* Serial chain base -> tip; replaces what Robot::from / make_fk_fn pull out of the MoveIt
 * RobotModel (src/robot.cpp:44-85, src/fk_moveit.cpp:11-35).  Fixed joints are collapsed into the
 * next joint's origin; the fixed links after the last actuated joint are collapsed into tip. */
typedef struct pikamd_chain {
    int32_t dof;
    const double* origin_xyz_rpy; /* [dof][6] URDF <origin xyz rpy> of each joint           */
    const double* axis;           /* [dof][3] joint axis (normalised internally)            */
    const int32_t* joint_type;    /* [dof] PIKAMD_JOINT_*; NULL = all revolute              */
    const double* tip_xyz_rpy;    /* [6]  fixed transform after the last joint              */
    const double* qmin;           /* [dof] position bounds                                  */
    const double* qmax;           /* [dof]                                                  */
    const double* vmax;           /* [dof] max velocity (minimal-displacement weights); NULL = 0 */
    const uint8_t* bounded;       /* [dof] position_bounded_; NULL = all bounded            */
} pikamd_chain;

/* Several tip links -- the plugin's tip_frames (src/pick_ik_plugin.cpp:57-69; one pose cost and one
 * frame test per tip, src/goal.cpp:27-49, 80-89; the active variables are the union of the joints on
 * the way to any tip, src/robot.cpp:130-160).  Each tip is described by the joints on ITS path from
 * the base, like a chain of its own; a joint shared by several tips (a torso) appears in every such
 * path with the same variable index.  With a multi-tip solver every goal / pose array of this API
 * holds n_tips consecutive poses per problem: goal_pos_quat [B][n_tips][7], fk -> [n][n_tips][7]. */
#define PIKAMD_MAX_TIPS 8
typedef struct pikamd_tip {
    int32_t n_joints;
    const int32_t* variable;      /* [n_joints] index of each joint's variable, strictly increasing */
    const double* origin_xyz_rpy; /* [n_joints][6] */
    const double* axis;           /* [n_joints][3] */
    const int32_t* joint_type;    /* [n_joints]; NULL = all revolute */
    const double* tip_xyz_rpy;    /* [6] */
} pikamd_tip;
typedef struct pikamd_multi_chain {
    int32_t dof;    /* number of active variables */
    int32_t n_tips; /* 1 .. PIKAMD_MAX_TIPS */
    const pikamd_tip* tips;
    const double* qmin; /* [dof] */
    const double* qmax;
    const double* vmax;     /* NULL = 0 */
    const uint8_t* bounded; /* NULL = all bounded */
} pikamd_multi_chain;

/* Mirrors src/pick_ik_parameters.yaml (same names, same defaults) minus the wall-clock limits
 * (memetic_gd_max_time, the plugin timeout): iteration budgets bind instead. */
typedef struct pikamd_params {
    int32_t mode; /* 0 = "global" (memetic, src/ik_memetic.cpp), 1 = "local" (src/ik_gradient.cpp) */
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
    int32_t memetic_num_threads; /* species (lock-step on the GPU); pow2ceil(it) * pow2ceil(elite_size) <= 64 */
    int32_t memetic_stop_on_first_solution;
    int32_t memetic_population_size;
    int32_t memetic_elite_size;
    double memetic_wipeout_fitness_tol;
    int32_t memetic_max_generations;
    int32_t memetic_gd_max_iters;
    int32_t return_approximate_solution; /* KinematicsQueryOptions::return_approximate_solution */
} pikamd_params;

/* per-problem counters (optional output) */
typedef struct pikamd_stats {
    int64_t cost_evals;  /* cost_fn invocations the reference algorithm makes for this solve */
    int32_t generations; /* memetic generations run / gd iterations in local mode */
    int32_t wipeouts;
    int32_t pool_erasures;
    int32_t reserved;
} pikamd_stats;

typedef struct pikamd_solver pikamd_solver;

/* yaml defaults */
void pikamd_default_params(pikamd_params* p);

/* Replaces PickIKPlugin::initialize's model extraction (src/pick_ik_plugin.cpp:22-71).
 * device_ordinal: HIP device index (>= 0). */
int32_t pikamd_create(const pikamd_chain* chain, int32_t device_ordinal, pikamd_solver** out);
int32_t pikamd_create_multi(const pikamd_multi_chain* chain, int32_t device_ordinal,
                            pikamd_solver** out);
int32_t pikamd_n_tips(const pikamd_solver* s); /* 1 for a solver made by pikamd_create */

/* Mimic joints on a tip path.  A mimic joint is no variable (src/robot.cpp:144-150 keeps it out of the active
 * variables), but the reference's forward kinematics moves it with its master: RobotState::setJointGroupPositions
 * (src/fk_moveit.cpp:22) ends in updateMimicJoints, i.e. the joint sits at multiplier * q[master] + offset.  Such a
 * joint is one more step of the chain product whose value follows a variable; it is declared here, behind the
 * variable whose joint it follows on the path.  The origins of the chain description are split accordingly: the
 * mimic joint's origin is the fixed transform from the previous moving joint of the path to it, and the next
 * joint's origin starts behind it.  (A mimic joint with multiplier 0 is a constant joint: fold it into the next
 * origin instead.)  Chains with mimic joints are solved by the exact kernels (like chains with a floating joint:
 * the Denavit-Hartenberg form has one variable per step).  Call before the first solve; n = 0 removes them. */
#define PIKAMD_MAX_MIMIC 4 /* per tip path */
typedef struct pikamd_mimic_joint {
    int32_t tip;             /* the tip path it lies on (0 for a solver made by pikamd_create) */
    int32_t after_variable;  /* it follows the joint of this variable on that path; -1 = in front of the first */
    int32_t master_variable; /* value = multiplier * q[master_variable] + offset */
    int32_t joint_type;      /* PIKAMD_JOINT_REVOLUTE or PIKAMD_JOINT_PRISMATIC */
    double origin_xyz_rpy[6];
    double axis[3];
    double multiplier, offset;
} pikamd_mimic_joint;
int32_t pikamd_set_mimic_joints(pikamd_solver* s, int32_t n, const pikamd_mimic_joint* joints);
void pikamd_destroy(pikamd_solver* s);
/* out [dof][7]: min max mid half_span max_velocity_rcp minimal_displacement_factor bounded
 * (Robot::Variable table, include/pick_ik/robot.hpp:15-37) */
int32_t pikamd_variables(const pikamd_solver* s, double* out);

/* make_fk_fn (src/fk_moveit.cpp:11-35): tip pose for n joint vectors.
 * q [n][dof] -> pos_quat [n][7] = x y z qw qx qy qz. */
int32_t pikamd_fk_batch(pikamd_solver* s, int64_t n, const double* q, double* pos_quat);

/* make_cost_fn / make_is_solution_test_fn (src/goal.cpp:188-203, 163-186) for n candidates.
 * goal [n][7], seed [n][dof] (minimal-displacement reference), q [n][dof] ->
 * cost [n] (may be NULL), is_solution [n] (may be NULL). */
int32_t pikamd_cost_batch(pikamd_solver* s, const pikamd_params* p, int64_t n,
                          const double* goal_pos_quat, const double* seed, const double* q,
                          double* cost, int32_t* is_solution);

/* One step() (src/ik_gradient.cpp:24-94) on n independent GradientIk states.
 * in/out: local [n][dof], best [n][dof], local_cost [n], best_cost [n];
 * out: gradient [n][dof], improved [n] (may be NULL). */
int32_t pikamd_gd_step_batch(pikamd_solver* s, const pikamd_params* p, int64_t n,
                             const double* goal_pos_quat, const double* seed, double* local,
                             double* best, double* local_cost, double* best_cost,
                             double* gradient, int32_t* improved);

/* ik_memetic (src/ik_memetic.cpp:285-373) or ik_gradient (src/ik_gradient.cpp:96-139), selected by
 * p->mode, for B independent problems.
 *   goal_pos_quat [B][7]  goal pose in the chain's base frame (x y z qw qx qy qz)
 *   seed          [B][dof] ik_seed_state
 *   rng_seed, problem_offset: random streams are keyed by (rng_seed, problem_offset + b), so a
 *                 batch sharded over several GPUs/calls gives the same answers as one call.
 *   solution [B][dof] (seed on failure, src/pick_ik_plugin.cpp:213-217), status [B],
 *   final_cost [B] (may be NULL), stats [B] (may be NULL).
 * Synchronous; staged through the library's own pinned buffers and stream (it never touches the
 * caller's slots of the device entry points). */
int32_t pikamd_solve_batch(pikamd_solver* s, const pikamd_params* p, int64_t B,
                           const double* goal_pos_quat, const double* seed, uint64_t rng_seed,
                           int64_t problem_offset, double* solution, int32_t* status,
                           double* final_cost, pikamd_stats* stats);

/* Same, on HBM-resident buffers; enqueues on `stream` (a hipStream_t, NULL = default stream) and
 * returns without synchronising.  Scratch memory is owned by the handle and reused across calls on
 * the same `slot` (0 <= slot < PIKAMD_MAX_SLOTS); calls on different slots may be in flight
 * concurrently on different streams. */
#define PIKAMD_MAX_SLOTS 128
int32_t pikamd_solve_batch_device(pikamd_solver* s, const pikamd_params* p, int64_t B,
                                  const double* d_goal_pos_quat, const double* d_seed,
                                  uint64_t rng_seed, int64_t problem_offset, double* d_solution,
                                  int32_t* d_status, double* d_final_cost, pikamd_stats* d_stats,
                                  void* stream, int32_t slot);
int32_t pikamd_fk_batch_device(pikamd_solver* s, int64_t n, const double* d_q, double* d_pos_quat,
                               void* stream);
/* Optional: allocate slot `slot`'s scratch for calls of up to B problems (all batches together) with
 * these parameters and upload the chain constants now, so that the first solve on the slot does not
 * allocate (hipMalloc synchronises the device). */
int32_t pikamd_reserve(pikamd_solver* s, const pikamd_params* p, int64_t B, int32_t slot,
                       void* stream);

/* ---- several batches per call -------------------------------------------------------------
 * The plugin solves one pose per call (src/pick_ik_plugin.cpp:73-294); a batch caller (a planner
 * sampling goal poses, a server collecting requests) has MANY independent batches.  One call here
 * takes up to PIKAMD_MAX_BATCHES of them and solves their problems as ONE pool: the persistent
 * wavefronts pull problems of all batches from one queue, so that the long tail of one batch (the
 * ~1 % of targets that run all memetic_max_generations) overlaps with the bulk of the others instead
 * of idling the chip.  Every batch gets exactly the answers a call of its own would give (random
 * streams are keyed by (rng_seed, problem_offset + b) per batch; asserted bit for bit by the tests).
 *
 * A batch also separates the two joint vectors the plugin keeps per call
 * (src/pick_ik_plugin.cpp:199-245): `seed` = ik_seed_state, which the minimal-displacement cost
 * (src/goal.cpp:131-144) measures against and which is returned on failure, and `initial_guess` =
 * init_state, where the search starts -- ik_seed_state on the first attempt, a random valid
 * configuration on restarts (:241-245).  NULL = seed.  On failure final_cost holds the cost of the
 * initial guess. */
#define PIKAMD_MAX_BATCHES 64
typedef struct pikamd_batch {
    int64_t B;
    const double* goal_pos_quat; /* [B][n_tips][7] */
    const double* seed;          /* [B][dof] ik_seed_state */
    const double* initial_guess; /* [B][dof] or NULL (= seed) */
    int64_t problem_offset;      /* random streams of problem b: (rng_seed, problem_offset + b) */
    double* solution;            /* [B][dof] */
    int32_t* status;             /* [B] */
    double* final_cost;          /* [B] or NULL */
    pikamd_stats* stats;         /* [B] or NULL */
    uint32_t* completed;         /* device entry point only, or NULL: a counter in device / pinned host
                                    memory that is incremented (release, system scope) once per
                                    finished problem of this batch, AFTER its results are in memory:
                                    completed == B  <=>  the batch is done, before the call is */
} pikamd_batch;

/* device pointers in every record; enqueues on `stream` and returns (see pikamd_solve_batch_device) */
int32_t pikamd_solve_batches_device(pikamd_solver* s, const pikamd_params* p, int32_t n_batches,
                                    const pikamd_batch* batches, uint64_t rng_seed, void* stream,
                                    int32_t slot);

/* Host pointers, asynchronous: inputs are copied into pinned staging memory, H2D copies, kernels and
 * D2H copies are enqueued on job `job`'s own stream (0 <= job < PIKAMD_MAX_HOST_JOBS), and the call
 * returns.  pikamd_wait(job) blocks until the job is done and copies the results into the batches'
 * output arrays (which must stay valid until then).  Jobs overlap each other's PCIe transfers and
 * kernels.  The input arrays may be reused as soon as the call returns. */
#define PIKAMD_MAX_HOST_JOBS 16
int32_t pikamd_solve_batches_async(pikamd_solver* s, const pikamd_params* p, int32_t n_batches,
                                   const pikamd_batch* batches, uint64_t rng_seed, int32_t job);
int32_t pikamd_wait(pikamd_solver* s, int32_t job);
/* = pikamd_solve_batches_async + pikamd_wait on an internal job */
int32_t pikamd_solve_batches(pikamd_solver* s, const pikamd_params* p, int32_t n_batches,
                             const pikamd_batch* batches, uint64_t rng_seed);

/* ---- several GPUs of one node ----------------------------------------------------------------
 * Every target pose is an independent problem (the reference solves one per call,
 * src/pick_ik_plugin.cpp:73-294; its only parallelism are the species threads of one solve,
 * src/ik_memetic.cpp:312-353), so a batch shards trivially: device r of n solves the contiguous range
 * pikamd_shard_bounds(B, r, n) with problem_offset + its first index as the random-stream key -- the
 * sharded call returns exactly what one call over the whole batch on one device returns, whatever n is.
 * solvers[r] is a handle created on device r's ordinal for the same chain (one handle per device; a
 * handle may not appear twice).  Host pointers: every device gets its own host thread, which stages
 * its shard in up to two chunks (option "shard_chunks"; asynchronous jobs 0, 1 of that handle: the second
 * one's PCIe transfers overlap the first one's kernels) and writes the results straight into the caller's arrays -- the "gather" is the
 * shards' device-to-host copies; no collective is needed for host-resident results (a caller who wants
 * the results resident on every GPU all-gathers them with RCCL, as bench.py does).  Synchronous. */
void pikamd_shard_bounds(int64_t total, int32_t rank, int32_t world, int64_t* lo, int64_t* hi);
int32_t pikamd_solve_batch_sharded(pikamd_solver* const* solvers, int32_t n_devices, const pikamd_params* p,
                                   int64_t B, const double* goal_pos_quat, const double* seed,
                                   const double* initial_guess /* NULL = seed */, uint64_t rng_seed,
                                   int64_t problem_offset, double* solution, int32_t* status,
                                   double* final_cost /* may be NULL */, pikamd_stats* stats /* may be NULL */);

/* ---- robot description -> solver ----------------------------------------------------------
 * Robot::from / get_link_indices / get_active_variable_indices (src/robot.cpp:44-160) over a URDF
 * document instead of a live MoveIt RobotModel: the actuated single-variable joints on the way from
 * base_link to the tip link(s), fixed joints folded into the next origin, mimic joints excluded
 * (src/robot.cpp:144-150; held at zero), continuous joints unbounded.  With several tips the variables
 * are numbered in the order they are first met walking the tips' paths in the order given (list the
 * tips so that shared joints come first).  pikamd_urdf_extract needs no device and returns the
 * description it found (the joint vector's order is variable_names); pikamd_create_from_urdf =
 * extract + pikamd_create / pikamd_create_multi. */
#define PIKAMD_MAX_NAME 64
typedef struct pikamd_urdf_model {
    int32_t dof, n_tips;
    char variable_names[PIKAMD_MAX_DOF][PIKAMD_MAX_NAME];
    double qmin[PIKAMD_MAX_DOF], qmax[PIKAMD_MAX_DOF], vmax[PIKAMD_MAX_DOF];
    uint8_t bounded[PIKAMD_MAX_DOF];
    struct {
        int32_t n_joints;
        int32_t variable[PIKAMD_MAX_DOF];
        double origin_xyz_rpy[PIKAMD_MAX_DOF][6];
        double axis[PIKAMD_MAX_DOF][3];
        int32_t joint_type[PIKAMD_MAX_DOF];
        double tip_xyz_rpy[6];
    } tips[PIKAMD_MAX_TIPS];
    int32_t n_mimic; /* mimic joints that follow a variable of their path (pikamd_set_mimic_joints) */
    pikamd_mimic_joint mimic[PIKAMD_MAX_TIPS * PIKAMD_MAX_MIMIC];
} pikamd_urdf_model;
int32_t pikamd_urdf_extract(const char* urdf_xml, const char* base_link, const char* const* tip_links,
                            int32_t n_tips, pikamd_urdf_model* out);
int32_t pikamd_create_from_urdf(const char* urdf_xml, const char* base_link, const char* const* tip_links,
                                int32_t n_tips, int32_t device_ordinal, pikamd_solver** out);

/* ---- a host cost function that takes part in the search ----------------------------------------------
 * Reference: kinematics::KinematicsBase::IKCostFn -- src/pick_ik_plugin.cpp:130-135 pushes one Goal of weight 1
 * per tip pose (make_ik_cost_fn, src/goal.cpp:146-161); the goal is summed into cost_fn (src/goal.cpp:188-203)
 * and must stay below cost_threshold^2 in solution_fn (src/goal.cpp:175-182).  It is an opaque host closure
 * evaluated at EVERY cost evaluation of the search, which no GPU kernel can call and no host round trip per
 * evaluation can afford (96 us each, 13 000 per solve).  pikamd_solve_batch_host therefore runs such queries ON
 * THE HOST: the reference's algorithm, one problem after the other on the calling thread, with the arithmetic of
 * the library's exact kernels compiled for the host and the library's random streams -- bit-identical to the CPU
 * oracle given the same callback (its math mode "fma"; "portable" for the verification library), and bit-identical
 * to the exact kernels' answer when the callback returns 0.  About 7 ms per default-parameter solve per core.
 *   cost_fn(q, dof, pose_index, user) -> the cost of joint vector q for tip pose `pose_index` (0 .. n_tips-1); it
 *   must be a pure function of its arguments.  cost_fn == NULL is refused: queries without a host cost function
 *   belong on the GPU (pikamd_solve_batch).  All arrays are host memory, [B][dof] / [B][n_tips][7] (a handle
 *   with joint_layout = soa is refused); initial_guess may be NULL (= seed).  A callback value that is not below
 *   cost_threshold^2 rejects a candidate exactly as the reference's `cost >= threshold` does (a NaN passes, as
 *   there).  Wall-clock limits: options host_max_time / host_gd_max_time below -- on THIS path the reference's
 *   clocks exist (a problem cut short returns what the reference returns at a timeout: its best individual if
 *   that is acceptable, else failure). */
typedef double (*pikamd_cost_fn)(const double* q, int32_t dof, int32_t pose_index, void* user);
int32_t pikamd_solve_batch_host(pikamd_solver* s, const pikamd_params* p, int64_t B, const double* goal_pos_quat,
                                const double* seed, const double* initial_guess, uint64_t rng_seed,
                                int64_t problem_offset, pikamd_cost_fn cost_fn, void* user, double* solution,
                                int32_t* status, double* final_cost, pikamd_stats* stats);

/* ---- scheduling options of a handle -------------------------------------------------------
 * How a call is cut into launches is chosen by the library (lanes per elite and
 * compaction passes from the number of problems still running, two wavefronts per SIMD from a size
 * threshold, latency- or throughput-greedy from the number of other calls in flight).  None of it
 * changes a result; a caller who knows better can pin each choice per handle.  Options are read when
 * a call is made.  THE LIBRARY never reads the environment; the Python test binding
 * (pick_ik_amd/solver.py ENV_OPTIONS) is what turns the PIK_* variables of the test suite and the tools
 * into pikamd_set_option calls -- a C or C++ caller sets options explicitly.  value NULL or "" restores
 * the default.
 *   "host_max_time", "host_gd_max_time"  wall-clock limits of pikamd_solve_batch_host in seconds ("0": none) --
 *                               the reference's max_time of a query / of one elite's descent, read in front
 *                               of every generation / descent step (src/ik_memetic.cpp:75-78, 226-228,
 *                               src/ik_gradient.cpp:112-115); the device paths have iteration budgets only
 *   "lanes_per_elite"           "0" adaptive | "1" "2" "4" "8" "16" (several tip frames: no "4"; a value a
 *                               call cannot have falls back to adaptive)
 *   "lanes_per_elite_schedule"  "g0:l0,g1:l1,..."  passes starting at generation >= g_i use l_i lanes (g0 = 0)
 *   "passes"                    "2,4,8,..." generation marks of the compaction passes | "none"
 *   "two_per_simd"              "0" never | "1" default threshold | "<n>" from n first-pass wavefronts on
 *   "regime"                    "adaptive" | "latency" | "throughput"
 *   "specialised"               "1" the kernels compiled for the common configuration (bounded revolute
 *                               variables, no joint goals, four elites, one species ...) serve the calls that
 *                               have it -- same bits, 15-19 % faster | "0" the general kernels always
 *   "shard_chunks"              "1".."8" host jobs per device of pikamd_solve_batch_sharded (default 2)
 *   "joint_layout"              "aos" (default): the joint-vector arrays of the solve entry points (seed,
 *                               initial_guess, solution) are [B][dof], one problem's vector contiguous -- what
 *                               a MoveIt caller holds | "soa": they are [dof][B] per batch (structure of
 *                               arrays); poses stay [B][n_tips][7].  The library transposes on the device in
 *                               front of and behind the kernels (180 bytes per problem); not with completion
 *                               counters, not with pikamd_solve_batch_sharded.  Unlike the options above this
 *                               one changes what the caller's arrays MEAN, not a result.
 *   "arithmetic"                "exact" (default): the exact kernels -- the reference's literal algorithm (MoveIt's
 *                               chain product, 2 dof + 3 cost evaluations per gradient step, IEEE square roots and
 *                               divisions) with fused multiply-adds at stated places; BIT-IDENTICAL to the CPU
 *                               oracle's math mode "fma" on every entry point: the joint vector a caller gets is
 *                               the one src/ik_memetic.cpp:356-370 returns | "fast" (opt-in): the
 *                               Denavit-Hartenberg kernels (frame-based gradient probes, in-house square roots;
 *                               whole solves agree with the reference statistically), about
 *                               twice the throughput.  (Chains with a floating or a mimic joint always run the
 *                               exact kernels.)
 *   "self_test"                 "auto" (default): the first pikamd_reserve or host-pointer solve of a KERNEL SET
 *                               (flavour, mode, species, elites, enabled cost terms -- not thresholds, weights or
 *                               budgets) that the general or the exact kernels serve runs a short pikamd_self_test
 *                               first (a dozen solves of 32 targets, four generations: ~10 ms, once per kernel set
 *                               and handle; pikamd_self_test_cost reports it).  The stream-ordered entry points
 *                               (pikamd_solve_batch[es]_device) never run it: they do not block -- reserve first.
 *                               | "off": only when pikamd_self_test is called
 * pikamd_set_option and pikamd_self_test change the handle's options: like every call on a handle they must not
 * run concurrently with another call on the same handle (a handle is used from one thread at a time).
 * The reference has no counterpart (its only scheduling parameter is memetic_num_threads,
 * src/ik_memetic.cpp:299-335, which pikamd_params carries). */
int32_t pikamd_set_option(pikamd_solver* s, const char* name, const char* value);

/* ---- self test ------------------------------------------------------------------------------
 * Every kernel variant re-schedules the same arithmetic, so on ANY input all of them must return the
 * same bits as the one-lane kernel -- a property that can be checked on the device, without the oracle.
 * pikamd_self_test solves n (1..4096) generated reachable targets of THIS chain with THESE parameters by
 * every variant the handle may choose (2 / 4 / 8 / 16 lanes per elite, with and without compaction passes,
 * the two-wavefronts-per-SIMD build) and compares solutions, status words, costs and counters bit for bit
 * with the one-lane kernel's.  A variant that disagrees is switched off for the handle (the adaptive
 * schedule then never picks it) and reported: *disabled_mask bit v = v lanes per elite (2, 4, 8, 16), bit 1
 * = the two-per-SIMD build, bit 32 = the common-configuration kernels against the general ones.  Why it exists: the multi-lane kernels of long chains compile at the register
 * cap, and twice during development such a kernel came out wrong after a change elsewhere in the source
 * (DESIGN.md section 3); the shipped kernels pass this test on every chain the tests generate -- a
 * deployment on its own robot can make sure in ~20 ms at start-up (the plugin shim does, once per group).
 * The reference has no counterpart. */
int32_t pikamd_self_test(pikamd_solver* s, const pikamd_params* p, int32_t n, uint32_t* disabled_mask);
/* what the AUTOMATIC self tests of this handle (option self_test = auto) have cost so far: how many ran, and their
 * wall-clock time in milliseconds (either pointer may be NULL) */
int32_t pikamd_self_test_cost(const pikamd_solver* s, int32_t* runs, double* total_ms);

/* library / kernel introspection for benches and tests */
const char* pikamd_last_error(void);
const char* pikamd_version(void);
/* name of the kernel pikamd_solve_batch* launches for this handle and these parameters, with the namespace
 * of the flavour that serves the call: `pik_common::memetic_kernel<7>` (kernels compiled for the common
 * configuration), `pik::...` (general), `pik_strict::...` (literal: floating-joint chains, and everything
 * in libpick_ik_amd_strict.so) -- for matching rocprof rows, and for tests that must know which ran */
const char* pikamd_kernel_name(const pikamd_solver* s, const pikamd_params* p);

#ifdef __cplusplus
}
#endif
#endif /* PICK_IK_AMD_H */
