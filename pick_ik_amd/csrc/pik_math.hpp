// pik_math.hpp -- FP64 kinematics / cost arithmetic of the pick_ik hot path for gfx950.
//
// Everything here is per-lane straight-line code over compile-time-sized arrays (D = DOF is a
// template parameter) so that joint vectors, frames and gradients live in VGPRs; the chain
// description (ChainK) and the solver parameters (ParamsK) are wave-uniform kernel arguments and
// are read through scalar loads.
//
// Reference semantics implemented (file:line in PickNikRobotics/pick_ik v1.1.2):
//   FK            src/fk_moveit.cpp:20-33 (MoveIt RobotState chain product), joint frames as in
//                 src/forward_kinematics.cpp:39-80
//   pose cost     src/goal.cpp:17-25, 51-78        frame tests  src/goal.cpp:27-36
//   joint costs   src/goal.cpp:91-144              cost/solution composition src/goal.cpp:163-203
//   clamp         src/robot.cpp:36-42
//
// The functions are also compilable for the host (PIK_HD) so tests/native can check the same
// source against the CPU oracle without a GPU.
#pragma once

#include <math.h>
#include <stdint.h>

// The strict-arithmetic verification build lives in its own namespace, so that its kernels carry
// their own names in profiles (pik_strict::memetic_kernel<...>) next to the product's.
#if defined(PIK_STRICT) && defined(PIK_EXACT_FMA) && PIK_EXACT_FMA
// ... the exact flavour with fused multiply-adds at stated places (PIK_XF below)
#define pik pik_exact
#elif defined(PIK_STRICT)
#define pik pik_strict
#elif defined(PIK_COMMON) && PIK_COMMON
// ... and so do the kernels specialised for the common configuration (see PIK_COMMON below), without and with
// the joint goals
#if defined(PIK_NO_GOALS) && !PIK_NO_GOALS
#define pik pik_common_goals
#else
#define pik pik_common
#endif
#endif

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PIK_HD __host__ __device__ __forceinline__
#else
#define PIK_HD inline
#endif

// Wave-uniform constants (chain, params) live in a device buffer that the kernels address through
// the CONSTANT address space: uniform address + constant memory => the compiler emits scalar
// loads (s_load_dwordx*) and feeds the values to the FP64 VALU ops as SGPR operands.
#if defined(__HIP_DEVICE_COMPILE__)
#define PIK_CONSTANT __attribute__((address_space(4)))
#else
#define PIK_CONSTANT
#endif

namespace pik {

enum AxisKind : uint32_t { AXIS_GENERAL = 0, AXIS_X = 1, AXIS_Y = 2, AXIS_Z = 3 };

// Coefficients of sincos_f64 / atan2_pos.  They are read from constant memory (scalar loads, SGPR
// operands) instead of being 64-bit literals: an FP64 VALU op cannot take a 64-bit literal, so
// every literal costs two v_mov_b32 issue slots per use.
//   0 1/(2pi)  1 2pi_hi  2 2pi_lo  3 2/pi  4..6 pi/2 (three-double split)
//   7..12 S1..S6   13..18 C1..C6   19..29 aT0..aT10   30..33 atan hi   34..37 atan lo
struct MathTab {
    double v[40];
};

// Serial chain, base -> tip (wave-uniform).
template <int D>
struct ChainK {
    double O[D][12];   // joint origin transform: rotation row-major [0..8], translation [9..11]
    double axis[D][3]; // normalised joint axis in the joint frame
    double tip[12];    // fixed transform after the last joint
    double qmin[D], qmax[D], mid[D], hspan[D], mdf[D];
    // clamp limits of the product build: qmin / qmax of a bounded variable, -inf / +inf otherwise (the
    // reference clamps an unbounded variable v to [v - span, v + span], i.e. leaves it as it is)
    double clo[D], chi[D];
    // Denavit-Hartenberg form used by the fast build (built on the host, pik_host.hpp build_dh):
    // frame A_j sits on joint j's axis (z = axis); the step to the next joint's frame is
    //   Rz(q_j + theta0) Tz(d) Tx(a) Rx(alpha)        dh[j] = {theta0, d, a, cos alpha, sin alpha, 0}
    // (a prismatic joint moves along z instead: Rz(theta0) Tz(q_j + d) ...), dh_base places the
    // first frame, dh_tip closes the chain to the tip link.  FK is branch-free, two column
    // rotations per joint, and the world joint axis is the third column of the running rotation.
    double dh[D][6];
    double dh_base[12];
    double dh_tip[12];
    // Joints whose axis is nearly but not exactly parallel to the next one (dh_general_mask): the
    // common normal's foot lies ~L / sin(angle) away and the DH offsets would cancel
    // catastrophically (1e-16 L / angle of FK error), so the step after such a joint's rotation is
    // the general rigid transform dhg[j] (12 constants) to a well-placed frame on the next axis.
    double dhg[D][12];
    MathTab mt;
    uint32_t origin_ident_mask; // bit j: origin transform is exactly the identity
    uint32_t prismatic_mask;    // bit j
    uint32_t bounded_mask;      // bit j
    uint32_t axis_kind;         // 2 bits per joint: AxisKind (only exact +x/+y/+z are specialised)
    uint32_t tip_ident;
    uint32_t active_mask; // bit j: variable j is a joint on the way to THIS tip (multi-tip chains)
    uint32_t dh_general_mask; // bit j: the step after joint j is dhg[j] instead of the DH constants
    // A floating joint (seven consecutive variables, include/pick_ik_amd.h) acts ONCE, at its seventh
    // variable: float_mask bit j = variable j is a floating joint's rot_w (O[j] = that joint's origin),
    // skip_mask bit j = variable j is one of its first six (no transform of its own).  Only the literal
    // forward kinematics (verification build) knows them; the Denavit-Hartenberg form has no such joint.
    uint32_t float_mask;
    uint32_t skip_mask;
    // Mimic joints of the path (include/pick_ik_amd.h pikamd_set_mimic_joints): up to MAX_MIMIC more steps of the
    // chain product, step m behind the joint of variable m_after[m] (-1: in front of the first), its value
    // mmult[m] * q[m_var[m]] + moff[m].  Only the literal forward kinematics (exact flavours) knows them.
    uint32_t m_count;
    double mO[4][12];
    double maxis[4][3];
    double mmult[4], moff[4];
    int32_t m_after[4], m_var[4];
    uint32_t m_ident_mask, m_pris_mask, m_kind; // bit m: identity origin / prismatic; 2 bits per joint: AxisKind
    // 1: every variable is a revolute joint about its frame's +z behind an origin that is not the identity, the tip
    // transform is not the identity, no floating / mimic joint, one tip frame; 2: the same with every axis exactly
    // +x, +y or +z -- the chain classes the exact flavour has specialised forms for (UZ / UA below and in
    // pik_exact.hpp; set by make_chain_k); 0: neither
    uint32_t uniform_z;
#if defined(PIK_STRICT) // (the exact flavours only: the fast flavours' source, and with it the hash beside their profiles, stays what it was)
    // Which entries of a fixed transform are EXACT ones / zeros (make_chain_k; read by the forms of classes 1 and 2
    // only, x_iso_mul below): 3 bits per joint 0..9 -- origin_kinds: IsoKind of the origin's rotation; origin_pmasks:
    // bit i set = component i of its translation is not zero; the same two for the tip transform
    uint32_t origin_kinds, origin_pmasks, tip_kind, tip_pmask;
#endif
};

// Solver parameters (wave-uniform), derived from pikamd_params on the host.
struct ParamsK {
    // block read by the gradient probes (one scalar load)
    double step_size;
    // rotation by the finite-difference step h about a joint axis (gradient probes):
    // sin h, 1 - cos h, sin h/2, cos h/2
    double sin_h, vers_h, sin_h2, cos_h2;
    double pos_scale, rot_scale;
    double min_cost_delta;
    // block read by the cost / solution test of an evaluation
    double pos_thr, ori_thr;
    double cost_thr_sq;
    double w_center_sq, w_limits_sq, w_disp_sq; // weight^2 (0 = goal disabled)
    double wipeout_tol;
    int32_t has_pos_thr, has_ori_thr;
    int32_t goal_mask; // bit0 center, bit1 avoid limits, bit2 minimal displacement
    int32_t stop_on_valid;
    int32_t stop_on_first; // memetic_stop_on_first_solution (species)
    int32_t approx;
    int32_t population, elites;
    int32_t max_generations, gd_max_iters;
    int32_t local_max_iters;
    // 1: the gradient step is small enough (<= 1e-3) for the line-search evaluations to take their
    // sines / cosines from the accepted point (sincos_delta).  An integer decided on the host: as a
    // floating-point comparison in the kernel it is a vector compare whose result lives in a lane mask
    int32_t line_delta;
};

// PIK_COMMON: the kernels compiled for the COMMON CONFIGURATION -- what pick_ik's yaml defaults and an
// industrial arm give: every variable a bounded revolute joint, no ill-conditioned pair of axes, no joint
// goal enabled (center / avoid-limits / minimal-displacement weights 0), both pose-cost terms on, four
// elites, one species, a gradient step small enough for the line-search angle addition.  Everything the
// general kernels decide at run time about these is a compile-time constant here, and the code of the
// untaken paths is gone.  Its mere PRESENCE in the loops costs the common configuration 15-19 %
// (measured: driver's command 2.92 -> 3.41 M solves/s, sustained 5.5 -> 6.5 M, one 4096-batch 12.85 ->
// 10.55 ms): instruction-cache footprint (the one-lane descent loop alone is 40 KB of a 64 KB cache shared
// by two CUs), scalar registers held by flag words and constants of paths never taken, selects on masks
// that are all zero.  The arithmetic of the taken path is the same expression for expression, so the two
// flavours return the same bits (tests/test_gpu_parity.py test_specialised_kernels_identical); the host
// picks per call (pik_amd.hip common_eligible), option "specialised" = "0" forces the general kernels.
// PIK_XF: the EXACT flavour with fused multiply-adds (-DPIK_STRICT -DPIK_EXACT_FMA, namespace pik_exact) -- the
// literal algorithm of PIK_STRICT (MoveIt's chain product, 2D + 3 evaluations per step, IEEE square roots and
// divisions) whose product-sums are fused AT STATED PLACES: the row products of the chain (xdot3 / xmad), the
// sums of squares of the distances and the quaternion product (xsumsq3, xmad), the cost accumulations, the
// gradient-step update, and the Horner forms of the sine / cosine / arctangent polynomials (the product build's
// sincos_f64 / atan2_pos).  The oracle's math mode 2 ("fma", oracle/pik_oracle.c) performs the same operations
// with C's fma(); the two agree bit for bit.  Why: pick_ik's own operation order is Eigen's and its contraction
// is the compiler's (gcc's default is -ffp-contract=fast: an aarch64 or -march=native build fuses, a baseline
// x86-64 build does not) -- both are "the reference"; the fused one costs a third fewer instructions.
#if defined(PIK_STRICT) && defined(PIK_EXACT_FMA) && PIK_EXACT_FMA
#define PIK_XF 1
#else
#define PIK_XF 0
#endif
#ifndef PIK_XF_TABLE2
#define PIK_XF_TABLE2 0 // (experiment: also the arc tangent's coefficients and the reduction constants from the table)
#endif
#ifndef PIK_XF_TABLE
#define PIK_XF_TABLE PIK_XF // (0: the scheduled-assembly Horner steps with literal coefficients, as the fast flavours)
#endif
#ifndef PIK_COMMON
#define PIK_COMMON 0
#endif
#ifndef PIK_NO_GOALS
#define PIK_NO_GOALS PIK_COMMON
#endif
// (PIK_COMMON: no joint goals, no general Denavit-Hartenberg step, line-search evaluations by angle addition)
#define PIK_LINE_DELTA(p) (PIK_COMMON ? true : ((p).line_delta != 0))
// ... all variables revolute and bounded, both pose-cost terms on (and with them both frame tests), one species
#define PIK_PRISMATIC(c) (PIK_COMMON ? 0u : (c).prismatic_mask)
#define PIK_POS_ON(p) (PIK_COMMON ? true : ((p).pos_scale > 0.0))
#define PIK_ROT_ON(p) (PIK_COMMON ? true : ((p).rot_scale > 0.0))
#define PIK_POS_TEST(p) (PIK_COMMON ? true : ((p).has_pos_thr != 0))
#define PIK_ORI_TEST(p) (PIK_COMMON ? true : ((p).has_ori_thr != 0))
#define PIK_GM(p) (PIK_NO_GOALS ? 0 : (p).goal_mask)

template <int D>
using CK = const PIK_CONSTANT ChainK<D>&;
using PK = const PIK_CONSTANT ParamsK&;
using CPtr = const PIK_CONSTANT double*;

// Returns the same reference through an opaque scalar-register copy of its address.  The chain
// holds ~150 doubles, far more than the 102 SGPRs of a wave: without this the compiler hoists
// every scalar load out of the solver loops and spills them into VGPR lanes (v_writelane /
// v_readlane per use).  Re-deriving the address per evaluation keeps the loads next to their
// uses, where they are cheap scalar-cache hits overlapped with the sincos arithmetic.
template <typename T>
PIK_HD const PIK_CONSTANT T& fresh(const PIK_CONSTANT T& r) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PIK_STRICT) // (strict build: evaluations are calls, no tuning)
    const PIK_CONSTANT T* p = &r;
    asm volatile("" : "+s"(p));
    return *p;
#else
    return r;
#endif
}

// fresh() pinned behind a vector value: the loads through the returned reference cannot be issued
// before `dep` has been computed.  Used to software-pipeline the chain constants by hand -- the
// loads of joint j+1 are issued in the middle of joint j and land while its sincos runs -- instead
// of letting the scheduler hoist every joint's loads to the top of the block and spill them.
template <typename T>
PIK_HD const PIK_CONSTANT T& fresh_after(const PIK_CONSTANT T& r, double dep) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PIK_STRICT)
    const PIK_CONSTANT T* p = &r;
    asm volatile("" : "+s"(p) : "v"(dep));
    return *p;
#else
    (void)dep;
    return r;
#endif
}

// Wave-uniform constants that reached a function through the arguments of a REAL call (the exact flavour's
// evaluations and descents are calls) arrive in vector registers, and every load through them would be a
// vector load.  Back into scalar registers: the address is the same in every lane.
template <typename T>
PIK_HD const PIK_CONSTANT T& scalar_ref(const PIK_CONSTANT T& r) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long a = (unsigned long long)&r;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return *(const PIK_CONSTANT T*)(((unsigned long long)hi << 32) | lo);
#else
    return r;
#endif
}
PIK_HD int scalar_int(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}

// Per-problem goal: translation + the goal frame's quaternion as the reference derives it
// (tf2::fromMsg pose -> matrix, then Eigen matrix -> quaternion inside angular_distance).
struct GoalK {
    double t[3];
    double q[4]; // w x y z
};

// Several tip frames (the plugin's tip_frames; reference src/pick_ik_plugin.cpp:57-69,
// src/goal.cpp:27-49, 80-89): one goal per tip, one chain description per tip.  Tip k's chain is
// its joint path PADDED to all D variables: a variable that is not on the path is a joint with an
// identity origin whose value is ignored (active_mask), so the same D-joint code computes every
// tip and the product is exactly MoveIt's product along the path (x * 1, x + 0 are exact).
constexpr int MAX_TIPS = 8;
// the goals of one problem of a multi-tip chain: n_tips x (x y z qw qx qy qz) in HBM.  They are
// re-derived per evaluation (7 loads + ~80 flops per tip against ~1000 for the tip's FK) rather
// than held in registers, and the tips are a real loop, so the multi-tip kernels are no larger
// than the single-tip ones.
struct GoalSet {
    const double* ptr;
};

// chain(s) + parameters of one call, uploaded into a device buffer and read by the kernels through
// the constant address space (scalar loads).  Single-tip kernels only ever touch chain / params.
template <int D>
struct ConstsK {
    ChainK<D> chain; // tip 0 (its joint limits etc. are the variables' for every tip)
    ParamsK params;
    int32_t n_tips;
    int32_t pad_;
    ChainK<D> more[MAX_TIPS - 1]; // tips 1..
};

template <int D>
PIK_HD int tip_count(CK<D> c0) {
    return reinterpret_cast<const PIK_CONSTANT ConstsK<D>*>(&c0)->n_tips;
}
template <int D>
PIK_HD CK<D> tip_chain(CK<D> c0, int k) {
    const PIK_CONSTANT ConstsK<D>* kc = reinterpret_cast<const PIK_CONSTANT ConstsK<D>*>(&c0);
    return k == 0 ? kc->chain : kc->more[k - 1];
}

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------

// Eigen 3.4 Quaternion::toRotationMatrix
PIK_HD void quat_to_matrix(const double (&q)[4], double (&R)[9]) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz);
    R[1] = txy - twz;
    R[2] = txz + twy;
    R[3] = txy + twz;
    R[4] = 1.0 - (txx + tzz);
    R[5] = tyz - twx;
    R[6] = txz - twy;
    R[7] = tyz + twx;
    R[8] = 1.0 - (txx + tyy);
}

PIK_HD double fma_f64(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fma(a, b, c);
#else
    return ::fma(a, b, c);
#endif
}

// ---- product-sums of the exact flavours: two roundings per term in the plain one (PIK_STRICT, compiled
// -ffp-contract=off), fused in PIK_XF.  (The fast build has its own forms: dh_row, iso_row.)
PIK_HD double xmad(double a, double b, double c) { // a * b + c
#if PIK_XF
    return fma_f64(a, b, c);
#else
    return a * b + c;
#endif
}
PIK_HD double xdot3(double a0, double b0, double a1, double b1, double a2, double b2) { // a0 b0 + a1 b1 + a2 b2
#if PIK_XF
    return fma_f64(a2, b2, fma_f64(a1, b1, a0 * b0));
#else
    return a0 * b0 + a1 * b1 + a2 * b2;
#endif
}
PIK_HD double xsumsq3(double a, double b, double c) { // a^2 + b^2 + c^2
#if PIK_XF
    return fma_f64(c, c, fma_f64(b, b, a * a));
#else
    return a * a + b * b + c * c;
#endif
}

// sqrt(x) and 0.5 / sqrt(x) together.  Product build on the device: the root from v_rsq_f64 (about 27 bits), one
// coupled Goldschmidt step and one residual correction -- eight instructions against 18 for the library's sqrt
// (two corrections, exponent scaling for subnormal inputs, a class test) -- and the half-inverse as the IEEE
// quotient 0.5 / root.  The root comes out CORRECTLY ROUNDED for the well-scaled arguments of this path (sums of
// squares of lengths and of unit-quaternion components): 2 x 10^7 inputs over [1e-12, 1e2] and [1, 4], none
// different from sqrt() (tests/native/sqrt_pair_check.hip, tests/test_gpu_product_arithmetic.py); x must be > 0
// and normal, see sqrt_pos.  Until round 4 the half-inverse was the Goldschmidt iterate h itself: three
// instructions cheaper than the divide, but up to 20 ulp off and -- depending on the bits of the hardware's
// reciprocal-square-root table -- the one number of the product's arithmetic a host could not reproduce.  With
// the quotient every operation of the product build is an IEEE operation or a correctly rounded root, so a
// sequential host loop over the same source returns the kernels' bits (pik_host_solve.hpp, fast flavour).
PIK_HD void sqrt_pair(double x, double& root, double& half_inv) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PIK_STRICT)
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = fma_f64(-h, g, 0.5);
    g = fma_f64(g, r, g);
    h = fma_f64(h, r, h);
    const double d = fma_f64(-g, g, x);
    root = fma_f64(d, h, g);
    // 0.5 / root, correctly rounded, from the iterate h (= 0.5 / sqrt(x) to ~2^-48) by two residual corrections with
    // 1 / root ~ 2 h: the first brings the quotient within an ulp, the second's residual is then exact and its
    // result the IEEE quotient -- the tail of the hardware's own division sequence, without its reciprocal
    // (a quarter-rate instruction), scaling and fix-up (root is a normal number in [1e-6, 10]): five
    // instructions instead of twelve.  Checked against 0.5 / sqrt(x) on 2 x 10^7 inputs like the root.
    const double yi = h + h;
    const double q1 = fma_f64(fma_f64(-root, h, 0.5), yi, h);
    half_inv = fma_f64(fma_f64(-root, q1, 0.5), yi, q1);
#else
    root = sqrt(x);
    half_inv = 0.5 / root;
#endif
}
// sqrt of a sum of squares (>= 0; anything below 1e-280 counts as zero)
PIK_HD double sqrt_pos(double x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PIK_STRICT)
    double g, h;
    sqrt_pair(x, g, h);
    return (x <= 1.0e-280) ? 0.0 : g; // (NaN stays NaN; +inf gives NaN: not a solution either way)
#else
    return sqrt(x);
#endif
}

// Eigen 3.4 rotation matrix -> quaternion (w x y z): the trace > 0 branch, else the branch of the
// largest diagonal element i with (i, j, k) cyclic:
//   t = sqrt(m_ii - m_jj - m_kk + 1); q_i = t/2; r = 0.5/t; w = (m_kj - m_jk) r;
//   q_j = (m_ji + m_ij) r; q_k = (m_ki + m_ik) r.
// All four cases are written out over statically indexed differences/sums and combined with
// selects between *computed scalars* (never between array elements, which the compiler would
// turn into a dynamically indexed -- i.e. scratch/LDS resident -- copy of R); one sqrt and one
// divide are shared by all cases and a wave never diverges here.
PIK_HD void matrix_to_quat(const double (&R)[9], double (&q)[4]) {
    const double m00 = R[0], m11 = R[4], m22 = R[8];
    const double tr = m00 + m11 + m22;
    const bool cW = tr > 0.0;
    const bool big1 = m11 > m00;
    const bool big2 = m22 > (big1 ? m11 : m00);
    const bool cZ = !cW && big2;
    const bool cY = !cW && !big2 && big1;
    const bool cX = !cW && !big2 && !big1;
    const double d0 = R[7] - R[5], d1 = R[2] - R[6], d2 = R[3] - R[1];
    const double s01 = R[3] + R[1], s02 = R[6] + R[2], s12 = R[7] + R[5];
    const double aW = tr + 1.0;
    const double aX = m00 - m11 - m22 + 1.0;
    const double aY = m11 - m22 - m00 + 1.0;
    const double aZ = m22 - m00 - m11 + 1.0;
    const double arg = cW ? aW : cX ? aX : cY ? aY : aZ;
#if defined(PIK_STRICT)
    const double t = sqrt(arg);
    const double r = 0.5 / t;
#else
    double t, r; // (arg >= 1 for a rotation matrix: the branch of the largest diagonal term)
    sqrt_pair(arg, t, r);
#endif
    const double h = 0.5 * t;
    const double nw = cX ? d0 : cY ? d1 : d2;
    const double nx = cW ? d0 : cY ? s01 : s02;
    const double ny = cW ? d1 : cX ? s01 : s12;
    const double nz = cW ? d2 : cX ? s02 : s12;
    q[0] = cW ? h : nw * r;
    q[1] = cX ? h : nx * r;
    q[2] = cY ? h : ny * r;
    q[3] = cZ ? h : nz * r;
}

// (R, t) <- (R, t) * (Ro, to)        [Eigen Isometry3d product]
PIK_HD void iso_mul(double (&R)[9], double (&t)[3], CPtr o) {
    double r[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            r[i * 3 + j] = xdot3(R[i * 3 + 0], o[0 * 3 + j], R[i * 3 + 1], o[1 * 3 + j], R[i * 3 + 2], o[2 * 3 + j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#if PIK_XF
        t[i] = fma_f64(R[i * 3 + 2], o[11], fma_f64(R[i * 3 + 1], o[10], fma_f64(R[i * 3 + 0], o[9], t[i])));
#else
        t[i] = R[i * 3 + 0] * o[9] + R[i * 3 + 1] * o[10] + R[i * 3 + 2] * o[11] + t[i];
#endif
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = r[i];
}

#if defined(PIK_STRICT)
// ---- fixed transforms with EXACT ones and zeros ---------------------------------------------------------------------
// The rotation of most fixed transforms of a robot description is one about a single axis of its own frame, or none:
// rpy = (alpha, 0, 0) -- the link twist of the Denavit-Hartenberg convention --, (0, beta, 0), (0, 0, gamma), (0, 0, 0),
// for which urdfdom's quaternion gives the entries 1 and 0 of [1 0 0; 0 a b; 0 c d] (etc.) EXACTLY, and most
// translations have components that are exactly zero.  iso_mul multiplies by them all the same.  The forms below leave
// those products out: r0 * 1 is r0, and a product by an exact zero is a zero that the next fused step adds to a rounded
// product -- element for element the bits iso_mul produces (the SIGN of a result that is exactly zero may differ; no
// value that is not a zero depends on the sign of one, and nothing on this path divides by a frame entry or takes its
// sign).  Classified on the host (make_chain_k: ChainK::origin_kinds / origin_pmasks / tip_kind / tip_pmask); the
// decisions are wave-uniform.  Fused flavour only (the order of the plain-IEEE translation sum is another).
#ifndef PIK_XSPARSE_K2
#define PIK_XSPARSE_K2 1 // (0: A/B experiments -- the transforms of class 2 and the tip transforms through iso_mul)
#endif
enum IsoKind : uint32_t { ISO_GENERAL = 0, ISO_RX = 1, ISO_RY = 2, ISO_RZ = 3, ISO_TRANS = 4 };

// row i of R <- row i of R * rotation of o, for a rotation of kind K
template <uint32_t K>
PIK_HD void iso_rot_row(double& r0, double& r1, double& r2, CPtr o) {
    const double a0 = r0, a1 = r1, a2 = r2;
    if constexpr (K == ISO_RX) {
        r1 = xmad(a2, o[7], a1 * o[4]);
        r2 = xmad(a2, o[8], a1 * o[5]);
    } else if constexpr (K == ISO_RY) {
        r0 = xmad(a2, o[6], a0 * o[0]);
        r2 = xmad(a2, o[8], a0 * o[2]);
    } else if constexpr (K == ISO_RZ) {
        r0 = xmad(a1, o[3], a0 * o[0]);
        r1 = xmad(a1, o[4], a0 * o[1]);
    } else if constexpr (K == ISO_GENERAL) {
        r0 = xdot3(a0, o[0], a1, o[3], a2, o[6]);
        r1 = xdot3(a0, o[1], a1, o[4], a2, o[7]);
        r2 = xdot3(a0, o[2], a1, o[5], a2, o[8]);
    }
}
// t <- t + R * translation of o, the components that are exact zeros left out (iso_mul's order: x, then y, then z).
// Behind PIK_XSPARSE_T, which is OFF: three wave-uniform branches per joint of the rolled fork loop, each with the
// scalar load of its component and the wait for it inside, measured 54.6 against 51.9 ms on the driver's pool (the
// Panda's origins would save 6.4 of 9 operations per frame and joint); with it off the mask is the constant 7.
PIK_HD void iso_trans_masked(const double (&R)[9], double (&t)[3], CPtr o, uint32_t pm) {
    if (pm & 1u) {
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = fma_f64(R[i * 3 + 0], o[9], t[i]);
    }
    if (pm & 2u) {
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = fma_f64(R[i * 3 + 1], o[10], t[i]);
    }
    if (pm & 4u) {
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = fma_f64(R[i * 3 + 2], o[11], t[i]);
    }
}
template <uint32_t K>
PIK_HD void iso_rot_k(double (&R)[9], CPtr o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) iso_rot_row<K>(R[i * 3 + 0], R[i * 3 + 1], R[i * 3 + 2], o);
}
// (R, t) <- (R, t) * o.  XM = 1 (chain class 1): every origin is of kind ISO_RX, `kind` is not read; XM = 2: one
// wave-uniform decision; XM = 0 or the plain-IEEE flavour: iso_mul
template <int XM>
PIK_HD void x_iso_mul(double (&R)[9], double (&t)[3], CPtr o, uint32_t kind, uint32_t pm) {
    if constexpr (PIK_XF && XM == 1) {
        (void)kind;
        iso_trans_masked(R, t, o, pm);
        iso_rot_k<ISO_RX>(R, o);
    } else if constexpr (PIK_XF && XM == 2 && PIK_XSPARSE_K2) {
        iso_trans_masked(R, t, o, pm);
        if (kind == ISO_TRANS) {
        } else if (kind == ISO_RX) {
            iso_rot_k<ISO_RX>(R, o);
        } else if (kind == ISO_RY) {
            iso_rot_k<ISO_RY>(R, o);
        } else if (kind == ISO_RZ) {
            iso_rot_k<ISO_RZ>(R, o);
        } else {
            iso_rot_k<ISO_GENERAL>(R, o);
        }
    } else {
        (void)kind;
        (void)pm;
        iso_mul(R, t, o);
    }
}
// two frames by the same transform: ONE decision with both products inside each answer (see x_rotate_pair)
template <int XM>
PIK_HD void x_iso_mul_pair(double (&Ra)[9], double (&ta)[3], double (&Rb)[9], double (&tb)[3], CPtr o, uint32_t kind,
                           uint32_t pm) {
    if constexpr (PIK_XF && XM == 1) {
        (void)kind;
        if (pm & 1u) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                ta[i] = fma_f64(Ra[i * 3 + 0], o[9], ta[i]);
                tb[i] = fma_f64(Rb[i * 3 + 0], o[9], tb[i]);
            }
        }
        if (pm & 2u) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                ta[i] = fma_f64(Ra[i * 3 + 1], o[10], ta[i]);
                tb[i] = fma_f64(Rb[i * 3 + 1], o[10], tb[i]);
            }
        }
        if (pm & 4u) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                ta[i] = fma_f64(Ra[i * 3 + 2], o[11], ta[i]);
                tb[i] = fma_f64(Rb[i * 3 + 2], o[11], tb[i]);
            }
        }
        iso_rot_k<ISO_RX>(Ra, o);
        iso_rot_k<ISO_RX>(Rb, o);
    } else if constexpr (PIK_XF && XM == 2 && PIK_XSPARSE_K2) {
        if (pm & 1u) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                ta[i] = fma_f64(Ra[i * 3 + 0], o[9], ta[i]);
                tb[i] = fma_f64(Rb[i * 3 + 0], o[9], tb[i]);
            }
        }
        if (pm & 2u) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                ta[i] = fma_f64(Ra[i * 3 + 1], o[10], ta[i]);
                tb[i] = fma_f64(Rb[i * 3 + 1], o[10], tb[i]);
            }
        }
        if (pm & 4u) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                ta[i] = fma_f64(Ra[i * 3 + 2], o[11], ta[i]);
                tb[i] = fma_f64(Rb[i * 3 + 2], o[11], tb[i]);
            }
        }
        if (kind == ISO_TRANS) {
        } else if (kind == ISO_RX) {
            iso_rot_k<ISO_RX>(Ra, o);
            iso_rot_k<ISO_RX>(Rb, o);
        } else if (kind == ISO_RY) {
            iso_rot_k<ISO_RY>(Ra, o);
            iso_rot_k<ISO_RY>(Rb, o);
        } else if (kind == ISO_RZ) {
            iso_rot_k<ISO_RZ>(Ra, o);
            iso_rot_k<ISO_RZ>(Rb, o);
        } else {
            iso_rot_k<ISO_GENERAL>(Ra, o);
            iso_rot_k<ISO_GENERAL>(Rb, o);
        }
    } else {
        (void)kind;
        (void)pm;
        iso_mul(Ra, ta, o);
        iso_mul(Rb, tb, o);
    }
}
// the tip transform.  Class 1 wants it to turn about its own z axis only ([a b 0; c d 0; 0 0 1] with the 1 and the 0
// exact: rpy = (0, 0, gamma), a tool flange -- the Panda's hand; make_chain_k) and leaves those products out at compile
// time, as for the origins; class 2 takes it through the decision of x_iso_mul<2> like its origins.  (Class 1 with ANY
// tip through that decision measured 51.8 against 51.1 ms on the driver's pool with the tip's product in full: in a
// chain whose origins need no decision, one per evaluation costs more than the products it saves; config 3 (UR5,
// class 2) measured 166.6 ms with the decision for origins and tip, 171.0 for the origins only, 169.5 with none.)
template <int XM>
PIK_HD void x_tip_mul(double (&R)[9], double (&t)[3], CPtr o, uint32_t kind, uint32_t pm) {
    if constexpr (PIK_XF && XM == 1) {
        (void)kind;
        (void)pm;
        iso_trans_masked(R, t, o, 7u);
        iso_rot_k<ISO_RZ>(R, o);
    } else {
        x_iso_mul<XM == 2 ? 2 : 0>(R, t, o, kind, pm);
    }
}
template <int XM>
PIK_HD void x_tip_mul_pair(double (&Ra)[9], double (&ta)[3], double (&Rb)[9], double (&tb)[3], CPtr o, uint32_t kind,
                           uint32_t pm) {
    if constexpr (PIK_XF && XM == 1) {
        (void)kind;
        (void)pm;
        iso_trans_masked(Ra, ta, o, 7u);
        iso_trans_masked(Rb, tb, o, 7u);
        iso_rot_k<ISO_RZ>(Ra, o);
        iso_rot_k<ISO_RZ>(Rb, o);
    } else {
        x_iso_mul_pair<XM == 2 ? 2 : 0>(Ra, ta, Rb, tb, o, kind, pm);
    }
}
// kind / mask of joint j's origin
#ifndef PIK_XSPARSE_T
#define PIK_XSPARSE_T 0 // (1: the zero components of the translations left out too -- measured SLOWER, see iso_trans_masked)
#endif
#define PIK_OKIND(c, j) (((c).origin_kinds >> (3 * (j))) & 7u)
#define PIK_OPM(c, j) (PIK_XSPARSE_T ? (((c).origin_pmasks >> (3 * (j))) & 7u) : 7u)
#define PIK_TPM(c) (PIK_XSPARSE_T ? (c).tip_pmask : 7u)
#endif // PIK_STRICT

// the same product with the right factor in registers (a floating joint's transform)
PIK_HD void iso_mul_r(double (&R)[9], double (&t)[3], const double (&o)[12]) {
    double r[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            r[i * 3 + j] = xdot3(R[i * 3 + 0], o[0 * 3 + j], R[i * 3 + 1], o[1 * 3 + j], R[i * 3 + 2], o[2 * 3 + j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#if PIK_XF
        t[i] = fma_f64(R[i * 3 + 2], o[11], fma_f64(R[i * 3 + 1], o[10], fma_f64(R[i * 3 + 0], o[9], t[i])));
#else
        t[i] = R[i * 3 + 0] * o[9] + R[i * 3 + 1] * o[10] + R[i * 3 + 2] * o[11] + t[i];
#endif
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = r[i];
}

// sin and cos of a joint angle.  Replaces libm's sin/cos (what MoveIt's
// RevoluteJointModel::computeTransform calls): Cody-Waite reduction by pi/2 with a three-double
// split and FMAs (exact to < 1 ulp of the reduced argument for |x| < ~1e5 rad, which covers every
// joint range; larger magnitudes are first folded by 2 pi), then the fdlibm minimax kernels on
// [-pi/4, pi/4].  No table, no stack array, no divergence: ~35 FP64 instructions versus the ~80
// plus scratch of the generic large-argument routine.
using MT = const PIK_CONSTANT MathTab&;

// The same 38 coefficients as compile-time literals.  Used as VALU multiplicands they are
// materialised by scalar moves (s_mov_b32 pairs, issued in the shadow of the 4-cycle vector ops and
// rematerialisable, i.e. never spilled), whereas the table in memory costs scalar loads whose
// results the register allocator parks in VGPR lanes (one v_readlane per dword per use) because the
// 102 scalar registers of a wave are already taken by the chain constants of the joint in flight.
#ifndef PIK_MT_LITERAL
#define PIK_MT_LITERAL (!PIK_XF_TABLE2) // (exact-fma flavour: from the table, see PIK_XF_TABLE)
#endif
PIK_HD constexpr double mt_lit(int k) {
    constexpr double v[38] = {
        0.15915494309189535, 6.283185307179586, 2.4492935982947064e-16, 0.6366197723675814,
        1.5707963267948966, 6.123233995736766e-17, -1.4973849048591698e-33,
        -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
        2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10,
        4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
        -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11,
        3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01,
        -1.11111104054623557880e-01, 9.09088713343650656196e-02, -7.69187620504482999495e-02,
        6.66107313738753120669e-02, -5.83357013379057348645e-02, 4.97687799461593236017e-02,
        -3.65315727442169155270e-02, 1.62858201153657823623e-02,
        4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01,
        1.57079632679489655800e+00, 2.26987774529616870924e-17, 3.06161699786838301793e-17,
        1.39033110312309984516e-17, 6.12323399573676603587e-17};
    return v[k];
}
#if PIK_MT_LITERAL
#define PIK_MV(m, k) (mt_lit(k))
#else
#define PIK_MV(m, k) ((m).v[k])
#endif

// Two Horner chains in x, interleaved, written as instructions (product build).
//
// Why assembly.  pa = (..(A0 x + A1) x + ..) + An has its coefficients as ADDENDS.  Left to itself the
// compiler selects v_fmac (dst = a * b + dst) and first copies every coefficient into the
// destination VGPR -- two v_mov_b32 per step on the vector unit, the unit these kernels are bound by
// (which is why the verification build's power sums, whose coefficients are multiplicands, used to be
// the product form too: six more multiplies per sine/cosine, ten per arctangent).  Here a step is one
// v_fma_f64 with the coefficient as its scalar operand.
// The scalar moves that build the coefficients are part of the statement on purpose: a coefficient
// handed in through an "s" operand may come out of an SGPR spill (v_readlane_b32), and gfx90a+ needs
// two wait states between a VALU instruction that writes an SGPR and a VALU instruction that reads it
// -- the hazard recogniser does not look inside inline assembly (seen as wrong results of the
// 8/16-lane kernels on chains with prismatic joints when the moves were left to the compiler).  They
// are issued one step ahead into two fixed register pairs, so a v_fma_f64 never waits for them, and
// the two dependent chains hide each other's latency.
// pa = S6 x^5 + .. + S1 (sine), pb = C6 x^5 + .. + C1 (cosine): mt_lit 12..7 and 18..13
PIK_HD void horner_sincos(double x, double& pa, double& pb) {
#if defined(__HIP_DEVICE_COMPILE__) && (!defined(PIK_STRICT) || PIK_XF) // (the fused exact flavour: the same operations)
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(12)) == 0x3de5d93a5acfd57cull, "coefficient 12");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(11)) == 0xbe5ae5e68a2b9cebull, "coefficient 11");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(10)) == 0x3ec71de357b1fe7dull, "coefficient 10");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(9)) == 0xbf2a01a019c161d5ull, "coefficient 9");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(8)) == 0x3f8111111110f8a6ull, "coefficient 8");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(7)) == 0xbfc5555555555549ull, "coefficient 7");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(18)) == 0xbda8fae9be8838d4ull, "coefficient 18");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(17)) == 0x3e21ee9ebdb4b1c4ull, "coefficient 17");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(16)) == 0xbe927e4f809c52adull, "coefficient 16");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(15)) == 0x3efa01a019cb1590ull, "coefficient 15");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(14)) == 0xbf56c16c16c15177ull, "coefficient 14");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(13)) == 0x3fa555555555554cull, "coefficient 13");
    asm(
        "s_mov_b32 s96, 0x5acfd57c\n\t"
        "s_mov_b32 s97, 0x3de5d93a\n\t"
        "s_mov_b32 s98, 0xbe8838d4\n\t"
        "s_mov_b32 s99, 0xbda8fae9\n\t"
        "v_mov_b64 %0, s[96:97]\n\t"
        "s_mov_b32 s96, 0x8a2b9ceb\n\t"
        "s_mov_b32 s97, 0xbe5ae5e6\n\t"
        "v_mov_b64 %1, s[98:99]\n\t"
        "s_mov_b32 s98, 0xbdb4b1c4\n\t"
        "s_mov_b32 s99, 0x3e21ee9e\n\t"
        "v_fma_f64 %0, %0, %2, s[96:97]\n\t"
        "s_mov_b32 s96, 0x57b1fe7d\n\t"
        "s_mov_b32 s97, 0x3ec71de3\n\t"
        "v_fma_f64 %1, %1, %2, s[98:99]\n\t"
        "s_mov_b32 s98, 0x809c52ad\n\t"
        "s_mov_b32 s99, 0xbe927e4f\n\t"
        "v_fma_f64 %0, %0, %2, s[96:97]\n\t"
        "s_mov_b32 s96, 0x19c161d5\n\t"
        "s_mov_b32 s97, 0xbf2a01a0\n\t"
        "v_fma_f64 %1, %1, %2, s[98:99]\n\t"
        "s_mov_b32 s98, 0x19cb1590\n\t"
        "s_mov_b32 s99, 0x3efa01a0\n\t"
        "v_fma_f64 %0, %0, %2, s[96:97]\n\t"
        "s_mov_b32 s96, 0x1110f8a6\n\t"
        "s_mov_b32 s97, 0x3f811111\n\t"
        "v_fma_f64 %1, %1, %2, s[98:99]\n\t"
        "s_mov_b32 s98, 0x16c15177\n\t"
        "s_mov_b32 s99, 0xbf56c16c\n\t"
        "v_fma_f64 %0, %0, %2, s[96:97]\n\t"
        "s_mov_b32 s96, 0x55555549\n\t"
        "s_mov_b32 s97, 0xbfc55555\n\t"
        "v_fma_f64 %1, %1, %2, s[98:99]\n\t"
        "s_mov_b32 s98, 0x5555554c\n\t"
        "s_mov_b32 s99, 0x3fa55555\n\t"
        "v_fma_f64 %0, %0, %2, s[96:97]\n\t"
        "v_fma_f64 %1, %1, %2, s[98:99]"
        : "=&v"(pa), "=&v"(pb)
        : "v"(x)
        : "s96", "s97", "s98", "s99");
#else
    pa = mt_lit(12);
    pa = fma_f64(pa, x, mt_lit(11));
    pa = fma_f64(pa, x, mt_lit(10));
    pa = fma_f64(pa, x, mt_lit(9));
    pa = fma_f64(pa, x, mt_lit(8));
    pa = fma_f64(pa, x, mt_lit(7));
    pb = mt_lit(18);
    pb = fma_f64(pb, x, mt_lit(17));
    pb = fma_f64(pb, x, mt_lit(16));
    pb = fma_f64(pb, x, mt_lit(15));
    pb = fma_f64(pb, x, mt_lit(14));
    pb = fma_f64(pb, x, mt_lit(13));
#endif
}

// pa = aT10 x^5 + aT8 x^4 + .. + aT0 (even coefficients), pb = aT9 x^4 + .. + aT1 (odd ones), x = z^2
PIK_HD void horner_atan(double x, double& pa, double& pb) {
#if defined(__HIP_DEVICE_COMPILE__) && (!defined(PIK_STRICT) || PIK_XF) // (the fused exact flavour: the same operations)
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(29)) == 0x3f90ad3ae322da11ull, "coefficient 29");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(27)) == 0x3fa97b4b24760debull, "coefficient 27");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(25)) == 0x3fb10d66a0d03d51ull, "coefficient 25");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(23)) == 0x3fb745cdc54c206eull, "coefficient 23");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(21)) == 0x3fc24924920083ffull, "coefficient 21");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(19)) == 0x3fd555555555550dull, "coefficient 19");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(28)) == 0xbfa2b4442c6a6c2full, "coefficient 28");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(26)) == 0xbfadde2d52defd9aull, "coefficient 26");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(24)) == 0xbfb3b0f2af749a6dull, "coefficient 24");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(22)) == 0xbfbc71c6fe231671ull, "coefficient 22");
    static_assert(__builtin_bit_cast(uint64_t, mt_lit(20)) == 0xbfc999999998ebc4ull, "coefficient 20");
    asm(
        "s_mov_b32 s96, 0xe322da11\n\t"
        "s_mov_b32 s97, 0x3f90ad3a\n\t"
        "s_mov_b32 s98, 0x2c6a6c2f\n\t"
        "s_mov_b32 s99, 0xbfa2b444\n\t"
        "v_mov_b64 %0, s[96:97]\n\t"
        "s_mov_b32 s96, 0x24760deb\n\t"
        "s_mov_b32 s97, 0x3fa97b4b\n\t"
        "v_mov_b64 %1, s[98:99]\n\t"
        "s_mov_b32 s98, 0x52defd9a\n\t"
        "s_mov_b32 s99, 0xbfadde2d\n\t"
        "v_fma_f64 %0, %0, %2, s[96:97]\n\t"
        "s_mov_b32 s96, 0xa0d03d51\n\t"
        "s_mov_b32 s97, 0x3fb10d66\n\t"
        "v_fma_f64 %1, %1, %2, s[98:99]\n\t"
        "s_mov_b32 s98, 0xaf749a6d\n\t"
        "s_mov_b32 s99, 0xbfb3b0f2\n\t"
        "v_fma_f64 %0, %0, %2, s[96:97]\n\t"
        "s_mov_b32 s96, 0xc54c206e\n\t"
        "s_mov_b32 s97, 0x3fb745cd\n\t"
        "v_fma_f64 %1, %1, %2, s[98:99]\n\t"
        "s_mov_b32 s98, 0xfe231671\n\t"
        "s_mov_b32 s99, 0xbfbc71c6\n\t"
        "v_fma_f64 %0, %0, %2, s[96:97]\n\t"
        "s_mov_b32 s96, 0x920083ff\n\t"
        "s_mov_b32 s97, 0x3fc24924\n\t"
        "v_fma_f64 %1, %1, %2, s[98:99]\n\t"
        "s_mov_b32 s98, 0x9998ebc4\n\t"
        "s_mov_b32 s99, 0xbfc99999\n\t"
        "v_fma_f64 %0, %0, %2, s[96:97]\n\t"
        "s_mov_b32 s96, 0x5555550d\n\t"
        "s_mov_b32 s97, 0x3fd55555\n\t"
        "v_fma_f64 %1, %1, %2, s[98:99]\n\t"
        "v_fma_f64 %0, %0, %2, s[96:97]"
        : "=&v"(pa), "=&v"(pb)
        : "v"(x)
        : "s96", "s97", "s98", "s99");
#else
    pa = mt_lit(29);
    pa = fma_f64(pa, x, mt_lit(27));
    pa = fma_f64(pa, x, mt_lit(25));
    pa = fma_f64(pa, x, mt_lit(23));
    pa = fma_f64(pa, x, mt_lit(21));
    pa = fma_f64(pa, x, mt_lit(19));
    pb = mt_lit(28);
    pb = fma_f64(pb, x, mt_lit(26));
    pb = fma_f64(pb, x, mt_lit(24));
    pb = fma_f64(pb, x, mt_lit(22));
    pb = fma_f64(pb, x, mt_lit(20));
#endif
}

// all lanes of the wavefront agree? (device: one ballot; host: the single value)
PIK_HD bool wave_all(bool v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PIK_STRICT)
    return __all(v);
#else
    return v;
#endif
}

// x with its sign bit xor-ed with bit 31 of `hi_mask` (0 or 0x80000000)
PIK_HD double flip_sign(double x, uint32_t hi_mask) {
    uint64_t u;
    __builtin_memcpy(&u, &x, sizeof u);
    u ^= (uint64_t)hi_mask << 32;
    __builtin_memcpy(&x, &u, sizeof u);
    return x;
}

// x - 2 pi rint(x / 2 pi) for |x| > 65536 (10^4 revolutions), x otherwise
PIK_HD double fold_2pi(MT m, double x) {
    const bool big = fabs(x) > 65536.0;
    const double k = big ? rint(x * PIK_MV(m, 0)) : 0.0;
    x = fma_f64(-k, PIK_MV(m, 1), x);
    return fma_f64(-k, PIK_MV(m, 2), x);
}

// FOLD = false: the caller guarantees |x| <= 65536 + pi (the product build's forward kinematics test
// all joint values of an evaluation at once -- one branch per evaluation that no real joint value
// takes; as a test per joint inside the chain the compiler turned it into ten select instructions
// per joint, a tenth of an evaluation)
template <bool FOLD = true>
PIK_HD void sincos_f64(MT m, double x, double& s, double& c) {
    // |x| > 65536 is first folded by 2 pi; a lane's result never depends on which other lanes share
    // its wavefront (fold_2pi leaves a small x untouched: k = 0 and x - 0 = x exactly)
    if (FOLD) {
        if (!wave_all(fabs(x) <= 65536.0)) x = fold_2pi(m, x);
    }
    const double fn = rint(x * PIK_MV(m, 3));
    const int n = (int)fn;
    double t = fma_f64(-fn, PIK_MV(m, 4), x);
    t = fma_f64(-fn, PIK_MV(m, 5), t);
    t = fma_f64(-fn, PIK_MV(m, 6), t);
    // fdlibm __kernel_sin / __kernel_cos minimax coefficients
    const double z = t * t;
#if defined(PIK_STRICT) && !PIK_XF
    // verification build: power sums with the smallest terms accumulated first, the operation order
    // of the oracle's portable math mode (oracle/pik_oracle.c)
    const double z2 = z * z, z3 = z2 * z, z4 = z2 * z2, z5 = z4 * z, z6 = z4 * z2, z7 = z6 * z;
    double as = PIK_MV(m, 12) * z6;
    as = as + PIK_MV(m, 11) * z5;
    as = as + PIK_MV(m, 10) * z4;
    as = as + PIK_MV(m, 9) * z3;
    as = as + PIK_MV(m, 8) * z2;
    as = as + PIK_MV(m, 7) * z;
    const double sn = t + t * as;
    double ac = PIK_MV(m, 18) * z7;
    ac = ac + PIK_MV(m, 17) * z6;
    ac = ac + PIK_MV(m, 16) * z5;
    ac = ac + PIK_MV(m, 15) * z4;
    ac = ac + PIK_MV(m, 14) * z3;
    ac = ac + PIK_MV(m, 13) * z2;
#else
    // product build: Horner (no powers of z: five instructions fewer per joint; the kernels are
    // bound by VALU issue)
    double rs, rc;
#if PIK_XF_TABLE
    // (exact-fma flavour: the same Horner steps with the coefficients as scalar-cache loads -- 24 `s_mov_b32` per
    //  sine / cosine are 8 % of the instructions of a one-lane descent step, and a lone wavefront pays a slot for each)
    rs = m.v[12];
    rs = fma_f64(rs, z, m.v[11]);
    rs = fma_f64(rs, z, m.v[10]);
    rs = fma_f64(rs, z, m.v[9]);
    rs = fma_f64(rs, z, m.v[8]);
    rs = fma_f64(rs, z, m.v[7]);
    rc = m.v[18];
    rc = fma_f64(rc, z, m.v[17]);
    rc = fma_f64(rc, z, m.v[16]);
    rc = fma_f64(rc, z, m.v[15]);
    rc = fma_f64(rc, z, m.v[14]);
    rc = fma_f64(rc, z, m.v[13]);
#else
    horner_sincos(z, rs, rc);
#endif
    const double sn = fma_f64(t * z, rs, t);
    const double zz = z * z;
#endif
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
#if defined(PIK_STRICT) && !PIK_XF
    const double cn = w + (((1.0 - w) - hz) + ac);
#else
    // (explicit: one evaluation order in every kernel variant, see dh_row)
    const double cn = w + fma_f64(zz, rc, (1.0 - w) - hz);
#endif
    const double a = (n & 1) ? cn : sn;
    const double b = (n & 1) ? sn : cn;
    // quadrant signs: flip the sign bit with an integer xor on the high word (exact, one instruction
    // instead of a select per word)
    s = flip_sign(a, (uint32_t)(n & 2) << 30);
    c = flip_sign(b, (uint32_t)((n + 1) & 2) << 30);
}

#if defined(PIK_STRICT)
// sincos_f64's fold in front of it, for one / two / D arguments under ONE branch that stays a branch: left to itself
// the compiler computes the fold for everybody and selects -- ten vector instructions per sine / cosine pair, 350 per
// descent step of a seven-joint chain, for a case no joint value ever is.  fold_2pi returns a small x as it is, so
// sincos_f64<false>(folded(x)) is sincos_f64(x), bit for bit, for every x.
#ifndef PIK_XFOLD_GUARD
#define PIK_XFOLD_GUARD 1 // (0: A/B experiments -- the compiler is free to flatten the branch again)
#endif
PIK_HD void keep_branch(double& x) {
#if defined(__HIP_DEVICE_COMPILE__) && PIK_XFOLD_GUARD
    asm volatile("" : "+v"(x)); // (something the block cannot be speculated past)
#else
    (void)x;
#endif
}
PIK_HD double folded(MT m, double x) {
    if (!(fabs(x) <= 65536.0)) {
        x = fold_2pi(m, x);
        keep_branch(x);
    }
    return x;
}
PIK_HD void folded2(MT m, double& a, double& b) {
    if (!(fabs(a) <= 65536.0 && fabs(b) <= 65536.0)) {
        a = fold_2pi(m, a);
        b = fold_2pi(m, b);
        keep_branch(a);
    }
}
template <int D>
PIK_HD void folded_all(MT m, const double (&q)[D], double (&qf)[D]) {
    bool small = true;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        small = small && fabs(q[j]) <= 65536.0;
        qf[j] = q[j];
    }
    if (!small) {
#pragma unroll
        for (int j = 0; j < D; ++j) qf[j] = fold_2pi(m, q[j]);
        keep_branch(qf[0]);
    }
}
#endif // PIK_STRICT

// R <- R * J(axis, angle): the revolute joint transform (MoveIt RevoluteJointModel::
// computeTransform), specialised for joints about +x/+y/+z where J only mixes two columns.
PIK_HD void rotate_about(double (&R)[9], uint32_t kind, CPtr a, double sn,
                         double cs) {
    if (kind == AXIS_Z) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double r0 = R[i * 3 + 0], r1 = R[i * 3 + 1];
            R[i * 3 + 0] = r0 * cs + r1 * sn;
            R[i * 3 + 1] = r1 * cs - r0 * sn;
        }
    } else if (kind == AXIS_Y) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double r0 = R[i * 3 + 0], r2 = R[i * 3 + 2];
            R[i * 3 + 0] = r0 * cs - r2 * sn;
            R[i * 3 + 2] = r2 * cs + r0 * sn;
        }
    } else if (kind == AXIS_X) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double r1 = R[i * 3 + 1], r2 = R[i * 3 + 2];
            R[i * 3 + 1] = r1 * cs + r2 * sn;
            R[i * 3 + 2] = r2 * cs - r1 * sn;
        }
    } else {
        const double x = a[0], y = a[1], z = a[2];
        const double tt = 1.0 - cs;
        const double txy = tt * (x * y), txz = tt * (x * z), tyz = tt * (y * z);
        const double zs = z * sn, ys = y * sn, xs = x * sn;
        double J[9];
        J[0] = tt * (x * x) + cs;
        J[3] = txy + zs;
        J[6] = txz - ys;
        J[1] = txy - zs;
        J[4] = tt * (y * y) + cs;
        J[7] = tyz + xs;
        J[2] = txz + ys;
        J[5] = tyz - xs;
        J[8] = tt * (z * z) + cs;
        double r[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                r[i * 3 + j] = xdot3(R[i * 3 + 0], J[j], R[i * 3 + 1], J[3 + j], R[i * 3 + 2], J[6 + j]);
            }
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = r[i];
    }
}

// ---- one ROW of the running frame (R[i][0..2], t[i]) through the steps of the fast forward
// kinematics.  Every product-sum is spelled out with explicit fused multiply-adds, so that the
// arithmetic does not depend on how the compiler contracts an expression in a given context: the
// one-lane-per-evaluation code (three rows per lane) and the cooperative evaluation of the wide
// kernels (one row per lane, pik_kernels.hpp gd_wide) produce the same bits by construction.
// Rz(q + theta0) Tz(tz) Tx(a) Rx(alpha) applied to a row
PIK_HD void dh_row(double& r0, double& r1, double& r2, double& t, double sn, double cs, double tz,
                   double a, double ca, double sa) {
    const double x0 = fma_f64(r0, cs, r1 * sn);     // Rz: columns 0, 1
    const double x1 = fma_f64(r1, cs, -(r0 * sn));
    t = fma_f64(x0, a, fma_f64(r2, tz, t));          // Tz, Tx
    const double n1 = fma_f64(x1, ca, r2 * sa);      // Rx: columns 1, 2
    const double n2 = fma_f64(r2, ca, -(x1 * sa));
    r0 = x0;
    r1 = n1;
    r2 = n2;
}
// Rz(q + theta0) Tz(tz) applied to a row (the joint part of a general step)
PIK_HD void rz_row(double& r0, double& r1, double& r2, double& t, double sn, double cs, double tz) {
    const double x0 = fma_f64(r0, cs, r1 * sn);
    const double x1 = fma_f64(r1, cs, -(r0 * sn));
    t = fma_f64(r2, tz, t);
    r0 = x0;
    r1 = x1;
}
// a row times a constant rigid transform o (rotation row-major [0..8], translation [9..11])
PIK_HD void iso_row(double& r0, double& r1, double& r2, double& t, const double (&o)[12]) {
    const double n0 = fma_f64(r2, o[6], fma_f64(r1, o[3], r0 * o[0]));
    const double n1 = fma_f64(r2, o[7], fma_f64(r1, o[4], r0 * o[1]));
    const double n2 = fma_f64(r2, o[8], fma_f64(r1, o[5], r0 * o[2]));
    t = fma_f64(r2, o[11], fma_f64(r1, o[10], fma_f64(r0, o[9], t)));
    r0 = n0;
    r1 = n1;
    r2 = n2;
}
// joint value -> rotation angle about / translation along the joint's z axis (branch-free: a
// prismatic joint rotates by theta0 and moves q + d, a revolute one rotates by q + theta0 and moves d)
// (q * 0 or q * 1 is exact, so the separate multiply and add round like the fused form; spelled
// unfused because a fused multiply-add with two scalar-register operands needs one of them copied
// into a vector register first)
PIK_HD double dh_angle(double q, double pm, double th0) {
#pragma clang fp contract(off)
    const double a = q * (1.0 - pm);
    return a + th0;
}
PIK_HD double dh_shift(double q, double pm, double d) {
#pragma clang fp contract(off)
    const double a = q * pm;
    return a + d;
}

// sin / cos of (theta + d) from sin / cos of theta for a SMALL d (|d| <= 1e-3: the truncated series
// are exact to < 1e-21): the two line-search evaluations of a gradient step sit at q -+ g with
// |g_j| < step size, so their joint angles are the accepted point's angles plus a tiny delta --
// 13 instructions per joint instead of a full sine / cosine (~38).
//   sin d = d + d^3 (-1/6 + d^2 / 120),   1 - cos d = d^2 (1/2 - d^2 / 24)
//   sin(theta + d) = s + (c sin d - s (1 - cos d)),   cos(theta + d) = c - (s sin d + c (1 - cos d))
PIK_HD void sincos_delta(double sn, double cs, double d, double& s2, double& c2) {
#pragma clang fp contract(off)
    const double d2 = d * d;
    const double u = d2 * (1.0 / 120.0) + (-1.0 / 6.0);
    const double sd = fma_f64(d * d2, u, d);
    const double ep = d2 * fma_f64(d2, -1.0 / 24.0, 0.5);
    s2 = sn + fma_f64(cs, sd, -(sn * ep));
    c2 = cs - fma_f64(sn, sd, cs * ep);
}

#if !defined(PIK_STRICT)
// The joints of the fast build's forward kinematics (Denavit-Hartenberg form).  GEN: the chain has
// ill-conditioned pairs of axes whose step is a general constant transform (ChainK::dhg); those
// chains run a copy of the loop with a chain-uniform branch per joint, every other chain the
// branch-free copy (one basic block -- the per-joint branch alone cost 13 % when it was in the
// common path: it stops the scheduler from overlapping one joint's sincos with the previous
// joint's products).
// SCM (sine / cosine mode): 0 plain; 1 also exports every joint's sine / cosine (bsn, bcs); 2 takes
// them from the exported values of the point qb plus the small difference q - qb (sincos_delta)
template <int D, bool WANT_FRAMES, bool MASKED, bool GEN, bool FRJ0 = true, int SCM = 0>
PIK_HD void fk_dh_joints(CK<D> c_in, const double (&q)[D], double (&R)[9], double (&t)[3], double* fr,
                         int stride, double (&o)[12], const double (&qb)[D], double (&bsn)[D],
                         double (&bcs)[D]) {
    const uint32_t active_mask = MASKED ? c_in.active_mask : ~0u;
    (void)active_mask;
    const uint32_t prismatic_mask = PIK_PRISMATIC(c_in);
    const uint32_t general_mask = GEN ? c_in.dh_general_mask : 0u;
    (void)general_mask;
    // fast build: Denavit-Hartenberg chain, one basic block, constants software-pipelined
    {
        CK<D> c0 = fresh(c_in);
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = c0.dh_base[i];
        t[0] = c0.dh_base[9];
        t[1] = c0.dh_base[10];
        t[2] = c0.dh_base[11];
    }
    double th0, dd, aa, ca, sa;
    {
        CK<D> c0 = fresh_after(c_in, q[0]);
        th0 = c0.dh[0][0]; dd = c0.dh[0][1]; aa = c0.dh[0][2]; ca = c0.dh[0][3]; sa = c0.dh[0][4];
    }
    // joint angles beyond 10^4 revolutions are folded by 2 pi first (see sincos_f64): one test for
    // the whole joint vector; the values of prismatic joints are lengths and stay as they are
    double qv[D];
    {
        double amax = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            qv[j] = q[j];
            amax = fmax(amax, fabs(q[j]));
        }
        if (!wave_all(amax <= 65536.0)) {
#pragma unroll
            for (int j = 0; j < D; ++j)
                if (!((prismatic_mask >> j) & 1u)) qv[j] = fold_2pi(c_in.mt, qv[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
        MT mt = c_in.mt; // unused: the coefficients are literals
        // FRJ0 = false: the first joint's frame is the chain constant dh_base and is not stored; the
        // rows hold joints 1 .. D-1 (six rows fewer: what lets eight one-lane wavefronts share a CU's
        // LDS at D = 7, see MemeticLds)
        if (WANT_FRAMES && (FRJ0 || j > 0)) {
            const int row0 = 6 * (FRJ0 ? j : j - 1);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                fr[(row0 + i) * stride] = R[i * 3 + 2]; // world joint axis = third column
                fr[(row0 + 3 + i) * stride] = t[i];     // a point on it
            }
        }
        // branch-free joint: a prismatic joint is a rotation by theta0 plus a translation q + d along
        // z, a revolute one a rotation by q + theta0 plus the translation d (x * 1.0, x + 0.0 exact)
        const double pm = ((prismatic_mask >> j) & 1u) ? 1.0 : 0.0;
        // a variable that is not on this tip's path: identity step (host) and a value of 0
        const double qj = MASKED ? (((active_mask >> j) & 1u) ? qv[j] : 0.0) : qv[j];
        double sn, cs;
        if (SCM == 2) {
            sincos_delta(bsn[j], bcs[j], (q[j] - qb[j]) * (1.0 - pm), sn, cs);
        } else {
            sincos_f64<false>(mt, dh_angle(qj, pm, th0), sn, cs);
            if (SCM == 1) {
                bsn[j] = sn;
                bcs[j] = cs;
            }
        }
        const double tz = dh_shift(qj, pm, dd);
        const double a_j = aa, ca_j = ca, sa_j = sa;
        // next joint's constants (or the tip transform): issued now, land during this joint's work
        {
            CK<D> cn = fresh_after(c_in, sn);
            if (j + 1 < D) {
                th0 = cn.dh[(j + 1 < D) ? j + 1 : 0][0];
                dd = cn.dh[(j + 1 < D) ? j + 1 : 0][1];
                aa = cn.dh[(j + 1 < D) ? j + 1 : 0][2];
                ca = cn.dh[(j + 1 < D) ? j + 1 : 0][3];
                sa = cn.dh[(j + 1 < D) ? j + 1 : 0][4];
            } else {
#pragma unroll
                for (int i = 0; i < 12; ++i) o[i] = cn.dh_tip[i];
            }
        }
        if (GEN && ((general_mask >> j) & 1u)) {
            // chain-uniform and rare (an ill-conditioned pair of axes): Rz / Tz of the joint, then
            // the general constant transform to the next joint's frame
            double og[12];
            {
                CK<D> cg = fresh_after(c_in, cs);
#pragma unroll
                for (int i = 0; i < 12; ++i) og[i] = cg.dhg[j][i];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                rz_row(R[i * 3 + 0], R[i * 3 + 1], R[i * 3 + 2], t[i], sn, cs, tz);
                iso_row(R[i * 3 + 0], R[i * 3 + 1], R[i * 3 + 2], t[i], og);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                dh_row(R[i * 3 + 0], R[i * 3 + 1], R[i * 3 + 2], t[i], sn, cs, tz, a_j, ca_j, sa_j);
        }
    }
}
#endif

#if defined(PIK_STRICT)
// ---- the steps of MoveIt's chain product, one joint at a time (verification build) ----
// R <- R * J(axis, angle): RevoluteJointModel::computeTransform (Rodrigues form: c, s, t = 1 - c) multiplied in
// from the right, operation for operation what rotate_about(AXIS_GENERAL) and the oracle's joint_transform +
// iso_mul perform.  For an axis that is exactly +x / +y / +z the terms of that formula that are exact zeros
// are not computed: t * (0 * a) = 0 and 0 * s = +-0 exactly, x + (+-0) = x, and t * (1 * 1) + c is the one
// diagonal entry that is NOT exactly 1 (kept: d below).  The results are the same numbers (a zero may come out
// with the other sign, which nothing downstream can see: sums, products, comparisons, square roots of sums);
// 23 instead of 65 operations per joint of an industrial arm.
PIK_HD void rotate_exact(double (&R)[9], uint32_t kind, CPtr a, double sn, double cs) {
    if (kind == AXIS_GENERAL) {
        rotate_about(R, AXIS_GENERAL, a, sn, cs);
        return;
    }
    const double tt = 1.0 - cs;
    const double d = tt + cs; // tt * (1 * 1) + cs
    if (kind == AXIS_Z) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double r0 = R[i * 3 + 0], r1 = R[i * 3 + 1], r2 = R[i * 3 + 2];
            R[i * 3 + 0] = xmad(r1, sn, r0 * cs);    // (fused: fma(r2, +0, fma(r1, sn, r0 cs)) is the same number)
            R[i * 3 + 1] = xmad(r1, cs, -(r0 * sn)); // fma(r2, +0, fma(r1, cs, r0 (-sn)))
            R[i * 3 + 2] = r2 * d;                   // fma(r2, d, +-0)
        }
    } else if (kind == AXIS_Y) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double r0 = R[i * 3 + 0], r1 = R[i * 3 + 1], r2 = R[i * 3 + 2];
            R[i * 3 + 0] = xmad(r2, -sn, r0 * cs); // fma(r2, -sn, fma(r1, +0, r0 cs))
            R[i * 3 + 1] = r1 * d;                 // fma(r2, +0, fma(r1, d, +-0))
            R[i * 3 + 2] = xmad(r2, cs, r0 * sn);  // fma(r2, cs, fma(r1, +0, r0 sn))
        }
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double r0 = R[i * 3 + 0], r1 = R[i * 3 + 1], r2 = R[i * 3 + 2];
            R[i * 3 + 0] = r0 * d;                    // fma(r2, +0, fma(r1, +0, r0 d))
            R[i * 3 + 1] = xmad(r2, sn, r1 * cs);     // fma(r2, sn, fma(r1, cs, +-0)): fma(r1, cs, 0) = r1 cs
            R[i * 3 + 2] = xmad(r2, cs, r1 * (-sn));  // fma(r2, cs, fma(r1, -sn, +-0))
        }
    }
}

// ---- UZ: the chain class "every variable a revolute joint about its frame's +z, no identity origin, a tip
// transform, no floating / mimic joint, one tip frame" (ChainK::uniform_z, decided on the host: Franka Panda, KUKA
// iiwa, any description written in the Denavit-Hartenberg convention).  What the general routines decide per joint
// at run time -- origin skipped?  prismatic?  which axis? -- is a compile-time constant in the UZ forms (here:
// the whole evaluation; pik_exact.hpp: the descent), and the arithmetic of the path taken is the same, operation for
// operation: the same bits.  Chain lengths that have them: up to ten variables (eight until round 6; the packing of
// ChainK::origin_kinds ends there, and the kernels of longer chains sit at the register cap in their general forms).
#ifndef PIK_XUZ_MAXD
#define PIK_XUZ_MAXD 10
#endif
// (the plain-IEEE verification library has them too -- PIK_XUZ_PLAIN: the forms delete decisions, not operations, and
//  every product-sum in them goes through xmad / xdot3 / iso_mul, which are the unfused ones there; only the products
//  by the exact ones and zeros of origins and tip stay with the fused flavour, whose accumulation order they need)
#ifndef PIK_XUZ_PLAIN
#define PIK_XUZ_PLAIN 1
#endif
#if defined(PIK_STRICT)
#define PIK_XUZ_D(D) ((PIK_XF || PIK_XUZ_PLAIN) && (D) <= PIK_XUZ_MAXD)
#else
#define PIK_XUZ_D(D) 0
#endif
// (ChainK::origin_kinds / origin_pmasks hold three bits per joint in 32-bit words -- make_chain_k packs joints 0..9 and
// PIK_OKIND shifts by 3 j: the specialised forms, their only readers, must not reach an eleventh joint)
static_assert(PIK_XUZ_MAXD <= 10, "PIK_XUZ_MAXD > 10: origin_kinds / origin_pmasks hold ten joints");
#ifndef PIK_XUA
#define PIK_XUA 1 // (0: chains of class 2 run the general forms -- A/B experiments)
#endif
// ... and UA (uniform_z = 2): the same with every axis exactly +x, +y OR +z (Universal Robots, most industrial arms
// whose description is not in the Denavit-Hartenberg convention): the joint's axis is one wave-uniform decision
// per joint, between three rotations of equal length.
// R <- R * Rz(angle): rotate_exact's AXIS_Z case
PIK_HD void rotate_z_exact(double (&R)[9], double sn, double cs) {
    const double tt = 1.0 - cs;
    const double d = tt + cs;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double r0 = R[i * 3 + 0], r1 = R[i * 3 + 1], r2 = R[i * 3 + 2];
        R[i * 3 + 0] = xmad(r1, sn, r0 * cs);
        R[i * 3 + 1] = xmad(r1, cs, -(r0 * sn));
        R[i * 3 + 2] = r2 * d;
    }
}
// R <- R * R_axis(angle), axis exactly +x / +y / +z (kind: wave-uniform, never AXIS_GENERAL): rotate_exact's three
// special cases
PIK_HD void rotate_axis_exact(double (&R)[9], uint32_t kind, double sn, double cs) {
    const double tt = 1.0 - cs;
    const double d = tt + cs;
    if (kind == AXIS_Z) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double r0 = R[i * 3 + 0], r1 = R[i * 3 + 1], r2 = R[i * 3 + 2];
            R[i * 3 + 0] = xmad(r1, sn, r0 * cs);
            R[i * 3 + 1] = xmad(r1, cs, -(r0 * sn));
            R[i * 3 + 2] = r2 * d;
        }
    } else if (kind == AXIS_Y) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double r0 = R[i * 3 + 0], r1 = R[i * 3 + 1], r2 = R[i * 3 + 2];
            R[i * 3 + 0] = xmad(r2, -sn, r0 * cs);
            R[i * 3 + 1] = r1 * d;
            R[i * 3 + 2] = xmad(r2, cs, r0 * sn);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double r0 = R[i * 3 + 0], r1 = R[i * 3 + 1], r2 = R[i * 3 + 2];
            R[i * 3 + 0] = r0 * d;
            R[i * 3 + 1] = xmad(r2, sn, r1 * cs);
            R[i * 3 + 2] = xmad(r2, cs, r1 * (-sn));
        }
    }
}
// XM = 1: about z; XM = 2: about the joint's axis
template <int XM>
PIK_HD void x_rotate(double (&R)[9], uint32_t kind, double sn, double cs) {
    if constexpr (XM == 1) {
        (void)kind;
        rotate_z_exact(R, sn, cs);
    } else {
        rotate_axis_exact(R, kind, sn, cs);
    }
}

// two frames through the same joint (a probe pair, the two line-search evaluations): ONE decision about the axis
// with both rotations inside each answer -- two decisions in a row left the compiler with register copies at
// every merge (18 moves per joint of a pair in the rolled fork loop)
template <int XM>
PIK_HD void x_rotate_pair(double (&Ra)[9], double (&Rb)[9], uint32_t kind, double sna, double csa, double snb, double csb) {
    if constexpr (XM == 1) {
        (void)kind;
        rotate_z_exact(Ra, sna, csa);
        rotate_z_exact(Rb, snb, csb);
    } else {
        if (kind == AXIS_Z) {
            rotate_axis_exact(Ra, AXIS_Z, sna, csa);
            rotate_axis_exact(Rb, AXIS_Z, snb, csb);
        } else if (kind == AXIS_Y) {
            rotate_axis_exact(Ra, AXIS_Y, sna, csa);
            rotate_axis_exact(Rb, AXIS_Y, snb, csb);
        } else {
            rotate_axis_exact(Ra, AXIS_X, sna, csa);
            rotate_axis_exact(Rb, AXIS_X, snb, csb);
        }
    }
}

// (R, t) <- (R, t) * origin of joint j, skipped when the origin is exactly the identity; `blank`: nothing has
// been multiplied in yet, the origin is copied (identity * o = o)
template <int D>
PIK_HD void chain_origin(CK<D> c, int j, double (&R)[9], double (&t)[3], bool blank) {
    if ((c.origin_ident_mask >> j) & 1u) return;
    CPtr o = c.O[j];
    if (blank) {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = o[i];
        t[0] = o[9];
        t[1] = o[10];
        t[2] = o[11];
    } else {
        iso_mul(R, t, o);
    }
}

// (R, t) <- (R, t) * transform of the revolute / prismatic joint j at value v (sn, cs = sin / cos of v)
template <int D>
PIK_HD void chain_joint(CK<D> c, int j, double (&R)[9], double (&t)[3], bool prismatic, uint32_t kind, double v,
                        double sn, double cs) {
    CPtr a = c.axis[j];
    if (prismatic) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#if PIK_XF
            t[i] = fma_f64(R[i * 3 + 2], a[2] * v, fma_f64(R[i * 3 + 1], a[1] * v, fma_f64(R[i * 3 + 0], a[0] * v, t[i])));
#else
            t[i] = R[i * 3 + 0] * (a[0] * v) + R[i * 3 + 1] * (a[1] * v) + R[i * 3 + 2] * (a[2] * v) + t[i];
#endif
        }
    } else {
        rotate_exact(R, kind, a, sn, cs);
    }
}

// the forward kinematics of a UZ chain: the D sines / cosines first (independent polynomial chains for the
// scheduler to interleave), then the chain product unrolled over the joints -- fk's PIK_STRICT path without its
// per-joint decisions
template <int D, int XM>
PIK_HD void fk_uz(CK<D> c, const double (&q)[D], double (&R)[9], double (&t)[3]) {
    const uint32_t kinds = XM == 1 ? 0u : c.axis_kind;
    double sn[D], cs[D];
    double qf[D];
    folded_all<D>(c.mt, q, qf);
#pragma unroll
    for (int j = 0; j < D; ++j) sincos_f64<false>(c.mt, qf[j], sn[j], cs[j]);
#pragma unroll
    for (int j = 0; j < D; ++j) {
        CPtr o = c.O[j];
        if (j == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = o[i];
            t[0] = o[9];
            t[1] = o[10];
            t[2] = o[11];
        } else {
            x_iso_mul<XM>(R, t, o, PIK_OKIND(c, j), PIK_OPM(c, j));
        }
        x_rotate<XM>(R, (kinds >> (2 * j)) & 3u, sn[j], cs[j]);
    }
    x_tip_mul<XM>(R, t, c.tip, c.tip_kind, PIK_TPM(c));
}
#endif

// Forward kinematics of the serial chain.  When WANT_FRAMES, also stores for every joint j the
// world-frame joint axis (rows 6j..6j+2) and joint origin (rows 6j+3..6j+5) into `fr` with element
// stride `stride` (on the GPU: one LDS column per lane, stride 64) -- the line a revolute joint
// rotates the tip about / the direction a prismatic joint moves it along.  The gradient probes of
// the fast step are built from these frames (the idea behind the reference's CachedJointFrames,
// src/forward_kinematics.cpp:102-125: a joint perturbation only moves that joint's frame).
template <int D, bool WANT_FRAMES, bool MASKED = false, bool FRJ0 = true, int SCM = 0>
PIK_HD void fk(CK<D> c_in, const double (&q)[D], double (&R)[9], double (&t)[3], double* fr,
               int stride, const double (&qb)[D], double (&bsn)[D], double (&bcs)[D]) {
    const uint32_t active_mask = MASKED ? c_in.active_mask : ~0u;
    (void)active_mask;
    // flag words: read once (a handful of SGPRs), not once per joint
    const uint32_t prismatic_mask = PIK_PRISMATIC(c_in);
#if !defined(PIK_STRICT)
    const uint32_t general_mask = c_in.dh_general_mask;
    (void)general_mask;
#endif
#if defined(PIK_STRICT)
    // strict-arithmetic build: MoveIt's chain product operation for operation (origin skipped
    // when it is the identity, Rodrigues joint matrix), bit-identical to the CPU oracle.  A ROLLED
    // loop over the joints (the chain constants are indexed by a wave-uniform register): one copy of
    // the joint's code whatever the chain length, shared with the memoised descent (pik_exact.hpp)
    const uint32_t float_mask = c_in.float_mask, skip_mask = c_in.skip_mask;
    const uint32_t kinds = c_in.axis_kind;
    R[0] = 1.0; R[1] = 0.0; R[2] = 0.0;
    R[3] = 0.0; R[4] = 1.0; R[5] = 0.0;
    R[6] = 0.0; R[7] = 0.0; R[8] = 1.0;
    t[0] = t[1] = t[2] = 0.0;
    bool blank = !MASKED; // nothing multiplied in yet: the first origin is copied (1 * x + 0 + 0 = x)
    // A floating joint takes the six variables in front of its rot_w.  They are NOT read as q[j - 6] .. q[j - 1]:
    // with that in the loop the compiler walks the joint vector with the pointer &q[j - 6] and reads q[j] as
    // p[6] -- and a FLAT access whose base register points below the lane's scratch frame (the first six
    // iterations, when q sits at the start of the frame) is an aperture violation on this chip, although
    // base + offset is a perfectly good address.  A window of the last six values in registers instead.
    const bool has_float = (float_mask | skip_mask) != 0u;
    double w1 = 0.0, w2 = 0.0, w3 = 0.0, w4 = 0.0, w5 = 0.0, w6 = 0.0; // q[j - 1] .. q[j - 6]
    // mimic joints: steps whose value follows a variable (RobotState::updateMimicJoints), behind the joint of
    // variable `after`
    const int n_mimic = (int)c_in.m_count;
    auto mimic_steps = [&](int after) {
#pragma unroll 1
        for (int m = 0; m < n_mimic; ++m) {
            if (c_in.m_after[m] != after) continue;
            const double v = c_in.mmult[m] * q[c_in.m_var[m]] + c_in.moff[m];
            if (!((c_in.m_ident_mask >> m) & 1u)) {
                if (blank) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) R[i] = c_in.mO[m][i];
                    t[0] = c_in.mO[m][9];
                    t[1] = c_in.mO[m][10];
                    t[2] = c_in.mO[m][11];
                } else {
                    iso_mul(R, t, c_in.mO[m]);
                }
            }
            CPtr a = c_in.maxis[m];
            if ((c_in.m_pris_mask >> m) & 1u) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#if PIK_XF
                    t[i] = fma_f64(R[i * 3 + 2], a[2] * v, fma_f64(R[i * 3 + 1], a[1] * v, fma_f64(R[i * 3 + 0], a[0] * v, t[i])));
#else
                    t[i] = R[i * 3 + 0] * (a[0] * v) + R[i * 3 + 1] * (a[1] * v) + R[i * 3 + 2] * (a[2] * v) + t[i];
#endif
                }
            } else {
                double sn, cs;
                sincos_f64(c_in.mt, v, sn, cs);
                rotate_exact(R, (c_in.m_kind >> (2 * m)) & 3u, a, sn, cs);
            }
            blank = false;
        }
    };
    if (n_mimic) mimic_steps(-1);
#pragma unroll 1
    for (int j = 0; j < D; ++j) {
        const double qj = q[j];
        const bool on_path = !MASKED || ((active_mask >> j) & 1u); // a joint of this tip's path
        const bool skip = (skip_mask >> j) & 1u; // one of the first six variables of a floating joint
        if (on_path && !skip) {
            if ((float_mask >> j) & 1u) {
                // FloatingJointModel::computeTransform: Translation(v0 v1 v2) * Quaterniond(w = v6, v3, v4, v5),
                // the quaternion as it is (Eigen toRotationMatrix); origin product first, as for every joint
                chain_origin<D>(c_in, j, R, t, blank);
                const double qq[4] = {qj, w3, w2, w1};
                double J[12];
                double JR[9];
                quat_to_matrix(qq, JR);
#pragma unroll
                for (int i = 0; i < 9; ++i) J[i] = JR[i];
                J[9] = w6;
                J[10] = w5;
                J[11] = w4;
                iso_mul_r(R, t, J);
            } else {
                chain_origin<D>(c_in, j, R, t, blank);
                const bool pj = (prismatic_mask >> j) & 1u;
                double sn = 0.0, cs = 1.0;
                if (!pj) sincos_f64(c_in.mt, qj, sn, cs);
                chain_joint<D>(c_in, j, R, t, pj, (kinds >> (2 * j)) & 3u, qj, sn, cs);
            }
            blank = false;
        }
        if (n_mimic) mimic_steps(j);
        if (has_float) {
            w6 = w5;
            w5 = w4;
            w4 = w3;
            w3 = w2;
            w2 = w1;
            w1 = qj;
        }
    }
    if (!c_in.tip_ident) iso_mul(R, t, c_in.tip);
    (void)fr;
    (void)stride;
#else
    // fast build: Denavit-Hartenberg chain (see fk_dh_joints)
    double o[12];
#if PIK_COMMON
    fk_dh_joints<D, WANT_FRAMES, MASKED, false, FRJ0, SCM>(c_in, q, R, t, fr, stride, o, qb, bsn, bcs);
#else
    if (general_mask != 0u) {
        fk_dh_joints<D, WANT_FRAMES, MASKED, true, FRJ0, SCM>(c_in, q, R, t, fr, stride, o, qb, bsn, bcs);
    } else {
        fk_dh_joints<D, WANT_FRAMES, MASKED, false, FRJ0, SCM>(c_in, q, R, t, fr, stride, o, qb, bsn, bcs);
    }
#endif
#pragma unroll
    for (int i = 0; i < 3; ++i) iso_row(R[i * 3 + 0], R[i * 3 + 1], R[i * 3 + 2], t[i], o);
#endif
}

// (plain form: no sine / cosine exchange)
template <int D, bool WANT_FRAMES, bool MASKED = false, bool FRJ0 = true>
PIK_HD void fk(CK<D> c_in, const double (&q)[D], double (&R)[9], double (&t)[3], double* fr,
               int stride) {
    double bsn[D], bcs[D]; // unused in mode 0
    fk<D, WANT_FRAMES, MASKED, FRJ0, 0>(c_in, q, R, t, fr, stride, q, bsn, bcs);
}

// d = a * conj(b)  (Eigen quaternion product), quaternions as w x y z
PIK_HD void quat_mul_conj(const double (&a)[4], const double (&b)[4], double (&d)[4]) {
    const double aw = a[0], ax = a[1], ay = a[2], az = a[3];
    const double bw = b[0], bx = -b[1], by = -b[2], bz = -b[3];
#if PIK_XF
    d[0] = fma_f64(-az, bz, fma_f64(-ay, by, fma_f64(-ax, bx, aw * bw)));
    d[1] = fma_f64(-az, by, fma_f64(ay, bz, fma_f64(ax, bw, aw * bx)));
    d[2] = fma_f64(-ax, bz, fma_f64(az, bx, fma_f64(ay, bw, aw * by)));
    d[3] = fma_f64(-ay, bx, fma_f64(ax, by, fma_f64(az, bw, aw * bz)));
#else
    d[0] = aw * bw - ax * bx - ay * by - az * bz;
    d[1] = aw * bx + ax * bw + ay * bz - az * by;
    d[2] = aw * by + ay * bw + az * bx - ax * bz;
    d[3] = aw * bz + az * bw + ax * by - ay * bx;
#endif
}

// atan2(y, x) for y >= 0, x >= 0 -- the only way the path uses it (Eigen angularDistance).
// fdlibm's atan scheme (breakpoints 7/16, 11/16, 19/16, 39/16; odd minimax polynomial; hi/lo
// table) with the interval reduction applied to the (y, x) pair so that a single divide serves
// both the quotient and the reduction; selects only, no divergence.  <= 1 ulp from libm (the
// verification build; the product build's shorter reduction is described in the function).
PIK_HD double atan2_pos(MT m, double y, double x) {
#if !defined(PIK_STRICT) || PIK_XF
    // Product build: two reduction steps instead of fdlibm's four breakpoints -- the smaller over the
    // larger argument (atan2 = pi/2 - atan(x / y) when y > x), then atan(a / b) = pi/4 + atan((a - b) /
    // (a + b)) above tan(pi/8); |r| <= tan(pi/8) < 7/16, so fdlibm's polynomial serves unchanged.  14
    // select instructions instead of 32 (the four-way chains picked among constants, each of which
    // had to be copied into a vector register first), still one divide; <= 2 ulp from libm.
    const bool sw = y > x;
    const double a = sw ? x : y, b = sw ? y : x; // (selects, not min / max: a NaN stays a NaN)
    const bool t = a > 0.41421356237309503 * b;
    const double num = t ? a - b : a;
    const double den = t ? a + b : b;
    const double r = num / den;
    const double z = r * r;
    const double w = z * z;
    double s1, s2;
#if PIK_XF_TABLE2
    s1 = m.v[29];
    s1 = fma_f64(s1, w, m.v[27]);
    s1 = fma_f64(s1, w, m.v[25]);
    s1 = fma_f64(s1, w, m.v[23]);
    s1 = fma_f64(s1, w, m.v[21]);
    s1 = fma_f64(s1, w, m.v[19]);
    s2 = m.v[28];
    s2 = fma_f64(s2, w, m.v[26]);
    s2 = fma_f64(s2, w, m.v[24]);
    s2 = fma_f64(s2, w, m.v[22]);
    s2 = fma_f64(s2, w, m.v[20]);
#else
    horner_atan(w, s1, s2);
#endif
    const double poly = fma_f64(z, s2, s1) * z; // z ((aT0 + aT2 w + ...) + z (aT1 + aT3 w + ...))
    const double p0 = fma_f64(-r, poly, r);     // atan(r)
    const double p1 = t ? (PIK_MV(m, 31) + (p0 + PIK_MV(m, 35))) : p0;   // + pi/4 (hi, lo)
    const double res = sw ? (PIK_MV(m, 33) - (p1 - PIK_MV(m, 37))) : p1; // pi/2 (hi, lo) - ...
    return (y == 0.0) ? 0.0 : res;
#else
    const double y16 = 16.0 * y;
    const bool c0 = y16 < 7.0 * x, c1 = y16 < 11.0 * x, c2 = y16 < 19.0 * x, c3 = y16 < 39.0 * x;
    const double num = c0 ? y : c1 ? (2.0 * y - x) : c2 ? (y - x) : c3 ? (y - 1.5 * x) : -x;
    const double den = c0 ? x : c1 ? (2.0 * x + y) : c2 ? (y + x) : c3 ? (x + 1.5 * y) : y;
    const double hi = c0 ? 0.0 : c1 ? PIK_MV(m, 30) : c2 ? PIK_MV(m, 31) : c3 ? PIK_MV(m, 32) : PIK_MV(m, 33);
    const double lo = c0 ? 0.0 : c1 ? PIK_MV(m, 34) : c2 ? PIK_MV(m, 35) : c3 ? PIK_MV(m, 36) : PIK_MV(m, 37);
    const double r = num / den;
    // atan(r) = r - r * sum_k aT_k z^(k+1), z = r^2
    const double z = r * r;
    // (power sum, smallest terms first: the oracle's portable math mode, see sincos_f64)
    const double z2 = z * z, z3 = z2 * z, z4 = z2 * z2, z5 = z4 * z, z6 = z4 * z2, z7 = z4 * z3,
                 z8 = z4 * z4, z9 = z8 * z, z10 = z8 * z2, z11 = z8 * z3;
    double a = PIK_MV(m, 29) * z11;
    a = a + PIK_MV(m, 28) * z10;
    a = a + PIK_MV(m, 27) * z9;
    a = a + PIK_MV(m, 26) * z8;
    a = a + PIK_MV(m, 25) * z7;
    a = a + PIK_MV(m, 24) * z6;
    a = a + PIK_MV(m, 23) * z5;
    a = a + PIK_MV(m, 22) * z4;
    a = a + PIK_MV(m, 21) * z3;
    a = a + PIK_MV(m, 20) * z2;
    a = a + PIK_MV(m, 19) * z;
    const double res = c0 ? (r - r * a) : (hi - ((r * a - lo) - r));
    return (y == 0.0) ? 0.0 : res;
#endif
}

// Eigen angularDistance from the relative quaternion: 2 atan2(|vec|, |w|)
PIK_HD double angle_of(MT m, const double (&d)[4], double& vnorm) {
#if PIK_XF
    vnorm = sqrt_pos(xsumsq3(d[1], d[2], d[3]));
#else
    vnorm = sqrt_pos(d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
#endif
    return 2.0 * atan2_pos(m, vnorm, fabs(d[0]));
}

struct PoseErr {
    double lin; // linear_distance(goal, frame)
    double ang; // angular_distance(goal, frame)
};

// make_pose_cost_fn -- src/goal.cpp:51-78 (terms dropped when the scale is <= 0)
PIK_HD double pose_cost(PK p, const PoseErr& e) {
    double c = 0.0;
    if (PIK_POS_ON(p)) {
        const double a = e.lin * p.pos_scale;
        c = a * a;
    }
    if (PIK_ROT_ON(p)) {
        const double a = e.ang * p.rot_scale;
#if PIK_XF
        c = fma_f64(a, a, c);
#else
        c = c + a * a;
#endif
    }
    return c;
}

// src/goal.cpp:91-144, each already multiplied by weight^2 as make_cost_fn does (:197-200)
template <int D>
PIK_HD double goal_cost_term(CK<D> c, PK p, int which,
                             const double (&q)[D], const double (&seed)[D]) {
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const bool bounded = (c.bounded_mask >> i) & 1u;
        double v;
        if (which == 0) {
            const double mid = (c.qmin[i] + c.qmax[i]) * 0.5;
            v = (q[i] - mid) * c.mdf[i];
            if (!bounded) v = 0.0;
        } else if (which == 1) {
            v = fmax(0.0, fabs(q[i] - c.mid[i]) * 2.0 - c.hspan[i]) * c.mdf[i];
            if (!bounded) v = 0.0;
        } else {
            v = (q[i] - seed[i]) * c.mdf[i];
        }
#if PIK_XF
        sum = fma_f64(v, v, sum);
#else
        sum += v * v;
#endif
    }
    return sum;
}

// Variable::clamp_to_limits -- src/robot.cpp:36-42
// Product build: max / min against per-variable limits (ChainK::clo / chi) and a NaN that stays a
// NaN -- five vector instructions.  The literal form below costs sixteen: its selects between a
// scalar-register constant and a vector register under a mask exceed the one scalar operand a
// gfx9 VALU instruction may read, so every constant is first copied into a vector register.
PIK_HD double clamp_lim(double v, double lo, double hi) {
    const double r = fmin(fmax(v, lo), hi);
    return (v != v) ? v : r;
}
template <int D>
PIK_HD double clamp_joint(CK<D> c, int j, double v) {
#if !defined(PIK_STRICT)
    return clamp_lim(v, c.clo[j], c.chi[j]);
#else
    // The literal form -- lo = bounded ? qmin : v - half_span, hi = bounded ? qmax : v + half_span,
    // (v < lo) ? lo : (hi < v) ? hi : v -- with the limits of an unbounded variable replaced by -inf / +inf (clo /
    // chi): v < v - half_span and v + half_span < v are false for every v (half_span = pi > 0; infinities and NaN
    // compare false as well), and so are v < -inf and +inf < v -- the same value in every case, without the two
    // selects and two additions per variable that only ever produced a comparison that fails.
    const double lo = c.clo[j], hi = c.chi[j];
    return (v < lo) ? lo : (hi < v) ? hi : v;
#endif
}

// ------------------------------------------------------------------------------------------
// One evaluation of a joint vector: cost_fn (src/goal.cpp:188-203) AND the solution_fn verdict
// (src/goal.cpp:163-186) from the same forward kinematics.  Carrying the verdict with every
// fitness value removes the separate FK the reference spends on solution_fn(best) each
// generation; the values are identical because both closures call the same fk(q).
// ------------------------------------------------------------------------------------------
struct EvalOut {
    double cost;
    double pc;         // the pose-cost part of `cost` (all tips), before the joint goals are added
    double lin, ang;   // linear / angular distance goal <-> tip
    double g0, g1, g2; // unweighted joint-goal sums (centre, avoid limits, minimal displacement)
    double vn;         // |vec(d0)| of the relative quaternion d0 = q_tip * conj(q_goal)
    bool sol;
};

// everything of an evaluation after the forward kinematics: pose cost, frame tests, joint goals
// (shared by the one-lane evaluation below and the cooperative one of the wide kernels)
#if defined(PIK_STRICT)
// (exact flavours) PROBE: the joint vector is q + dh e_jsel, a finite-difference probe.  It is only formed where it
// is read, by the joint goals -- built by the caller it cost ~45 scalar instructions per evaluation with no goal on.
// NG: the caller knows that no joint goal is on (PIK_GM(p) == 0) -- a compile-time fact instead of a wave-uniform branch
// around code whose operands (the joint vector, the seed, three weights) then need no registers
template <int D, bool PROBE = false, bool NG = false>
PIK_HD void pose_tail(CK<D> c_in, PK p_in, const GoalK& g, const double (&seed)[D], const double (&q_in)[D],
                      const double (&R)[9], const double (&tipt)[3], EvalOut& e, double (&d0)[4], int jsel = -1,
                      double dh = 0.0) {
#else
template <int D>
PIK_HD void pose_tail(CK<D> c_in, PK p_in, const GoalK& g, const double (&seed)[D], const double (&q)[D],
                      const double (&R)[9], const double (&tipt)[3], EvalOut& e, double (&d0)[4]) {
#endif
    // constants of the cost phase: (re)loaded here, behind the forward kinematics, so that they are
    // not hoisted out of the solver loops and parked in spilled scalar registers (one v_readlane
    // per use); the loads land during the square roots / divide below
    PK p = fresh_after(p_in, tipt[0]);
    CK<D> c = fresh_after(c_in, tipt[1]);
    const double dx = g.t[0] - tipt[0], dy = g.t[1] - tipt[1], dz = g.t[2] - tipt[2];
#if PIK_XF
    e.lin = sqrt_pos(xsumsq3(dx, dy, dz));
#else
    e.lin = sqrt_pos(dx * dx + dy * dy + dz * dz);
#endif
    double qt[4];
    matrix_to_quat(R, qt);
    quat_mul_conj(qt, g.q, d0);
    e.ang = angle_of(c.mt, d0, e.vn);
    PoseErr pe;
    pe.lin = e.lin;
    pe.ang = e.ang;
    double cost = pose_cost(p, pe);
    e.pc = cost;
    bool ok = (!PIK_POS_TEST(p) || e.lin <= p.pos_thr) && (!PIK_ORI_TEST(p) || fabs(e.ang) <= p.ori_thr);
    e.g0 = e.g1 = e.g2 = 0.0;
#if defined(PIK_STRICT)
    if (!NG && PIK_GM(p)) {
#else
    if (PIK_GM(p)) {
#endif
#if defined(PIK_STRICT)
        double q[D];
#pragma unroll
        for (int k = 0; k < D; ++k) q[k] = PROBE ? q_in[k] + ((k == jsel) ? dh : 0.0) : q_in[k];
#endif
        double gc = 0.0;
        if (PIK_GM(p) & 1) {
            e.g0 = goal_cost_term<D>(c, p, 0, q, seed);
            const double w = e.g0 * p.w_center_sq;
            gc = gc + w;
            ok = ok && (w < p.cost_thr_sq);
        }
        if (PIK_GM(p) & 2) {
            e.g1 = goal_cost_term<D>(c, p, 1, q, seed);
            const double w = e.g1 * p.w_limits_sq;
            gc = gc + w;
            ok = ok && (w < p.cost_thr_sq);
        }
        if (PIK_GM(p) & 4) {
            e.g2 = goal_cost_term<D>(c, p, 2, q, seed);
            const double w = e.g2 * p.w_disp_sq;
            gc = gc + w;
            ok = ok && (w < p.cost_thr_sq);
        }
        cost = cost + gc;
    }
    e.cost = cost;
    e.sol = ok;
}

template <int D, bool WANT_FRAMES, bool FRJ0 = true>
PIK_HD void eval_pose(CK<D> c_in, PK p_in, const GoalK& g, const double (&seed)[D],
                      const double (&q)[D], EvalOut& e, double (&tipt)[3], double (&d0)[4], double* fr,
                      int stride) {
    double R[9];
    fk<D, WANT_FRAMES, false, FRJ0>(c_in, q, R, tipt, fr, stride);
    pose_tail<D>(c_in, p_in, g, seed, q, R, tipt, e, d0);
}
// ... with the sine / cosine exchange of fk_dh_joints (SCM 1: export at q; 2: q is qb plus a small step)
template <int D, bool WANT_FRAMES, bool FRJ0, int SCM>
PIK_HD void eval_pose_sc(CK<D> c_in, PK p_in, const GoalK& g, const double (&seed)[D],
                         const double (&q)[D], EvalOut& e, double (&tipt)[3], double (&d0)[4], double* fr,
                         int stride, const double (&qb)[D], double (&bsn)[D], double (&bcs)[D]) {
    double R[9];
    fk<D, WANT_FRAMES, false, FRJ0, SCM>(c_in, q, R, tipt, fr, stride, qb, bsn, bcs);
    pose_tail<D>(c_in, p_in, g, seed, q, R, tipt, e, d0);
}

// The 2D central-difference probes of step() (src/ik_gradient.cpp:28-43) without 2D forward
// kinematics.  c(q +- h e_j) only moves joint j, i.e. it rotates the tip by +-h about joint j's
// world axis a through its world origin o (or translates it by +-h a for a prismatic joint):
//     t(+-)  = t + (+-sin h) (a x r) + (1 - cos h) (a (a.r) - r),          r = t - o
//     d(+-)  = (cos h/2, +-sin h/2 a) * d0,   d0 = q_tip * conj(q_goal)    (relative quaternion)
// with (a, o) per joint, t and d0 taken from the evaluation of q itself.
//
// probe_joint returns g_j = c(q + h e_j) - c(q - h e_j) for ONE joint whose data is passed by
// value, so the joint index may be per-lane (several lanes sharing the probes of one elite).
// The difference is formed term by term instead of subtracting two evaluated costs:
//   position   |dp|^2 - |dm|^2 = 4 su (mid . u)            with dp/dm = mid +- su u
//   rotation   the half angles alpha(+-) = atan2(|v(+-)|, |w(+-)|) differ from the base alpha0 by
//              delta(+-) = asin((|v(+-)| |w0| - |v0| |w(+-)|) / |d0|^2)   (sine of a difference;
//              |delta| <= h/2, so the arcsine is a 4-term series), and
//              (2 rs alpha+)^2 - (2 rs alpha-)^2 = 4 rs^2 (delta+ - delta-)(2 alpha0 + delta+ + delta-)
//   joint goals only joint j's term changes: w (tp^2 - tm^2)
// Same mathematics as the literal central difference, no cancellation of two O(1) costs, and no
// atan2 / divide per joint (the literal form needs 14 of each per step at D = 7).
struct JointGoalConsts {
    double qmin, qmax, mid, hspan, mdf, seed;
    bool bounded;
};

// per-evaluation quantities shared by the D probes
struct ProbeBase {
    double dt0[3];  // tip - goal translation
    double aw0;     // |w(d0)|
    double vn2;     // |vec(d0)|^2
    double inv_n2;  // 1 / |d0|^2
};

PIK_HD void make_probe_base(const GoalK& g, const double (&tipt)[3], const double (&d0)[4],
                            const EvalOut& base, ProbeBase& b) {
    b.dt0[0] = tipt[0] - g.t[0];
    b.dt0[1] = tipt[1] - g.t[1];
    b.dt0[2] = tipt[2] - g.t[2];
    b.aw0 = fabs(d0[0]);
    b.vn2 = d0[1] * d0[1] + d0[2] * d0[2] + d0[3] * d0[3]; // (the sum base.vn is the root of)
    // |d0| = |q_goal| (the tip quaternion is unit): 1 unless the caller passed a goal quaternion
    // that is not normalised; 1/n2 = 2 - n2 to rounding in the normal case, a divide otherwise
    // (per-lane choice: a lane's result does not depend on its neighbours)
    const double n2 = d0[0] * d0[0] + base.vn * base.vn;
    const double dev = 1.0 - n2;
    b.inv_n2 = (fabs(dev) < 1.0e-8) ? (1.0 + dev) : (1.0 / n2);
}

PIK_HD double probe_joint(PK p, const EvalOut& base, const ProbeBase& pb, const double (&tipt)[3],
                          const double (&d0)[4], const double (&a)[3], const double (&o)[3],
                          bool prismatic, double qj, const JointGoalConsts& jc,
                          bool with_pose = true, bool with_goals = true) {
    const double h = p.step_size;
    double diff = 0.0;
    // position part: |dp|^2 - |dm|^2 = 4 su (mid . u) with mid = dt0 + sw wv, u = a x r, wv = a (a . r) - r.
    // wv is orthogonal to u (a x r is orthogonal to both a and r), so mid . u = dt0 . u: the triple
    // product dt0 . (a x r), and the second-order displacement never has to be formed.
    if (with_pose && PIK_POS_ON(p)) {
        double u[3];
        if (prismatic) {
#pragma unroll
            for (int i = 0; i < 3; ++i) u[i] = a[i];
        } else {
            const double r[3] = {tipt[0] - o[0], tipt[1] - o[1], tipt[2] - o[2]};
            u[0] = a[1] * r[2] - a[2] * r[1];
            u[1] = a[2] * r[0] - a[0] * r[2];
            u[2] = a[0] * r[1] - a[1] * r[0];
        }
        const double su = prismatic ? h : p.sin_h;
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) acc += pb.dt0[i] * u[i];
        diff = 4.0 * su * acc * (p.pos_scale * p.pos_scale);
    }
    // orientation part.  d(+-) = (c, +-s a) * d0 with c = cos h/2, s = sin h/2, d0 = (w0, v0), A = a . v0:
    //   w(+-)     = c w0 -+ s A
    //   |v(+-)|^2 = |v0|^2 + s^2 (w0^2 - A^2) +- 2 c s w0 A          (a is a unit vector, c^2 + s^2 = 1;
    //               v0 . (a x v0) = 0, |a x v0|^2 = |v0|^2 - A^2)
    // i.e. the leading term plus two small corrections (s^2 = (1 - cos h) / 2, 2 c s = sin h) -- no
    // vector v(+-) to build, and no cancellation: the corrections carry their own relative precision.
    if (with_pose && PIK_ROT_ON(p)) {
        const double sh2 = prismatic ? 0.0 : p.sin_h2;
        const double ch2 = prismatic ? 1.0 : p.cos_h2;
        const double s2 = prismatic ? 0.0 : 0.5 * p.vers_h;
        const double cs2 = prismatic ? 0.0 : p.sin_h;
        const double w0 = d0[0];
        const double A = a[0] * d0[1] + a[1] * d0[2] + a[2] * d0[3];
        const double cw = ch2 * w0;
        const double wp = cw - sh2 * A, wm = cw + sh2 * A;
        const double k2 = pb.vn2 + s2 * ((w0 - A) * (w0 + A));
        const double l2 = cs2 * (w0 * A);
        const double vp2 = k2 + l2, vm2 = k2 - l2;
        const double sp = (sqrt_pos(vp2) * pb.aw0 - base.vn * fabs(wp)) * pb.inv_n2;
        const double sm = (sqrt_pos(vm2) * pb.aw0 - base.vn * fabs(wm)) * pb.inv_n2;
        // asin x = x + x^3/6 + 3x^5/40 + 15x^7/336 (|x| <= sin(h/2): the next term is < 1e-16
        // relative for every step size up to 1e-2)
        const double sp2 = sp * sp, sm2 = sm * sm;
        const double dp = sp + sp * sp2 * (1.0 / 6.0 + sp2 * (3.0 / 40.0 + sp2 * (15.0 / 336.0)));
        const double dm = sm + sm * sm2 * (1.0 / 6.0 + sm2 * (3.0 / 40.0 + sm2 * (15.0 / 336.0)));
        const double rs = p.rot_scale;
        diff += 4.0 * (rs * rs) * (dp - dm) * (base.ang + (dp + dm)); // base.ang = 2 alpha0
    }
    if (with_goals && PIK_GM(p)) {
        // only joint j's term of each joint goal changes
        const double qp = qj + h, qm = qj - h;
        if (PIK_GM(p) & 1) {
#if defined(PIK_STRICT)
            const double mid = (jc.qmin + jc.qmax) * 0.5, m = jc.bounded ? jc.mdf : 0.0;
#else
            const double mid = jc.mid, m = jc.bounded ? jc.mdf : 0.0; // (ChainK::mid is this midpoint)
#endif
            const double tp = (qp - mid) * m, tm = (qm - mid) * m;
            diff += (tp * tp - tm * tm) * p.w_center_sq;
        }
        if (PIK_GM(p) & 2) {
            const double m = jc.bounded ? jc.mdf : 0.0;
            const double tp = fmax(0.0, fabs(qp - jc.mid) * 2.0 - jc.hspan) * m;
            const double tm = fmax(0.0, fabs(qm - jc.mid) * 2.0 - jc.hspan) * m;
            diff += (tp * tp - tm * tm) * p.w_limits_sq;
        }
        if (PIK_GM(p) & 4) {
            const double tp = (qp - jc.seed) * jc.mdf, tm = (qm - jc.seed) * jc.mdf;
            diff += (tp * tp - tm * tm) * p.w_disp_sq;
        }
    }
    return diff;
}

// all D probes by one lane (joint index known at compile time).  FRJ0 = false: the rows hold the
// frames of joints 1 .. D-1 and the first joint's frame is the chain constant (see fk_dh_joints)
template <int D, bool FRJ0 = true>
PIK_HD void probe_gradient(CK<D> c_in, PK p_in, const GoalK& g, const double (&seed)[D],
                           const double (&q)[D], const EvalOut& base, const double (&tipt)[3],
                           const double (&d0)[4], const double* fr, int stride,
                           double (&grad)[D]) {
    PK p = fresh_after(p_in, base.cost); // probe constants: one scalar load, not hoisted + spilled
    ProbeBase pb;
    make_probe_base(g, tipt, d0, base, pb);
    const uint32_t prismatic_mask = PIK_PRISMATIC(c_in), bounded_mask = c_in.bounded_mask;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double a[3], o[3];
        if (!FRJ0 && j == 0) {
            CK<D> cb = fresh_after(c_in, base.cost);
            a[0] = cb.dh_base[2]; a[1] = cb.dh_base[5]; a[2] = cb.dh_base[8];
            o[0] = cb.dh_base[9]; o[1] = cb.dh_base[10]; o[2] = cb.dh_base[11];
        } else {
            const int row0 = 6 * (FRJ0 ? j : j - 1);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                a[i] = fr[(row0 + i) * stride];
                o[i] = fr[(row0 + 3 + i) * stride];
            }
        }
        JointGoalConsts jc;
        jc.qmin = jc.qmax = jc.mid = jc.hspan = jc.mdf = jc.seed = 0.0;
        jc.bounded = (bounded_mask >> j) & 1u;
        if (PIK_GM(p)) {
            CK<D> c = fresh(c_in);
            jc.qmin = c.qmin[j];
            jc.qmax = c.qmax[j];
            jc.mid = c.mid[j];
            jc.hspan = c.hspan[j];
            jc.mdf = c.mdf[j];
            jc.seed = seed[j];
        }
        grad[j] = probe_joint(p, base, pb, tipt, d0, a, o, (prismatic_mask >> j) & 1u, q[j], jc);
    }
}

// goal pose message -> GoalK.  tf2::fromMsg: Translation * Quaterniond(w,x,y,z) (not normalised)
// -> goal frame matrix; angular_distance then re-derives the quaternion from that matrix
// (src/goal.cpp:22-23).
PIK_HD void make_goal(const double* g7, GoalK& g) {
    g.t[0] = g7[0];
    g.t[1] = g7[1];
    g.t[2] = g7[2];
    const double q[4] = {g7[3], g7[4], g7[5], g7[6]};
    double R[9];
    quat_to_matrix(q, R);
    matrix_to_quat(R, g.q);
}

// ------------------------------------------------------------------------------------------
// Several tip frames: cost = sum_k pose_cost_k(FK_k(q)) + joint goals, verdict = every frame test
// and every goal test (make_cost_fn / make_is_solution_test_fn, src/goal.cpp:163-203, with the
// vectors make_pose_cost_functions / make_frame_tests build, :27-49, 80-89).  One padded D-joint
// chain per tip (see GoalSet).  WANT_GRAD (fast build): also the central-difference gradient of
// step() from each tip's joint frames -- a joint contributes the probes of every tip it moves.
// ------------------------------------------------------------------------------------------
template <int D, bool WANT_GRAD>
PIK_HD void eval_multi(CK<D> c0, PK p_in, const GoalSet& gs, const double (&seed)[D],
                       const double (&q)[D], EvalOut& e, double* fr, int stride,
                       double (&grad)[D]) {
    const int n_tips = tip_count<D>(c0);
    double pc = 0.0;
    bool ok = true;
    if (WANT_GRAD) {
#pragma unroll
        for (int j = 0; j < D; ++j) grad[j] = 0.0;
    }
    e.lin = e.ang = e.vn = 0.0;
#pragma unroll 1
    for (int k = 0; k < n_tips; ++k) { // wave-uniform trip count
        {
            CK<D> ck = tip_chain<D>(c0, k);
            double R[9], tipt[3], d0[4];
            fk<D, WANT_GRAD, true>(ck, q, R, tipt, fr, stride);
            PK p = fresh_after(p_in, tipt[0]);
            GoalK g;
            make_goal(gs.ptr + 7 * k, g);
            const double dx = g.t[0] - tipt[0], dy = g.t[1] - tipt[1], dz = g.t[2] - tipt[2];
            EvalOut ek;
#if PIK_XF
            ek.lin = sqrt_pos(xsumsq3(dx, dy, dz));
#else
            ek.lin = sqrt_pos(dx * dx + dy * dy + dz * dz);
#endif
            double qt[4];
            matrix_to_quat(R, qt);
            quat_mul_conj(qt, g.q, d0);
            ek.ang = angle_of(ck.mt, d0, ek.vn);
            PoseErr pe;
            pe.lin = ek.lin;
            pe.ang = ek.ang;
            pc = pc + pose_cost(p, pe);
            ok = ok && (!PIK_POS_TEST(p) || ek.lin <= p.pos_thr) &&
                 (!PIK_ORI_TEST(p) || fabs(ek.ang) <= p.ori_thr);
#if !defined(PIK_STRICT)
            if (WANT_GRAD) {
                ek.g0 = ek.g1 = ek.g2 = 0.0;
                ProbeBase pb;
                make_probe_base(g, tipt, d0, ek, pb);
                const uint32_t prismatic_mask = PIK_PRISMATIC(ck), active_mask = ck.active_mask;
                JointGoalConsts jc;
                jc.qmin = jc.qmax = jc.mid = jc.hspan = jc.mdf = jc.seed = 0.0;
                jc.bounded = false;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const double a[3] = {fr[(6 * j + 0) * stride], fr[(6 * j + 1) * stride], fr[(6 * j + 2) * stride]};
                    const double o[3] = {fr[(6 * j + 3) * stride], fr[(6 * j + 4) * stride], fr[(6 * j + 5) * stride]};
                    const double dj = probe_joint(p, ek, pb, tipt, d0, a, o, (prismatic_mask >> j) & 1u,
                                                  q[j], jc, true, false);
                    grad[j] += ((active_mask >> j) & 1u) ? dj : 0.0;
                }
            }
#endif
        }
    }
    PK p = fresh_after(p_in, pc);
    CK<D> c = fresh_after(c0, pc);
    double cost = pc;
    e.pc = pc;
    e.g0 = e.g1 = e.g2 = 0.0;
    if (PIK_GM(p)) {
        double gc = 0.0;
        if (PIK_GM(p) & 1) {
            e.g0 = goal_cost_term<D>(c, p, 0, q, seed);
            const double w = e.g0 * p.w_center_sq;
            gc = gc + w;
            ok = ok && (w < p.cost_thr_sq);
        }
        if (PIK_GM(p) & 2) {
            e.g1 = goal_cost_term<D>(c, p, 1, q, seed);
            const double w = e.g1 * p.w_limits_sq;
            gc = gc + w;
            ok = ok && (w < p.cost_thr_sq);
        }
        if (PIK_GM(p) & 4) {
            e.g2 = goal_cost_term<D>(c, p, 2, q, seed);
            const double w = e.g2 * p.w_disp_sq;
            gc = gc + w;
            ok = ok && (w < p.cost_thr_sq);
        }
        cost = cost + gc;
#if !defined(PIK_STRICT)
        if (WANT_GRAD) {
            // joint-goal part of the probes, once per variable
            const uint32_t bounded_mask = c.bounded_mask;
            ProbeBase pb0;
            pb0.dt0[0] = pb0.dt0[1] = pb0.dt0[2] = 0.0;
            pb0.aw0 = 0.0;
            pb0.vn2 = 0.0;
            pb0.inv_n2 = 1.0;
            const double z3[3] = {0.0, 0.0, 0.0}, z4[4] = {1.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < D; ++j) {
                JointGoalConsts jc;
                jc.bounded = (bounded_mask >> j) & 1u;
                jc.qmin = c.qmin[j];
                jc.qmax = c.qmax[j];
                jc.mid = c.mid[j];
                jc.hspan = c.hspan[j];
                jc.mdf = c.mdf[j];
                jc.seed = seed[j];
                grad[j] += probe_joint(p, e, pb0, z3, z4, z3, z3, false, q[j], jc, false, true);
            }
        }
#endif
    }
    e.cost = cost;
    e.sol = ok;
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based RNG (Salmon et al., SC'11); stream/slot layout documented in
// DESIGN.md "Random streams" and mirrored by the oracle.
// ------------------------------------------------------------------------------------------
struct U4 {
    uint32_t x, y, z, w;
};

// One 32x32 -> 64 bit product per Philox multiplier (a single v_mad_u64_u32 on gfx950) instead of
// a mul_hi + mul_lo pair: the 32-bit integer multiplier is quarter rate, so this halves the cost
// of the generator, which is ~1/3 of the child-generation phase.
PIK_HD U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                        uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0;
        c1 = (uint32_t)p1;
        c2 = n2;
        c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

constexpr uint32_t STREAM_INIT = 1u;
constexpr uint32_t STREAM_REPRODUCE = 2u;

PIK_HD U4 rng_block(uint64_t seed, uint32_t stream, uint64_t problem, uint32_t epoch,
                    uint32_t individual, uint32_t block) {
    return philox4x32_10(block, individual, epoch, (uint32_t)problem, (uint32_t)seed ^ stream,
                         (uint32_t)(seed >> 32) + (uint32_t)(problem >> 32));
}

PIK_HD double u01_from_words(uint32_t lo, uint32_t hi) {
    const uint64_t x = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)x * (1.0 / 9007199254740992.0);
}
// w / 2^32: built from the bits (1 + w 2^-32 is exact in binary64, so is the subtraction) -- the same
// value as (double)w * 2^-32 without the quarter-rate integer-to-double conversion
PIK_HD double u01_from_word(uint32_t w) {
    uint64_t u = 0x3FF0000000000000ull | ((uint64_t)w << 20);
    double d;
    __builtin_memcpy(&d, &u, sizeof d);
    return d - 1.0;
}
PIK_HD double uniform_real(double a, double b, double u) { return (b - a) * u + a; }

} // namespace pik
