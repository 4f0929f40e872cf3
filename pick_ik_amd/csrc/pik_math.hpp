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

// Serial chain, base -> tip (wave-uniform).
template <int D>
struct ChainK {
    double O[D][12];   // joint origin transform: rotation row-major [0..8], translation [9..11]
    double axis[D][3]; // normalised joint axis in the joint frame
    double tip[12];    // fixed transform after the last joint
    double qmin[D], qmax[D], mid[D], hspan[D], mdf[D];
    uint32_t origin_ident_mask; // bit j: origin transform is exactly the identity
    uint32_t prismatic_mask;    // bit j
    uint32_t bounded_mask;      // bit j
    uint32_t axis_kind;         // 2 bits per joint: AxisKind (only exact +x/+y/+z are specialised)
    uint32_t tip_ident;
    uint32_t pad_;
};

// Solver parameters (wave-uniform), derived from pikamd_params on the host.
struct ParamsK {
    double step_size;
    double min_cost_delta;
    double pos_thr, ori_thr;
    double cost_thr_sq;
    double pos_scale, rot_scale;
    double w_center_sq, w_limits_sq, w_disp_sq; // weight^2 (0 = goal disabled)
    double wipeout_tol;
    // rotation by the finite-difference step h about a joint axis (gradient probes):
    // sin h, 1 - cos h, sin h/2, cos h/2
    double sin_h, vers_h, sin_h2, cos_h2;
    int32_t has_pos_thr, has_ori_thr;
    int32_t goal_mask; // bit0 center, bit1 avoid limits, bit2 minimal displacement
    int32_t stop_on_valid;
    int32_t approx;
    int32_t population, elites;
    int32_t max_generations, gd_max_iters;
    int32_t local_max_iters;
    int32_t pad_;
};

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
#if defined(__HIP_DEVICE_COMPILE__)
    const PIK_CONSTANT T* p = &r;
    asm volatile("" : "+s"(p));
    return *p;
#else
    return r;
#endif
}

// Per-problem goal: translation + the goal frame's quaternion as the reference derives it
// (tf2::fromMsg pose -> matrix, then Eigen matrix -> quaternion inside angular_distance).
struct GoalK {
    double t[3];
    double q[4]; // w x y z
};

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
    const double t = sqrt(arg);
    const double h = 0.5 * t;
    const double r = 0.5 / t;
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
            r[i * 3 + j] = R[i * 3 + 0] * o[0 * 3 + j] + R[i * 3 + 1] * o[1 * 3 + j] +
                           R[i * 3 + 2] * o[2 * 3 + j];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        t[i] = R[i * 3 + 0] * o[9] + R[i * 3 + 1] * o[10] + R[i * 3 + 2] * o[11] + t[i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = r[i];
}

PIK_HD double fma_f64(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fma(a, b, c);
#else
    return ::fma(a, b, c);
#endif
}

// sin and cos of a joint angle.  Replaces libm's sin/cos (what MoveIt's
// RevoluteJointModel::computeTransform calls): Cody-Waite reduction by pi/2 with a three-double
// split and FMAs (exact to < 1 ulp of the reduced argument for |x| < ~1e5 rad, which covers every
// joint range; larger magnitudes are first folded by 2 pi), then the fdlibm minimax kernels on
// [-pi/4, pi/4].  No table, no stack array, no divergence: ~35 FP64 instructions versus the ~80
// plus scratch of the generic large-argument routine.
PIK_HD void sincos_f64(double x, double& s, double& c) {
    if (fabs(x) > 65536.0) {
        const double k = rint(x * 0.15915494309189535);
        x = fma_f64(-k, 6.283185307179586, x);
        x = fma_f64(-k, 2.4492935982947064e-16, x);
    }
    const double fn = rint(x * 0.6366197723675814);
    const int n = (int)fn;
    double t = fma_f64(-fn, 1.5707963267948966, x);
    t = fma_f64(-fn, 6.123233995736766e-17, t);
    t = fma_f64(-fn, -1.4973849048591698e-33, t);
    const double z = t * t;
    // fdlibm __kernel_sin / __kernel_cos coefficients
    const double rs = 8.33333333332248946124e-03 +
                      z * (-1.98412698298579493134e-04 +
                           z * (2.75573137070700676789e-06 +
                                z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double sn = t + (z * t) * (-1.66666666666666324348e-01 + z * rs);
    const double rc = z * (4.16666666666666019037e-02 +
                           z * (-1.38888888888741095749e-03 +
                                z * (2.48015872894767294178e-05 +
                                     z * (-2.75573143513906633035e-07 +
                                          z * (2.08757232129817482790e-09 +
                                               z * -1.13596475577881948265e-11)))));
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cn = w + (((1.0 - w) - hz) + z * rc);
    const double a = (n & 1) ? cn : sn;
    const double b = (n & 1) ? sn : cn;
    s = (n & 2) ? -a : a;
    c = ((n + 1) & 2) ? -b : b;
}

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
                r[i * 3 + j] = R[i * 3 + 0] * J[j] + R[i * 3 + 1] * J[3 + j] +
                               R[i * 3 + 2] * J[6 + j];
            }
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = r[i];
    }
}

// Forward kinematics of the serial chain.  When WANT_FRAMES, also stores for every joint j the
// world-frame joint axis (rows 6j..6j+2) and joint origin (rows 6j+3..6j+5) into `fr` with element
// stride `stride` (on the GPU: one LDS column per lane, stride 64) -- the line a revolute joint
// rotates the tip about / the direction a prismatic joint moves it along.  The gradient probes of
// the fast step are built from these frames (the idea behind the reference's CachedJointFrames,
// src/forward_kinematics.cpp:102-125: a joint perturbation only moves that joint's frame).
template <int D, bool WANT_FRAMES>
PIK_HD void fk(CK<D> c_in, const double (&q)[D], double (&R)[9], double (&t)[3], double* fr,
               int stride) {
    R[0] = 1.0; R[1] = 0.0; R[2] = 0.0;
    R[3] = 0.0; R[4] = 1.0; R[5] = 0.0;
    R[6] = 0.0; R[7] = 0.0; R[8] = 1.0;
    t[0] = t[1] = t[2] = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        CK<D> c = fresh(c_in); // joint j's constants are (re)loaded here, not hoisted
        if (!((c.origin_ident_mask >> j) & 1u)) {
            if (j == 0) {
#pragma unroll
                for (int i = 0; i < 9; ++i) R[i] = c.O[0][i];
                t[0] = c.O[0][9]; t[1] = c.O[0][10]; t[2] = c.O[0][11];
            } else {
                iso_mul(R, t, c.O[j]);
            }
        }
#if defined(PIK_STRICT)
        // strict-arithmetic build: always the generic Rodrigues product, operation for operation
        // what the oracle (and MoveIt) compute, so results are bit-identical to the CPU
        const uint32_t kind = AXIS_GENERAL;
#else
        const uint32_t kind = (c.axis_kind >> (2 * j)) & 3u;
#endif
        CPtr a = c.axis[j];
        if (WANT_FRAMES) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                fr[(6 * j + i) * stride] = R[i * 3 + 0] * a[0] + R[i * 3 + 1] * a[1] + R[i * 3 + 2] * a[2];
                fr[(6 * j + 3 + i) * stride] = t[i];
            }
        }
        if ((c.prismatic_mask >> j) & 1u) {
            const double v = q[j];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                t[i] = R[i * 3 + 0] * (a[0] * v) + R[i * 3 + 1] * (a[1] * v) +
                       R[i * 3 + 2] * (a[2] * v) + t[i];
            }
        } else {
            double sn, cs;
            sincos_f64(q[j], sn, cs);
            rotate_about(R, kind, a, sn, cs);
        }
    }
    CK<D> ct = fresh(c_in);
    if (!ct.tip_ident) iso_mul(R, t, ct.tip);
}

// d = a * conj(b)  (Eigen quaternion product), quaternions as w x y z
PIK_HD void quat_mul_conj(const double (&a)[4], const double (&b)[4], double (&d)[4]) {
    const double aw = a[0], ax = a[1], ay = a[2], az = a[3];
    const double bw = b[0], bx = -b[1], by = -b[2], bz = -b[3];
    d[0] = aw * bw - ax * bx - ay * by - az * bz;
    d[1] = aw * bx + ax * bw + ay * bz - az * by;
    d[2] = aw * by + ay * bw + az * bx - ax * bz;
    d[3] = aw * bz + az * bw + ax * by - ay * bx;
}

// atan2(y, x) for y >= 0, x >= 0 -- the only way the path uses it (Eigen angularDistance).
// fdlibm's atan scheme (breakpoints 7/16, 11/16, 19/16, 39/16; odd minimax polynomial; hi/lo
// table) with the interval reduction applied to the (y, x) pair so that a single divide serves
// both the quotient and the reduction; selects only, no divergence.  <= 1 ulp from libm.
PIK_HD double atan2_pos(double y, double x) {
    const double y16 = 16.0 * y;
    const bool c0 = y16 < 7.0 * x, c1 = y16 < 11.0 * x, c2 = y16 < 19.0 * x, c3 = y16 < 39.0 * x;
    const double num = c0 ? y : c1 ? (2.0 * y - x) : c2 ? (y - x) : c3 ? (y - 1.5 * x) : -x;
    const double den = c0 ? x : c1 ? (2.0 * x + y) : c2 ? (y + x) : c3 ? (x + 1.5 * y) : y;
    const double hi = c0   ? 0.0
                      : c1 ? 4.63647609000806093515e-01
                      : c2 ? 7.85398163397448278999e-01
                      : c3 ? 9.82793723247329054082e-01
                           : 1.57079632679489655800e+00;
    const double lo = c0   ? 0.0
                      : c1 ? 2.26987774529616870924e-17
                      : c2 ? 3.06161699786838301793e-17
                      : c3 ? 1.39033110312309984516e-17
                           : 6.12323399573676603587e-17;
    const double r = num / den;
    const double z = r * r;
    const double w = z * z;
    const double s1 =
        z * (3.33333333333329318027e-01 +
             w * (1.42857142725034663711e-01 +
                  w * (9.09088713343650656196e-02 +
                       w * (6.66107313738753120669e-02 +
                            w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    const double s2 = w * (-1.99999999998764832476e-01 +
                           w * (-1.11111104054623557880e-01 +
                                w * (-7.69187620504482999495e-02 +
                                     w * (-5.83357013379057348645e-02 +
                                          w * -3.65315727442169155270e-02))));
    const double res = c0 ? (r - r * (s1 + s2)) : (hi - ((r * (s1 + s2) - lo) - r));
    return (y == 0.0) ? 0.0 : res;
}

// Eigen angularDistance from the relative quaternion: 2 atan2(|vec|, |w|)
PIK_HD double angle_of(const double (&d)[4]) {
    return 2.0 * atan2_pos(sqrt(d[1] * d[1] + d[2] * d[2] + d[3] * d[3]), fabs(d[0]));
}

struct PoseErr {
    double lin; // linear_distance(goal, frame)
    double ang; // angular_distance(goal, frame)
};

PIK_HD PoseErr pose_error(const GoalK& g, const double (&R)[9], const double (&t)[3]) {
    PoseErr e;
    const double dx = g.t[0] - t[0], dy = g.t[1] - t[1], dz = g.t[2] - t[2];
    e.lin = sqrt(dx * dx + dy * dy + dz * dz);
    double qt[4], d[4];
    matrix_to_quat(R, qt);
    quat_mul_conj(qt, g.q, d);
    e.ang = angle_of(d);
    return e;
}

// make_pose_cost_fn -- src/goal.cpp:51-78 (terms dropped when the scale is <= 0)
PIK_HD double pose_cost(PK p, const PoseErr& e) {
    double c = 0.0;
    if (p.pos_scale > 0.0) {
        const double a = e.lin * p.pos_scale;
        c = a * a;
    }
    if (p.rot_scale > 0.0) {
        const double a = e.ang * p.rot_scale;
        c = c + a * a;
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
        sum += v * v;
    }
    return sum;
}

template <int D>
PIK_HD double goals_cost(CK<D> c, PK p, const double (&q)[D],
                         const double (&seed)[D]) {
    double gc = 0.0;
    if (p.goal_mask & 1) gc = gc + goal_cost_term<D>(c, p, 0, q, seed) * p.w_center_sq;
    if (p.goal_mask & 2) gc = gc + goal_cost_term<D>(c, p, 1, q, seed) * p.w_limits_sq;
    if (p.goal_mask & 4) gc = gc + goal_cost_term<D>(c, p, 2, q, seed) * p.w_disp_sq;
    return gc;
}

// make_cost_fn -- src/goal.cpp:188-203
template <int D>
PIK_HD double cost_fn(CK<D> c, PK p, const GoalK& g, const double (&seed)[D],
                      const double (&q)[D]) {
    double R[9], t[3];
    fk<D, false>(c, q, R, t, nullptr, 0);
    const PoseErr e = pose_error(g, R, t);
    double cost = pose_cost(p, e);
    if (p.goal_mask) cost = cost + goals_cost<D>(c, p, q, seed);
    return cost;
}

// make_is_solution_test_fn -- src/goal.cpp:163-186
template <int D>
PIK_HD bool solution_fn(CK<D> c, PK p, const GoalK& g,
                        const double (&seed)[D], const double (&q)[D]) {
    double R[9], t[3];
    fk<D, false>(c, q, R, t, nullptr, 0);
    const PoseErr e = pose_error(g, R, t);
    bool ok = (!p.has_pos_thr || e.lin <= p.pos_thr) && (!p.has_ori_thr || fabs(e.ang) <= p.ori_thr);
    if (p.goal_mask & 1) ok = ok && (goal_cost_term<D>(c, p, 0, q, seed) * p.w_center_sq < p.cost_thr_sq);
    if (p.goal_mask & 2) ok = ok && (goal_cost_term<D>(c, p, 1, q, seed) * p.w_limits_sq < p.cost_thr_sq);
    if (p.goal_mask & 4) ok = ok && (goal_cost_term<D>(c, p, 2, q, seed) * p.w_disp_sq < p.cost_thr_sq);
    return ok;
}

// Variable::clamp_to_limits -- src/robot.cpp:36-42
template <int D>
PIK_HD double clamp_joint(CK<D> c, int j, double v) {
    const bool bounded = (c.bounded_mask >> j) & 1u;
    const double lo = bounded ? c.qmin[j] : v - c.hspan[j];
    const double hi = bounded ? c.qmax[j] : v + c.hspan[j];
    return (v < lo) ? lo : (hi < v) ? hi : v;
}

// ------------------------------------------------------------------------------------------
// One evaluation of a joint vector: cost_fn (src/goal.cpp:188-203) AND the solution_fn verdict
// (src/goal.cpp:163-186) from the same forward kinematics.  Carrying the verdict with every
// fitness value removes the separate FK the reference spends on solution_fn(best) each
// generation; the values are identical because both closures call the same fk(q).
// ------------------------------------------------------------------------------------------
struct EvalOut {
    double cost;
    double lin, ang;   // linear / angular distance goal <-> tip
    double g0, g1, g2; // unweighted joint-goal sums (centre, avoid limits, minimal displacement)
    bool sol;
};

template <int D, bool WANT_FRAMES>
PIK_HD void eval_pose(CK<D> c, PK p, const GoalK& g, const double (&seed)[D], const double (&q)[D],
                      EvalOut& e, double (&tipt)[3], double (&d0)[4], double* fr, int stride) {
    double R[9];
    fk<D, WANT_FRAMES>(c, q, R, tipt, fr, stride);
    const double dx = g.t[0] - tipt[0], dy = g.t[1] - tipt[1], dz = g.t[2] - tipt[2];
    e.lin = sqrt(dx * dx + dy * dy + dz * dz);
    double qt[4];
    matrix_to_quat(R, qt);
    quat_mul_conj(qt, g.q, d0);
    e.ang = angle_of(d0);
    PoseErr pe;
    pe.lin = e.lin;
    pe.ang = e.ang;
    double cost = pose_cost(p, pe);
    bool ok = (!p.has_pos_thr || e.lin <= p.pos_thr) && (!p.has_ori_thr || fabs(e.ang) <= p.ori_thr);
    e.g0 = e.g1 = e.g2 = 0.0;
    if (p.goal_mask) {
        double gc = 0.0;
        if (p.goal_mask & 1) {
            e.g0 = goal_cost_term<D>(c, p, 0, q, seed);
            const double w = e.g0 * p.w_center_sq;
            gc = gc + w;
            ok = ok && (w < p.cost_thr_sq);
        }
        if (p.goal_mask & 2) {
            e.g1 = goal_cost_term<D>(c, p, 1, q, seed);
            const double w = e.g1 * p.w_limits_sq;
            gc = gc + w;
            ok = ok && (w < p.cost_thr_sq);
        }
        if (p.goal_mask & 4) {
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

// The 2D central-difference probes of step() (src/ik_gradient.cpp:28-43) without 2D forward
// kinematics.  c(q +- h e_j) only moves joint j, i.e. it rotates the tip by +-h about joint j's
// world axis a through its world origin o (or translates it by +-h a for a prismatic joint):
//     t(+-)  = t + (+-sin h) (a x r) + (1 - cos h) (a (a.r) - r),          r = t - o
//     d(+-)  = (cos h/2, +-sin h/2 a) * d0,   d0 = q_tip * conj(q_goal)    (relative quaternion)
// with (a, o) per joint, t and d0 taken from the evaluation of q itself.  Same mathematics as
// evaluating the cost at the perturbed joint vector, ~10x fewer FP64 instructions.
template <int D>
PIK_HD void probe_gradient(CK<D> c_in, PK p, const GoalK& g, const double (&seed)[D],
                           const double (&q)[D], const EvalOut& base, const double (&tipt)[3],
                           const double (&d0)[4], const double* fr, int stride,
                           double (&grad)[D]) {
    const double h = p.step_size;
    const double dt0[3] = {tipt[0] - g.t[0], tipt[1] - g.t[1], tipt[2] - g.t[2]};
    const double ps2 = p.pos_scale * p.pos_scale;
    const bool use_pos = p.pos_scale > 0.0, use_rot = p.rot_scale > 0.0;
    const double rot0 = base.ang * p.rot_scale;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        CK<D> c = fresh(c_in);
        const double a[3] = {fr[(6 * j + 0) * stride], fr[(6 * j + 1) * stride], fr[(6 * j + 2) * stride]};
        const double o[3] = {fr[(6 * j + 3) * stride], fr[(6 * j + 4) * stride], fr[(6 * j + 5) * stride]};
        double cp = 0.0, cm = 0.0; // cost at +h / -h
        if ((c.prismatic_mask >> j) & 1u) {
            if (use_pos) {
                double lp = 0.0, lm = 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double dp = dt0[i] + h * a[i], dm = dt0[i] - h * a[i];
                    lp += dp * dp;
                    lm += dm * dm;
                }
                cp = lp * ps2;
                cm = lm * ps2;
            }
            if (use_rot) {
                cp += rot0 * rot0;
                cm += rot0 * rot0;
            }
        } else {
            if (use_pos) {
                const double r[3] = {tipt[0] - o[0], tipt[1] - o[1], tipt[2] - o[2]};
                const double u[3] = {a[1] * r[2] - a[2] * r[1], a[2] * r[0] - a[0] * r[2],
                                     a[0] * r[1] - a[1] * r[0]};
                const double ar = a[0] * r[0] + a[1] * r[1] + a[2] * r[2];
                double lp = 0.0, lm = 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double mid = dt0[i] + p.vers_h * (a[i] * ar - r[i]);
                    const double dp = mid + p.sin_h * u[i], dm = mid - p.sin_h * u[i];
                    lp += dp * dp;
                    lm += dm * dm;
                }
                cp = lp * ps2;
                cm = lm * ps2;
            }
            if (use_rot) {
                const double dv[3] = {d0[1], d0[2], d0[3]};
                const double A = a[0] * dv[0] + a[1] * dv[1] + a[2] * dv[2];
                const double Bv[3] = {d0[0] * a[0] + (a[1] * dv[2] - a[2] * dv[1]),
                                      d0[0] * a[1] + (a[2] * dv[0] - a[0] * dv[2]),
                                      d0[0] * a[2] + (a[0] * dv[1] - a[1] * dv[0])};
                const double cw = p.cos_h2 * d0[0];
                const double wp = cw - p.sin_h2 * A, wm = cw + p.sin_h2 * A;
                double vp2 = 0.0, vm2 = 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double cv = p.cos_h2 * dv[i];
                    const double vp = cv + p.sin_h2 * Bv[i], vm = cv - p.sin_h2 * Bv[i];
                    vp2 += vp * vp;
                    vm2 += vm * vm;
                }
                const double ap = 2.0 * atan2_pos(sqrt(vp2), fabs(wp)) * p.rot_scale;
                const double am = 2.0 * atan2_pos(sqrt(vm2), fabs(wm)) * p.rot_scale;
                cp += ap * ap;
                cm += am * am;
            }
        }
        if (p.goal_mask) {
            // only joint j's term of each joint goal changes
            const bool bounded = (c.bounded_mask >> j) & 1u;
            const double qj = q[j], qp = qj + h, qm = qj - h;
            double gp = 0.0, gm = 0.0;
            if (p.goal_mask & 1) {
                const double mid = (c.qmin[j] + c.qmax[j]) * 0.5, m = bounded ? c.mdf[j] : 0.0;
                const double t0 = (qj - mid) * m, tp = (qp - mid) * m, tm = (qm - mid) * m;
                gp += (base.g0 - t0 * t0 + tp * tp) * p.w_center_sq;
                gm += (base.g0 - t0 * t0 + tm * tm) * p.w_center_sq;
            }
            if (p.goal_mask & 2) {
                const double m = bounded ? c.mdf[j] : 0.0;
                const double t0 = fmax(0.0, fabs(qj - c.mid[j]) * 2.0 - c.hspan[j]) * m;
                const double tp = fmax(0.0, fabs(qp - c.mid[j]) * 2.0 - c.hspan[j]) * m;
                const double tm = fmax(0.0, fabs(qm - c.mid[j]) * 2.0 - c.hspan[j]) * m;
                gp += (base.g1 - t0 * t0 + tp * tp) * p.w_limits_sq;
                gm += (base.g1 - t0 * t0 + tm * tm) * p.w_limits_sq;
            }
            if (p.goal_mask & 4) {
                const double t0 = (qj - seed[j]) * c.mdf[j], tp = (qp - seed[j]) * c.mdf[j],
                             tm = (qm - seed[j]) * c.mdf[j];
                gp += (base.g2 - t0 * t0 + tp * tp) * p.w_disp_sq;
                gm += (base.g2 - t0 * t0 + tm * tm) * p.w_disp_sq;
            }
            cp += gp;
            cm += gm;
        }
        grad[j] = cp - cm;
    }
}

// ------------------------------------------------------------------------------------------
// GradientIk + step() -- include/pick_ik/ik_gradient.hpp:25-34, src/ik_gradient.cpp:24-94
// ------------------------------------------------------------------------------------------
template <int D>
struct GradState {
    double local[D];
    double best[D];
    double gradient[D];
    double local_cost;
    double best_cost;
};

// Literal step(): 2D + 3 full cost evaluations.  The evaluation loops are kept rolled
// (`#pragma unroll 1`) with unrolled selects for the perturbed joint, so the kernel holds two
// inlined FK bodies instead of 2D + 3 and the hot loop stays inside the instruction cache.
template <int D>
PIK_HD bool gd_step_literal(CK<D> c, PK p, const GoalK& g,
                            const double (&seed)[D], GradState<D>& s) {
    const double h = p.step_size;
    // compute gradient direction -- src/ik_gradient.cpp:28-43
    // (the perturbed joint is selected with 0/1 masks instead of indexed writes: x + 0.0 and
    //  0.0 + 1.0 * g are exact, and nothing gets demoted to an indexable scratch array)
#pragma unroll
    for (int j = 0; j < D; ++j) s.gradient[j] = 0.0;
#pragma unroll 1
    for (int i = 0; i < D; ++i) {
        double pm0 = 0.0, pm1 = 0.0;
#pragma unroll 1
        for (int sg = 0; sg < 2; ++sg) {
            const double dh = sg ? h : -h;
            double working[D];
#pragma unroll
            for (int j = 0; j < D; ++j) working[j] = s.local[j] + ((j == i) ? dh : 0.0);
            const double v = cost_fn<D>(c, p, g, seed, working);
            pm0 = sg ? pm0 : v;
            pm1 = sg ? v : pm1;
        }
        const double gi = pm1 - pm0;
#pragma unroll
        for (int j = 0; j < D; ++j) s.gradient[j] += ((j == i) ? 1.0 : 0.0) * gi;
    }
    // normalize gradient direction -- :46-54
    double sum = h;
#pragma unroll
    for (int i = 0; i < D; ++i) sum = sum + fabs(s.gradient[i]);
    const double f = 1.0 / sum * h;
#pragma unroll
    for (int i = 0; i < D; ++i) s.gradient[i] = s.gradient[i] * f;

    // line search probes, step, accept -- :57-85
    double p1 = 0.0, p3 = 0.0;
#pragma unroll 1
    for (int k = 0; k < 3; ++k) {
        double working[D];
        if (k == 2) {
            const double p2 = (p1 + p3) * 0.5;
            const double cost_diff = (p3 - p1) * 0.5;
            double joint_diff = p2 / cost_diff;
            if (!isfinite(joint_diff)) joint_diff = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                s.local[i] = clamp_joint<D>(c, i, s.local[i] - s.gradient[i] * joint_diff);
                working[i] = s.local[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i)
                working[i] = (k == 0) ? s.local[i] - s.gradient[i] : s.local[i] + s.gradient[i];
        }
        const double v = cost_fn<D>(c, p, g, seed, working);
        if (k == 0) p1 = v;
        else if (k == 1) p3 = v;
        else s.local_cost = v;
    }
    // update best -- :88-93
    if (s.local_cost < s.best_cost) {
#pragma unroll
        for (int i = 0; i < D; ++i) s.best[i] = s.local[i];
        s.best_cost = s.local_cost;
        return true;
    }
    return false;
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based RNG (Salmon et al., SC'11); stream/slot layout documented in
// DESIGN.md "Random streams" and mirrored by the oracle.
// ------------------------------------------------------------------------------------------
PIK_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

struct U4 {
    uint32_t x, y, z, w;
};

PIK_HD U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                        uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = mulhi32(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = mulhi32(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0;
        const uint32_t n2 = h0 ^ c3 ^ k1;
        c0 = n0;
        c1 = l1;
        c2 = n2;
        c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

constexpr uint32_t STREAM_INIT = 1u;
constexpr uint32_t STREAM_REPRODUCE = 2u;
constexpr uint32_t REPRO_IDXB_BLOCK0 = 0x10000u;

PIK_HD U4 rng_block(uint64_t seed, uint32_t stream, uint64_t problem, uint32_t epoch,
                    uint32_t individual, uint32_t block) {
    return philox4x32_10(block, individual, epoch, (uint32_t)problem, (uint32_t)seed ^ stream,
                         (uint32_t)(seed >> 32) + (uint32_t)(problem >> 32));
}

PIK_HD double u01_from_words(uint32_t lo, uint32_t hi) {
    const uint64_t x = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)x * (1.0 / 9007199254740992.0);
}
PIK_HD double u01_from_word(uint32_t w) { return (double)w * (1.0 / 4294967296.0); }
PIK_HD double uniform_real(double a, double b, double u) { return (b - a) * u + a; }

} // namespace pik
