/*
 * pik_oracle.c -- CPU ORACLE (test infrastructure; see pik_oracle.h for scope and parity status).
 *
 * Statement-by-statement restatement of the pick_ik hot path.  Citations are relative to the
 * reference tree (PickNikRobotics/pick_ik v1.1.2).  Compiled with -ffp-contract=off so that the
 * arithmetic is plain IEEE-754 binary64 like the reference's default x86-64 build.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* sincos() */
#endif
#include "pik_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * Transcendental functions.
 *
 * math mode 0 ("libm", default): glibc sin / cos / atan2 -- what the reference's dependencies call.
 * math mode 1 ("portable"): sincos and atan2 restated operation for operation as the GPU library
 *   implements them (pick_ik_amd/csrc/pik_math.hpp: Cody-Waite reduction with explicit FMAs +
 *   fdlibm minimax kernels; fdlibm-style atan with the interval reduction applied to the (y, x)
 *   pair).  Every other operation of the path is +,-,*,/,sqrt, which IEEE-754 defines exactly, so
 *   in this mode the oracle and a GPU build without FMA contraction must agree BIT FOR BIT; that
 *   is how the kernels' control flow (random streams, mating pool, selection, wipeouts) is
 *   verified despite the chaotic sensitivity of the gradient descent (DESIGN.md "Parity").
 *   Mode 1 differs from mode 0 by <= 1 ulp per call (tests/test_oracle_golden.py).
 * math mode 2 ("fma"): mode 1's algorithm with fused multiply-adds AT STATED PLACES -- the row
 *   products of the chain (iso_mul), the sums of squares of the two distances, the relative
 *   quaternion, the cost accumulations (pose cost, joint-goal sums), the gradient-step update -- and
 *   the sine / cosine / arctangent polynomials in Horner form (the product build's sincos_f64 /
 *   atan2_pos).  Every fused operation is C's fma(), i.e. correctly rounded whatever the host: what
 *   the GPU's exact-fma flavour (-DPIK_STRICT -DPIK_EXACT_FMA, pik_math.hpp PIK_XF) executes, bit for
 *   bit.  pick_ik's own contraction is its compiler's (gcc defaults to -ffp-contract=fast: a build for
 *   a machine with FMA fuses, a baseline x86-64 build does not); modes 1 and 2 are those two builds.
 * ---------------------------------------------------------------------------------------- */
static int g_math_mode = 0;
void pko_set_math_mode(int32_t mode) { g_math_mode = mode; }
int32_t pko_get_math_mode(void) { return g_math_mode; }

static void portable_sincos(double x, double* s, double* c) {
    if (fabs(x) > 65536.0) {
        const double k = rint(x * 0.15915494309189535);
        x = __builtin_fma(-k, 6.283185307179586, x);
        x = __builtin_fma(-k, 2.4492935982947064e-16, x);
    }
    const double fn = rint(x * 0.6366197723675814);
    const int n = (int)fn;
    double t = __builtin_fma(-fn, 1.5707963267948966, x);
    t = __builtin_fma(-fn, 6.123233995736766e-17, t);
    t = __builtin_fma(-fn, -1.4973849048591698e-33, t);
    /* power sums, smallest terms first -- operation for operation pik_math.hpp sincos_f64 */
    const double z = t * t;
    const double z2 = z * z, z3 = z2 * z, z4 = z2 * z2, z5 = z4 * z, z6 = z4 * z2, z7 = z6 * z;
    double as = 1.58969099521155010221e-10 * z6;
    as = as + -2.50507602534068634195e-08 * z5;
    as = as + 2.75573137070700676789e-06 * z4;
    as = as + -1.98412698298579493134e-04 * z3;
    as = as + 8.33333333332248946124e-03 * z2;
    as = as + -1.66666666666666324348e-01 * z;
    const double sn = t + t * as;
    double ac = -1.13596475577881948265e-11 * z7;
    ac = ac + 2.08757232129817482790e-09 * z6;
    ac = ac + -2.75573143513906633035e-07 * z5;
    ac = ac + 2.48015872894767294178e-05 * z4;
    ac = ac + -1.38888888888741095749e-03 * z3;
    ac = ac + 4.16666666666666019037e-02 * z2;
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cn = w + (((1.0 - w) - hz) + ac);
    const double a = (n & 1) ? cn : sn;
    const double b = (n & 1) ? sn : cn;
    *s = (n & 2) ? -a : a;
    *c = ((n + 1) & 2) ? -b : b;
}

/* atan2(y, x) for y >= 0, x >= 0 (the only call site: 2 atan2(|vec|, |w|)) */
static double portable_atan2_pos(double y, double x) {
    const double y16 = 16.0 * y;
    const int c0 = y16 < 7.0 * x, c1 = y16 < 11.0 * x, c2 = y16 < 19.0 * x, c3 = y16 < 39.0 * x;
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
    const double z2 = z * z, z3 = z2 * z, z4 = z2 * z2, z5 = z4 * z, z6 = z4 * z2, z7 = z4 * z3,
                 z8 = z4 * z4, z9 = z8 * z, z10 = z8 * z2, z11 = z8 * z3;
    double a = 1.62858201153657823623e-02 * z11;
    a = a + -3.65315727442169155270e-02 * z10;
    a = a + 4.97687799461593236017e-02 * z9;
    a = a + -5.83357013379057348645e-02 * z8;
    a = a + 6.66107313738753120669e-02 * z7;
    a = a + -7.69187620504482999495e-02 * z6;
    a = a + 9.09088713343650656196e-02 * z5;
    a = a + -1.11111104054623557880e-01 * z4;
    a = a + 1.42857142725034663711e-01 * z3;
    a = a + -1.99999999998764832476e-01 * z2;
    a = a + 3.33333333333329318027e-01 * z;
    const double res = c0 ? (r - r * a) : (hi - ((r * a - lo) - r));
    return (y == 0.0) ? 0.0 : res;
}

/* ---- mode 2 ---- */
#define FMA(a, b, c) __builtin_fma((a), (b), (c))
/* a0 b0 + a1 b1 + a2 b2 and a^2 + b^2 + c^2: pik_math.hpp xdot3 / xsumsq3 */
static double dot3(double a0, double b0, double a1, double b1, double a2, double b2) {
    if (g_math_mode == 2) return FMA(a2, b2, FMA(a1, b1, a0 * b0));
    return a0 * b0 + a1 * b1 + a2 * b2;
}
static double sumsq3(double a, double b, double c) {
    if (g_math_mode == 2) return FMA(c, c, FMA(b, b, a * a));
    return a * a + b * b + c * c;
}

/* pik_math.hpp sincos_f64, product / PIK_XF branch: the same reduction, Horner polynomials */
static void fma_sincos(double x, double* s, double* c) {
    if (fabs(x) > 65536.0) {
        const double k = rint(x * 0.15915494309189535);
        x = FMA(-k, 6.283185307179586, x);
        x = FMA(-k, 2.4492935982947064e-16, x);
    }
    const double fn = rint(x * 0.6366197723675814);
    const int n = (int)fn;
    double t = FMA(-fn, 1.5707963267948966, x);
    t = FMA(-fn, 6.123233995736766e-17, t);
    t = FMA(-fn, -1.4973849048591698e-33, t);
    const double z = t * t;
    double rs = 1.58969099521155010221e-10;
    rs = FMA(rs, z, -2.50507602534068634195e-08);
    rs = FMA(rs, z, 2.75573137070700676789e-06);
    rs = FMA(rs, z, -1.98412698298579493134e-04);
    rs = FMA(rs, z, 8.33333333332248946124e-03);
    rs = FMA(rs, z, -1.66666666666666324348e-01);
    double rc = -1.13596475577881948265e-11;
    rc = FMA(rc, z, 2.08757232129817482790e-09);
    rc = FMA(rc, z, -2.75573143513906633035e-07);
    rc = FMA(rc, z, 2.48015872894767294178e-05);
    rc = FMA(rc, z, -1.38888888888741095749e-03);
    rc = FMA(rc, z, 4.16666666666666019037e-02);
    const double sn = FMA(t * z, rs, t);
    const double zz = z * z;
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cn = w + FMA(zz, rc, (1.0 - w) - hz);
    const double a = (n & 1) ? cn : sn;
    const double b = (n & 1) ? sn : cn;
    *s = (n & 2) ? -a : a;
    *c = ((n + 1) & 2) ? -b : b;
}

/* pik_math.hpp atan2_pos, product / PIK_XF branch (y >= 0, x >= 0): the smaller over the larger
 * argument, then atan(a / b) = pi/4 + atan((a - b) / (a + b)) above tan(pi/8); fdlibm's polynomial
 * split into its even and odd coefficients, both in Horner form */
static double fma_atan2_pos(double y, double x) {
    const int sw = y > x;
    const double a = sw ? x : y, b = sw ? y : x;
    const int t = a > 0.41421356237309503 * b;
    const double num = t ? a - b : a;
    const double den = t ? a + b : b;
    const double r = num / den;
    const double z = r * r;
    const double w = z * z;
    double s1 = 1.62858201153657823623e-02;
    s1 = FMA(s1, w, 4.97687799461593236017e-02);
    s1 = FMA(s1, w, 6.66107313738753120669e-02);
    s1 = FMA(s1, w, 9.09088713343650656196e-02);
    s1 = FMA(s1, w, 1.42857142725034663711e-01);
    s1 = FMA(s1, w, 3.33333333333329318027e-01);
    double s2 = -3.65315727442169155270e-02;
    s2 = FMA(s2, w, -5.83357013379057348645e-02);
    s2 = FMA(s2, w, -7.69187620504482999495e-02);
    s2 = FMA(s2, w, -1.11111104054623557880e-01);
    s2 = FMA(s2, w, -1.99999999998764832476e-01);
    const double poly = FMA(z, s2, s1) * z;
    const double p0 = FMA(-r, poly, r);
    const double p1 = t ? (7.85398163397448278999e-01 + (p0 + 3.06161699786838301793e-17)) : p0;
    const double res = sw ? (1.57079632679489655800e+00 - (p1 - 6.12323399573676603587e-17)) : p1;
    return (y == 0.0) ? 0.0 : res;
}

static void sincos_dispatch(double x, double* s, double* c) {
    if (g_math_mode == 2) {
        fma_sincos(x, s, c);
    } else if (g_math_mode == 1) {
        portable_sincos(x, s, c);
    } else {
        *c = cos(x);
        *s = sin(x);
    }
}
static double atan2_dispatch(double y, double x) {
    return g_math_mode == 2 ? fma_atan2_pos(y, x) : g_math_mode == 1 ? portable_atan2_pos(y, x) : atan2(y, x);
}
void pko_sincos(double x, double* s, double* c) { sincos_dispatch(x, s, c); }
double pko_atan2(double y, double x) { return atan2_dispatch(y, x); }

/* ------------------------------------------------------------------------------------------
 * Types
 * ---------------------------------------------------------------------------------------- */

/* Eigen::Isometry3d restated as rotation (row-major) + translation. */
typedef struct {
    double R[9];
    double t[3];
} iso_t;

/* pick_ik::Robot::Variable -- include/pick_ik/robot.hpp:15-37 */
typedef struct {
    double min, max, mid;
    int bounded;
    double half_span;
    double max_velocity_rcp;
    double minimal_displacement_factor;
} variable_t;

/* The joints between the base and ONE tip link, in order: what RobotState::updateLinkTransforms
 * multiplies to get that link's global transform.  `var[j]` is the index of joint j's variable in
 * the active-variable vector (src/robot.cpp:130-160: the union over all tips, in model order). */
typedef struct {
    int n;
    int var[PKO_MAX_DOF];
    iso_t origin[PKO_MAX_DOF];       /* LinkModel::getJointOriginTransform() of each joint's child link */
    int origin_is_identity[PKO_MAX_DOF];
    double axis[PKO_MAX_DOF][3];     /* normalised (RevoluteJointModel::setAxis) */
    int joint_type[PKO_MAX_DOF];
    iso_t tip;                       /* fixed transform(s) after the last joint */
    int tip_is_identity;
    /* mimic joints of the path: no variables (src/robot.cpp:144-150), but RobotState moves them with their
     * master (setJointGroupPositions -> updateMimicJoints, src/fk_moveit.cpp:22): value = mult * q[var] + off.
     * m_after[m] = the VARIABLE whose joint the mimic joint follows on the path (-1: in front of the first) */
    int n_mimic;
    int m_after[PKO_MAX_MIMIC], m_var[PKO_MAX_MIMIC], m_type[PKO_MAX_MIMIC];
    iso_t m_origin[PKO_MAX_MIMIC];
    int m_origin_is_identity[PKO_MAX_MIMIC];
    double m_axis[PKO_MAX_MIMIC][3];
    double m_mult[PKO_MAX_MIMIC], m_off[PKO_MAX_MIMIC];
} path_t;

struct pko_chain {
    int dof;
    int n_tips;                      /* tip_frames of the plugin (src/pick_ik_plugin.cpp:57-69) */
    path_t path[PKO_MAX_TIPS];
    variable_t var[PKO_MAX_DOF];
};

/* ------------------------------------------------------------------------------------------
 * Eigen / urdfdom / MoveIt arithmetic restated from their published definitions
 * ---------------------------------------------------------------------------------------- */

/* Eigen 3.4 Quaternion::toRotationMatrix (Eigen/src/Geometry/Quaternion.h). q = (w,x,y,z). */
static void quat_to_matrix(const double q[4], double R[9]) {
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

/* Eigen 3.4 quaternionbase_assign_impl<Other,3,3>::run (rotation matrix -> quaternion), the
 * conversion `Eigen::Quaterniond(frame.rotation())` performs in src/goal.cpp:22-23. */
static void matrix_to_quat(const double R[9], double q[4]) {
#define M(i, j) R[(i) * 3 + (j)]
    double t = M(0, 0) + M(1, 1) + M(2, 2);
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (M(2, 1) - M(1, 2)) * t;
        q[2] = (M(0, 2) - M(2, 0)) * t;
        q[3] = (M(1, 0) - M(0, 1)) * t;
    } else {
        int i = 0;
        if (M(1, 1) > M(0, 0)) i = 1;
        if (M(2, 2) > M(i, i)) i = 2;
        int j = (i + 1) % 3;
        int k = (j + 1) % 3;
        t = sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (M(k, j) - M(j, k)) * t;
        q[1 + j] = (M(j, i) + M(i, j)) * t;
        q[1 + k] = (M(k, i) + M(i, k)) * t;
    }
#undef M
}

/* urdf::Rotation::setFromRPY (urdfdom_headers urdf_model/pose.h) followed by normalize(), then
 * the Eigen conversion MoveIt applies when it builds LinkModel::joint_origin_transform_. */
static void rpy_xyz_to_iso(const double xyz_rpy[6], iso_t* out) {
    const double phi = xyz_rpy[3] / 2.0, the = xyz_rpy[4] / 2.0, psi = xyz_rpy[5] / 2.0;
    /* glibc sincos(): what GCC emits for the sin/cos pairs of setFromRPY; called explicitly so the
     * result does not depend on whether a compiler merges the pair (glibc's cos() and sincos()
     * differ by 1 ulp for some arguments) */
    double sphi, cphi, sthe, cthe, spsi, cpsi;
    sincos(phi, &sphi, &cphi);
    sincos(the, &sthe, &cthe);
    sincos(psi, &spsi, &cpsi);
    double q[4];
    q[1] = sphi * cthe * cpsi - cphi * sthe * spsi;
    q[2] = cphi * sthe * cpsi + sphi * cthe * spsi;
    q[3] = cphi * cthe * spsi - sphi * sthe * cpsi;
    q[0] = cphi * cthe * cpsi + sphi * sthe * spsi;
    const double s = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (s == 0.0) {
        q[0] = 1.0;
        q[1] = q[2] = q[3] = 0.0;
    } else {
        for (int i = 0; i < 4; ++i) q[i] /= s;
    }
    quat_to_matrix(q, out->R);
    out->t[0] = xyz_rpy[0];
    out->t[1] = xyz_rpy[1];
    out->t[2] = xyz_rpy[2];
}

static int iso_is_identity(const iso_t* a) {
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; ++i)
        if (a->R[i] != I[i]) return 0;
    return a->t[0] == 0.0 && a->t[1] == 0.0 && a->t[2] == 0.0;
}

/* Eigen Isometry3d product: linear = A.linear*B.linear, translation = A.linear*B.t + A.t */
static void iso_mul(const iso_t* a, const iso_t* b, iso_t* out) {
    iso_t r;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            r.R[i * 3 + j] = dot3(a->R[i * 3 + 0], b->R[0 * 3 + j], a->R[i * 3 + 1], b->R[1 * 3 + j],
                                  a->R[i * 3 + 2], b->R[2 * 3 + j]);
        }
        if (g_math_mode == 2)
            r.t[i] = FMA(a->R[i * 3 + 2], b->t[2], FMA(a->R[i * 3 + 1], b->t[1], FMA(a->R[i * 3 + 0], b->t[0], a->t[i])));
        else
            r.t[i] = a->R[i * 3 + 0] * b->t[0] + a->R[i * 3 + 1] * b->t[1] + a->R[i * 3 + 2] * b->t[2] +
                     a->t[i];
    }
    *out = r;
}

/* moveit::core::RevoluteJointModel::computeTransform (Rodrigues form with c, s, t = 1 - c) and
 * PrismaticJointModel::computeTransform; the per-joint frame pick_ik's own (dead) FK states in
 * src/forward_kinematics.cpp:39-80 is the same rotation written as a half-angle quaternion. */
static void joint_transform_axis(const double axis[3], int joint_type, double v, iso_t* out);
static void joint_transform(const path_t* c, int j, double v, iso_t* out) {
    joint_transform_axis(c->axis[j], c->joint_type[j], v, out);
}
static void joint_transform_axis(const double axis[3], int joint_type, double v, iso_t* out) {
    const double x = axis[0], y = axis[1], z = axis[2];
    if (joint_type == PKO_JOINT_PRISMATIC) {
        static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        memcpy(out->R, I, sizeof I);
        out->t[0] = x * v;
        out->t[1] = y * v;
        out->t[2] = z * v;
        return;
    }
    double cs, sn;
    sincos_dispatch(v, &sn, &cs);
    const double t = 1.0 - cs;
    const double txy = t * (x * y);
    const double txz = t * (x * z);
    const double tyz = t * (y * z);
    const double zs = z * sn, ys = y * sn, xs = x * sn;
    out->R[0] = t * (x * x) + cs;
    out->R[3] = txy + zs;
    out->R[6] = txz - ys;
    out->R[1] = txy - zs;
    out->R[4] = t * (y * y) + cs;
    out->R[7] = tyz + xs;
    out->R[2] = txz + ys;
    out->R[5] = tyz - xs;
    out->R[8] = t * (z * z) + cs;
    out->t[0] = out->t[1] = out->t[2] = 0.0;
}

/* The live FK: src/fk_moveit.cpp:20-33 -> RobotState::updateLinkTransforms():
 *   global(link) = global(parent) * joint_origin(link) * joint_transform(q)   (left to right),
 * skipping the origin product when it is the identity, then the fixed links up to the tip. */
/* the mimic joints that follow the joint of variable `after`: origin product, then the joint's transform at
 * multiplier * q[master] + offset (JointModel::computeTransform of a revolute / prismatic joint) */
static void mimic_steps(const path_t* c, int after, const double* q, iso_t* g) {
    for (int m = 0; m < c->n_mimic; ++m) {
        if (c->m_after[m] != after) continue;
        const double v = c->m_mult[m] * q[c->m_var[m]] + c->m_off[m];
        iso_t jt;
        joint_transform_axis(c->m_axis[m], c->m_type[m], v, &jt);
        if (!c->m_origin_is_identity[m]) iso_mul(g, &c->m_origin[m], g);
        iso_mul(g, &jt, g);
    }
}

static void fk_path(const path_t* c, const double* q, iso_t* tip) {
    iso_t g;
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(g.R, I, sizeof I);
    g.t[0] = g.t[1] = g.t[2] = 0.0;
    if (c->n_mimic) mimic_steps(c, -1, q, &g);
    for (int j = 0; j < c->n; ++j) {
        iso_t jt;
        if (c->joint_type[j] >= PKO_JOINT_FLOATING_TX && c->joint_type[j] < PKO_JOINT_FLOATING_RW) {
            if (c->n_mimic) mimic_steps(c, c->var[j], q, &g);
            continue; /* the first six variables of a floating joint: the joint acts at its seventh */
        }
        if (c->joint_type[j] == PKO_JOINT_FLOATING_RW) {
            /* FloatingJointModel::computeTransform: Translation(v0 v1 v2) * Quaterniond(v6, v3, v4, v5) */
            const double qq[4] = {q[c->var[j]], q[c->var[j - 3]], q[c->var[j - 2]], q[c->var[j - 1]]};
            quat_to_matrix(qq, jt.R);
            jt.t[0] = q[c->var[j - 6]];
            jt.t[1] = q[c->var[j - 5]];
            jt.t[2] = q[c->var[j - 4]];
        } else {
            joint_transform(c, j, q[c->var[j]], &jt);
        }
        if (!c->origin_is_identity[j]) iso_mul(&g, &c->origin[j], &g);
        iso_mul(&g, &jt, &g);
        if (c->n_mimic) mimic_steps(c, c->var[j], q, &g);
    }
    if (!c->tip_is_identity) iso_mul(&g, &c->tip, &g);
    *tip = g;
}

/* make_fk_fn -- src/fk_moveit.cpp:11-35: one frame per tip link, in tip_frames order */
static void fk(const pko_chain* c, const double* q, iso_t* tips) {
    for (int k = 0; k < c->n_tips; ++k) fk_path(&c->path[k], q, &tips[k]);
}

/* ------------------------------------------------------------------------------------------
 * src/goal.cpp
 * ---------------------------------------------------------------------------------------- */

/* linear_distance -- src/goal.cpp:17-19 */
static double linear_distance(const iso_t* f1, const iso_t* f2) {
    const double dx = f1->t[0] - f2->t[0], dy = f1->t[1] - f2->t[1], dz = f1->t[2] - f2->t[2];
    return sqrt(sumsq3(dx, dy, dz));
}

/* angular_distance -- src/goal.cpp:21-25: q_2.angularDistance(q_1) with
 * Eigen 3.4 QuaternionBase::angularDistance: d = (*this) * other.conjugate();
 * return 2 * atan2(d.vec().norm(), abs(d.w())). */
static double angular_distance(const iso_t* f1, const iso_t* f2) {
    double q1[4], q2[4];
    matrix_to_quat(f1->R, q1);
    matrix_to_quat(f2->R, q2);
    /* d = q2 * conj(q1);  conj(q1) = (w, -x, -y, -z) */
    const double aw = q2[0], ax = q2[1], ay = q2[2], az = q2[3];
    const double bw = q1[0], bx = -q1[1], by = -q1[2], bz = -q1[3];
    double dw, dx, dy, dz;
    if (g_math_mode == 2) { /* pik_math.hpp quat_mul_conj, PIK_XF */
        dw = FMA(-az, bz, FMA(-ay, by, FMA(-ax, bx, aw * bw)));
        dx = FMA(-az, by, FMA(ay, bz, FMA(ax, bw, aw * bx)));
        dy = FMA(-ax, bz, FMA(az, bx, FMA(ay, bw, aw * by)));
        dz = FMA(-ay, bx, FMA(ax, by, FMA(az, bw, aw * bz)));
    } else {
        dw = aw * bw - ax * bx - ay * by - az * bz;
        dx = aw * bx + ax * bw + ay * bz - az * by;
        dy = aw * by + ay * bw + az * bx - ax * bz;
        dz = aw * bz + az * bw + ax * by - ay * bx;
    }
    return 2.0 * atan2_dispatch(sqrt(sumsq3(dx, dy, dz)), fabs(dw));
}

/* make_frame_test_fn -- src/goal.cpp:27-36 */
static int frame_test(const iso_t* goal, const iso_t* tip, int has_pos, double pos_thr, int has_ori,
                      double ori_thr) {
    return (!has_pos || linear_distance(goal, tip) <= pos_thr) &&
           (!has_ori || fabs(angular_distance(goal, tip)) <= ori_thr);
}

/* make_pose_cost_fn -- src/goal.cpp:51-78 (the four scale>0 branches) */
static double pose_cost(const iso_t* goal, const iso_t* frame, double position_scale,
                        double rotation_scale) {
    if (position_scale > 0.0) {
        if (rotation_scale > 0.0) {
            if (g_math_mode == 2) {
                const double a = linear_distance(goal, frame) * position_scale;
                const double b = angular_distance(goal, frame) * rotation_scale;
                return FMA(b, b, a * a);
            }
            return pow(linear_distance(goal, frame) * position_scale, 2) +
                   pow(angular_distance(goal, frame) * rotation_scale, 2);
        }
        return pow(linear_distance(goal, frame) * position_scale, 2);
    }
    if (rotation_scale > 0.0) {
        return pow(angular_distance(goal, frame) * rotation_scale, 2);
    }
    return 0.0;
}

/* sum += pow(v, 2) of the three joint-goal sums (mode 2: one fused operation) */
static double sq_acc(double sum, double v) {
    if (g_math_mode == 2) return FMA(v, v, sum);
    return sum + pow(v, 2);
}

/* make_center_joints_cost_fn -- src/goal.cpp:91-108 */
static double center_joints_cost(const pko_chain* c, const double* q) {
    double sum = 0;
    for (int i = 0; i < c->dof; ++i) {
        const variable_t* v = &c->var[i];
        if (!v->bounded) continue;
        const double mid = (v->min + v->max) * 0.5;
        sum = sq_acc(sum, (q[i] - mid) * v->minimal_displacement_factor);
    }
    return sum;
}

/* make_avoid_joint_limits_cost_fn -- src/goal.cpp:110-129 */
static double avoid_joint_limits_cost(const pko_chain* c, const double* q) {
    double sum = 0;
    for (int i = 0; i < c->dof; ++i) {
        const variable_t* v = &c->var[i];
        if (!v->bounded) continue;
        sum = sq_acc(sum, fmax(0.0, fabs(q[i] - v->mid) * 2.0 - v->half_span) * v->minimal_displacement_factor);
    }
    return sum;
}

/* make_minimal_displacement_cost_fn -- src/goal.cpp:131-144 */
static double minimal_displacement_cost(const pko_chain* c, const double* q, const double* guess) {
    double sum = 0;
    for (int i = 0; i < c->dof; ++i) {
        sum = sq_acc(sum, (q[i] - guess[i]) * c->var[i].minimal_displacement_factor);
    }
    return sum;
}

/* One IK problem: what src/pick_ik_plugin.cpp:88-142 assembles into cost_fn / solution_fn. */
typedef struct {
    const pko_chain* chain;
    const pko_params* params;
    iso_t goal[PKO_MAX_TIPS]; /* goal frames in the base frame, one per tip */
    const double* seed; /* ik_seed_state (minimal-displacement reference) */
    int has_pos_thr, has_ori_thr;
    int64_t evals;
    /* host cost function (kinematics::KinematicsBase::IKCostFn): one more Goal of weight 1 per tip pose, pushed
     * behind the joint goals -- src/pick_ik_plugin.cpp:130-135, make_ik_cost_fn src/goal.cpp:146-161 */
    pko_cost_fn cb;
    void* cb_user;
} problem_t;

/* the three optional goals, in the order the plugin pushes them (src/pick_ik_plugin.cpp:118-129) */
static int n_goals(const problem_t* pb, double w[3], int kind[3]) {
    int n = 0;
    if (pb->params->center_joints_weight > 0.0) {
        w[n] = pb->params->center_joints_weight;
        kind[n++] = 0;
    }
    if (pb->params->avoid_joint_limits_weight > 0.0) {
        w[n] = pb->params->avoid_joint_limits_weight;
        kind[n++] = 1;
    }
    if (pb->params->minimal_displacement_weight > 0.0) {
        w[n] = pb->params->minimal_displacement_weight;
        kind[n++] = 2;
    }
    return n;
}

static double goal_eval(const problem_t* pb, int kind, const double* q) {
    switch (kind) {
        case 0: return center_joints_cost(pb->chain, q);
        case 1: return avoid_joint_limits_cost(pb->chain, q);
        default: return minimal_displacement_cost(pb->chain, q, pb->seed);
    }
}

/* make_cost_fn -- src/goal.cpp:188-203: pose_cost + sum goal.eval * weight^2 */
static double cost_fn(problem_t* pb, const double* q) {
    iso_t tip[PKO_MAX_TIPS];
    pb->evals++;
    fk(pb->chain, q, tip);
    /* std::accumulate(pose_cost_functions, 0.0, sum + fn(tip_frames)) -- one per goal frame
     * (make_pose_cost_functions, src/goal.cpp:80-89) */
    double pc = 0.0;
    for (int k = 0; k < pb->chain->n_tips; ++k)
        pc = pc + pose_cost(&pb->goal[k], &tip[k], pb->params->position_scale, pb->params->rotation_scale);
    double w[3];
    int kind[3];
    const int n = n_goals(pb, w, kind);
    double gc = 0.0;
    for (int g = 0; g < n; ++g) gc = gc + goal_eval(pb, kind[g], q) * pow(w[g], 2);
    if (pb->cb)
        for (int k = 0; k < pb->chain->n_tips; ++k) gc = gc + pb->cb(q, pb->chain->dof, k, pb->cb_user) * pow(1.0, 2);
    return pc + gc;
}

/* make_is_solution_test_fn -- src/goal.cpp:163-186 */
static int solution_fn(problem_t* pb, const double* q) {
    iso_t tip[PKO_MAX_TIPS];
    fk(pb->chain, q, tip);
    for (int k = 0; k < pb->chain->n_tips; ++k) {
        if (!frame_test(&pb->goal[k], &tip[k], pb->has_pos_thr, pb->params->position_threshold,
                        pb->has_ori_thr, pb->params->orientation_threshold))
            return 0;
    }
    const double cost_threshold_sq = pow(pb->params->cost_threshold, 2);
    double w[3];
    int kind[3];
    const int n = n_goals(pb, w, kind);
    for (int g = 0; g < n; ++g) {
        const double cost = goal_eval(pb, kind[g], q) * pow(w[g], 2);
        if (cost >= cost_threshold_sq) return 0;
    }
    if (pb->cb)
        for (int k = 0; k < pb->chain->n_tips; ++k) {
            const double cost = pb->cb(q, pb->chain->dof, k, pb->cb_user) * pow(1.0, 2);
            if (cost >= cost_threshold_sq) return 0;
        }
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * src/robot.cpp
 * ---------------------------------------------------------------------------------------- */

/* Variable::clamp_to_limits -- src/robot.cpp:36-42 (std::clamp(v, lo, hi): v<lo?lo : hi<v?hi : v) */
static double clamp_to_limits(const variable_t* v, double val) {
    double lo, hi;
    if (v->bounded) {
        lo = v->min;
        hi = v->max;
    } else {
        lo = val - v->half_span;
        hi = val + v->half_span;
    }
    return (val < lo) ? lo : (hi < val) ? hi : val;
}

/* Variable::is_valid -- src/robot.cpp:32-34 (kept for completeness; used by the plugin-level
 * seed check src/pick_ik_plugin.cpp:153-159) */
static int is_valid(const variable_t* v, double val) {
    return (!v->bounded) || (val <= v->max && val >= v->min);
}

/* ------------------------------------------------------------------------------------------
 * RNG: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11),
 * replacing rsl::uniform_real / rsl::uniform_int (call sites src/ik_memetic.cpp:131-159,
 * src/robot.cpp:25-28).
 * ---------------------------------------------------------------------------------------- */
void pko_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

#define STREAM_INIT 1u      /* initPopulation random elites */
#define STREAM_REPRODUCE 2u /* reproduce() draws */

/* Philox block for (seed, stream, problem, epoch, individual, block).
 *   counter = {block, individual, epoch, problem[31:0]}
 *   key     = {seed[31:0] ^ stream, seed[63:32] + problem[63:32]} */
static void rng_block(uint64_t seed, uint32_t stream, uint64_t problem, uint32_t epoch,
                      uint32_t individual, uint32_t block, uint32_t out[4]) {
    const uint32_t ctr[4] = {block, individual, epoch, (uint32_t)problem};
    const uint32_t key[2] = {(uint32_t)seed ^ stream,
                             (uint32_t)(seed >> 32) + (uint32_t)(problem >> 32)};
    pko_philox4x32_10(ctr, key, out);
}

/* two 32-bit words -> double in [0,1) with 53 random bits */
static double u01_from_words(uint32_t lo, uint32_t hi) {
    const uint64_t x = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)x * (1.0 / 9007199254740992.0);
}

/* one 32-bit word -> double in [0,1) */
static double u01_from_word(uint32_t w) { return (double)w * (1.0 / 4294967296.0); }

/* slot s addresses double number (s & 1) of block (s >> 1) */
double pko_rng_u01(uint64_t seed, uint32_t stream, uint64_t problem, uint32_t epoch,
                   uint32_t individual, uint32_t slot) {
    uint32_t w[4];
    rng_block(seed, stream, problem, epoch, individual, slot >> 1, w);
    return (slot & 1u) ? u01_from_words(w[2], w[3]) : u01_from_words(w[0], w[1]);
}

/* std::uniform_real_distribution(a,b): (b - a) * u + a */
static double uniform_real(double a, double b, double u) { return (b - a) * u + a; }

typedef struct {
    uint64_t seed;
    uint64_t problem;
    uint32_t species; /* folded into the individual index: individual | species << 20 */
} rng_t;

/* Slot layout of the REPRODUCE stream for child i of generation g (epoch = g):
 *   block 0          : word0 -> idxA, word1 -> idxB (uniform over the other pool members),
 *                      words 2,3 -> mix_ratio (53 bit)
 *   block 1+j (gene j): word0 -> parentA gradient coefficient, word1 -> parentB's,
 *                       word2 -> mutation test, word3 -> mutation amount      (32-bit uniforms)
 *                       (words 2,3 as one 53-bit draw = the random configuration value when the
 *                        mating pool is empty)
 * Slot layout of the INIT stream for elite i of init epoch e: 53-bit double slot j -> joint j. */

/* ------------------------------------------------------------------------------------------
 * src/ik_gradient.cpp
 * ---------------------------------------------------------------------------------------- */

/* GradientIk -- include/pick_ik/ik_gradient.hpp:25-34 */
typedef struct {
    double gradient[PKO_MAX_DOF];
    double working[PKO_MAX_DOF];
    double local[PKO_MAX_DOF];
    double best[PKO_MAX_DOF];
    double local_cost;
    double best_cost;
} gradient_ik_t;

/* GradientIk::from -- src/ik_gradient.cpp:14-22 */
static void gradient_ik_from(gradient_ik_t* self, problem_t* pb, const double* initial_guess) {
    const int n = pb->chain->dof;
    const double initial_cost = cost_fn(pb, initial_guess);
    for (int i = 0; i < n; ++i) {
        self->gradient[i] = 0.0;
        self->working[i] = initial_guess[i];
        self->local[i] = initial_guess[i];
        self->best[i] = initial_guess[i];
    }
    self->local_cost = initial_cost;
    self->best_cost = initial_cost;
}

/* step -- src/ik_gradient.cpp:24-94 */
static int gd_step(gradient_ik_t* self, problem_t* pb, double step_size) {
    const int count = pb->chain->dof;

    /* compute gradient direction :28-43 */
    for (int i = 0; i < count; ++i) {
        self->working[i] = self->local[i] - step_size;
        const double p1 = cost_fn(pb, self->working);
        self->working[i] = self->local[i] + step_size;
        const double p3 = cost_fn(pb, self->working);
        self->working[i] = self->local[i];
        self->gradient[i] = p3 - p1;
    }

    /* normalize gradient direction :46-54 */
    double sum = step_size;
    for (int i = 0; i < count; ++i) sum = sum + fabs(self->gradient[i]);
    const double f = 1.0 / sum * step_size;
    for (int i = 0; i < count; ++i) self->gradient[i] = self->gradient[i] * f;

    /* initialize line search :57-66 */
    for (int i = 0; i < count; ++i) self->working[i] = self->local[i] - self->gradient[i];
    const double p1 = cost_fn(pb, self->working);
    for (int i = 0; i < count; ++i) self->working[i] = self->local[i] + self->gradient[i];
    const double p3 = cost_fn(pb, self->working);
    const double p2 = (p1 + p3) * 0.5;

    /* linear step size estimation :69-73 */
    const double cost_diff = (p3 - p1) * 0.5;
    double joint_diff = p2 / cost_diff;
    if (!isfinite(joint_diff)) joint_diff = 0.0;

    /* apply optimization step :77-81 */
    for (int i = 0; i < count; ++i) {
        const double updated_value = g_math_mode == 2 ? FMA(-self->gradient[i], joint_diff, self->local[i])
                                                      : self->local[i] - self->gradient[i] * joint_diff;
        self->working[i] = clamp_to_limits(&pb->chain->var[i], updated_value);
    }

    /* always accept :84-85 */
    for (int i = 0; i < count; ++i) self->local[i] = self->working[i];
    self->local_cost = cost_fn(pb, self->local);

    /* update best :88-93 */
    if (self->local_cost < self->best_cost) {
        for (int i = 0; i < count; ++i) self->best[i] = self->local[i];
        self->best_cost = self->local_cost;
        return 1;
    }
    return 0;
}

/* ik_gradient -- src/ik_gradient.cpp:96-139 (wall-clock limit disabled).
 * Returns 1 and fills out[] when the reference would return a value; *valid says whether the value
 * passed solution_fn (1) or is the approximate best (0). */
static int ik_gradient(problem_t* pb, const double* initial_guess, int approx_solution,
                       double* out, double* out_cost, int* valid, int* iterations) {
    const pko_params* p = pb->params;
    const int n = pb->chain->dof;
    *iterations = 0;
    if (p->stop_optimization_on_valid_solution && solution_fn(pb, initial_guess)) {
        memcpy(out, initial_guess, sizeof(double) * (size_t)n);
        *out_cost = cost_fn(pb, initial_guess);
        pb->evals--; /* bookkeeping only: the reference does not evaluate the cost here */
        *valid = 1;
        return 1;
    }

    gradient_ik_t ik;
    gradient_ik_from(&ik, pb, initial_guess);

    int num_iterations = 0;
    double previous_cost = 0.0;
    while (num_iterations < p->gd_max_iters) {
        if (gd_step(&ik, pb, p->gd_step_size)) {
            if (p->stop_optimization_on_valid_solution && solution_fn(pb, ik.best)) {
                memcpy(out, ik.best, sizeof(double) * (size_t)n);
                *out_cost = ik.best_cost;
                *valid = 1;
                *iterations = num_iterations + 1;
                return 1;
            }
        }
        if (fabs(ik.local_cost - previous_cost) <= p->gd_min_cost_delta) break;
        previous_cost = ik.local_cost;
        num_iterations++;
    }
    *iterations = num_iterations;

    if (!p->stop_optimization_on_valid_solution && solution_fn(pb, ik.best)) {
        memcpy(out, ik.best, sizeof(double) * (size_t)n);
        *out_cost = ik.best_cost;
        *valid = 1;
        return 1;
    }
    if (approx_solution) {
        memcpy(out, ik.best, sizeof(double) * (size_t)n);
        *out_cost = ik.best_cost;
        *valid = 0;
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * src/ik_memetic.cpp
 * ---------------------------------------------------------------------------------------- */

/* Individual -- include/pick_ik/ik_memetic.hpp:19-24 (+ slot: the pre-sort population index,
 * used only as the deterministic tie-break of the reference's unstable std::sort). */
typedef struct {
    double genes[PKO_MAX_DOF];
    double fitness;
    double extinction;
    double gradient[PKO_MAX_DOF];
    int slot;
} individual_t;

/* MemeticIk -- include/pick_ik/ik_memetic.hpp:47-85 */
typedef struct {
    individual_t* population; /* population_size */
    int* mating_pool;         /* indices into population (pointers in the reference) */
    int pool_size;
    individual_t best;      /* best_ */
    individual_t best_curr; /* best_curr_ */
    int has_previous_fitness;
    double previous_fitness;
    double* extinction_grading;
    double inverse_gene_size;
    int population_size, elite_size;
    rng_t rng;
    uint32_t init_epoch; /* number of initPopulation calls so far */
    int wipeouts, erasures;
} memetic_t;

/* MemeticIk::MemeticIk / from -- src/ik_memetic.cpp:18-41 */
static void memetic_from(memetic_t* ik, problem_t* pb, const double* initial_guess, rng_t rng) {
    const pko_params* p = pb->params;
    const int n = pb->chain->dof;
    memset(ik, 0, sizeof *ik);
    ik->population_size = p->memetic_population_size;
    ik->elite_size = p->memetic_elite_size;
    ik->population = (individual_t*)calloc((size_t)ik->population_size, sizeof(individual_t));
    ik->mating_pool = (int*)calloc((size_t)ik->elite_size, sizeof(int));
    ik->extinction_grading = (double*)calloc((size_t)ik->population_size, sizeof(double));
    for (int i = 0; i < n; ++i) {
        ik->best.genes[i] = initial_guess[i];
        ik->best.gradient[i] = 0.0;
    }
    ik->best.fitness = cost_fn(pb, initial_guess);
    ik->best.extinction = 0.0;
    ik->best_curr = ik->best;
    for (int i = 0; i < ik->population_size; ++i) {
        ik->extinction_grading[i] = (double)i / (double)(ik->population_size - 1);
    }
    ik->inverse_gene_size = 1.0 / (double)n;
    ik->rng = rng;
}

static void memetic_free(memetic_t* ik) {
    free(ik->population);
    free(ik->mating_pool);
    free(ik->extinction_grading);
}

/* MemeticIk::checkWipeout -- src/ik_memetic.cpp:43-55 */
static int check_wipeout(memetic_t* ik, const pko_params* p) {
    if (ik->has_previous_fitness) {
        const int improved =
            (ik->best_curr.fitness < ik->previous_fitness - p->memetic_wipeout_fitness_tol);
        if (!improved) return 1;
    }
    ik->previous_fitness = ik->best_curr.fitness;
    ik->has_previous_fitness = 1;
    return 0;
}

/* MemeticIk::computeExtinctions -- src/ik_memetic.cpp:57-64 */
static void compute_extinctions(memetic_t* ik) {
    const double min_fitness = ik->population[0].fitness;
    const double max_fitness = ik->population[ik->population_size - 1].fitness;
    for (int i = 0; i < ik->population_size; ++i) {
        ik->population[i].extinction =
            (ik->population[i].fitness + min_fitness * (ik->extinction_grading[i] - 1)) /
            max_fitness;
    }
}

/* MemeticIk::gradientDescent -- src/ik_memetic.cpp:66-91 (5 ms wall budget disabled) */
static void gradient_descent(memetic_t* ik, int i, problem_t* pb) {
    const pko_params* p = pb->params;
    const int n = pb->chain->dof;
    individual_t* individual = &ik->population[i];
    gradient_ik_t local_ik;
    gradient_ik_from(&local_ik, pb, individual->genes);

    int num_iterations = 0;
    double previous_cost = 0;
    while (num_iterations < p->memetic_gd_max_iters) {
        gd_step(&local_ik, pb, p->gd_step_size);
        if (fabs(local_ik.local_cost - previous_cost) <= p->gd_min_cost_delta) break;
        previous_cost = local_ik.local_cost;
        num_iterations++;
    }
    for (int j = 0; j < n; ++j) individual->genes[j] = local_ik.best[j];
    individual->fitness = cost_fn(pb, individual->genes);
    for (int j = 0; j < n; ++j) individual->gradient[j] = local_ik.gradient[j];
}

/* Robot::set_random_valid_configuration -- src/robot.cpp:87-95, Variable::generate_valid_value
 * -- src/robot.cpp:23-30.  u(j) supplies the j-th uniform [0,1) draw. */
static void set_random_valid_configuration(const pko_chain* c, double* config, uint64_t seed,
                                           uint32_t stream, uint64_t problem, uint32_t epoch,
                                           uint32_t individual, int repro_layout) {
    for (int j = 0; j < c->dof; ++j) {
        const variable_t* v = &c->var[j];
        const uint32_t slot = repro_layout ? (uint32_t)(2 * (1 + j) + 1) : (uint32_t)j;
        const double u = pko_rng_u01(seed, stream, problem, epoch, individual, slot);
        if (v->bounded) {
            config[j] = uniform_real(v->min, v->max, u);
        } else {
            config[j] = uniform_real(config[j] - M_PI, config[j] + M_PI, u);
        }
    }
}

/* MemeticIk::initPopulation -- src/ik_memetic.cpp:93-117 */
static void init_population(memetic_t* ik, problem_t* pb, const double* initial_guess) {
    const int n = pb->chain->dof;
    double guess[PKO_MAX_DOF];
    memcpy(guess, initial_guess, sizeof(double) * (size_t)n); /* may alias ik->best.genes */
    const uint32_t epoch = ik->init_epoch++;
    for (int i = 0; i < ik->elite_size; ++i) {
        individual_t* ind = &ik->population[i];
        double genotype[PKO_MAX_DOF];
        memcpy(genotype, guess, sizeof(double) * (size_t)n);
        if (i > 0) {
            set_random_valid_configuration(pb->chain, genotype, ik->rng.seed, STREAM_INIT,
                                           ik->rng.problem, epoch,
                                           (uint32_t)i | (ik->rng.species << 20), 0);
        }
        memcpy(ind->genes, genotype, sizeof(double) * (size_t)n);
        ind->fitness = cost_fn(pb, genotype);
        ind->extinction = 1.0;
        for (int j = 0; j < n; ++j) ind->gradient[j] = 0.0;
        ind->slot = i;
    }
    for (int i = ik->elite_size; i < ik->population_size; ++i) {
        individual_t* ind = &ik->population[i];
        memcpy(ind->genes, guess, sizeof(double) * (size_t)n);
        ind->fitness = 0.0;
        ind->extinction = 1.0;
        for (int j = 0; j < n; ++j) ind->gradient[j] = 0.0;
        ind->slot = i;
    }
    for (int i = 0; i < ik->population_size; ++i) {
        ik->population[i].fitness = cost_fn(pb, ik->population[i].genes);
    }
    compute_extinctions(ik);
    ik->has_previous_fitness = 0;
}

static void pool_erase(memetic_t* ik, int individual_index) {
    for (int k = 0; k < ik->pool_size; ++k) {
        if (ik->mating_pool[k] == individual_index) {
            for (int m = k; m + 1 < ik->pool_size; ++m) ik->mating_pool[m] = ik->mating_pool[m + 1];
            ik->pool_size--;
            ik->erasures++;
            return;
        }
    }
}

/* MemeticIk::reproduce -- src/ik_memetic.cpp:119-190 */
static void reproduce(memetic_t* ik, problem_t* pb, uint32_t generation) {
    const pko_chain* c = pb->chain;
    const int n = c->dof;
    const uint64_t seed = ik->rng.seed, problem = ik->rng.problem;
    ik->pool_size = ik->elite_size;
    for (int i = 0; i < ik->elite_size; ++i) ik->mating_pool[i] = i;

    for (int i = ik->elite_size; i < ik->population_size; ++i) {
        individual_t* child = &ik->population[i];
        const uint32_t ind = (uint32_t)i | (ik->rng.species << 20);
        child->slot = i;
        if (ik->pool_size > 0) {
            uint32_t w[4];
            rng_block(seed, STREAM_REPRODUCE, problem, generation, ind, 0, w);
            const uint32_t pool = (uint32_t)ik->pool_size;
            const int idxA = (int)(((uint64_t)w[0] * pool) >> 32);
            const double mix_ratio = u01_from_words(w[2], w[3]);
            /* idxB: "draw until different from idxA" (src/ik_memetic.cpp:132-135) has the uniform
             * distribution over the OTHER pool members; drawn here directly from that distribution
             * with one word (index among the others, skipping idxA) -- a rejection loop of data-
             * dependent length is what a lock-step GPU pays most for (6 % of the whole solve). */
            int idxB = idxA;
            if (ik->pool_size > 1) {
                idxB = (int)(((uint64_t)w[1] * (pool - 1u)) >> 32);
                idxB += (idxB >= idxA) ? 1 : 0;
            }
            const int ia = ik->mating_pool[idxA], ib = ik->mating_pool[idxB];
            const individual_t* parentA = &ik->population[ia];
            const individual_t* parentB = &ik->population[ib];

            const double extinction = 0.5 * (parentA->extinction + parentB->extinction);
            const double mutation_prob =
                extinction * (1.0 - ik->inverse_gene_size) + ik->inverse_gene_size;

            for (int j = 0; j < n; ++j) {
                uint32_t wj[4];
                rng_block(seed, STREAM_REPRODUCE, problem, generation, ind, (uint32_t)(1 + j), wj);
                const variable_t* joint = &c->var[j];
                double gene =
                    mix_ratio * parentA->genes[j] + (1.0 - mix_ratio) * parentB->genes[j];
                gene += u01_from_word(wj[0]) * parentA->gradient[j] +
                        u01_from_word(wj[1]) * parentB->gradient[j];
                const double original_gene = gene;
                if (u01_from_word(wj[2]) < mutation_prob) {
                    gene += extinction * joint->half_span *
                            uniform_real(-1.0, 1.0, u01_from_word(wj[3]));
                }
                gene = clamp_to_limits(joint, gene);
                child->genes[j] = gene;
                child->gradient[j] = gene - original_gene;
            }

            child->fitness = cost_fn(pb, child->genes);
            const double fa = parentA->fitness, fb = parentB->fitness;
            if (child->fitness < fa) pool_erase(ik, ia);
            if (child->fitness < fb) pool_erase(ik, ib);
        } else {
            set_random_valid_configuration(c, child->genes, seed, STREAM_REPRODUCE, problem,
                                           generation, ind, 1);
            child->fitness = cost_fn(pb, child->genes);
            for (int j = 0; j < n; ++j) child->gradient[j] = 0.0;
        }
    }
}

static int cmp_individual(const void* a, const void* b) {
    const individual_t* x = (const individual_t*)a;
    const individual_t* y = (const individual_t*)b;
    if (x->fitness < y->fitness) return -1;
    if (y->fitness < x->fitness) return 1;
    return (x->slot > y->slot) - (x->slot < y->slot);
}

/* MemeticIk::sortPopulation -- src/ik_memetic.cpp:200-209.  std::sort is unstable; ties are
 * broken here by pre-sort slot index so that every implementation agrees. */
static void sort_population(memetic_t* ik) {
    for (int i = 0; i < ik->population_size; ++i) ik->population[i].slot = i;
    qsort(ik->population, (size_t)ik->population_size, sizeof(individual_t), cmp_individual);
    compute_extinctions(ik);
    ik->best_curr = ik->population[0];
    if (ik->best_curr.fitness < ik->best.fitness) ik->best = ik->best_curr;
}

/* one pass of the body of the while loop in ik_memetic_impl -- src/ik_memetic.cpp:229-261.
 * returns 1 when the reference would `return ik.best()` at :252-255. */
static int memetic_generation(memetic_t* ik, problem_t* pb, uint32_t iter) {
    const pko_params* p = pb->params;
    for (int i = 0; i < ik->elite_size; ++i) gradient_descent(ik, i, pb);
    reproduce(ik, pb, iter);
    sort_population(ik);
    if (p->stop_optimization_on_valid_solution && solution_fn(pb, ik->best.genes)) return 1;
    if (check_wipeout(ik, p)) {
        ik->wipeouts++;
        init_population(ik, pb, ik->best.genes);
    }
    return 0;
}

/* ik_memetic -- src/ik_memetic.cpp:285-373 with ik_memetic_impl :211-283 inlined.
 * num_threads > 1 ("species") is restated as a lock-step schedule of the reference's race: all
 * species advance one generation at a time; the first generation in which any species returns a
 * solution ends the race (`terminate`), the lowest-numbered such species is the one "popped first",
 * and the remaining species contribute their best only when approx_solution is set (:356-370). */
static int ik_memetic(problem_t* pb, const double* initial_guess, uint64_t rng_seed,
                      uint64_t problem_index, int approx_solution, double* out, double* out_cost,
                      int* valid, pko_stats* st) {
    const pko_params* p = pb->params;
    const int n = pb->chain->dof;
    if (p->stop_optimization_on_valid_solution && solution_fn(pb, initial_guess)) {
        memcpy(out, initial_guess, sizeof(double) * (size_t)n);
        *out_cost = cost_fn(pb, initial_guess);
        pb->evals--;
        *valid = 1;
        return 1;
    }
    const int S = p->memetic_num_threads <= 1 ? 1 : p->memetic_num_threads;
    memetic_t* ik = (memetic_t*)calloc((size_t)S, sizeof(memetic_t));
    int* state = (int*)calloc((size_t)S, sizeof(int)); /* 0 running, 1 returned solution */
    for (int s = 0; s < S; ++s) {
        rng_t rng = {rng_seed, problem_index, (uint32_t)s};
        memetic_from(&ik[s], pb, initial_guess, rng);
        init_population(&ik[s], pb, initial_guess);
    }

    int iter = 0;
    int terminate = 0;
    while (iter < p->memetic_max_generations && !terminate) {
        int running = 0;
        for (int s = 0; s < S; ++s) {
            if (state[s]) continue; /* this species already returned (only when !stop_on_first) */
            if (memetic_generation(&ik[s], pb, (uint32_t)iter)) {
                state[s] = 1;
                if (S == 1 || p->memetic_stop_on_first_solution) terminate = 1;
            } else {
                running++;
            }
        }
        if (running == 0) terminate = 1;
        iter++;
    }
    if (st) {
        st->generations = iter;
        st->wipeouts = ik[0].wipeouts;
        st->pool_erasures = ik[0].erasures;
    }

    /* collect results the way :299-311 (S==1) / :337-371 (S>1) do */
    int have = 0;
    double min_cost = 1.7976931348623157e308;
    *valid = 0;
    for (int s = 0; s < S; ++s) {
        int has_value = 0, is_valid = 0;
        if (state[s]) {
            has_value = 1;
            is_valid = 1;
        } else {
            /* post-loop of ik_memetic_impl :272-282 */
            if (!p->stop_optimization_on_valid_solution && solution_fn(pb, ik[s].best.genes)) {
                has_value = 1;
                is_valid = 1;
            } else if (approx_solution) {
                has_value = 1;
            }
        }
        if (has_value && ik[s].best.fitness < min_cost) {
            min_cost = ik[s].best.fitness;
            memcpy(out, ik[s].best.genes, sizeof(double) * (size_t)n);
            *out_cost = ik[s].best.fitness;
            *valid = is_valid;
            have = 1;
        }
    }
    for (int s = 0; s < S; ++s) memetic_free(&ik[s]);
    free(ik);
    free(state);
    return have;
}

/* ------------------------------------------------------------------------------------------
 * Public API
 * ---------------------------------------------------------------------------------------- */

void pko_default_params(pko_params* p) {
    /* src/pick_ik_parameters.yaml defaults */
    p->mode = 0;
    p->gd_step_size = 0.0001;
    p->gd_max_iters = 100;
    p->gd_min_cost_delta = 1.0e-12;
    p->position_threshold = 0.001;
    p->orientation_threshold = 0.001;
    p->cost_threshold = 0.001;
    p->position_scale = 1.0;
    p->rotation_scale = 0.5;
    p->center_joints_weight = 0.0;
    p->avoid_joint_limits_weight = 0.0;
    p->minimal_displacement_weight = 0.0;
    p->stop_optimization_on_valid_solution = 1;
    p->memetic_num_threads = 1;
    p->memetic_stop_on_first_solution = 1;
    p->memetic_population_size = 16;
    p->memetic_elite_size = 4;
    p->memetic_wipeout_fitness_tol = 0.00001;
    p->memetic_max_generations = 100;
    p->memetic_gd_max_iters = 25;
    p->return_approximate_solution = 0;
}

/* Robot::from -- src/robot.cpp:44-85 (variable table and minimal displacement factors) */
static void variables_init(pko_chain* c, int dof, const double* qmin, const double* qmax,
                           const double* vmax, const uint8_t* bounded) {
    /* Robot::from -- src/robot.cpp:44-85 */
    double minimal_displacement_divisor = 0.0;
    for (int j = 0; j < dof; ++j) {
        variable_t* v = &c->var[j];
        v->bounded = bounded ? bounded[j] : 1;
        v->min = qmin[j];
        v->max = qmax[j];
        v->mid = 0.5 * (v->min + v->max);
        v->half_span = v->bounded ? (v->max - v->min) / 2.0 : M_PI;
        const double max_velocity = vmax ? vmax[j] : 0.0;
        v->max_velocity_rcp = max_velocity > 0.0 ? 1.0 / max_velocity : 0.0;
        v->minimal_displacement_factor = 1.0 / (double)dof;
        minimal_displacement_divisor += v->max_velocity_rcp;
    }
    if (minimal_displacement_divisor > 0) {
        for (int j = 0; j < dof; ++j) {
            c->var[j].minimal_displacement_factor =
                c->var[j].max_velocity_rcp / minimal_displacement_divisor;
        }
    }
}

static void path_init(path_t* p, int n, const int32_t* variable, const double* origin_xyz_rpy,
                      const double* axis, const int32_t* joint_type, const double* tip_xyz_rpy) {
    p->n = n;
    for (int j = 0; j < n; ++j) {
        p->var[j] = variable ? variable[j] : j;
        rpy_xyz_to_iso(origin_xyz_rpy + 6 * j, &p->origin[j]);
        p->origin_is_identity[j] = iso_is_identity(&p->origin[j]);
        const double ax = axis[3 * j], ay = axis[3 * j + 1], az = axis[3 * j + 2];
        const double nrm = sqrt(ax * ax + ay * ay + az * az);
        p->axis[j][0] = ax / nrm;
        p->axis[j][1] = ay / nrm;
        p->axis[j][2] = az / nrm;
        p->joint_type[j] = joint_type ? joint_type[j] : PKO_JOINT_REVOLUTE;
        /* PlanarJointModel::computeTransform = Translation(x, y, 0) * AngleAxis(theta, UnitZ): the
         * three variables as elementary joints of the joint frame (see pik_oracle.h) */
        if (p->joint_type[j] >= PKO_JOINT_PLANAR_X && p->joint_type[j] <= PKO_JOINT_PLANAR_THETA) {
            const int k = p->joint_type[j] - PKO_JOINT_PLANAR_X; /* 0 x, 1 y, 2 theta */
            if (k > 0) {
                static const double zero6[6] = {0, 0, 0, 0, 0, 0};
                rpy_xyz_to_iso(zero6, &p->origin[j]);
                p->origin_is_identity[j] = 1;
            }
            p->axis[j][0] = k == 0 ? 1.0 : 0.0;
            p->axis[j][1] = k == 1 ? 1.0 : 0.0;
            p->axis[j][2] = k == 2 ? 1.0 : 0.0;
            p->joint_type[j] = k == 2 ? PKO_JOINT_REVOLUTE : PKO_JOINT_PRISMATIC;
        }
        /* a floating joint acts at its seventh variable, with the origin its first one carries */
        if (p->joint_type[j] == PKO_JOINT_FLOATING_RW && j >= 6) {
            rpy_xyz_to_iso(origin_xyz_rpy + 6 * (j - 6), &p->origin[j]);
            p->origin_is_identity[j] = iso_is_identity(&p->origin[j]);
        }
        if (p->joint_type[j] >= PKO_JOINT_FLOATING_TX && p->joint_type[j] <= PKO_JOINT_FLOATING_RW) {
            p->axis[j][0] = p->axis[j][1] = 0.0; /* (unused) */
            p->axis[j][2] = 1.0;
        }
    }
    rpy_xyz_to_iso(tip_xyz_rpy, &p->tip);
    p->tip_is_identity = iso_is_identity(&p->tip);
}

pko_chain* pko_chain_create(int32_t dof, const double* origin_xyz_rpy, const double* axis,
                            const int32_t* joint_type, const double* tip_xyz_rpy,
                            const double* qmin, const double* qmax, const double* vmax,
                            const uint8_t* bounded) {
    if (dof < 1 || dof > PKO_MAX_DOF) return NULL;
    pko_chain* c = (pko_chain*)calloc(1, sizeof *c);
    c->dof = dof;
    c->n_tips = 1;
    path_init(&c->path[0], dof, NULL, origin_xyz_rpy, axis, joint_type, tip_xyz_rpy);
    variables_init(c, dof, qmin, qmax, vmax, bounded);
    return c;
}

/* Several tip links (tip_frames of the plugin): tip k is reached through tip_n_joints[k] joints,
 * described like a chain of their own, the j-th of which moves variable variable[...] of the
 * active-variable vector.  The per-tip arrays are concatenated in tip order. */
pko_chain* pko_chain_create_multi(int32_t dof, int32_t n_tips, const int32_t* tip_n_joints,
                                  const int32_t* variable, const double* origin_xyz_rpy,
                                  const double* axis, const int32_t* joint_type,
                                  const double* tip_xyz_rpy, const double* qmin, const double* qmax,
                                  const double* vmax, const uint8_t* bounded) {
    if (dof < 1 || dof > PKO_MAX_DOF || n_tips < 1 || n_tips > PKO_MAX_TIPS) return NULL;
    pko_chain* c = (pko_chain*)calloc(1, sizeof *c);
    c->dof = dof;
    c->n_tips = n_tips;
    int off = 0;
    for (int k = 0; k < n_tips; ++k) {
        const int n = tip_n_joints[k];
        if (n < 0 || n > dof) {
            free(c);
            return NULL;
        }
        for (int j = 0; j < n; ++j) {
            const int v = variable[off + j];
            if (v < 0 || v >= dof || (j > 0 && v <= variable[off + j - 1])) {
                free(c);
                return NULL;
            }
        }
        path_init(&c->path[k], n, variable + off, origin_xyz_rpy + 6 * off, axis + 3 * off,
                  joint_type ? joint_type + off : NULL, tip_xyz_rpy + 6 * k);
        off += n;
    }
    variables_init(c, dof, qmin, qmax, vmax, bounded);
    return c;
}

int32_t pko_chain_n_tips(const pko_chain* c) { return c->n_tips; }

/* mimic joints (pick_ik_amd.h pikamd_set_mimic_joints: same meaning, same layout) */
int32_t pko_chain_set_mimic(pko_chain* c, int32_t n, const pko_mimic_joint* joints) {
    if (!c || n < 0 || (n > 0 && !joints)) return -1;
    for (int k = 0; k < c->n_tips; ++k) c->path[k].n_mimic = 0;
    for (int i = 0; i < n; ++i) {
        const pko_mimic_joint* m = &joints[i];
        if (m->tip < 0 || m->tip >= c->n_tips) return -2;
        path_t* p = &c->path[m->tip];
        if (p->n_mimic >= PKO_MAX_MIMIC || m->after_variable < -1 || m->after_variable >= c->dof ||
            m->master_variable < 0 || m->master_variable >= c->dof ||
            (m->joint_type != PKO_JOINT_REVOLUTE && m->joint_type != PKO_JOINT_PRISMATIC))
            return -3;
        const int k = p->n_mimic++;
        p->m_after[k] = m->after_variable;
        p->m_var[k] = m->master_variable;
        p->m_type[k] = m->joint_type;
        rpy_xyz_to_iso(m->origin_xyz_rpy, &p->m_origin[k]);
        p->m_origin_is_identity[k] = iso_is_identity(&p->m_origin[k]);
        const double nrm = sqrt(m->axis[0] * m->axis[0] + m->axis[1] * m->axis[1] + m->axis[2] * m->axis[2]);
        if (!(nrm > 0.0)) return -4;
        for (int a = 0; a < 3; ++a) p->m_axis[k][a] = m->axis[a] / nrm;
        p->m_mult[k] = m->multiplier;
        p->m_off[k] = m->offset;
    }
    return 0;
}

void pko_chain_destroy(pko_chain* c) { free(c); }

void pko_chain_variables(const pko_chain* c, double* out) {
    for (int j = 0; j < c->dof; ++j) {
        const variable_t* v = &c->var[j];
        out[7 * j + 0] = v->min;
        out[7 * j + 1] = v->max;
        out[7 * j + 2] = v->mid;
        out[7 * j + 3] = v->half_span;
        out[7 * j + 4] = v->max_velocity_rcp;
        out[7 * j + 5] = v->minimal_displacement_factor;
        out[7 * j + 6] = (double)v->bounded;
    }
}

static void iso_to12(const iso_t* a, double* p) {
    memcpy(p, a->R, sizeof a->R);
    memcpy(p + 9, a->t, sizeof a->t);
}
static void iso_from12(const double* p, iso_t* a) {
    memcpy(a->R, p, sizeof a->R);
    memcpy(a->t, p + 9, sizeof a->t);
}

void pko_fk_matrix(const pko_chain* c, const double* q, double* pose12) {
    iso_t tip[PKO_MAX_TIPS];
    fk(c, q, tip);
    for (int k = 0; k < c->n_tips; ++k) iso_to12(&tip[k], pose12 + 12 * k);
}

/* pos_quat [n][n_tips][7] */
void pko_fk_batch(const pko_chain* c, int64_t n, const double* q, double* pos_quat) {
    for (int64_t i = 0; i < n; ++i) {
        iso_t tip[PKO_MAX_TIPS];
        fk(c, q + i * c->dof, tip);
        for (int k = 0; k < c->n_tips; ++k) {
            double qt[4];
            matrix_to_quat(tip[k].R, qt);
            double* o = pos_quat + 7 * (i * c->n_tips + k);
            o[0] = tip[k].t[0];
            o[1] = tip[k].t[1];
            o[2] = tip[k].t[2];
            o[3] = qt[0];
            o[4] = qt[1];
            o[5] = qt[2];
            o[6] = qt[3];
        }
    }
}

/* tf2::fromMsg(geometry_msgs::Pose, Eigen::Isometry3d): Translation * Quaterniond(w,x,y,z)
 * (no normalisation), as used by transform_poses_to_frames -- src/robot.cpp:169-181. */
static void pose_from_pos_quat(const double* pq, iso_t* out) {
    const double q[4] = {pq[3], pq[4], pq[5], pq[6]};
    quat_to_matrix(q, out->R);
    out->t[0] = pq[0];
    out->t[1] = pq[1];
    out->t[2] = pq[2];
}

void pko_pose_from_pos_quat(const double* pos_quat7, double* pose12) {
    iso_t a;
    pose_from_pos_quat(pos_quat7, &a);
    iso_to12(&a, pose12);
}

double pko_linear_distance(const double* a12, const double* b12) {
    iso_t a, b;
    iso_from12(a12, &a);
    iso_from12(b12, &b);
    return linear_distance(&a, &b);
}
double pko_angular_distance(const double* a12, const double* b12) {
    iso_t a, b;
    iso_from12(a12, &a);
    iso_from12(b12, &b);
    return angular_distance(&a, &b);
}
double pko_pose_cost(const double* goal12, const double* frame12, double position_scale,
                     double rotation_scale) {
    iso_t a, b;
    iso_from12(goal12, &a);
    iso_from12(frame12, &b);
    return pose_cost(&a, &b, position_scale, rotation_scale);
}
int32_t pko_frame_test(const double* goal12, const double* frame12, int32_t has_pos, double pos_thr,
                       int32_t has_ori, double ori_thr) {
    iso_t a, b;
    iso_from12(goal12, &a);
    iso_from12(frame12, &b);
    return frame_test(&a, &b, has_pos, pos_thr, has_ori, ori_thr);
}
double pko_center_joints_cost(const pko_chain* c, const double* q) {
    return center_joints_cost(c, q);
}
double pko_avoid_joint_limits_cost(const pko_chain* c, const double* q) {
    return avoid_joint_limits_cost(c, q);
}
double pko_minimal_displacement_cost(const pko_chain* c, const double* q, const double* seed) {
    return minimal_displacement_cost(c, q, seed);
}

static void problem_init(problem_t* pb, const pko_chain* c, const pko_params* p,
                         const double* goal_pos_quat, const double* seed) {
    pb->chain = c;
    pb->params = p;
    for (int k = 0; k < c->n_tips; ++k) pose_from_pos_quat(goal_pos_quat + 7 * k, &pb->goal[k]);
    pb->seed = seed;
    /* thresholds are only set when the matching scale is > 0 -- src/pick_ik_plugin.cpp:97-106 */
    pb->has_pos_thr = p->position_scale > 0;
    pb->has_ori_thr = p->rotation_scale > 0;
    pb->evals = 0;
    pb->cb = NULL;
    pb->cb_user = NULL;
}

void pko_cost_batch(const pko_chain* c, const pko_params* p, const double* goal_pos_quat,
                    const double* seed, int64_t n, const double* q, double* cost,
                    int32_t* is_solution) {
    problem_t pb;
    problem_init(&pb, c, p, goal_pos_quat, seed);
    for (int64_t i = 0; i < n; ++i) {
        if (cost) cost[i] = cost_fn(&pb, q + i * c->dof);
        if (is_solution) is_solution[i] = solution_fn(&pb, q + i * c->dof);
    }
}

void pko_gd_step_batch(const pko_chain* c, const pko_params* p, int64_t n,
                       const double* goal_pos_quat, const double* seed, double* local,
                       double* best, double* local_cost, double* best_cost, double* gradient,
                       int32_t* improved) {
    const int d = c->dof;
    for (int64_t i = 0; i < n; ++i) {
        problem_t pb;
        problem_init(&pb, c, p, goal_pos_quat + 7 * c->n_tips * i, seed + i * d);
        gradient_ik_t g;
        memset(&g, 0, sizeof g);
        for (int j = 0; j < d; ++j) {
            g.local[j] = local[i * d + j];
            g.working[j] = local[i * d + j];
            g.best[j] = best[i * d + j];
        }
        g.local_cost = local_cost[i];
        g.best_cost = best_cost[i];
        const int imp = gd_step(&g, &pb, p->gd_step_size);
        for (int j = 0; j < d; ++j) {
            local[i * d + j] = g.local[j];
            best[i * d + j] = g.best[j];
            gradient[i * d + j] = g.gradient[j];
        }
        local_cost[i] = g.local_cost;
        best_cost[i] = g.best_cost;
        if (improved) improved[i] = imp;
    }
}

int32_t pko_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* The plugin keeps TWO joint vectors per call (src/pick_ik_plugin.cpp:199-245): ik_seed_state, which
 * the minimal-displacement cost measures against and which is returned on failure, and init_state,
 * the start of the search -- equal to ik_seed_state on the first attempt, re-randomised on restarts
 * (:241-245).  `seed` is the former, `initial_guess` the latter (NULL = seed). */
int32_t pko_solve_batch_guess(const pko_chain* c, const pko_params* p, int64_t B,
                              const double* goal_pos_quat, const double* seed,
                              const double* initial_guess, uint64_t rng_seed,
                              int64_t problem_offset, double* solution, int32_t* status,
                              double* final_cost, pko_stats* stats, int32_t num_threads) {
    return pko_solve_batch_cost_fn(c, p, B, goal_pos_quat, seed, initial_guess, rng_seed, problem_offset, NULL, NULL,
                                   solution, status, final_cost, stats, num_threads);
}

/* ... with a host cost function (NULL: none).  With one, the problems are solved on ONE thread (the callback may
 * be a Python function). */
int32_t pko_solve_batch_cost_fn(const pko_chain* c, const pko_params* p, int64_t B,
                                const double* goal_pos_quat, const double* seed,
                                const double* initial_guess, uint64_t rng_seed,
                                int64_t problem_offset, pko_cost_fn cost_function, void* user,
                                double* solution, int32_t* status,
                                double* final_cost, pko_stats* stats, int32_t num_threads) {
    if (cost_function) num_threads = 1;
    if (!c || !p || B < 0) return -1;
    if (p->mode == 0 && (p->memetic_elite_size < 1 ||
                         p->memetic_population_size <= p->memetic_elite_size))
        return -2;
    const int d = c->dof;
    (void)is_valid;
#ifdef _OPENMP
    if (num_threads < 1) num_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads)
#endif
    for (int64_t b = 0; b < B; ++b) {
        problem_t pb;
        const double* sd = seed + b * d;
        const double* ig = initial_guess ? initial_guess + b * d : sd;
        problem_init(&pb, c, p, goal_pos_quat + 7 * c->n_tips * b, sd);
        pb.cb = cost_function;
        pb.cb_user = user;
        double out[PKO_MAX_DOF];
        double out_cost = 0.0;
        int valid = 0;
        int have;
        pko_stats st;
        memset(&st, 0, sizeof st);
        if (p->mode == 0) {
            have = ik_memetic(&pb, ig, rng_seed, (uint64_t)(problem_offset + b),
                              p->return_approximate_solution, out, &out_cost, &valid, &st);
        } else {
            int iters = 0;
            have = ik_gradient(&pb, ig, p->return_approximate_solution, out, &out_cost, &valid,
                               &iters);
            st.generations = iters;
        }
        st.cost_evals = pb.evals;
        if (have) {
            memcpy(solution + b * d, out, sizeof(double) * (size_t)d);
            status[b] = valid ? PKO_SUCCESS : PKO_APPROXIMATE;
            if (final_cost) final_cost[b] = out_cost;
        } else {
            /* solution = ik_seed_state on failure -- src/pick_ik_plugin.cpp:213-217 */
            memcpy(solution + b * d, sd, sizeof(double) * (size_t)d);
            status[b] = PKO_NO_IK_SOLUTION;
            if (final_cost) final_cost[b] = cost_fn(&pb, ig); /* cost of the initial guess */
        }
        if (stats) stats[b] = st;
    }
    return 0;
}

int32_t pko_solve_batch(const pko_chain* c, const pko_params* p, int64_t B,
                        const double* goal_pos_quat, const double* seed, uint64_t rng_seed,
                        int64_t problem_offset, double* solution, int32_t* status,
                        double* final_cost, pko_stats* stats, int32_t num_threads) {
    return pko_solve_batch_guess(c, p, B, goal_pos_quat, seed, NULL, rng_seed, problem_offset,
                                 solution, status, final_cost, stats, num_threads);
}
