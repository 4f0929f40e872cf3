// pik_host.hpp -- host-side model extraction for the C ABI (no device code).
//
// Turns a pikamd_chain (URDF-style origins/axes/limits) into the wave-uniform ChainK<D> the kernels
// take as a kernel argument, and pikamd_params into ParamsK.
//
// Reference semantics:
//   Robot::from                src/robot.cpp:44-85   (variable table, minimal displacement factors)
//   joint origin transforms    urdf::Rotation::setFromRPY + Eigen toRotationMatrix, as MoveIt's
//                              LinkModel::joint_origin_transform_ is built
//   parameter mapping          src/pick_ik_plugin.cpp:97-106 (thresholds unset when scale is 0),
//                              :118-129 (goal order), :165-196 (solver params)
#pragma once

#include <cmath>
#include <cstring>

#include "../../include/pick_ik_amd.h"
#include "pik_math.hpp"

// Everything in this header runs on the HOST and produces the constants the kernels (and the host executors of
// tests / pik_host_solve.hpp) read.  Its arithmetic must not depend on how a translation unit is compiled -- the
// library's host pass has no fused multiply-add, a test program built with -mfma would fuse and get Denavit-
// Hartenberg constants that differ in the last bit --, so every function here switches contraction off for itself.
namespace pik {

struct ChainHost {
    int dof = 0;
    double O[PIKAMD_MAX_DOF][12];
    double axis[PIKAMD_MAX_DOF][3];
    double tip[12];
    // Denavit-Hartenberg form used by the fast build (see ChainK::dh)
    double dh_base[12];
    double dh[PIKAMD_MAX_DOF][6]; // theta0, d, a, cos(alpha), sin(alpha), 0
    double dh_tip[12];
    double dhg[PIKAMD_MAX_DOF][12]; // general step after joint j (dh_general_mask), see ChainK::dhg
    double qmin[PIKAMD_MAX_DOF], qmax[PIKAMD_MAX_DOF], mid[PIKAMD_MAX_DOF], hspan[PIKAMD_MAX_DOF],
        mdf[PIKAMD_MAX_DOF], vrcp[PIKAMD_MAX_DOF];
    uint32_t origin_ident_mask = 0, prismatic_mask = 0, bounded_mask = 0, axis_kind = 0,
             tip_ident = 0, active_mask = 0, dh_general_mask = 0;
    uint32_t float_mask = 0, skip_mask = 0; // floating joints (see ChainK)
    // mimic joints of this path (pikamd_set_mimic_joints): steps of the chain product whose value follows a variable
    int n_mimic = 0;
    double mO[PIKAMD_MAX_MIMIC][12], maxis[PIKAMD_MAX_MIMIC][3], mmult[PIKAMD_MAX_MIMIC], moff[PIKAMD_MAX_MIMIC];
    int m_after[PIKAMD_MAX_MIMIC], m_var[PIKAMD_MAX_MIMIC];
    uint32_t m_ident_mask = 0, m_pris_mask = 0, m_kind = 0;
};

inline void xyz_rpy_to_iso12(const double* xyz_rpy, double* o12) {
#pragma clang fp contract(off) // (host-side model arithmetic: the same numbers whatever the compiler flags)

    const double phi = xyz_rpy[3] / 2.0, the = xyz_rpy[4] / 2.0, psi = xyz_rpy[5] / 2.0;
    // glibc sincos() -- what GCC emits for a sin/cos pair of one argument, i.e. the usual build of
    // urdfdom's setFromRPY.  Called explicitly: clang keeps separate sin() and cos() calls, and
    // glibc's cos() differs from its sincos() by 1 ulp for some arguments, which made the origin
    // matrices (and every FK after them) depend on the host compiler.
    double sphi, cphi, sthe, cthe, spsi, cpsi;
    ::sincos(phi, &sphi, &cphi);
    ::sincos(the, &sthe, &cthe);
    ::sincos(psi, &spsi, &cpsi);
    double q[4];
    q[1] = sphi * cthe * cpsi - cphi * sthe * spsi;
    q[2] = cphi * sthe * cpsi + sphi * cthe * spsi;
    q[3] = cphi * cthe * spsi - sphi * sthe * cpsi;
    q[0] = cphi * cthe * cpsi + sphi * sthe * spsi;
    const double s = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (s == 0.0) {
        q[0] = 1.0;
        q[1] = q[2] = q[3] = 0.0;
    } else {
        for (double& v : q) v /= s;
    }
    double R[9];
    quat_to_matrix(q, R);
    std::memcpy(o12, R, sizeof R);
    o12[9] = xyz_rpy[0];
    o12[10] = xyz_rpy[1];
    o12[11] = xyz_rpy[2];
}

static_assert(PIKAMD_MAX_MIMIC == 4, "ChainK (pik_math.hpp) holds four mimic joints per path");

// one mimic joint into a path's description (order of declaration = order along the path among the joints that
// follow the same variable)
inline const char* add_mimic_joint(ChainHost& c, const pikamd_mimic_joint& m) {
#pragma clang fp contract(off) // (host-side model arithmetic: the same numbers whatever the compiler flags)

    if (c.n_mimic >= PIKAMD_MAX_MIMIC) return "too many mimic joints on one path (PIKAMD_MAX_MIMIC)";
    if (m.after_variable < -1 || m.after_variable >= c.dof) return "mimic joint: after_variable out of range";
    if (m.master_variable < 0 || m.master_variable >= c.dof) return "mimic joint: master_variable out of range";
    if (m.joint_type != PIKAMD_JOINT_REVOLUTE && m.joint_type != PIKAMD_JOINT_PRISMATIC)
        return "mimic joint: must be revolute or prismatic";
    bool finite = std::isfinite(m.multiplier) && std::isfinite(m.offset);
    for (int i = 0; i < 6; ++i) finite = finite && std::isfinite(m.origin_xyz_rpy[i]);
    for (int i = 0; i < 3; ++i) finite = finite && std::isfinite(m.axis[i]);
    if (!finite) return "mimic joint: non-finite origin / axis / multiplier / offset";
    const int k = c.n_mimic;
    xyz_rpy_to_iso12(m.origin_xyz_rpy, c.mO[k]);
    const double n = std::sqrt(m.axis[0] * m.axis[0] + m.axis[1] * m.axis[1] + m.axis[2] * m.axis[2]);
    if (!(n > 0.0)) return "mimic joint: zero axis";
    for (int i = 0; i < 3; ++i) c.maxis[k][i] = m.axis[i] / n;
    uint32_t kind = 0; // AXIS_GENERAL
    if (c.maxis[k][0] == 1.0 && c.maxis[k][1] == 0.0 && c.maxis[k][2] == 0.0) kind = 1;
    if (c.maxis[k][0] == 0.0 && c.maxis[k][1] == 1.0 && c.maxis[k][2] == 0.0) kind = 2;
    if (c.maxis[k][0] == 0.0 && c.maxis[k][1] == 0.0 && c.maxis[k][2] == 1.0) kind = 3;
    c.m_kind |= kind << (2 * k);
    c.mmult[k] = m.multiplier;
    c.moff[k] = m.offset;
    c.m_after[k] = m.after_variable;
    c.m_var[k] = m.master_variable;
    bool ident = true;
    for (int i = 0; i < 12; ++i) ident = ident && c.mO[k][i] == ((i == 0 || i == 4 || i == 8) ? 1.0 : 0.0);
    if (ident) c.m_ident_mask |= 1u << k;
    if (m.joint_type == PIKAMD_JOINT_PRISMATIC) c.m_pris_mask |= 1u << k;
    c.n_mimic = k + 1;
    return nullptr;
}
inline void clear_mimic_joints(ChainHost& c) {
    c.n_mimic = 0;
    c.m_ident_mask = c.m_pris_mask = c.m_kind = 0;
}


inline bool iso12_is_identity(const double* o) {
    static const double I[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    for (int i = 0; i < 12; ++i)
        if (o[i] != I[i]) return false;
    return true;
}

inline void fill_math_tab(MathTab& m) {
    std::memset(&m, 0, sizeof m);
    const double v[38] = {
        0.15915494309189535, 6.283185307179586, 2.4492935982947064e-16, 0.6366197723675814,
        1.5707963267948966, 6.123233995736766e-17, -1.4973849048591698e-33,
        // S1..S6
        -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
        2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10,
        // C1..C6
        4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
        -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11,
        // aT0..aT10
        3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01,
        -1.11111104054623557880e-01, 9.09088713343650656196e-02, -7.69187620504482999495e-02,
        6.66107313738753120669e-02, -5.83357013379057348645e-02, 4.97687799461593236017e-02,
        -3.65315727442169155270e-02, 1.62858201153657823623e-02,
        // atan hi / lo
        4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01,
        1.57079632679489655800e+00, 2.26987774529616870924e-17, 3.06161699786838301793e-17,
        1.39033110312309984516e-17, 6.12323399573676603587e-17};
    std::memcpy(m.v, v, sizeof v);
}

// ---- small 3-vector helpers for the DH construction ----
inline void v_cross(const double* a, const double* b, double* o) {
#pragma clang fp contract(off) // (host-side model arithmetic: the same numbers whatever the compiler flags)

    const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
inline double v_dot(const double* a, const double* b) {
#pragma clang fp contract(off) // (host-side model arithmetic: the same numbers whatever the compiler flags)
 return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double v_norm(const double* a) {
#pragma clang fp contract(off) // (host-side model arithmetic: the same numbers whatever the compiler flags)
 return std::sqrt(v_dot(a, a)); }
// pose12 (R row-major | t) from frame axes x, z (unit, orthogonal) and origin
inline void pose_from_xz(const double* x, const double* z, const double* org, double* p12) {
#pragma clang fp contract(off) // (host-side model arithmetic: the same numbers whatever the compiler flags)

    double y[3];
    v_cross(z, x, y);
    for (int i = 0; i < 3; ++i) {
        p12[i * 3 + 0] = x[i];
        p12[i * 3 + 1] = y[i];
        p12[i * 3 + 2] = z[i];
        p12[9 + i] = org[i];
    }
}
// out = a * b for pose12
inline void pose_mul(const double* a, const double* b, double* out) {
#pragma clang fp contract(off) // (host-side model arithmetic: the same numbers whatever the compiler flags)

    double r[12];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            r[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
        r[9 + i] = a[i * 3 + 0] * b[9] + a[i * 3 + 1] * b[10] + a[i * 3 + 2] * b[11] + a[9 + i];
    }
    std::memcpy(out, r, sizeof r);
}
// out = a^-1 * b for pose12 (a rigid)
inline void pose_inv_mul(const double* a, const double* b, double* out) {
#pragma clang fp contract(off) // (host-side model arithmetic: the same numbers whatever the compiler flags)

    double r[12];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            r[i * 3 + j] = a[0 * 3 + i] * b[0 * 3 + j] + a[1 * 3 + i] * b[1 * 3 + j] + a[2 * 3 + i] * b[2 * 3 + j];
        r[9 + i] = a[0 * 3 + i] * (b[9] - a[9]) + a[1 * 3 + i] * (b[10] - a[10]) + a[2 * 3 + i] * (b[11] - a[11]);
    }
    std::memcpy(out, r, sizeof r);
}

// The chain in Denavit-Hartenberg form (fast build).  With frame A_j sitting on joint j's axis
// (z = the axis), the step to the next joint's frame is Rz(q_j + theta0) Tz(d) Tx(a) Rx(alpha):
// two column rotations and two axis translations (30 multiply-adds) instead of a general rigid
// product plus the joint rotation (51), and 5 constants per joint instead of 12.  The frames are
// built at q = 0 from the joint axis lines: x of A_{j+1} is the common normal of axes j and j+1,
// its origin the foot of that normal on axis j+1 (any choice along / about a joint's own axis
// commutes with the joint's motion).  Variables that are not joints of this chain (multi-tip
// padding) get the identity step.  `base` places A_first, `tip` closes the chain to the tip link.
inline void build_dh(ChainHost& c) {
#pragma clang fp contract(off) // (host-side model arithmetic: the same numbers whatever the compiler flags)

    const int D = c.dof;
    if (c.float_mask != 0u) {
        // no Denavit-Hartenberg form with a floating joint on the chain: such chains run the literal
        // forward kinematics only (the fields stay defined: identity steps)
        const double I12[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
        std::memcpy(c.dh_base, I12, sizeof I12);
        std::memcpy(c.dh_tip, I12, sizeof I12);
        c.dh_general_mask = 0;
        for (int j = 0; j < D; ++j) {
            c.dh[j][0] = c.dh[j][1] = c.dh[j][2] = c.dh[j][4] = c.dh[j][5] = 0.0;
            c.dh[j][3] = 1.0;
            std::memcpy(c.dhg[j], I12, sizeof I12);
        }
        return;
    }
    // world pose of every URDF joint frame at q = 0, axis lines
    double W[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    double P[PIKAMD_MAX_DOF][3], Z[PIKAMD_MAX_DOF][3];
    for (int j = 0; j < D; ++j) {
        pose_mul(W, c.O[j], W);
        for (int i = 0; i < 3; ++i) {
            P[j][i] = W[9 + i];
            Z[j][i] = W[i * 3 + 0] * c.axis[j][0] + W[i * 3 + 1] * c.axis[j][1] + W[i * 3 + 2] * c.axis[j][2];
        }
        const double n = v_norm(Z[j]);
        for (int i = 0; i < 3; ++i) Z[j][i] /= n;
    }
    double Wtip[12];
    pose_mul(W, c.tip, Wtip);
    c.dh_general_mask = 0;
    for (int j = 0; j < D; ++j) {
        c.dh[j][0] = c.dh[j][1] = c.dh[j][2] = c.dh[j][4] = c.dh[j][5] = 0.0;
        c.dh[j][3] = 1.0;
        for (int i = 0; i < 12; ++i) c.dhg[j][i] = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
    }
    int act[PIKAMD_MAX_DOF], m = 0;
    for (int j = 0; j < D; ++j)
        if ((c.active_mask >> j) & 1u) act[m++] = j;
    if (m == 0) { // nothing moves: base = identity, tip = the whole constant chain
        const double I12[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
        std::memcpy(c.dh_base, I12, sizeof I12);
        std::memcpy(c.dh_tip, Wtip, sizeof Wtip);
        return;
    }
    // A_0: on the first axis, x = any unit vector orthogonal to it
    double o[3], x[3], z[3];
    {
        const int j = act[0];
        for (int i = 0; i < 3; ++i) {
            o[i] = P[j][i];
            z[i] = Z[j][i];
        }
        const int k = (std::fabs(z[0]) <= std::fabs(z[1]) && std::fabs(z[0]) <= std::fabs(z[2])) ? 0
                      : (std::fabs(z[1]) <= std::fabs(z[2]) ? 1 : 2);
        double e[3] = {0, 0, 0};
        e[k] = 1.0;
        const double dz = v_dot(e, z);
        for (int i = 0; i < 3; ++i) x[i] = e[i] - dz * z[i];
        const double n = v_norm(x);
        for (int i = 0; i < 3; ++i) x[i] /= n;
    }
    pose_from_xz(x, z, o, c.dh_base);
    for (int a = 0; a + 1 < m; ++a) {
        const int j = act[a], jn = act[a + 1];
        const double* zn = Z[jn];
        double delta[3], w[3], nrm[3], foot2[3];
        for (int i = 0; i < 3; ++i) delta[i] = P[jn][i] - o[i];
        v_cross(z, zn, w);
        const double sw2 = v_dot(w, w);
        double d = 0.0, aa = 0.0;
        if (sw2 > 1e-24 && sw2 < 1e-6) {
            // Nearly but not exactly parallel axes (e.g. rpy = "1.57079632679" written for pi/2 is
            // 4.9e-12 rad off): the common normal's foot is ~L / sin(angle) away, the DH offsets d
            // of this step and the next cancel to 1e-16 L / angle -- 3e-5 m at 3e-12 rad.  Such a
            // pair gets a general constant step instead: the next frame sits at the point of the
            // next axis closest to this frame's origin, x = this x made orthogonal to the next axis.
            double An[12], Aj[12], xn[3], on[3];
            const double along = v_dot(delta, zn);
            for (int i = 0; i < 3; ++i) on[i] = P[jn][i] - along * zn[i];
            const double dzx = v_dot(x, zn);
            for (int i = 0; i < 3; ++i) xn[i] = x[i] - dzx * zn[i];
            const double n = v_norm(xn);
            for (int i = 0; i < 3; ++i) xn[i] /= n;
            pose_from_xz(x, z, o, Aj);
            pose_from_xz(xn, zn, on, An);
            pose_inv_mul(Aj, An, c.dhg[j]);
            c.dh_general_mask |= 1u << j;
            for (int i = 0; i < 3; ++i) {
                o[i] = on[i];
                x[i] = xn[i];
                z[i] = zn[i];
            }
            continue;
        }
        if (sw2 > 1e-24) { // skew or intersecting axes
            const double sw = std::sqrt(sw2);
            for (int i = 0; i < 3; ++i) nrm[i] = w[i] / sw;
            aa = v_dot(delta, nrm);
            if (aa < 0.0) {
                aa = -aa;
                for (int i = 0; i < 3; ++i) nrm[i] = -nrm[i];
            }
            double t1[3], t2[3];
            v_cross(delta, zn, t1);
            v_cross(delta, z, t2);
            const double s = v_dot(t1, w) / sw2, t = v_dot(t2, w) / sw2;
            d = s;
            for (int i = 0; i < 3; ++i) foot2[i] = P[jn][i] + t * zn[i] - o[i] + o[i];
        } else { // parallel (or anti-parallel) axes
            const double dz = v_dot(delta, z);
            double perp[3];
            for (int i = 0; i < 3; ++i) perp[i] = delta[i] - dz * z[i];
            aa = v_norm(perp);
            if (aa > 1e-12) {
                for (int i = 0; i < 3; ++i) nrm[i] = perp[i] / aa;
            } else { // the same line
                aa = 0.0;
                for (int i = 0; i < 3; ++i) nrm[i] = x[i];
            }
            d = 0.0;
            for (int i = 0; i < 3; ++i) foot2[i] = o[i] + aa * nrm[i];
        }
        // re-orthogonalise the normal against z (rounding) and measure the angles
        {
            const double dz = v_dot(nrm, z);
            for (int i = 0; i < 3; ++i) nrm[i] -= dz * z[i];
            const double n = v_norm(nrm);
            for (int i = 0; i < 3; ++i) nrm[i] /= n;
        }
        double xc[3], zc[3];
        v_cross(x, nrm, xc);
        v_cross(z, zn, zc);
        const double th0 = std::atan2(v_dot(xc, z), v_dot(x, nrm));
        const double alpha = std::atan2(v_dot(zc, nrm), v_dot(z, zn));
        c.dh[j][0] = th0;
        c.dh[j][1] = d;
        c.dh[j][2] = aa;
        ::sincos(alpha, &c.dh[j][4], &c.dh[j][3]);
        // next frame: origin = foot on the next axis, x = common normal, z = next axis
        for (int i = 0; i < 3; ++i) {
            o[i] = foot2[i];
            x[i] = nrm[i];
            z[i] = zn[i];
        }
        // x must be orthogonal to the new z as well (it is, up to rounding: normal to both axes)
        {
            const double dz = v_dot(x, z);
            for (int i = 0; i < 3; ++i) x[i] -= dz * z[i];
            const double n = v_norm(x);
            for (int i = 0; i < 3; ++i) x[i] /= n;
        }
    }
    // close the chain: tip pose relative to the last joint's frame
    double Alast[12];
    pose_from_xz(x, z, o, Alast);
    pose_inv_mul(Alast, Wtip, c.dh_tip);
}

// returns nullptr on success, else an error message
inline const char* build_chain(const pikamd_chain* in, ChainHost& c) {
#pragma clang fp contract(off) // (host-side model arithmetic: the same numbers whatever the compiler flags)

    if (!in) return "chain is NULL";
    if (in->dof < 1 || in->dof > PIKAMD_MAX_DOF) return "dof out of range [1, PIKAMD_MAX_DOF]";
    if (!in->origin_xyz_rpy || !in->axis || !in->tip_xyz_rpy || !in->qmin || !in->qmax)
        return "chain has NULL arrays";
    c.dof = in->dof;
    c.active_mask = (in->dof >= 32) ? ~0u : ((1u << in->dof) - 1u);
    double divisor = 0.0;
    for (int j = 0; j < c.dof; ++j) {
        int jt = in->joint_type ? in->joint_type[j] : PIKAMD_JOINT_REVOLUTE;
        double origin6[6], ax, ay, az;
        std::memcpy(origin6, in->origin_xyz_rpy + 6 * j, sizeof origin6);
        ax = in->axis[3 * j];
        ay = in->axis[3 * j + 1];
        az = in->axis[3 * j + 2];
        if (jt >= PIKAMD_JOINT_PLANAR_X && jt <= PIKAMD_JOINT_PLANAR_THETA) {
            // PlanarJointModel::computeTransform = Translation(x, y, 0) * AngleAxis(theta, UnitZ):
            // three elementary joints of the joint frame, the first one carrying the joint's origin
            const int k = jt - PIKAMD_JOINT_PLANAR_X;
            const int prev = j > 0 ? in->joint_type[j - 1] : -1;
            if (k > 0 && prev != jt - 1) return "planar joint: its x, y, theta variables must be consecutive";
            if (k < 2 && (j + 1 >= c.dof || in->joint_type[j + 1] != jt + 1))
                return "planar joint: its x, y, theta variables must be consecutive";
            if (k > 0) std::memset(origin6, 0, sizeof origin6);
            ax = k == 0 ? 1.0 : 0.0;
            ay = k == 1 ? 1.0 : 0.0;
            az = k == 2 ? 1.0 : 0.0;
            jt = k == 2 ? PIKAMD_JOINT_REVOLUTE : PIKAMD_JOINT_PRISMATIC;
        }
        bool floating_var = false;
        if (jt >= PIKAMD_JOINT_FLOATING_TX && jt <= PIKAMD_JOINT_FLOATING_RW) {
            // FloatingJointModel: seven consecutive variables, ONE transform applied at the seventh (rot_w)
            // with the origin the first one (trans_x) carries
            const int k = jt - PIKAMD_JOINT_FLOATING_TX;
            if (!in->joint_type) return "floating joint without joint types";
            if (j - k < 0 || j - k + 6 >= c.dof) return "floating joint: its seven variables must be consecutive";
            for (int m = 0; m < 7; ++m)
                if (in->joint_type[j - k + m] != PIKAMD_JOINT_FLOATING_TX + m)
                    return "floating joint: its seven variables must be consecutive (trans_x .. rot_w)";
            if (k < 6) {
                std::memset(origin6, 0, sizeof origin6);
                c.skip_mask |= 1u << j;
            } else {
                std::memcpy(origin6, in->origin_xyz_rpy + 6 * (j - 6), sizeof origin6);
                c.float_mask |= 1u << j;
            }
            ax = ay = 0.0;
            az = 1.0;
            jt = PIKAMD_JOINT_REVOLUTE; // (the entries below are unused for these variables)
            floating_var = true;
        }
        (void)floating_var;
        xyz_rpy_to_iso12(origin6, c.O[j]);
        if (iso12_is_identity(c.O[j])) c.origin_ident_mask |= 1u << j;
        const double n = std::sqrt(ax * ax + ay * ay + az * az);
        if (!(n > 0.0)) return "zero joint axis";
        c.axis[j][0] = ax / n;
        c.axis[j][1] = ay / n;
        c.axis[j][2] = az / n;
        uint32_t kind = AXIS_GENERAL;
        if (c.axis[j][0] == 1.0 && c.axis[j][1] == 0.0 && c.axis[j][2] == 0.0) kind = AXIS_X;
        if (c.axis[j][0] == 0.0 && c.axis[j][1] == 1.0 && c.axis[j][2] == 0.0) kind = AXIS_Y;
        if (c.axis[j][0] == 0.0 && c.axis[j][1] == 0.0 && c.axis[j][2] == 1.0) kind = AXIS_Z;
        c.axis_kind |= kind << (2 * j);
        if (jt == PIKAMD_JOINT_PRISMATIC)
            c.prismatic_mask |= 1u << j;
        else if (jt != PIKAMD_JOINT_REVOLUTE)
            return "unsupported joint type";
        const bool bounded = in->bounded ? in->bounded[j] != 0 : true;
        if (bounded) c.bounded_mask |= 1u << j;
        c.qmin[j] = in->qmin[j];
        c.qmax[j] = in->qmax[j];
        c.mid[j] = 0.5 * (c.qmin[j] + c.qmax[j]);
        c.hspan[j] = bounded ? (c.qmax[j] - c.qmin[j]) / 2.0 : M_PI;
        const double vmax = in->vmax ? in->vmax[j] : 0.0;
        c.vrcp[j] = vmax > 0.0 ? 1.0 / vmax : 0.0;
        c.mdf[j] = 1.0 / static_cast<double>(c.dof);
        divisor += c.vrcp[j];
    }
    if (divisor > 0) {
        for (int j = 0; j < c.dof; ++j) c.mdf[j] = c.vrcp[j] / divisor;
    }
    xyz_rpy_to_iso12(in->tip_xyz_rpy, c.tip);
    c.tip_ident = iso12_is_identity(c.tip) ? 1u : 0u;
    build_dh(c);
    return nullptr;
}

// Tip k of a multi-tip chain as a chain over ALL variables: the joints of its path at their
// variable indices, every other variable a joint with an identity origin that is ignored
// (active_mask).  The variables' limits are taken from the common arrays.
inline const char* build_tip_chain(const pikamd_multi_chain* in, int k, ChainHost& c) {
#pragma clang fp contract(off) // (host-side model arithmetic: the same numbers whatever the compiler flags)

    if (!in || !in->tips) return "multi-tip chain is NULL";
    if (in->dof < 1 || in->dof > PIKAMD_MAX_DOF) return "dof out of range [1, PIKAMD_MAX_DOF]";
    const pikamd_tip& t = in->tips[k];
    if (t.n_joints < 0 || t.n_joints > in->dof) return "tip path longer than the variable vector";
    if (t.n_joints > 0 && (!t.variable || !t.origin_xyz_rpy || !t.axis)) return "tip has NULL arrays";
    if (!t.tip_xyz_rpy) return "tip has no tip transform";
    double origin[PIKAMD_MAX_DOF * 6] = {0.0};
    double axis[PIKAMD_MAX_DOF * 3];
    int32_t jt[PIKAMD_MAX_DOF];
    for (int i = 0; i < in->dof; ++i) {
        axis[3 * i] = axis[3 * i + 1] = 0.0;
        axis[3 * i + 2] = 1.0;
        jt[i] = PIKAMD_JOINT_REVOLUTE;
    }
    uint32_t active = 0;
    for (int j = 0; j < t.n_joints; ++j) {
        const int v = t.variable[j];
        if (v < 0 || v >= in->dof) return "tip variable index out of range";
        if (j > 0 && v <= t.variable[j - 1]) return "tip variable indices must be strictly increasing";
        std::memcpy(origin + 6 * v, t.origin_xyz_rpy + 6 * j, 6 * sizeof(double));
        std::memcpy(axis + 3 * v, t.axis + 3 * j, 3 * sizeof(double));
        jt[v] = t.joint_type ? t.joint_type[j] : PIKAMD_JOINT_REVOLUTE;
        active |= 1u << v;
    }
    pikamd_chain padded{in->dof, origin, axis, jt, t.tip_xyz_rpy, in->qmin, in->qmax, in->vmax, in->bounded};
    if (const char* m = build_chain(&padded, c)) return m;
    c.active_mask = active;
    build_dh(c); // again, now that the joints of the path are known
    return nullptr;
}

template <int D>
inline ChainK<D> make_chain_k(const ChainHost& h) {
    ChainK<D> k;
    std::memset(&k, 0, sizeof k);
    for (int j = 0; j < D; ++j) {
        std::memcpy(k.O[j], h.O[j], sizeof k.O[j]);
        std::memcpy(k.axis[j], h.axis[j], sizeof k.axis[j]);
        k.qmin[j] = h.qmin[j];
        k.qmax[j] = h.qmax[j];
        k.mid[j] = h.mid[j];
        k.hspan[j] = h.hspan[j];
        const bool bounded = (h.bounded_mask >> j) & 1u;
        k.clo[j] = bounded ? h.qmin[j] : -HUGE_VAL;
        k.chi[j] = bounded ? h.qmax[j] : HUGE_VAL;
        k.mdf[j] = h.mdf[j];
    }
    std::memcpy(k.tip, h.tip, sizeof k.tip);
    for (int j = 0; j < D; ++j) std::memcpy(k.dh[j], h.dh[j], sizeof k.dh[j]);
    std::memcpy(k.dh_base, h.dh_base, sizeof k.dh_base);
    std::memcpy(k.dh_tip, h.dh_tip, sizeof k.dh_tip);
    for (int j = 0; j < D; ++j) std::memcpy(k.dhg[j], h.dhg[j], sizeof k.dhg[j]);
    k.dh_general_mask = h.dh_general_mask;
    fill_math_tab(k.mt);
    k.origin_ident_mask = h.origin_ident_mask;
    k.prismatic_mask = h.prismatic_mask;
    k.bounded_mask = h.bounded_mask;
    k.axis_kind = h.axis_kind;
    k.tip_ident = h.tip_ident;
    k.active_mask = h.active_mask;
    k.float_mask = h.float_mask;
    k.skip_mask = h.skip_mask;
    k.m_count = (uint32_t)h.n_mimic;
    for (int m = 0; m < h.n_mimic; ++m) {
        std::memcpy(k.mO[m], h.mO[m], sizeof k.mO[m]);
        std::memcpy(k.maxis[m], h.maxis[m], sizeof k.maxis[m]);
        k.mmult[m] = h.mmult[m];
        k.moff[m] = h.moff[m];
        k.m_after[m] = h.m_after[m];
        k.m_var[m] = h.m_var[m];
    }
    k.m_ident_mask = h.m_ident_mask;
    k.m_pris_mask = h.m_pris_mask;
    k.m_kind = h.m_kind;
    {
        // (the class is a property of ONE tip's path; with several tip frames every path gets its own ChainK and the
        // kernels take the several-tips overloads, which do not read it)
        bool uz = h.origin_ident_mask == 0 && h.prismatic_mask == 0 && h.tip_ident == 0 && h.float_mask == 0 &&
                  h.skip_mask == 0 && h.n_mimic == 0;
        bool all_z = true, all_axis = true;
        for (int j = 0; j < D; ++j) {
            const uint32_t kind = (h.axis_kind >> (2 * j)) & 3u;
            all_z = all_z && kind == (uint32_t)AXIS_Z;
            all_axis = all_axis && kind != (uint32_t)AXIS_GENERAL;
        }
#if defined(PIK_STRICT) // (the exact flavours: the only readers of the chain class)
        // class 1 also wants every origin's rotation to be one about its own x axis with the entries 1 / 0 exact
        // ([1 0 0; 0 a b; 0 c d]: rpy = (alpha, 0, 0), the link twist of the Denavit-Hartenberg convention) --
        // x_iso_mul (pik_math.hpp) leaves the products by those entries out; a z chain with other origins is a
        // chain of class 2
        bool all_rx = true;
        for (int j = 0; j < D; ++j) {
            const double* o = h.O[j];
            all_rx = all_rx && o[0] == 1.0 && o[1] == 0.0 && o[2] == 0.0 && o[3] == 0.0 && o[6] == 0.0;
        }
        // ... and the tip transform to turn about its own z axis only ([a b 0; c d 0; 0 0 1]: rpy = (0, 0, gamma))
        const bool tip_rz = h.tip[8] == 1.0 && h.tip[2] == 0.0 && h.tip[5] == 0.0 && h.tip[6] == 0.0 && h.tip[7] == 0.0;
        k.uniform_z = !uz ? 0u : (all_z && all_rx && tip_rz) ? 1u : all_axis ? 2u : 0u;
        // the exact ones and zeros of the fixed transforms (x_iso_mul, pik_math.hpp)
        auto iso_kind = [](const double* o) -> uint32_t {
            const bool e0 = o[0] == 1.0 && o[1] == 0.0 && o[2] == 0.0 && o[3] == 0.0 && o[6] == 0.0;
            const bool e1 = o[4] == 1.0 && o[1] == 0.0 && o[3] == 0.0 && o[5] == 0.0 && o[7] == 0.0;
            const bool e2 = o[8] == 1.0 && o[2] == 0.0 && o[5] == 0.0 && o[6] == 0.0 && o[7] == 0.0;
            return (e0 && e1 && e2) ? 4u : e0 ? 1u : e1 ? 2u : e2 ? 3u : 0u;
        };
        auto iso_pmask = [](const double* o) -> uint32_t {
            return (o[9] != 0.0 ? 1u : 0u) | (o[10] != 0.0 ? 2u : 0u) | (o[11] != 0.0 ? 4u : 0u);
        };
        for (int j = 0; j < D && j < 10; ++j) {
            k.origin_kinds |= iso_kind(h.O[j]) << (3 * j);
            k.origin_pmasks |= iso_pmask(h.O[j]) << (3 * j);
        }
        k.tip_kind = iso_kind(h.tip);
        k.tip_pmask = iso_pmask(h.tip);
#else
        k.uniform_z = !uz ? 0u : all_z ? 1u : all_axis ? 2u : 0u;
#endif
    }
    return k;
}

inline const char* make_params_k(const pikamd_params* p, ParamsK& k) {
#pragma clang fp contract(off) // (host-side model arithmetic: the same numbers whatever the compiler flags)

    if (!p) return "params is NULL";
    if (p->mode != 0 && p->mode != 1) return "mode must be 0 (global) or 1 (local)";
    if (!(p->gd_step_size > 0.0)) return "gd_step_size must be > 0";
    std::memset(&k, 0, sizeof k);
    k.step_size = p->gd_step_size;
    k.min_cost_delta = p->gd_min_cost_delta;
    k.pos_thr = p->position_threshold;
    k.ori_thr = p->orientation_threshold;
    k.cost_thr_sq = p->cost_threshold * p->cost_threshold;
    k.pos_scale = p->position_scale;
    k.rot_scale = p->rotation_scale;
    k.has_pos_thr = p->position_scale > 0.0;
    k.has_ori_thr = p->rotation_scale > 0.0;
    k.goal_mask = 0;
    if (p->center_joints_weight > 0.0) {
        k.goal_mask |= 1;
        k.w_center_sq = p->center_joints_weight * p->center_joints_weight;
    }
    if (p->avoid_joint_limits_weight > 0.0) {
        k.goal_mask |= 2;
        k.w_limits_sq = p->avoid_joint_limits_weight * p->avoid_joint_limits_weight;
    }
    if (p->minimal_displacement_weight > 0.0) {
        k.goal_mask |= 4;
        k.w_disp_sq = p->minimal_displacement_weight * p->minimal_displacement_weight;
    }
    k.wipeout_tol = p->memetic_wipeout_fitness_tol;
    {
        const double h = p->gd_step_size;
        double s2, c2, ch;
        ::sincos(0.5 * h, &s2, &c2); // explicit glibc sincos: same constants from every host compiler
        ::sincos(h, &k.sin_h, &ch);
        (void)ch;
        k.vers_h = 2.0 * s2 * s2; // 1 - cos h without cancellation
        k.sin_h2 = s2;
        k.cos_h2 = c2;
    }
    k.stop_on_valid = p->stop_optimization_on_valid_solution != 0;
    k.approx = p->return_approximate_solution != 0;
    k.stop_on_first = p->memetic_stop_on_first_solution != 0;
    k.population = p->memetic_population_size;
    k.elites = p->memetic_elite_size;
    k.max_generations = p->memetic_max_generations;
    k.gd_max_iters = p->memetic_gd_max_iters;
    k.local_max_iters = p->gd_max_iters;
    k.line_delta = (p->gd_step_size <= 1.0e-3) ? 1 : 0;
    if (p->mode == 0) {
        if (k.elites < 1 || k.elites > 64) return "memetic_elite_size must be in [1, 64]";
        if (k.population <= k.elites) return "memetic_population_size must exceed memetic_elite_size";
        if (k.population > (1 << 20)) return "memetic_population_size too large";
        if (p->memetic_num_threads > 1) {
            int gs = 1, sp = 1;
            while (gs < k.elites) gs <<= 1;
            while (sp < p->memetic_num_threads) sp <<= 1;
            if (gs * sp > 64) return "memetic_num_threads x pow2ceil(memetic_elite_size) must fit one 64-lane wavefront";
        }
    }
    return nullptr;
}

} // namespace pik
