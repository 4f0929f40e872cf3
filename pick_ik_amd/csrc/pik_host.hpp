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

namespace pik {

struct ChainHost {
    int dof = 0;
    double O[PIKAMD_MAX_DOF][12];
    double axis[PIKAMD_MAX_DOF][3];
    double tip[12];
    double qmin[PIKAMD_MAX_DOF], qmax[PIKAMD_MAX_DOF], mid[PIKAMD_MAX_DOF], hspan[PIKAMD_MAX_DOF],
        mdf[PIKAMD_MAX_DOF], vrcp[PIKAMD_MAX_DOF];
    uint32_t origin_ident_mask = 0, prismatic_mask = 0, bounded_mask = 0, axis_kind = 0,
             tip_ident = 0;
};

inline void xyz_rpy_to_iso12(const double* xyz_rpy, double* o12) {
    const double phi = xyz_rpy[3] / 2.0, the = xyz_rpy[4] / 2.0, psi = xyz_rpy[5] / 2.0;
    double q[4];
    q[1] = std::sin(phi) * std::cos(the) * std::cos(psi) - std::cos(phi) * std::sin(the) * std::sin(psi);
    q[2] = std::cos(phi) * std::sin(the) * std::cos(psi) + std::sin(phi) * std::cos(the) * std::sin(psi);
    q[3] = std::cos(phi) * std::cos(the) * std::sin(psi) - std::sin(phi) * std::sin(the) * std::cos(psi);
    q[0] = std::cos(phi) * std::cos(the) * std::cos(psi) + std::sin(phi) * std::sin(the) * std::sin(psi);
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

inline bool iso12_is_identity(const double* o) {
    static const double I[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    for (int i = 0; i < 12; ++i)
        if (o[i] != I[i]) return false;
    return true;
}

// returns nullptr on success, else an error message
inline const char* build_chain(const pikamd_chain* in, ChainHost& c) {
    if (!in) return "chain is NULL";
    if (in->dof < 1 || in->dof > PIKAMD_MAX_DOF) return "dof out of range [1, PIKAMD_MAX_DOF]";
    if (!in->origin_xyz_rpy || !in->axis || !in->tip_xyz_rpy || !in->qmin || !in->qmax)
        return "chain has NULL arrays";
    c.dof = in->dof;
    double divisor = 0.0;
    for (int j = 0; j < c.dof; ++j) {
        xyz_rpy_to_iso12(in->origin_xyz_rpy + 6 * j, c.O[j]);
        if (iso12_is_identity(c.O[j])) c.origin_ident_mask |= 1u << j;
        const double ax = in->axis[3 * j], ay = in->axis[3 * j + 1], az = in->axis[3 * j + 2];
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
        const int jt = in->joint_type ? in->joint_type[j] : PIKAMD_JOINT_REVOLUTE;
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
        k.mdf[j] = h.mdf[j];
    }
    std::memcpy(k.tip, h.tip, sizeof k.tip);
    k.origin_ident_mask = h.origin_ident_mask;
    k.prismatic_mask = h.prismatic_mask;
    k.bounded_mask = h.bounded_mask;
    k.axis_kind = h.axis_kind;
    k.tip_ident = h.tip_ident;
    return k;
}

inline const char* make_params_k(const pikamd_params* p, ParamsK& k) {
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
        const double h = p->gd_step_size, s2 = std::sin(0.5 * h);
        k.sin_h = std::sin(h);
        k.vers_h = 2.0 * s2 * s2; // 1 - cos h without cancellation
        k.sin_h2 = s2;
        k.cos_h2 = std::cos(0.5 * h);
    }
    k.stop_on_valid = p->stop_optimization_on_valid_solution != 0;
    k.approx = p->return_approximate_solution != 0;
    k.population = p->memetic_population_size;
    k.elites = p->memetic_elite_size;
    k.max_generations = p->memetic_max_generations;
    k.gd_max_iters = p->memetic_gd_max_iters;
    k.local_max_iters = p->gd_max_iters;
    if (p->mode == 0) {
        if (k.elites < 1 || k.elites > 64) return "memetic_elite_size must be in [1, 64]";
        if (k.population <= k.elites) return "memetic_population_size must exceed memetic_elite_size";
        if (k.population > (1 << 20)) return "memetic_population_size too large";
        if (p->memetic_num_threads > 1) return "memetic_num_threads > 1 (species) is not implemented on the GPU yet";
    }
    return nullptr;
}

} // namespace pik
