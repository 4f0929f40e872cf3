// TEST STUB (tests/native/ros2_stubs/README.md) -- moveit::core::RobotState: joint values of one
// group + forward kinematics along parent links (T_link = T_parent * origin * joint(q)).
#pragma once
#include <array>
#include <cmath>
#include <map>
#include <moveit/robot_model/robot_model.h>
#include <stdexcept>
namespace moveit::core {
class RobotState {
  public:
    explicit RobotState(RobotModelConstPtr model) : model_(std::move(model)) {}
    void setToDefaultValues() { q_.clear(); }
    void setJointGroupPositions(JointModelGroup const* g, std::vector<double> const& v) {
        size_t n = 0;
        for (auto const* j : g->getActiveJointModels()) n += j->getVariableCount();
        if (v.size() != n) throw std::invalid_argument("setJointGroupPositions: size");
        size_t i = 0;
        for (auto const* j : g->getActiveJointModels()) {
            q_[j] = v[i];
            if (j->getType() == JointModel::PLANAR) planar_[j] = {v[i], v[i + 1], v[i + 2]};
            if (j->getType() == JointModel::FLOATING)
                floating_[j] = {v[i], v[i + 1], v[i + 2], v[i + 3], v[i + 4], v[i + 5], v[i + 6]};
            i += j->getVariableCount();
        }
    }
    void copyJointGroupPositions(JointModelGroup const* g, std::vector<double>& out) const {
        out.clear();
        for (auto const* j : g->getActiveJointModels()) {
            if (j->getType() == JointModel::PLANAR) {
                auto p = planar_.find(j);
                for (int k = 0; k < 3; ++k) out.push_back(p == planar_.end() ? 0.0 : p->second[k]);
            } else if (j->getType() == JointModel::FLOATING) {
                auto p = floating_.find(j);
                for (int k = 0; k < 7; ++k) out.push_back(p == floating_.end() ? (k == 6 ? 1.0 : 0.0) : p->second[static_cast<size_t>(k)]);
            } else {
                auto v = q_.find(j);
                out.push_back(v == q_.end() ? 0.0 : v->second);
            }
        }
    }
    void update() {}
    Eigen::Isometry3d getGlobalLinkTransform(std::string const& name) const {
        LinkModel const* l = model_->getLinkModel(name);
        if (!l) throw std::invalid_argument("no such link: " + name);
        std::vector<LinkModel const*> up;
        for (; l; l = l->getParentLinkModel()) up.push_back(l);
        Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
        for (auto it = up.rbegin(); it != up.rend(); ++it) {
            T = T * (*it)->getJointOriginTransform();
            JointModel const* j = (*it)->getParentJointModel();
            // a joint nobody set sits at its default position; a mimic joint at factor * master + offset
            // (RobotState::updateMimicJoints)
            auto const value_of = [&](JointModel const* jj) {
                auto v = q_.find(jj);
                double dflt[7] = {0, 0, 0, 0, 0, 0, 0};
                if (jj->getVariableCount() > 0) jj->getVariableDefaultPositions(dflt);
                return v == q_.end() ? dflt[0] : v->second;
            };
            double const q = j->getMimic() ? j->getMimicFactor() * value_of(j->getMimic()) + j->getMimicOffset() : value_of(j);
            Eigen::Isometry3d J = Eigen::Isometry3d::Identity();
            if (j->getType() == JointModel::REVOLUTE) {
                double const c = std::cos(q), s = std::sin(q), t = 1 - c, x = j->axis_.x(), y = j->axis_.y(), z = j->axis_.z();
                J.R(0, 0) = t * x * x + c; J.R(0, 1) = t * x * y - z * s; J.R(0, 2) = t * x * z + y * s;
                J.R(1, 0) = t * x * y + z * s; J.R(1, 1) = t * y * y + c; J.R(1, 2) = t * y * z - x * s;
                J.R(2, 0) = t * x * z - y * s; J.R(2, 1) = t * y * z + x * s; J.R(2, 2) = t * z * z + c;
            } else if (j->getType() == JointModel::PRISMATIC) {
                J.t = Eigen::Vector3d(j->axis_.x() * q, j->axis_.y() * q, j->axis_.z() * q);
            } else if (j->getType() == JointModel::PLANAR) {
                auto p = planar_.find(j);
                double const x = p == planar_.end() ? 0.0 : p->second[0], y = p == planar_.end() ? 0.0 : p->second[1],
                             th = p == planar_.end() ? 0.0 : p->second[2];
                J.t = Eigen::Vector3d(x, y, 0.0);
                J.R(0, 0) = std::cos(th); J.R(0, 1) = -std::sin(th);
                J.R(1, 0) = std::sin(th); J.R(1, 1) = std::cos(th);
            } else if (j->getType() == JointModel::FLOATING) {
                // FloatingJointModel::computeTransform: Translation(v0 v1 v2) * Quaterniond(v6, v3, v4, v5)
                auto p = floating_.find(j);
                std::array<double, 7> const v = p == floating_.end() ? std::array<double, 7>{0, 0, 0, 0, 0, 0, 1} : p->second;
                J.t = Eigen::Vector3d(v[0], v[1], v[2]);
                J.R = Eigen::Quaterniond(v[6], v[3], v[4], v[5]).toRotationMatrix();
            }
            T = T * J;
        }
        return T;
    }

  private:
    RobotModelConstPtr model_;
    std::map<JointModel const*, double> q_;
    std::map<JointModel const*, std::array<double, 3>> planar_;
    std::map<JointModel const*, std::array<double, 7>> floating_;
};
} // namespace moveit::core
