// bio_ik/robot_info.h — per-variable joint-limit information as goals read it through GoalContext::getRobotInfo()
// (reference include/bio_ik/robot_info.h:45-124: clip range, span, bounds, velocity limits, joint type per variable).
// Host side of the MI355X build: the same numbers the problem compiler puts into the device's joint program
// (bio_ik_amd/csrc/bioik_compile.cpp), here for goals that are evaluated on the host.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <vector>

namespace bio_ik {

class RobotInfo {
    struct VariableInfo {
        double clip_min, clip_max, span, min, max, max_velocity, max_velocity_rcp;
        bool revolute, prismatic;
    };
    std::vector<VariableInfo> variables;

public:
    RobotInfo() {}
    // one entry per robot variable: bounds as MoveIt's VariableBounds holds them, and the type of the variable's joint
    void addVariable(double min_position, double max_position, bool position_bounded, double max_velocity, bool revolute, bool prismatic) {
        VariableInfo info;
        bool bounded = position_bounded;
        if (revolute && max_position - min_position >= 2 * M_PI * 0.9999) bounded = false;  // a full turn is no limit (robot_info.h:85-87)
        info.min = min_position, info.max = max_position;
        info.clip_min = bounded ? min_position : -DBL_MAX;
        info.clip_max = bounded ? max_position : +DBL_MAX;
        info.span = max_position - min_position;
        if (!(info.span >= 0 && info.span < FLT_MAX)) info.span = 1;
        info.max_velocity = max_velocity;
        info.max_velocity_rcp = max_velocity > 0.0 ? 1.0 / max_velocity : 0.0;
        info.revolute = revolute, info.prismatic = prismatic;
        variables.push_back(info);
    }
    size_t getVariableCount() const { return variables.size(); }
    double clip(double p, size_t i) const { return p < variables[i].clip_min ? variables[i].clip_min : (p > variables[i].clip_max ? variables[i].clip_max : p); }
    double getSpan(size_t i) const { return variables[i].span; }
    double getClipMin(size_t i) const { return variables[i].clip_min; }
    double getClipMax(size_t i) const { return variables[i].clip_max; }
    double getMin(size_t i) const { return variables[i].min; }
    double getMax(size_t i) const { return variables[i].max; }
    bool isRevolute(size_t i) const { return variables[i].revolute; }
    bool isPrismatic(size_t i) const { return variables[i].prismatic; }
    double getMaxVelocity(size_t i) const { return variables[i].max_velocity; }
    double getMaxVelocityRcp(size_t i) const { return variables[i].max_velocity_rcp; }
};

}  // namespace bio_ik
