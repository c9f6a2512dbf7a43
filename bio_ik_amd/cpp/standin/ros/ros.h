// stand-in for ros/ros.h: the private-namespace parameter lookup the plugin uses for its kinematics.yaml keys, and the wall clock
// (see ../README.md).  param_store() plays the parameter server: tests put the yaml keys there.
#pragma once
#include <chrono>
#include <map>
#include <string>
namespace ros {
struct ParamValue {
    enum { NONE, BOOL, INT, DOUBLE, STRING } kind = NONE;
    bool b = false;
    int i = 0;
    double d = 0;
    std::string s;
};
inline std::map<std::string, ParamValue>& param_store() {
    static std::map<std::string, ParamValue> store;
    return store;
}
inline void set_param(const std::string& k, bool v) { ParamValue p; p.kind = ParamValue::BOOL, p.b = v; param_store()[k] = p; }
inline void set_param(const std::string& k, int v) { ParamValue p; p.kind = ParamValue::INT, p.i = v; param_store()[k] = p; }
inline void set_param(const std::string& k, double v) { ParamValue p; p.kind = ParamValue::DOUBLE, p.d = v; param_store()[k] = p; }
inline void set_param(const std::string& k, const char* v) { ParamValue p; p.kind = ParamValue::STRING, p.s = v; param_store()[k] = p; }
class NodeHandle {
    std::string ns_;
    const ParamValue* find(const std::string& name) const {
        auto it = param_store().find(name);
        return it == param_store().end() ? nullptr : &it->second;
    }

public:
    explicit NodeHandle(const std::string& ns = std::string()) : ns_(ns) {}
    bool param(const std::string& name, const bool& dflt) const { auto* p = find(name); return p && p->kind == ParamValue::BOOL ? p->b : dflt; }
    int param(const std::string& name, const int& dflt) const { auto* p = find(name); return p && p->kind == ParamValue::INT ? p->i : dflt; }
    double param(const std::string& name, const double& dflt) const {
        auto* p = find(name);
        if (p && p->kind == ParamValue::DOUBLE) return p->d;
        if (p && p->kind == ParamValue::INT) return (double)p->i;
        return dflt;
    }
    std::string param(const std::string& name, const std::string& dflt) const { auto* p = find(name); return p && p->kind == ParamValue::STRING ? p->s : dflt; }
};
struct WallTime {
    double t = 0;
    static WallTime now() {
        WallTime w;
        w.t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        return w;
    }
    double toSec() const { return t; }
};
}  // namespace ros
