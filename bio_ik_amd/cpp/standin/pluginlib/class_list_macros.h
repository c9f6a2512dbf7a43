// stand-in for pluginlib/class_list_macros.h: PLUGINLIB_EXPORT_CLASS registers a factory for the class under its C++ type name, as
// class_loader does when the library is loaded; pluginlib::registry() is what a pluginlib::ClassLoader<Base> would consult after
// reading the package's plugin description xml (see ../README.md).
#pragma once
#include <map>
#include <string>
namespace pluginlib {
typedef void* (*Factory)();
inline std::map<std::string, std::pair<std::string, Factory>>& registry() {  // derived type name -> (base type name, factory)
    static std::map<std::string, std::pair<std::string, Factory>> r;
    return r;
}
struct Registrar {
    Registrar(const char* derived, const char* base, Factory f) { registry()[derived] = std::make_pair(std::string(base), f); }
};
}  // namespace pluginlib
#define PLUGINLIB_EXPORT_CLASS_CAT2(a, b) a##b
#define PLUGINLIB_EXPORT_CLASS_CAT(a, b) PLUGINLIB_EXPORT_CLASS_CAT2(a, b)
#define PLUGINLIB_EXPORT_CLASS(Derived, Base)                                                                                      \
    namespace {                                                                                                                    \
    void* PLUGINLIB_EXPORT_CLASS_CAT(pluginlib_factory_, __LINE__)() { return static_cast<Base*>(new Derived()); }                 \
    pluginlib::Registrar PLUGINLIB_EXPORT_CLASS_CAT(pluginlib_registrar_, __LINE__)(#Derived, #Base, &PLUGINLIB_EXPORT_CLASS_CAT(pluginlib_factory_, __LINE__)); \
    }
// the C entry point a loader can dlsym without sharing inline statics across the library boundary
extern "C" void* pluginlib_standin_create(const char* derived_type_name);
