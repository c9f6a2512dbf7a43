// stand-in for XmlRpc (only named by the unused XmlRpcReader helper of src/utils.h)
#pragma once
#include <string>
namespace XmlRpc {
struct XmlRpcException {};
struct XmlRpcValue {
    enum Type { TypeInvalid, TypeBoolean, TypeInt, TypeDouble, TypeString };
    Type getType() const { return TypeInvalid; }
    bool hasMember(const char*) const { return false; }
    XmlRpcValue& operator[](int) { return *this; }
    XmlRpcValue& operator[](const char*) { return *this; }
    operator bool() const { return false; }
    operator int() const { return 0; }
    operator double() const { return 0.0; }
    operator std::string() const { return std::string(); }
};
}  // namespace XmlRpc
