#pragma once   // TEST-ONLY stub (see README.md)
#include <string>
struct StubJsonValue {
    StubJsonValue &operator=(bool) { return *this; }
    operator bool() const { return false; }
};
struct StubJson {
    bool contains(const std::string &) const { return false; }
    StubJsonValue &operator[](const std::string &) { return v; }
    StubJsonValue v;
};
class ConfigManager {
public:
    void acquire() {}
    void release(bool /*modified*/ = false) {}
    StubJson conf;
};
