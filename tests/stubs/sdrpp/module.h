#pragma once   // TEST-ONLY stub (see README.md)
#include <string>
namespace ModuleManager {
class Instance {
public:
    virtual ~Instance() {}
    virtual void postInit() = 0;
    virtual void enable() = 0;
    virtual void disable() = 0;
    virtual bool isEnabled() = 0;
};
}
