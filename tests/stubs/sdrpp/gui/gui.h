#pragma once   // TEST-ONLY stub (see README.md)
#include <string>
namespace gui {
struct Menu {
    void registerEntry(const std::string &, void (*)(void *), void * = nullptr, void * = nullptr) {}
};
extern Menu menu;
struct WaterFall {
    std::string selectedVFO;
};
extern WaterFall waterfall;
}
