#pragma once   // TEST-ONLY stub (see README.md)
#include <string>
#include "dsp/stream.h"
namespace ImGui { struct WaterfallVFO { enum { REF_LOWER, REF_CENTER, REF_UPPER }; }; }
class VFOManager {
public:
    struct VFO { dsp::stream<dsp::complex_t> *output = nullptr; };
    VFO *createVFO(const std::string &, int, double, double, double, double, double, bool) { return nullptr; }
    void deleteVFO(VFO *) {}
};
