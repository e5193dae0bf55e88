#pragma once   // TEST-ONLY stub (see README.md)
#include "dsp/stream.h"
namespace ImGui {
class ConstellationDiagram {
public:
    dsp::complex_t *acquireBuffer();
    void releaseBuffer();
    void draw();
};
}
