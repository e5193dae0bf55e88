#pragma once   // TEST-ONLY stub (see README.md)
namespace style {
inline void beginDisabled() {}
inline void endDisabled() {}
}
