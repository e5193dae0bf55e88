#pragma once   // TEST-ONLY stub (see ../README.md)
#include <string>
namespace fmt {
template <class... Args>
std::string format(const char *, Args &&...);
}
