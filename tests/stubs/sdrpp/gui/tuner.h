#pragma once   // TEST-ONLY stub (see README.md)
#include <string>
namespace tuner {
enum { TUNER_MODE_CENTER, TUNER_MODE_NORMAL };
void tune(int mode, const std::string &vfo_name, double freq);
}
