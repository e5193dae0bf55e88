#pragma once   // TEST-ONLY stub (see README.md)
#include "../imgui.h"
