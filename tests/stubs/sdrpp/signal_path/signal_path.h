#pragma once   // TEST-ONLY stub (see README.md)
#include "signal_path/sink.h"
#include "signal_path/vfo_manager.h"
namespace sigpath {
extern VFOManager vfoManager;
extern SinkManager sinkManager;
}
