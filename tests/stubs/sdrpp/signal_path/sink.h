#pragma once   // TEST-ONLY stub (see README.md)
#include <string>
#include "dsp/stream.h"
template <class T>
struct EventHandler {
    void (*handler)(T, void *) = nullptr;
    void *ctx = nullptr;
};
class SinkManager {
public:
    class Stream {
    public:
        void init(dsp::stream<dsp::stereo_t> *, EventHandler<float> *, float) {}
        void setVolume(float) {}
        void start() {}
        void stop() {}
    };
    void registerStream(const std::string &, Stream *) {}
    void unregisterStream(const std::string &) {}
};
