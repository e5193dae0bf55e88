#pragma once   // TEST-ONLY stub (see README.md)
#include "dsp/stream.h"
namespace dsp {
template <class T>
class Sink {
public:
    virtual ~Sink() {}
    virtual int run() = 0;
    void init(stream<T> *in) { _in = in; _block_init = true; }
    void setInput(stream<T> *in) { _in = in; }
    void start() {}
    void stop() {}
protected:
    bool _block_init = false;
    stream<T> *_in = nullptr;
};
}
