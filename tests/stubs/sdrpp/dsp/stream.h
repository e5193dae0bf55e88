#pragma once   // TEST-ONLY stub (see README.md)
namespace dsp {
struct complex_t { float re, im; };
struct stereo_t { float l, r; };
template <class T>
class stream {
public:
    T *writeBuf = nullptr;
    T *readBuf = nullptr;
    bool swap(int) { return true; }
    int read() { return -1; }
    void flush() {}
    void stopReader() {}
    void stopWriter() {}
    void clearReadStop() {}
    void clearWriteStop() {}
};
}
