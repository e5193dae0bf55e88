// utility/gpu_buffers.h -- what the host mirror needs around libdabgpu besides the calls themselves: which GPU a
// new OFDM_Demod / BasicRadio uses, and page-locked host buffers (dabgpu_host_alloc) for everything that crosses
// PCIe every frame, so the copies are plain DMA transfers instead of going through the runtime's bounce buffers.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <new>
#include "dabgpu.h"

// Device ordinal for objects created from now on: SetDabGpuDefaultDevice(), else the DABGPU_DEVICE environment
// variable, else 0.  The reference creates one Radio_Block per module instance (`Max instances -1`,
// /root/reference/src/main.cpp:12; /root/reference/src/dab_module.cpp:92): instances can be spread over GPUs.
inline int &dabgpu_default_device_ref() {
    static int device = [] {
        const char *e = std::getenv("DABGPU_DEVICE");
        return e ? std::atoi(e) : 0;
    }();
    return device;
}
inline int GetDabGpuDefaultDevice() { return dabgpu_default_device_ref(); }
inline void SetDabGpuDefaultDevice(int device) { dabgpu_default_device_ref() = device; }

// fixed-size array in page-locked memory (zero-initialised)
template <class T>
class PinnedBuffer {
public:
    PinnedBuffer() = default;
    explicit PinnedBuffer(size_t n) { resize(n); }
    ~PinnedBuffer() { dabgpu_host_free(m_p); }
    PinnedBuffer(const PinnedBuffer &) = delete;
    PinnedBuffer &operator=(const PinnedBuffer &) = delete;
    PinnedBuffer(PinnedBuffer &&o) noexcept : m_p(o.m_p), m_n(o.m_n) { o.m_p = nullptr; o.m_n = 0; }
    PinnedBuffer &operator=(PinnedBuffer &&o) noexcept {
        if (this != &o) { dabgpu_host_free(m_p); m_p = o.m_p; m_n = o.m_n; o.m_p = nullptr; o.m_n = 0; }
        return *this;
    }
    void resize(size_t n) {
        dabgpu_host_free(m_p);
        m_p = nullptr;
        m_n = 0;
        if (n == 0) return;
        m_p = static_cast<T *>(dabgpu_host_alloc(n * sizeof(T)));
        if (!m_p) throw std::bad_alloc();
        m_n = n;
        for (size_t i = 0; i < n; i++) m_p[i] = T();
    }
    T *data() { return m_p; }
    const T *data() const { return m_p; }
    size_t size() const { return m_n; }
    T &operator[](size_t i) { return m_p[i]; }
    const T &operator[](size_t i) const { return m_p[i]; }
    T *begin() { return m_p; }
    T *end() { return m_p + m_n; }
    const T *begin() const { return m_p; }
    const T *end() const { return m_p + m_n; }

private:
    T *m_p = nullptr;
    size_t m_n = 0;
};
