// get_DAB_PRS_reference(transmission_mode, span<complex<float>>[nb_fft]) -- /root/reference/src/radio_block.cpp:18-19
#pragma once
#include <complex>
#include <stdexcept>
#include "dabgpu.h"
#include "utility/span.h"

inline void get_DAB_PRS_reference(const int transmission_mode, tcb::span<std::complex<float>> buf) {
    if (dabgpu_get_prs_reference(transmission_mode, reinterpret_cast<float *>(buf.data()), int(buf.size())) != DABGPU_OK)
        throw std::runtime_error("get_DAB_PRS_reference: bad mode or buffer size");
}
