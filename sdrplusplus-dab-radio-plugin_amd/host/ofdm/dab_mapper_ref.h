// get_DAB_mapper_ref(span<int>[nb_data_carriers], nb_fft) -- /root/reference/src/radio_block.cpp:20-21
#pragma once
#include <stdexcept>
#include "dabgpu.h"
#include "utility/span.h"

inline void get_DAB_mapper_ref(tcb::span<int> carrier_map, const int nb_fft) {
    static_assert(sizeof(int) == sizeof(int32_t), "int is 32 bit");
    if (dabgpu_get_mapper_reference(reinterpret_cast<int32_t *>(carrier_map.data()), int(carrier_map.size()), nb_fft) != DABGPU_OK)
        throw std::runtime_error("get_DAB_mapper_ref: bad sizes");
}
