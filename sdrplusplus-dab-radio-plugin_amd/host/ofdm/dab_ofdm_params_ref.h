// get_DAB_OFDM_params(transmission_mode) -- /root/reference/src/radio_block.cpp:12
#pragma once
#include <stdexcept>
#include "dabgpu.h"
#include "ofdm/ofdm_params.h"

inline OFDM_Params get_DAB_OFDM_params(const int transmission_mode) {
    dabgpu_ofdm_params p;
    if (dabgpu_get_ofdm_params(transmission_mode, &p) != DABGPU_OK)
        throw std::runtime_error("unsupported DAB transmission mode (only mode 1 is built)");
    OFDM_Params o;
    o.nb_frame_symbols = size_t(p.nb_frame_symbols);
    o.nb_symbol_period = size_t(p.nb_symbol_period);
    o.nb_null_period = size_t(p.nb_null_period);
    o.nb_fft = size_t(p.nb_fft);
    o.nb_cyclic_prefix = size_t(p.nb_cyclic_prefix);
    o.nb_data_carriers = size_t(p.nb_data_carriers);
    o.freq_carrier_spacing = p.freq_carrier_spacing;
    return o;
}
