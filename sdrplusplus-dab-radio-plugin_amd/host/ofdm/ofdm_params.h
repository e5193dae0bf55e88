#pragma once
#include <cstddef>
// Field names used by the reference: nb_fft, nb_data_carriers (/root/reference/src/radio_block.cpp:18,20)
struct OFDM_Params {
    size_t nb_frame_symbols;
    size_t nb_symbol_period;
    size_t nb_null_period;
    size_t nb_fft;
    size_t nb_cyclic_prefix;
    size_t nb_data_carriers;
    int freq_carrier_spacing;
};
