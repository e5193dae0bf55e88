// DAB_Parameters / get_dab_parameters(mode) -- /root/reference/src/radio_block.cpp:2,13; field nb_frame_bits :23,34
#pragma once
#include <stdexcept>
#include "dabgpu.h"

struct DAB_Parameters {
    int nb_frame_bits;
    int nb_symbols;
    int nb_fic_symbols;
    int nb_msc_symbols;
    int nb_sym_bits;
    int nb_fic_bits;
    int nb_msc_bits;
    int nb_fibs;
    int nb_cifs;
    int nb_fib_bits;
    int nb_fib_cif_bits;
    int nb_fibs_per_cif;
    int nb_cif_bits;
};

inline DAB_Parameters get_dab_parameters(const int transmission_mode) {
    dabgpu_dab_params p;
    if (dabgpu_get_dab_params(transmission_mode, &p) != DABGPU_OK)
        throw std::runtime_error("unsupported DAB transmission mode (only mode 1 is built)");
    return DAB_Parameters{p.nb_frame_bits, p.nb_symbols, p.nb_fic_symbols, p.nb_msc_symbols, p.nb_sym_bits,
                          p.nb_fic_bits, p.nb_msc_bits, p.nb_fibs, p.nb_cifs, p.nb_fib_bits, p.nb_fib_cif_bits,
                          p.nb_fibs_per_cif, p.nb_cif_bits};
}
