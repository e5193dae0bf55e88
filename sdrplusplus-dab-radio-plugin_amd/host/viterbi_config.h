// Soft-decision bit type handed from OFDM_Demod to BasicRadio
// (`viterbi_bit_t`, /root/reference/src/radio_block.h:19).  int8: +127 = 1, -127 = 0, 0 = erased.
#pragma once
#include <cstdint>
typedef int8_t viterbi_bit_t;
constexpr viterbi_bit_t SOFT_DECISION_VITERBI_HIGH = +127;
constexpr viterbi_bit_t SOFT_DECISION_VITERBI_LOW = -127;
constexpr viterbi_bit_t SOFT_DECISION_VITERBI_PUNCTURED = 0;
