// dab/audio/aac_frame_processor.h -- the one type the reference's formatters take from the AAC layer
// (/root/reference/src/render_formatters.h:4, render_formatters.cpp:91-104): the MPEG Surround configuration of
// a DAB+ super-frame header (TS 102 563 clause 5.2, mps_config).  The AAC decoder itself is not part of the path.
#pragma once
#include <cstdint>

enum class MPEG_Surround : uint8_t { NOT_USED, SURROUND_51, SURROUND_71, SURROUND_OTHER, RFA };

inline MPEG_Surround mpeg_surround_from_config(uint8_t mps_config) {
    switch (mps_config & 7) {
    case 0: return MPEG_Surround::NOT_USED;
    case 1: return MPEG_Surround::SURROUND_51;
    case 2: return MPEG_Surround::SURROUND_71;
    case 7: return MPEG_Surround::SURROUND_OTHER;
    default: return MPEG_Surround::RFA;
    }
}
