// dab/constants/subchannel_protection_tables.h -- the two helpers the reference's formatters use to turn a
// sub-channel entity into a bit rate (/root/reference/src/render_formatters.cpp:18-25):
//   GetUEPDescriptor(subchannel)   short form: row `uep_prot_index` of the 64 UEP protection profiles
//   CalculateEEPBitrate(subchannel) long form: size = k * n capacity units, bit rate = 8n (option A) / 32n (option B)
// The UEP table itself lives in libdabgpu (csrc/dab_tables.hpp, consistency-checked at build time); this header
// reads it through the C ABI, which needs no GPU for that.
#pragma once
#include <cstdint>
#include "dab/database/dab_database_entities.h"
#include "dabgpu.h"

struct UEP_Descriptor {
    uint16_t subchannel_size = 0;      // capacity units
    uint16_t bitrate = 0;              // kbit/s
    uint8_t protection_level = 0;      // 1..5
};

inline UEP_Descriptor GetUEPDescriptor(const Subchannel &subchannel) {
    UEP_Descriptor d;
    dabgpu_subchannel sc{};
    if (dabgpu_uep_subchannel(subchannel.uep_prot_index, 0, &sc) == DABGPU_OK) {
        d.subchannel_size = uint16_t(sc.length);
        d.bitrate = uint16_t(sc.bitrate_kbps);
        d.protection_level = uint8_t(sc.protection_level);
    }
    return d;
}

// 0 when the size is not a multiple of the profile's unit (not a valid EEP sub-channel)
inline uint32_t CalculateEEPBitrate(const Subchannel &subchannel) {
    static const int per_unit_a[4] = {12, 8, 6, 4}, per_unit_b[4] = {27, 21, 18, 15};   // CUs per 8 / 32 kbit/s
    if (subchannel.eep_prot_level > 3) return 0;
    const bool type_b = subchannel.eep_type == EEP_Type::TYPE_B;
    const int k = (type_b ? per_unit_b : per_unit_a)[subchannel.eep_prot_level];
    if (subchannel.length == 0 || subchannel.length % k) return 0;
    return uint32_t(subchannel.length / k) * (type_b ? 32u : 8u);
}
