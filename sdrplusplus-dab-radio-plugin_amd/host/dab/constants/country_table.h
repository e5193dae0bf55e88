// dab/constants/country_table.h -- country names from the extended country code and the country id (the top nibble
// of an ensemble / service identifier): /root/reference/src/render_formatters.cpp:74-76,
// src/render_radio_block.cpp:570-578, 799-802.  The European broadcasting area (ECC 0xE0..0xE4) of ETSI TS 101 756
// table 3 restated from memory, plus Australia; anything else prints "Unknown" -- parity unpinned, and incomplete on
// purpose (the table is not on the hot path).
#pragma once
#include <string>
#include "dab/database/dab_database_entities.h"

inline const std::string &GetCountryName(extended_country_id_t ecc, country_id_t country_id) {
    static const std::string UNKNOWN = "Unknown";
    static const std::string E[5][16] = {
        /* E0 */ {"", "Germany", "Algeria", "Andorra", "Israel", "Italy", "Belgium", "Russian Federation", "Palestine", "Albania",
                  "Austria", "Hungary", "Malta", "Germany", "", "Egypt"},
        /* E1 */ {"", "Greece", "Cyprus", "San Marino", "Switzerland", "Jordan", "Finland", "Luxembourg", "Bulgaria", "Denmark",
                  "Gibraltar", "Iraq", "United Kingdom", "Libya", "Romania", "France"},
        /* E2 */ {"", "Morocco", "Czech Republic", "Poland", "Vatican", "Slovakia", "Syria", "Tunisia", "", "Liechtenstein",
                  "Iceland", "Monaco", "Lithuania", "Serbia", "Spain", "Norway"},
        /* E3 */ {"", "Montenegro", "Ireland", "Turkey", "Macedonia", "", "", "", "Netherlands", "Latvia", "Lebanon", "Azerbaijan",
                  "Croatia", "Kazakhstan", "Sweden", "Belarus"},
        /* E4 */ {"", "Moldova", "Estonia", "Kyrgyzstan", "", "", "Ukraine", "", "Portugal", "Slovenia", "Armenia", "", "Georgia",
                  "", "", "Bosnia Herzegovina"}};
    static const std::string AUSTRALIA = "Australia";
    if (ecc >= 0xE0 && ecc <= 0xE4 && country_id < 16 && !E[ecc - 0xE0][country_id].empty()) return E[ecc - 0xE0][country_id];
    if (ecc == 0xF0 && country_id >= 1 && country_id <= 8) return AUSTRALIA;
    return UNKNOWN;
}
