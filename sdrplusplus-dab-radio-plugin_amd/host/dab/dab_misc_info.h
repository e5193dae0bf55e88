// dab/dab_misc_info.h -- what BasicRadio::GetMiscInfo() returns (/root/reference/src/render_radio_block.cpp:812-836):
// date and time from FIG 0/10 (EN 300 401 clause 8.1.3.1: modified Julian date + UTC) and the CIF counter from
// FIG 0/0 (upper part modulo 20, lower part modulo 250).
#pragma once
#include <cstdint>

struct DAB_Date_Time {
    int day = 0, month = 0, year = 0;
    uint8_t hours = 0, minutes = 0, seconds = 0;
    uint16_t milliseconds = 0;
};

struct DAB_CIF_Counter {
    uint8_t upper_count = 0;      // 0..19
    uint8_t lower_count = 0;      // 0..249
    uint32_t GetTotalCount() const { return uint32_t(upper_count) * 250u + lower_count; }
};

struct DAB_Misc_Info {
    DAB_Date_Time datetime;
    DAB_CIF_Counter cif_counter;
};

// modified Julian date -> calendar date (the standard's annex algorithm, valid 1900..2100)
inline void mjd_to_ymd(int mjd, int &year, int &month, int &day) {
    const int yp = int((double(mjd) - 15078.2) / 365.25);
    const int mp = int((double(mjd) - 14956.1 - double(int(double(yp) * 365.25))) / 30.6001);
    day = mjd - 14956 - int(double(yp) * 365.25) - int(double(mp) * 30.6001);
    const int k = (mp == 14 || mp == 15) ? 1 : 0;
    year = 1900 + yp + k;
    month = mp - 1 - k * 12;
}
