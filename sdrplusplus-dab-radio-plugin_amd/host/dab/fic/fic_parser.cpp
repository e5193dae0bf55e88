#include "dab/fic/fic_parser.h"

#include <string>

uint16_t FIC_Parser::CRC16(const uint8_t *data, size_t n) {
    uint16_t crc = 0xFFFF;                                  // x^16 + x^12 + x^5 + 1, initial ones, result inverted
    for (size_t i = 0; i < n; i++) {
        crc ^= uint16_t(data[i]) << 8;
        for (int b = 0; b < 8; b++) crc = (crc & 0x8000) ? uint16_t((crc << 1) ^ 0x1021) : uint16_t(crc << 1);
    }
    return uint16_t(~crc);
}

bool FIC_Parser::ProcessFIB(tcb::span<const uint8_t> fib) {
    if (fib.size() != 32) return false;
    const uint8_t *d = fib.data();
    if (CRC16(d, 30) != uint16_t((d[30] << 8) | d[31])) return false;
    int i = 0;
    while (i < 30) {
        const uint8_t hdr = d[i];
        if (hdr == 0xFF) break;                             // end marker
        const int type = hdr >> 5, len = hdr & 0x1F;
        if (len == 0 || i + 1 + len > 30) break;
        const uint8_t *body = d + i + 1;
        i += 1 + len;
        m_total_figs++;
        if (type == 0) fig0(body, len);
        else if (type == 1) fig1(body, len);
    }
    return true;
}

void FIC_Parser::fig0(const uint8_t *d, int n) {
    const bool pd = (d[0] >> 5) & 1;
    const int ext = d[0] & 0x1F;
    switch (ext) {
    case 0: fig0_0(d + 1, n - 1); break;
    case 1: fig0_1(d + 1, n - 1); break;
    case 2: fig0_2(d + 1, n - 1, pd); break;
    case 10: fig0_10(d + 1, n - 1); break;
    default: break;
    }
}

void FIC_Parser::fig0_0(const uint8_t *d, int n) {
    if (n < 4) return;
    m_updater.SetEnsembleId(uint16_t((d[0] << 8) | d[1]));
    m_updater.SetEnsembleCIFCounter(uint8_t(d[2] & 0x1F), d[3]);
}

void FIC_Parser::fig0_1(const uint8_t *d, int n) {
    int i = 0;
    while (i + 3 <= n) {
        const subchannel_id_t id = d[i] >> 2;
        const uint16_t start = uint16_t(((d[i] & 3) << 8) | d[i + 1]);
        Subchannel v;
        v.id = id;
        v.start_address = start;
        if (d[i + 2] & 0x80) {                              // long form: EEP
            if (i + 4 > n) return;
            v.is_uep = false;
            v.eep_type = ((d[i + 2] >> 4) & 7) == 0 ? EEP_Type::TYPE_A : EEP_Type::TYPE_B;
            if (((d[i + 2] >> 4) & 7) > 1) { i += 4; continue; }      // reserved option: not an entry we can use
            v.eep_prot_level = uint8_t((d[i + 2] >> 2) & 3);        // 0..3 as transmitted
            v.length = uint16_t(((d[i + 2] & 3) << 8) | d[i + 3]);
            i += 4;
        } else {                                            // short form: UEP table index
            v.is_uep = true;
            v.uep_prot_index = d[i + 2] & 0x3F;
            i += 3;
        }
        bool is_new;
        Subchannel &s = m_updater.GetSubchannel(id, &is_new);
        if (is_new) s = v;
        else
            m_updater.same(s.start_address == v.start_address && s.length == v.length && s.is_uep == v.is_uep &&
                           s.uep_prot_index == v.uep_prot_index && s.eep_type == v.eep_type &&
                           s.eep_prot_level == v.eep_prot_level);
    }
}

void FIC_Parser::fig0_2(const uint8_t *d, int n, bool pd) {
    const int idlen = pd ? 4 : 2;
    int i = 0;
    while (i + idlen + 1 <= n) {
        uint32_t sid = 0;
        for (int k = 0; k < idlen; k++) sid = (sid << 8) | d[i + k];
        const int nc = d[i + idlen] & 0x0F;
        i += idlen + 1;
        if (i + 2 * nc > n) return;
        m_updater.GetService(sid);
        for (int c = 0; c < nc; c++, i += 2) {
            const uint8_t b0 = d[i], b1 = d[i + 1];
            if ((b0 >> 6) != 0) continue;                   // only MSC stream audio is followed
            bool is_new;
            ServiceComponent &sc = m_updater.GetServiceComponent(sid, b1 >> 2, &is_new);
            const auto ascty = static_cast<AudioServiceType>(b0 & 0x3F);     // 0 = DAB, 63 = DAB+, others kept as sent
            if (is_new) {
                sc.component_id = service_component_id_t(c);
                sc.transport_mode = TransportMode::STREAM_MODE_AUDIO;
                sc.audio_service_type = ascty;
                sc.is_primary = (b1 & 2) != 0;
            } else {
                m_updater.same(sc.audio_service_type == ascty);
            }
        }
    }
}

// FIG 0/10 date and time: Rfu(1) MJD(17) LSI(1) Rfu(1) UTC flag(1), then hours(5) minutes(6) and, in the long
// form, seconds(6) milliseconds(10)
void FIC_Parser::fig0_10(const uint8_t *d, int n) {
    if (n < 4) return;
    const uint32_t w = (uint32_t(d[0]) << 24) | (uint32_t(d[1]) << 16) | (uint32_t(d[2]) << 8) | d[3];
    const int mjd = int((w >> 14) & 0x1FFFF);
    const bool long_form = (w >> 11) & 1;
    DAB_Date_Time dt;
    mjd_to_ymd(mjd, dt.year, dt.month, dt.day);
    dt.hours = uint8_t((w >> 6) & 0x1F);
    dt.minutes = uint8_t(w & 0x3F);
    if (long_form && n >= 6) {
        dt.seconds = uint8_t(d[4] >> 2);
        dt.milliseconds = uint16_t(((d[4] & 3) << 8) | d[5]);
    }
    if (dt.hours < 24 && dt.minutes < 60 && dt.seconds < 61) m_updater.SetDateTime(dt);
}

void FIC_Parser::fig1(const uint8_t *d, int n) {
    if (n < 21) return;
    const int ext = d[0] & 7;
    const uint16_t ident = uint16_t((d[1] << 8) | d[2]);
    std::string label(reinterpret_cast<const char *>(d + 3), 16);
    while (!label.empty() && label.back() == ' ') label.pop_back();
    if (ext == 0) {
        m_updater.SetEnsembleId(ident);
        m_updater.SetEnsembleLabel(label);
    } else if (ext == 1) {
        m_updater.SetServiceLabel(ident, label);
    }
}
