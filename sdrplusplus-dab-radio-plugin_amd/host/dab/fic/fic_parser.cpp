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
    case 5: fig0_5(d + 1, n - 1); break;
    case 6: fig0_6(d + 1, n - 1, pd); break;
    case 8: fig0_8(d + 1, n - 1, pd); break;
    case 9: fig0_9(d + 1, n - 1); break;
    case 10: fig0_10(d + 1, n - 1); break;
    case 17: fig0_17(d + 1, n - 1); break;
    case 21: fig0_21(d + 1, n - 1); break;
    case 24: fig0_24(d + 1, n - 1, pd); break;
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
        m_updater.GetService(sid, pd);
        for (int c = 0; c < nc; c++, i += 2) {
            const uint8_t b0 = d[i], b1 = d[i + 1];
            if ((b0 >> 6) != 0) continue;                   // only MSC stream audio is followed
            bool is_new;
            ServiceComponent &sc = m_updater.GetServiceComponent(sid, b1 >> 2, &is_new);
            const auto ascty = static_cast<AudioServiceType>(b0 & 0x3F);     // 0 = DAB, 63 = DAB+, others kept as sent
            if (is_new) {
                sc.service_id.type = pd ? ServiceIdType::BITS32 : ServiceIdType::BITS16;
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

// FIG 0/5 service component language.  Short form: L/S=0, MSC/FIC flag, SubChId(6), language(8); long form: L/S=1,
// Rfa(3), SCId(12), language(8) (packet-mode components: not followed).
void FIC_Parser::fig0_5(const uint8_t *d, int n) {
    int i = 0;
    while (i < n) {
        if (d[i] & 0x80) {
            if (i + 3 > n) return;
            i += 3;
            continue;
        }
        if (i + 2 > n) return;
        const bool fic = (d[i] >> 6) & 1;
        const subchannel_id_t id = d[i] & 0x3F;
        const language_id_t lang = d[i + 1];
        i += 2;
        if (fic) continue;
        for (auto &c : m_updater.Components())
            if (c.subchannel_id == id && c.transport_mode == TransportMode::STREAM_MODE_AUDIO) {
                if (c.language == 0 && lang != 0) { c.language = lang; m_updater.changed(); }
                else m_updater.same(c.language == lang);
            }
    }
}

// FIG 0/6 service linking: Id list flag(1) LA(1) S/H(1) ILS(1) LSN(12); with a list: Rfu(1) IdLQ(2) Rfa(1) count(4)
// then ids of 16 bits (P/D = 0, national), 8 + 16 bits (P/D = 0, international: ECC + id) or 32 bits (P/D = 1).
// IdLQ: 0 = DAB service ids, 1 = RDS PI codes, 3 = DRM service ids.
void FIC_Parser::fig0_6(const uint8_t *d, int n, bool pd) {
    int i = 0;
    while (i + 2 <= n) {
        const unsigned w = unsigned(d[i] << 8) | d[i + 1];
        const bool has_list = (w >> 15) & 1, la = (w >> 14) & 1, hard = (w >> 13) & 1, ils = (w >> 12) & 1;
        const lsn_t lsn = lsn_t(w & 0x0FFF);
        i += 2;
        if (!has_list) continue;                             // a bare status entry: nothing to record
        if (i + 1 > n) return;
        const int idlq = (d[i] >> 5) & 3, count = d[i] & 0x0F;
        i += 1;
        const int idlen = pd ? 4 : (ils ? 3 : 2);
        if (i + idlen * count > n) return;
        bool is_new;
        LinkService &link = m_updater.GetLinkService(lsn, &is_new);
        if (is_new) {
            link.is_active_link = la; link.is_hard_link = hard; link.is_international = ils;
        } else {
            m_updater.same(link.is_active_link == la && link.is_hard_link == hard && link.is_international == ils);
        }
        for (int k = 0; k < count; k++, i += idlen) {
            uint32_t id = 0;
            for (int b = 0; b < idlen; b++) id = (id << 8) | d[i + b];
            if (pd || idlq == 0) {
                if (k == 0) {                                // the DAB list is anchored on its first service
                    ServiceId sid;
                    sid.value = pd ? id : (id & 0xFFFF);
                    sid.type = pd ? ServiceIdType::BITS32 : ServiceIdType::BITS16;
                    if (!link.has_service_id) { link.service_id = sid; link.has_service_id = true; m_updater.changed(); }
                    else m_updater.same(link.service_id.value == sid.value);
                }
            } else if (idlq == 1) {
                FM_Service &fm = m_updater.GetFMService(uint16_t(id & 0xFFFF));
                m_updater.set_once(fm.linkage_set_number, fm.has_linkage, lsn);
            } else if (idlq == 3) {
                DRM_Service &drm = m_updater.GetDRMService(id & 0xFFFFFF);
                m_updater.set_once(drm.linkage_set_number, drm.has_linkage, lsn);
            }
        }
    }
}

// FIG 0/8 service component global definition: SId, Ext flag(1) Rfa(3) SCIdS(4), then L/S=0: MSC/FIC(1) SubChId(6)
// or L/S=1: Rfa(3) SCId(12); one more Rfa byte when the Ext flag is set.
void FIC_Parser::fig0_8(const uint8_t *d, int n, bool pd) {
    const int idlen = pd ? 4 : 2;
    int i = 0;
    while (i + idlen + 2 <= n) {
        uint32_t sid = 0;
        for (int k = 0; k < idlen; k++) sid = (sid << 8) | d[i + k];
        const bool ext = d[i + idlen] >> 7;
        const service_component_id_t scids = d[i + idlen] & 0x0F;
        const uint8_t c = d[i + idlen + 1];
        const bool long_form = c >> 7;
        const int size = idlen + 2 + (long_form ? 1 : 0) + (ext ? 1 : 0);
        if (i + size > n) return;
        i += size;
        if (long_form || ((c >> 6) & 1)) continue;           // packet mode / FIDC: not followed
        for (auto &comp : m_updater.Components())
            if (comp.service_id.value == sid && comp.subchannel_id == (c & 0x3F))
                m_updater.set_once(comp.component_id, comp.has_component_id, scids);
    }
}

// FIG 0/9 country, LTO and international table: Ext flag(1) Rfa(1) LTO(6: sign + half hours), ECC(8), table id(8)
void FIC_Parser::fig0_9(const uint8_t *d, int n) {
    if (n < 3) return;
    const int mag = d[0] & 0x1F;
    const int lto = ((d[0] >> 5) & 1) ? -5 * mag : 5 * mag;          // tenths of an hour
    m_updater.SetCountryInfo(d[1], lto, d[2]);
}

// FIG 0/17 programme type: SId(16), S/D(1) Rfa(1) L flag(1) CC flag(1) Rfa(4), [language(8)], Rfa(3) int code(5),
// [Rfa(3) complementary code(5)].  Later editions of the standard keep the two flags at zero.
void FIC_Parser::fig0_17(const uint8_t *d, int n) {
    int i = 0;
    while (i + 4 <= n) {
        const uint32_t sid = uint32_t(d[i] << 8) | d[i + 1];
        const bool has_lang = (d[i + 2] >> 5) & 1, has_cc = (d[i + 2] >> 4) & 1;
        const int size = 4 + (has_lang ? 1 : 0) + (has_cc ? 1 : 0);
        if (i + size > n) return;
        const language_id_t lang = has_lang ? d[i + 3] : 0;
        const programme_id_t code = d[i + 3 + (has_lang ? 1 : 0)] & 0x1F;
        i += size;
        Service *sv = m_updater.FindService(sid);
        if (!sv) continue;                                   // FIG 0/2 has not introduced it yet: it will be repeated
        m_updater.set_once(sv->programme_type, sv->has_programme_type, code);
        if (has_lang) {
            if (sv->language == 0 && lang != 0) { sv->language = lang; m_updater.changed(); }
            else m_updater.same(sv->language == lang);
        }
    }
}

// FIG 0/21 frequency information: blocks of Rfa(11) length(5); inside, entries Id(16) R&M(4) continuity(1)
// list length(3) + list.  R&M 0: DAB ensemble, 3 bytes control(5) frequency(19) x 16 kHz; 8: FM with RDS, 1 byte,
// 87.5 MHz + 100 kHz steps; 6: DRM, one byte extending the id to 24 bits, then 2 bytes Rfu(1) frequency(15) in kHz.
void FIC_Parser::fig0_21(const uint8_t *d, int n) {
    int i = 0;
    while (i + 2 <= n) {
        const int fi_len = d[i + 1] & 0x1F;
        i += 2;
        if (i + fi_len > n) return;
        const uint8_t *blk = d + i;
        i += fi_len;
        int j = 0;
        while (j + 3 <= fi_len) {
            const uint16_t id = uint16_t((blk[j] << 8) | blk[j + 1]);
            const int rm = blk[j + 2] >> 4, len = blk[j + 2] & 7;
            const bool cont = (blk[j + 2] >> 3) & 1;
            j += 3;
            if (j + len > fi_len) break;
            const uint8_t *fl = blk + j;
            j += len;
            if (rm == 0) {
                OtherEnsemble &oe = m_updater.GetOtherEnsemble(id);
                oe.is_continuous_output = oe.is_continuous_output || cont;
                for (int k = 0; k + 3 <= len; k += 3)
                    m_updater.add_unique(oe.frequencies, freq_t((fl[k] & 7u) << 16 | unsigned(fl[k + 1]) << 8 | fl[k + 2]) * 16000u);
            } else if (rm == 8) {
                FM_Service &fm = m_updater.GetFMService(id);
                fm.is_time_compensated = fm.is_time_compensated || cont;
                for (int k = 0; k < len; k++) m_updater.add_unique(fm.frequencies, freq_t(87500000u + 100000u * fl[k]));
            } else if (rm == 6) {
                if (len < 1) continue;
                DRM_Service &drm = m_updater.GetDRMService((uint32_t(fl[0]) << 16) | id);
                drm.is_time_compensated = drm.is_time_compensated || cont;
                for (int k = 1; k + 2 <= len; k += 2)
                    m_updater.add_unique(drm.frequencies, freq_t((((fl[k] & 0x7Fu) << 8) | fl[k + 1]) * 1000u));
            }
        }
    }
}

// FIG 0/24 services carried in other ensembles: SId, Rfa(1) CAId(3) number of EIds(4), EIds
void FIC_Parser::fig0_24(const uint8_t *d, int n, bool pd) {
    const int idlen = pd ? 4 : 2;
    int i = 0;
    while (i + idlen + 1 <= n) {
        uint32_t sid = 0;
        for (int k = 0; k < idlen; k++) sid = (sid << 8) | d[i + k];
        const int count = d[i + idlen] & 0x0F;
        i += idlen + 1;
        if (i + 2 * count > n) return;
        for (int k = 0; k < count; k++, i += 2)
            m_updater.add_unique(m_updater.GetOtherEnsemble(uint16_t((d[i] << 8) | d[i + 1])).services, sid);
    }
}

static std::string label_of(const uint8_t *p) {
    std::string label(reinterpret_cast<const char *>(p), 16);
    while (!label.empty() && label.back() == ' ') label.pop_back();
    return label;
}

// FIG 1: charset(4) Rfu(1) extension(3), identifier, 16 characters, 16 flag bits.  Extension 0: ensemble (EId),
// 1: programme service (SId 16), 4: service component (P/D(1) Rfa(3) SCIdS(4), SId 16 or 32), 5: data service (SId 32).
void FIC_Parser::fig1(const uint8_t *d, int n) {
    const int ext = d[0] & 7;
    if (ext == 0 || ext == 1) {
        if (n < 21) return;
        const uint16_t ident = uint16_t((d[1] << 8) | d[2]);
        if (ext == 0) {
            m_updater.SetEnsembleId(ident);
            m_updater.SetEnsembleLabel(label_of(d + 3));
        } else {
            m_updater.SetServiceLabel(ident, label_of(d + 3));
        }
    } else if (ext == 5) {
        if (n < 23) return;
        const uint32_t sid = (uint32_t(d[1]) << 24) | (uint32_t(d[2]) << 16) | (uint32_t(d[3]) << 8) | d[4];
        Service *sv = m_updater.FindService(sid);
        if (sv) m_updater.SetServiceLabel(sid, label_of(d + 5));
    } else if (ext == 4) {
        if (n < 2) return;
        const bool pd = d[1] >> 7;
        const service_component_id_t scids = d[1] & 0x0F;
        const int idlen = pd ? 4 : 2;
        if (n < 2 + idlen + 18) return;
        uint32_t sid = 0;
        for (int k = 0; k < idlen; k++) sid = (sid << 8) | d[2 + k];
        for (auto &c : m_updater.Components())
            if (c.service_id.value == sid && c.has_component_id && c.component_id == scids)
                m_updater.SetComponentLabel(c, label_of(d + 2 + idlen));
    }
}
