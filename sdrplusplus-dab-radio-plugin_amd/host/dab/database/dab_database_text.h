// dab/database/dab_database_text.h -- the database in a canonical text form, one entity per line, sorted by
// identifier: what tests compare with the Python restatement (oracle/fig_oracle.py Database.lines()) and what the
// demo writes to <prefix>.db.
#pragma once
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>
#include "dab/dab_misc_info.h"
#include "dab/database/dab_database.h"

inline std::string freq_list_text(const std::vector<freq_t> &f) {
    std::string s;
    for (size_t i = 0; i < f.size(); i++) s += (i ? "," : "") + std::to_string(f[i]);
    return s;
}

inline void print_database(std::FILE *f, const DAB_Database &db, const DAB_Misc_Info &mi) {
    if (db.ensemble.cif_counter >= 0 || !db.ensemble.label.empty())
        std::fprintf(f, "ensemble id=%04X label=[%s]\n", unsigned(db.ensemble.id.value), escape_label(db.ensemble.label).c_str());
    if (db.ensemble.has_country_info)
        std::fprintf(f, "ensemble_info ecc=%02X lto=%d inter_table=%d\n", unsigned(db.ensemble.extended_country_code),
                     db.ensemble.local_time_offset, int(db.ensemble.international_table_id));
    auto subs = db.subchannels;
    std::sort(subs.begin(), subs.end(), [](const Subchannel &a, const Subchannel &b) { return a.id < b.id; });
    for (const auto &s : subs)
        std::fprintf(f, "subchannel id=%d start=%d length=%d uep=%d uep_index=%d eep_type=%d eep_level=%d\n", s.id,
                     s.start_address, s.length, int(s.is_uep), s.uep_prot_index, int(s.eep_type), s.eep_prot_level);
    auto svs = db.services;
    std::sort(svs.begin(), svs.end(), [](const Service &a, const Service &b) { return a.id.value < b.id.value; });
    for (const auto &sv : svs) {
        std::fprintf(f, "service id=%04X label=[%s] pty=%d lang=%d bits32=%d\n", unsigned(sv.id.value), escape_label(sv.label).c_str(),
                     sv.has_programme_type ? int(sv.programme_type) : -1, int(sv.language), int(sv.id.type == ServiceIdType::BITS32));
        for (const auto &c : db.service_components)
            if (c.service_id.value == sv.id.value)
                std::fprintf(f, "component service=%04X subchannel=%d tmid=%d ascty=%d primary=%d scids=%d lang=%d label=[%s]\n",
                             unsigned(sv.id.value), c.subchannel_id, int(c.transport_mode), int(c.audio_service_type),
                             int(c.is_primary), c.has_component_id ? int(c.component_id) : -1, int(c.language),
                             escape_label(c.label).c_str());
    }
    auto links = db.link_services;
    std::sort(links.begin(), links.end(), [](const LinkService &a, const LinkService &b) { return a.id < b.id; });
    for (const auto &l : links) {
        std::fprintf(f, "link lsn=%d active=%d hard=%d intl=%d service=", int(l.id), int(l.is_active_link), int(l.is_hard_link),
                     int(l.is_international));
        if (l.has_service_id) std::fprintf(f, "%04X\n", unsigned(l.service_id.value));
        else std::fprintf(f, "none\n");
    }
    auto fms = db.fm_services;
    std::sort(fms.begin(), fms.end(), [](const FM_Service &a, const FM_Service &b) { return a.RDS_PI_code < b.RDS_PI_code; });
    for (const auto &m : fms)
        std::fprintf(f, "fm pi=%04X lsn=%d tc=%d freqs=%s\n", unsigned(m.RDS_PI_code), m.has_linkage ? int(m.linkage_set_number) : -1,
                     int(m.is_time_compensated), freq_list_text(m.frequencies).c_str());
    auto drms = db.drm_services;
    std::sort(drms.begin(), drms.end(), [](const DRM_Service &a, const DRM_Service &b) { return a.drm_code < b.drm_code; });
    for (const auto &m : drms)
        std::fprintf(f, "drm code=%06X lsn=%d tc=%d freqs=%s\n", unsigned(m.drm_code), m.has_linkage ? int(m.linkage_set_number) : -1,
                     int(m.is_time_compensated), freq_list_text(m.frequencies).c_str());
    auto oes = db.other_ensembles;
    std::sort(oes.begin(), oes.end(), [](const OtherEnsemble &a, const OtherEnsemble &b) { return a.id < b.id; });
    for (const auto &o : oes) {
        std::string sv;
        for (size_t i = 0; i < o.services.size(); i++) {
            char buf[16];
            std::snprintf(buf, sizeof(buf), "%s%04X", i ? "," : "", unsigned(o.services[i]));
            sv += buf;
        }
        std::fprintf(f, "other_ensemble id=%04X cont=%d freqs=%s services=%s\n", unsigned(o.id), int(o.is_continuous_output),
                     freq_list_text(o.frequencies).c_str(), sv.c_str());
    }
    if (mi.datetime.year)
        std::fprintf(f, "datetime %04d-%02d-%02d %02u:%02u:%02u.%03u cif=%u\n", mi.datetime.year, mi.datetime.month, mi.datetime.day,
                     mi.datetime.hours, mi.datetime.minutes, mi.datetime.seconds, mi.datetime.milliseconds,
                     mi.cif_counter.GetTotalCount());
}
