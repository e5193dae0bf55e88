// dab/database/dab_database_updater.h -- find-or-create access to the database for the FIG parser.
// A field is written once; a later FIG carrying a different value for it counts as a conflict and is ignored
// (a corrupted FIB that slipped through the CRC must not rewrite the multiplex).
#pragma once
#include <string>
#include "dab/dab_misc_info.h"
#include "dab/database/dab_database.h"

class DAB_Database_Updater {
public:
    explicit DAB_Database_Updater(DAB_Database &db) : m_db(db) {}
    // pending / completed are recounted from the database on every call
    const DAB_Database_Statistics &GetStatistics() const {
        size_t done = (m_have_eid && !m_db.ensemble.label.empty()) ? 1 : 0, all = 1;
        for (const auto &s : m_db.services) { all++; done += s.label.empty() ? 0 : 1; }
        for (const auto &c : m_db.service_components) {
            all++;
            for (const auto &sc : m_db.subchannels)
                if (sc.id == c.subchannel_id) { done++; break; }
        }
        all += m_db.subchannels.size();
        done += m_db.subchannels.size();
        m_stats.nb_completed = done;
        m_stats.nb_pending = all - done;
        return m_stats;
    }

    void SetEnsembleId(uint16_t eid) {
        if (!m_have_eid) { m_db.ensemble.id.value = eid; m_have_eid = true; changed(); }
        else same(m_db.ensemble.id.value == eid);
    }
    void SetEnsembleCIFCounter(uint8_t upper, uint8_t lower) {
        m_db.ensemble.cif_counter = int(upper) * 250 + int(lower);
        m_misc.cif_counter.upper_count = upper;
        m_misc.cif_counter.lower_count = lower;
    }
    void SetDateTime(const DAB_Date_Time &dt) { m_misc.datetime = dt; }
    const DAB_Misc_Info &GetMiscInfo() const { return m_misc; }
    void SetEnsembleLabel(const std::string &label) { set_string(m_db.ensemble.label, label); }

    Subchannel &GetSubchannel(subchannel_id_t id, bool *is_new) {
        for (auto &s : m_db.subchannels)
            if (s.id == id) { *is_new = false; return s; }
        m_db.subchannels.emplace_back();
        m_db.subchannels.back().id = id;
        *is_new = true;
        changed();
        return m_db.subchannels.back();
    }
    Service *FindService(uint32_t sid) {
        for (auto &s : m_db.services)
            if (s.id.value == sid) return &s;
        return nullptr;
    }
    Service &GetService(uint32_t sid, bool bits32 = false) {
        for (auto &s : m_db.services)
            if (s.id.value == sid) return s;
        m_db.services.emplace_back();
        m_db.services.back().id.value = sid;
        m_db.services.back().id.type = bits32 ? ServiceIdType::BITS32 : ServiceIdType::BITS16;
        m_db.ensemble.nb_services = uint8_t(m_db.services.size());
        changed();
        return m_db.services.back();
    }
    void SetServiceLabel(uint32_t sid, const std::string &label) { set_string(GetService(sid).label, label); }
    ServiceComponent &GetServiceComponent(uint32_t sid, subchannel_id_t subchannel_id, bool *is_new) {
        for (auto &c : m_db.service_components)
            if (c.service_id.value == sid && c.subchannel_id == subchannel_id) { *is_new = false; return c; }
        m_db.service_components.emplace_back();
        auto &c = m_db.service_components.back();
        c.service_id.value = sid;
        c.subchannel_id = subchannel_id;
        *is_new = true;
        changed();
        return c;
    }
    // a value written once; a later different value is a conflict and is ignored
    template <class T>
    void set_once(T &dst, bool &has, const T &v) {
        if (!has) { dst = v; has = true; changed(); }
        else same(dst == v);
    }
    void SetCountryInfo(uint8_t ecc, int lto_tenths, uint8_t inter_table) {
        Ensemble &e = m_db.ensemble;
        if (!e.has_country_info) {
            e.extended_country_code = ecc; e.local_time_offset = lto_tenths; e.international_table_id = inter_table;
            e.has_country_info = true;
            changed();
        } else {
            same(e.extended_country_code == ecc && e.local_time_offset == lto_tenths && e.international_table_id == inter_table);
        }
    }
    LinkService &GetLinkService(lsn_t lsn, bool *is_new) {
        for (auto &l : m_db.link_services)
            if (l.id == lsn) { *is_new = false; return l; }
        m_db.link_services.emplace_back();
        m_db.link_services.back().id = lsn;
        *is_new = true;
        changed();
        return m_db.link_services.back();
    }
    FM_Service &GetFMService(uint16_t pi) {
        for (auto &f : m_db.fm_services)
            if (f.RDS_PI_code == pi) return f;
        m_db.fm_services.emplace_back();
        m_db.fm_services.back().RDS_PI_code = pi;
        changed();
        return m_db.fm_services.back();
    }
    DRM_Service &GetDRMService(uint32_t code) {
        for (auto &f : m_db.drm_services)
            if (f.drm_code == code) return f;
        m_db.drm_services.emplace_back();
        m_db.drm_services.back().drm_code = code;
        changed();
        return m_db.drm_services.back();
    }
    OtherEnsemble &GetOtherEnsemble(uint16_t eid) {
        for (auto &o : m_db.other_ensembles)
            if (o.id == eid) return o;
        m_db.other_ensembles.emplace_back();
        m_db.other_ensembles.back().id = eid;
        changed();
        return m_db.other_ensembles.back();
    }
    template <class T>
    void add_unique(std::vector<T> &v, T x) {
        for (const T &y : v)
            if (y == x) return;
        if (v.size() < 64) { v.push_back(x); changed(); }      // a hostile FIC must not grow the lists without bound
    }
    void SetComponentLabel(ServiceComponent &c, const std::string &label) { set_string(c.label, label); }
    std::vector<ServiceComponent> &Components() { return m_db.service_components; }
    void same(bool ok) { m_stats.nb_total++; if (!ok) m_stats.nb_conflicts++; }
    void changed() { m_stats.nb_total++; m_stats.nb_updates++; }

private:
    void set_string(std::string &dst, const std::string &v) {
        if (dst.empty() && !v.empty()) { dst = v; changed(); }
        else same(dst == v);
    }
    DAB_Database &m_db;
    mutable DAB_Database_Statistics m_stats;
    DAB_Misc_Info m_misc;
    bool m_have_eid = false;
};
