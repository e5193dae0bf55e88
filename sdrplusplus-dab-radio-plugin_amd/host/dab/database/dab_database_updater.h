// dab/database/dab_database_updater.h -- find-or-create access to the database for the FIG parser.
// A field is written once; a later FIG carrying a different value for it counts as a conflict and is ignored
// (a corrupted FIB that slipped through the CRC must not rewrite the multiplex).
#pragma once
#include <algorithm>
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

    // Entities are found or created by id with a linear search, under the radio mutex: every list is capped, so a
    // hostile or garbage FIC that passes the CRC cannot grow them (24-bit DRM ids, 16-bit PI codes / EIds, 32-bit
    // service ids ...) into quadratic work.  An entity beyond the cap lands in a scratch object that nobody lists, and
    // counts as a conflict.
    static constexpr size_t MAX_ENTITIES = 1024;

    Subchannel &GetSubchannel(subchannel_id_t id, bool *is_new) {
        for (auto &s : m_db.subchannels)
            if (s.id == id) { *is_new = false; return s; }
        *is_new = true;
        Subchannel &s = create(m_db.subchannels, m_sink_subchannel);
        s.id = id;
        return s;
    }
    Service *FindService(uint32_t sid) {
        for (auto &s : m_db.services)
            if (s.id.value == sid) return &s;
        return nullptr;
    }
    Service &GetService(uint32_t sid, bool bits32 = false) {
        for (auto &s : m_db.services)
            if (s.id.value == sid) return s;
        Service &s = create(m_db.services, m_sink_service);
        s.id.value = sid;
        s.id.type = bits32 ? ServiceIdType::BITS32 : ServiceIdType::BITS16;
        m_db.ensemble.nb_services = uint8_t(std::min<size_t>(m_db.services.size(), 255));
        return s;
    }
    void SetServiceLabel(uint32_t sid, const std::string &label) { set_string(GetService(sid).label, label); }
    ServiceComponent &GetServiceComponent(uint32_t sid, subchannel_id_t subchannel_id, bool *is_new) {
        for (auto &c : m_db.service_components)
            if (c.service_id.value == sid && c.subchannel_id == subchannel_id) { *is_new = false; return c; }
        auto &c = create(m_db.service_components, m_sink_component);
        c.service_id.value = sid;
        c.subchannel_id = subchannel_id;
        *is_new = true;
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
        *is_new = true;
        LinkService &l = create(m_db.link_services, m_sink_link);
        l.id = lsn;
        return l;
    }
    FM_Service &GetFMService(uint16_t pi) {
        for (auto &f : m_db.fm_services)
            if (f.RDS_PI_code == pi) return f;
        FM_Service &f = create(m_db.fm_services, m_sink_fm);
        f.RDS_PI_code = pi;
        return f;
    }
    DRM_Service &GetDRMService(uint32_t code) {
        for (auto &f : m_db.drm_services)
            if (f.drm_code == code) return f;
        DRM_Service &f = create(m_db.drm_services, m_sink_drm);
        f.drm_code = code;
        return f;
    }
    OtherEnsemble &GetOtherEnsemble(uint16_t eid) {
        for (auto &o : m_db.other_ensembles)
            if (o.id == eid) return o;
        OtherEnsemble &o = create(m_db.other_ensembles, m_sink_other);
        o.id = eid;
        return o;
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
    // a new entity at the end of its list, or (list full) the scratch object, reset
    template <class T>
    T &create(std::vector<T> &v, T &sink) {
        if (v.size() >= MAX_ENTITIES) {
            same(false);
            sink = T{};
            return sink;
        }
        v.emplace_back();
        changed();
        return v.back();
    }
    Subchannel m_sink_subchannel;
    Service m_sink_service;
    ServiceComponent m_sink_component;
    LinkService m_sink_link;
    FM_Service m_sink_fm;
    DRM_Service m_sink_drm;
    OtherEnsemble m_sink_other;
    void set_string(std::string &dst, const std::string &v) {
        if (dst.empty() && !v.empty()) { dst = v; changed(); }
        else same(dst == v);
    }
    DAB_Database &m_db;
    mutable DAB_Database_Statistics m_stats;
    DAB_Misc_Info m_misc;
    bool m_have_eid = false;
};
