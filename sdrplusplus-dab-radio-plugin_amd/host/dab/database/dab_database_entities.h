// dab/database/dab_database_entities.h -- the part of the reference's database the hot path's callers read:
// field names as the GUI uses them (/root/reference/src/render_radio_block.cpp:239-306, 490-752;
// src/render_formatters.cpp:9-25 for Subchannel{is_uep, uep_prot_index, eep_type, eep_prot_level,
// start_address, length}; :553-578 service description; :599-700 linked services; :730-745 component;
// :780-810 ensemble).  Filled from FIG 0/0, 0/1, 0/2, 0/5, 0/6, 0/8, 0/9, 0/10, 0/17, 0/21, 0/24, 1/0, 1/1, 1/4, 1/5
// (SURVEY.md 8f-4).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

typedef uint8_t subchannel_id_t;
typedef uint8_t service_component_id_t;
typedef uint8_t programme_id_t;
typedef uint8_t language_id_t;
typedef uint8_t country_id_t;
typedef uint8_t extended_country_id_t;

enum class TransportMode : uint8_t { STREAM_MODE_AUDIO = 0, STREAM_MODE_DATA = 1, FIDC = 2, PACKET_MODE_DATA = 3 };
enum class AudioServiceType : uint8_t { DAB = 0, DAB_PLUS = 63, UNDEFINED = 255 };   // other ASCTy values are kept as sent
enum class DataServiceType : uint8_t { UNDEFINED = 0, TRANSPARENT_CHANNEL = 5, MPEG2 = 24, MOT = 60, PROPRIETARY = 63 };
enum class EEP_Type : uint8_t { TYPE_A = 0, TYPE_B = 1 };

struct EnsembleId {
    uint16_t value = 0;
    uint8_t get_country_code() const { return uint8_t(value >> 12); }
    uint16_t get_reference() const { return value & 0x0FFF; }
    uint32_t get_unique_identifier() const { return value; }
};

enum class ServiceIdType : uint8_t { BITS16 = 0, BITS32 = 1 };   // programme services / data services (P/D flag)

struct ServiceId {
    uint32_t value = 0;
    ServiceIdType type = ServiceIdType::BITS16;
    uint32_t get_unique_identifier() const { return value; }
    // 16 bits: country(4) reference(12); 32 bits: ECC(8) country(4) reference(20)
    country_id_t get_country_code() const {
        return country_id_t(type == ServiceIdType::BITS32 ? (value >> 20) & 0xF : (value >> 12) & 0xF);
    }
    extended_country_id_t get_extended_country_code() const {
        return extended_country_id_t(type == ServiceIdType::BITS32 ? value >> 24 : 0);
    }
    bool operator==(const ServiceId &o) const { return value == o.value; }
};

struct Ensemble {
    EnsembleId id;
    std::string label;
    uint8_t nb_services = 0;
    uint16_t reconfiguration_count = 0;
    int cif_counter = -1;          // (upper mod 20) * 250 + (lower mod 250), -1 until FIG 0/0 was seen
    // FIG 0/9
    bool has_country_info = false;
    extended_country_id_t extended_country_code = 0;
    int local_time_offset = 0;     // tenths of an hour (the GUI prints local_time_offset / 10 hours)
    uint8_t international_table_id = 0;
};

struct Service {
    ServiceId id;
    std::string label;
    programme_id_t programme_type = 0;     // FIG 0/17 international code
    bool has_programme_type = false;
    language_id_t language = 0;            // FIG 0/17 (when its language flag is set)
};

struct ServiceComponent {
    ServiceId service_id;
    service_component_id_t component_id = 0;      // SCIdS, from FIG 0/8 (0 until then)
    bool has_component_id = false;
    uint16_t global_id = 0;                       // SCId of packet-mode components (not followed here)
    subchannel_id_t subchannel_id = 0;
    TransportMode transport_mode = TransportMode::STREAM_MODE_AUDIO;
    AudioServiceType audio_service_type = AudioServiceType::UNDEFINED;
    DataServiceType data_service_type = DataServiceType::UNDEFINED;
    bool is_primary = true;
    language_id_t language = 0;                   // FIG 0/5
    std::string label;                            // FIG 1/4
};

// Service linking (FIG 0/6) and the frequencies of what is linked (FIG 0/21): the "Linked Services" tab,
// /root/reference/src/render_radio_block.cpp:599-700.
typedef uint16_t lsn_t;
typedef uint64_t freq_t;                          // Hz (19 bits x 16 kHz does not fit 32 bits)

struct LinkService {
    lsn_t id = 0;                                 // linkage set number
    bool is_active_link = false;
    bool is_hard_link = false;
    bool is_international = false;
    ServiceId service_id;                         // the DAB service the set is anchored on (first id of its DAB list)
    bool has_service_id = false;
};

struct FM_Service {
    uint16_t RDS_PI_code = 0;
    lsn_t linkage_set_number = 0;
    bool has_linkage = false;
    bool is_time_compensated = false;
    std::vector<freq_t> frequencies;
};

struct DRM_Service {
    uint32_t drm_code = 0;                        // 24 bits
    lsn_t linkage_set_number = 0;
    bool has_linkage = false;
    bool is_time_compensated = false;
    std::vector<freq_t> frequencies;
};

// another ensemble carrying (some of) the same services: FIG 0/21 frequencies, FIG 0/24 services
struct OtherEnsemble {
    uint16_t id = 0;
    bool is_continuous_output = false;
    std::vector<freq_t> frequencies;
    std::vector<uint32_t> services;
};

// printable form of a label for logs and tests: anything outside plain ASCII (and the brackets / backslash used as
// delimiters) as \xNN -- any byte may arrive in a damaged FIB
inline std::string escape_label(const std::string &label) {
    static const char hex[] = "0123456789ABCDEF";
    std::string out;
    for (unsigned char c : label) {
        if (c >= 0x20 && c < 0x7F && c != '\\' && c != '[' && c != ']') out.push_back(char(c));
        else { out += "\\x"; out.push_back(hex[c >> 4]); out.push_back(hex[c & 15]); }
    }
    return out;
}

struct Subchannel {
    subchannel_id_t id = 0;
    uint16_t start_address = 0;    // capacity units
    uint16_t length = 0;           // capacity units (0 for a UEP entry: its size comes from the UEP table)
    bool is_uep = false;
    uint8_t uep_prot_index = 0;
    EEP_Type eep_type = EEP_Type::TYPE_A;
    uint8_t eep_prot_level = 0;    // 0..3 as transmitted (the GUI prints eep_prot_level + 1, render_formatters.cpp:14)
};
