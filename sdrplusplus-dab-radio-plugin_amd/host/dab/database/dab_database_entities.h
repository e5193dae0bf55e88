// dab/database/dab_database_entities.h -- the part of the reference's database the hot path's callers read:
// field names as the GUI uses them (/root/reference/src/render_radio_block.cpp:239-306, 490-752;
// src/render_formatters.cpp:9-25 for Subchannel{is_uep, uep_prot_index, eep_type, eep_prot_level,
// start_address, length}).  Filled from FIG 0/0, 0/1, 0/2, 0/10, 1/0, 1/1.
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

struct ServiceId {
    uint32_t value = 0;
    uint32_t get_unique_identifier() const { return value; }
    bool operator==(const ServiceId &o) const { return value == o.value; }
};

struct Ensemble {
    EnsembleId id;
    std::string label;
    uint8_t nb_services = 0;
    uint16_t reconfiguration_count = 0;
    int cif_counter = -1;          // (upper mod 20) * 250 + (lower mod 250), -1 until FIG 0/0 was seen
};

struct Service {
    ServiceId id;
    std::string label;
};

struct ServiceComponent {
    ServiceId service_id;
    service_component_id_t component_id = 0;      // position in the service's FIG 0/2 entry
    subchannel_id_t subchannel_id = 0;
    TransportMode transport_mode = TransportMode::STREAM_MODE_AUDIO;
    AudioServiceType audio_service_type = AudioServiceType::UNDEFINED;
    bool is_primary = true;
    std::string label;
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
