// dab/constants/programme_type_table.h -- programme type names the GUI prints for Service::programme_type
// (/root/reference/src/render_formatters.cpp:66-68, src/render_radio_block.cpp:559-561).  International table 1
// (ETSI TS 101 756 table 12, the RDS PTY set) restated from memory -- parity unpinned; table 2 (RBDS, North
// America) is not restated and falls back to table 1.
#pragma once
#include <string>
#include "dab/database/dab_database_entities.h"

struct ProgrammeTypeName {
    std::string long_label;
    std::string short_label;
};

inline const ProgrammeTypeName &GetProgrammeTypeName(uint8_t /*inter_table_id*/, programme_id_t programme_id) {
    static const ProgrammeTypeName TABLE[32] = {
        {"No programme type", "None"}, {"News", "News"}, {"Current Affairs", "Affairs"}, {"Information", "Info"},
        {"Sport", "Sport"}, {"Education", "Educate"}, {"Drama", "Drama"}, {"Culture", "Arts"},
        {"Science", "Science"}, {"Varied", "Talk"}, {"Pop Music", "Pop"}, {"Rock Music", "Rock"},
        {"Easy Listening Music", "Easy"}, {"Light Classical", "Classics"}, {"Serious Classical", "Classics"},
        {"Other Music", "Other_M"}, {"Weather/meteorology", "Weather"}, {"Finance/Business", "Finance"},
        {"Children's programmes", "Children"}, {"Social Affairs", "Factual"}, {"Religion", "Religion"},
        {"Phone In", "Phone_In"}, {"Travel", "Travel"}, {"Leisure", "Leisure"}, {"Jazz Music", "Jazz"},
        {"Country Music", "Country"}, {"National Music", "Nation_M"}, {"Oldies Music", "Oldies"},
        {"Folk Music", "Folk"}, {"Documentary", "Document"}, {"Not used", "Unused"}, {"Not used", "Unused"}};
    return TABLE[programme_id & 31];
}
