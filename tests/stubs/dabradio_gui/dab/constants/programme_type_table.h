#pragma once   // TEST-ONLY stub (see ../../README.md)
#include <string>
#include "dab/database/dab_database_entities.h"
struct ProgrammeTypeName {
    std::string long_label;
    std::string short_label;
};
const ProgrammeTypeName &GetProgrammeTypeName(uint8_t inter_table_id, programme_id_t programme_id);
