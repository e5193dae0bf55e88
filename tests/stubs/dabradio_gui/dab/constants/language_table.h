#pragma once   // TEST-ONLY stub (see ../../README.md)
#include <string>
#include "dab/database/dab_database_entities.h"
const std::string &GetLanguageName(language_id_t language_id);
