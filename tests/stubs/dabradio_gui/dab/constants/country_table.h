#pragma once   // TEST-ONLY stub (see ../../README.md)
#include <string>
#include "dab/database/dab_database_entities.h"
const std::string &GetCountryName(extended_country_id_t ecc, country_id_t country_id);
