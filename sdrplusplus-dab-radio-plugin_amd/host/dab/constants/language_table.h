// dab/constants/language_table.h -- language names for ServiceComponent::language / Service::language
// (/root/reference/src/render_formatters.cpp:70-72).  ETSI TS 101 756 tables 9 (European languages written in Latin
// script, 0x01..0x2B) and 10 (other languages, 0x7F downwards) restated from memory -- parity unpinned.
#pragma once
#include <string>
#include "dab/database/dab_database_entities.h"

inline const std::string &GetLanguageName(language_id_t language_id) {
    static const std::string UNKNOWN = "Unknown";
    static const std::string LOW[0x2C] = {
        "Unknown", "Albanian", "Breton", "Catalan", "Croatian", "Welsh", "Czech", "Danish", "German", "English", "Spanish",
        "Esperanto", "Estonian", "Basque", "Faroese", "French", "Frisian", "Irish", "Gaelic", "Galician", "Icelandic",
        "Italian", "Sami", "Latin", "Latvian", "Luxembourgian", "Lithuanian", "Hungarian", "Maltese", "Dutch", "Norwegian",
        "Occitan", "Polish", "Portuguese", "Romanian", "Romansh", "Serbian", "Slovak", "Slovene", "Finnish", "Swedish",
        "Turkish", "Flemish", "Walloon"};
    static const std::string HIGH[0x7F - 0x45 + 1] = {      // index 0x7F - id
        "Amharic", "Arabic", "Armenian", "Assamese", "Azerbaijani", "Bambora", "Belorussian", "Bengali", "Bulgarian",
        "Burmese", "Chinese", "Chuvash", "Dari", "Fulani", "Georgian", "Greek", "Gujurati", "Gurani", "Hausa", "Hebrew",
        "Hindi", "Indonesian", "Japanese", "Kannada", "Kazakh", "Khmer", "Korean", "Laotian", "Macedonian", "Malagasay",
        "Malaysian", "Moldavian", "Marathi", "Ndebele", "Nepali", "Oriya", "Papiamento", "Persian", "Punjabi", "Pushtu",
        "Quechua", "Russian", "Rusyn", "Serbo-Croat", "Shona", "Sinhalese", "Somali", "Sranan Tongo", "Swahili", "Tadzhik",
        "Tamil", "Tatar", "Telugu", "Thai", "Ukrainian", "Urdu", "Uzbek", "Vietnamese", "Zulu"};
    if (language_id < 0x2C) return LOW[language_id];
    if (language_id >= 0x45 && language_id <= 0x7F) return HIGH[0x7F - language_id];
    return UNKNOWN;
}
