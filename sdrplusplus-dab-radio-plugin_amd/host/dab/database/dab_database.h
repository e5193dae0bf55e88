// dab/database/dab_database.h -- what BasicRadio::GetDatabase() returns
// (/root/reference/src/render_radio_block.cpp:158, 239, 491, 781).
#pragma once
#include <vector>
#include "dab/database/dab_database_entities.h"

struct DAB_Database {
    Ensemble ensemble;
    std::vector<Service> services;
    std::vector<ServiceComponent> service_components;
    std::vector<Subchannel> subchannels;
    std::vector<LinkService> link_services;        // render_radio_block.cpp:601
    std::vector<FM_Service> fm_services;           // :628
    std::vector<DRM_Service> drm_services;         // :667
    std::vector<OtherEnsemble> other_ensembles;
};

struct DAB_Database_Statistics {           // GetDatabaseStatistics(), render_radio_block.cpp:755
    size_t nb_total = 0;                   // entity fields written
    size_t nb_pending = 0;                 // entities still missing something the FIC has yet to say (label, sub-channel)
    size_t nb_completed = 0;               // entities that are complete
    size_t nb_conflicts = 0;               // a field that was already set arrived with a different value
    size_t nb_updates = 0;                 // writes that changed something
};
