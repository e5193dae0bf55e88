// test_fig_parser -- CPU-only: reads FIBs (32 bytes each) from a file, runs the host FIG parser and prints the
// database in the canonical text form tests/test_fig.py compares with the oracle's (oracle/fig_oracle.py).
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <vector>
#include "dab/fic/fic_parser.h"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    std::ifstream in(argv[1], std::ios::binary);
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    DAB_Database db;
    DAB_Database_Updater updater(db);
    FIC_Parser parser(updater);
    int bad = 0;
    for (size_t i = 0; i + 32 <= buf.size(); i += 32)
        if (!parser.ProcessFIB(tcb::span<const uint8_t>(buf.data() + i, 32))) bad++;
    if (db.ensemble.cif_counter >= 0 || !db.ensemble.label.empty())
        std::printf("ensemble id=%04X label=[%s]\n", unsigned(db.ensemble.id.value), escape_label(db.ensemble.label).c_str());
    auto subs = db.subchannels;
    std::sort(subs.begin(), subs.end(), [](const Subchannel &a, const Subchannel &b) { return a.id < b.id; });
    for (const auto &s : subs)
        std::printf("subchannel id=%d start=%d length=%d uep=%d uep_index=%d eep_type=%d eep_level=%d\n", s.id,
                    s.start_address, s.length, int(s.is_uep), s.uep_prot_index, int(s.eep_type), s.eep_prot_level);
    auto svs = db.services;
    std::sort(svs.begin(), svs.end(), [](const Service &a, const Service &b) { return a.id.value < b.id.value; });
    for (const auto &sv : svs) {
        std::printf("service id=%04X label=[%s]\n", unsigned(sv.id.value), escape_label(sv.label).c_str());
        for (const auto &c : db.service_components)
            if (c.service_id.value == sv.id.value)
                std::printf("component service=%04X subchannel=%d tmid=%d ascty=%d primary=%d\n", unsigned(sv.id.value),
                            c.subchannel_id, int(c.transport_mode), int(c.audio_service_type), int(c.is_primary));
    }
    const auto &mi = updater.GetMiscInfo();
    if (mi.datetime.year)
        std::printf("datetime %04d-%02d-%02d %02u:%02u:%02u.%03u cif=%u\n", mi.datetime.year, mi.datetime.month, mi.datetime.day,
                    mi.datetime.hours, mi.datetime.minutes, mi.datetime.seconds, mi.datetime.milliseconds,
                    mi.cif_counter.GetTotalCount());
    std::printf("# fibs_bad_crc=%d figs=%d conflicts=%zu\n", bad, parser.GetTotalFIGs(), updater.GetStatistics().nb_conflicts);
    return 0;
}
