// test_fig_parser -- CPU-only: reads FIBs (32 bytes each) from a file, runs the host FIG parser and prints the
// database in the canonical text form tests/test_fig.py compares with the oracle's (oracle/fig_oracle.py).
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <vector>
#include "dab/database/dab_database_text.h"
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
    print_database(stdout, db, updater.GetMiscInfo());
    std::printf("# fibs_bad_crc=%d figs=%d conflicts=%zu\n", bad, parser.GetTotalFIGs(), updater.GetStatistics().nb_conflicts);
    return 0;
}
