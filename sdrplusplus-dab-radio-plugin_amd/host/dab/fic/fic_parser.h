// dab/fic/fic_parser.h -- FIB -> FIGs -> database: the step the reference's BasicRadio runs on
// the decoded FIC (absent vendor/DAB-Radio sub-module) so that the GUI can list the multiplex through
// radio.GetDatabase() (/root/reference/src/render_radio_block.cpp:239-306).  Handles the FIGs a receiver needs to
// find and decode audio services -- 0/0 ensemble, 0/1 sub-channel organisation, 0/2 service organisation, 0/10 date
// and time, 1/0 and 1/1 labels (ETSI EN 300 401 clauses 5.2, 6.2.1, 6.3.1, 6.4, 8.1.3.1, 8.1.13-14) -- and the ones
// behind the rest of what the GUI prints: 0/5 component language, 0/6 service linking, 0/8 component identifiers,
// 0/9 country / local time offset / international table, 0/17 programme type, 0/21 frequency information, 0/24
// services in other ensembles, 1/4 component labels, 1/5 data-service labels (clauses 8.1.2, 8.1.15, 6.3.5, 8.1.3.2,
// 8.1.5, 8.1.8, 8.1.10.2, 8.1.14).  Everything else is skipped by its length field.
#pragma once
#include <cstdint>
#include "dab/database/dab_database_updater.h"
#include "utility/span.h"

class FIC_Parser {
public:
    explicit FIC_Parser(DAB_Database_Updater &updater) : m_updater(updater) {}
    // fib: 32 bytes (30 data + CRC16).  Returns false, touching nothing, when the CRC fails.
    bool ProcessFIB(tcb::span<const uint8_t> fib);
    static uint16_t CRC16(const uint8_t *data, size_t n);
    int GetTotalFIGs() const { return m_total_figs; }

private:
    void fig0(const uint8_t *d, int n);
    void fig0_0(const uint8_t *d, int n);
    void fig0_1(const uint8_t *d, int n);
    void fig0_2(const uint8_t *d, int n, bool pd);
    void fig0_5(const uint8_t *d, int n);
    void fig0_6(const uint8_t *d, int n, bool pd);
    void fig0_8(const uint8_t *d, int n, bool pd);
    void fig0_9(const uint8_t *d, int n);
    void fig0_10(const uint8_t *d, int n);
    void fig0_17(const uint8_t *d, int n);
    void fig0_21(const uint8_t *d, int n);
    void fig0_24(const uint8_t *d, int n, bool pd);
    void fig1(const uint8_t *d, int n);
    DAB_Database_Updater &m_updater;
    int m_total_figs = 0;
};
