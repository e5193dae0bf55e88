// wire_mirror_check.cpp -- compile-only proof that dab_upstream_dump.h fits the two hand-over points with the types the
// plugin uses there: the On_OFDM_Frame payload (tcb::span<const viterbi_bit_t>, /root/reference/src/radio_block.cpp:25)
// and, for this repository's host mirror, the FIB / logical-frame observers of BasicRadio (what upstream produces inside
// BasicRadio::Process, src/radio_block.cpp:42).  Compiled with -fsyntax-only by tests/test_upstream_dump.py.
#include "dab_upstream_dump.h"

#include "basic_radio/basic_radio.h"
#include "ofdm/ofdm_demodulator.h"

void wire_dump(OFDM_Demod &demod, BasicRadio &radio, const dabgpu_subchannel &sc, int subchannel_id) {
    dab_upstream_dump::open("/tmp/dabdump", 32);
    demod.On_OFDM_Frame().Attach([&radio](tcb::span<const viterbi_bit_t> buf) {
        dab_upstream_dump::soft_frame(buf.data(), buf.size());                       // radio_block.cpp:25 (OFDM-only builds)
        dab_upstream_dump::begin_radio_frame(buf.data(), buf.size());                // radio_block.cpp:42, before ...
        radio.Process(buf);                                                          // ... radio->Process(data)
    });
    radio.On_FIC().Attach([](tcb::span<const uint8_t> fibs, tcb::span<const uint8_t> crc) {
        for (size_t i = 0; i < crc.size(); i++) dab_upstream_dump::fib(fibs.data() + 32 * i, crc[i] != 0);
    });
    const int index = radio.AddSubchannel(sc);
    radio.On_MSC_Frame().Attach([=](int which, tcb::span<const uint8_t> lf) {
        if (which != index) return;
        const int desc[6] = {sc.start_address, sc.length, sc.is_uep, sc.eep_type, sc.protection_level, sc.bitrate_kbps};
        dab_upstream_dump::msc_logical_frame(subchannel_id, desc, lf.data(), lf.size());
    });
}
