// CPU-only unit test of the boundary types that need no GPU: tcb::span conversions, Observable, and the
// blocking SPSC ThreadedRingBuffer semantics Radio_Block relies on (/root/reference/src/radio_block.cpp:23-44,53).
#include <cassert>
#include <complex>
#include <cstdio>
#include <mutex>
#include <numeric>
#include <thread>
#include <memory>
#include <vector>
#include "dab/database/dab_database_updater.h"
#include "dab/audio/aac_frame_processor.h"
#include "dab/constants/subchannel_protection_tables.h"
#include "utility/lru_cache.h"

#include "app_helpers/app_io_buffers.h"
#include "dab/constants/dab_parameters.h"
#include "ofdm/dab_mapper_ref.h"
#include "ofdm/dab_ofdm_params_ref.h"
#include "ofdm/dab_prs_ref.h"
#include "basic_radio/basic_data_packet_channel.h"
#include "dab/constants/country_table.h"
#include "dab/constants/language_table.h"
#include "dab/constants/programme_type_table.h"
#include "utility/observable.h"
#include "utility/span.h"

int main() {
    {   // identifiers and the name tables behind the GUI's service / ensemble description
        ServiceId s16, s32;
        s16.value = 0xC221;
        s32.value = 0xE0D12345; s32.type = ServiceIdType::BITS32;
        assert(s16.get_country_code() == 0xC && s16.get_extended_country_code() == 0);
        assert(s32.get_country_code() == 0xD && s32.get_extended_country_code() == 0xE0);
        assert(GetCountryName(0xE1, 0xC) == "United Kingdom" && GetCountryName(0xE0, 0xD) == "Germany" &&
               GetCountryName(0x12, 3) == "Unknown" && GetCountryName(0xF0, 3) == "Australia");
        assert(GetLanguageName(0x09) == "English" && GetLanguageName(0x0F) == "French" && GetLanguageName(0x7F) == "Amharic" &&
               GetLanguageName(0x45) == "Zulu" && GetLanguageName(0x56) == "Russian" && GetLanguageName(0x35) == "Unknown");
        assert(GetProgrammeTypeName(1, 1).long_label == "News" && GetProgrammeTypeName(1, 24).short_label == "Jazz" &&
               GetProgrammeTypeName(1, 29).long_label == "Documentary");
        Subchannel sc;
        Basic_Data_Packet_Channel ch(sc);
        std::lock_guard<std::mutex> lock(ch.GetSlideshowManager().GetSlideshowsMutex());
        assert(ch.GetSlideshowManager().GetSlideshows().empty());
    }
    // span: mutable -> const conversion (dab_module.cpp passes span<complex<float>> to Process(span<const ...>))
    std::vector<std::complex<float>> v(8);
    tcb::span<std::complex<float>> m(v.data(), v.size());
    tcb::span<const std::complex<float>> c = m;
    assert(c.size() == 8 && c.data() == v.data());
    auto s2 = tcb::span(v);
    assert(s2.size() == 8);
    // observable
    Observable<int, int> obs;
    int acc = 0;
    obs.Attach([&](int a, int b) { acc += a * b; });
    obs.Attach([&](int a, int b) { acc += a + b; });
    obs.Notify(3, 4);
    assert(acc == 19);
    // ring buffer: producer blocks when 2 "frames" are queued, consumer sees data in order, close() unblocks
    const size_t frame = 1000;
    ThreadedRingBuffer<int> ring(frame * 2);
    std::thread prod([&] {
        std::vector<int> buf(frame);
        for (int f = 0; f < 50; f++) {
            std::iota(buf.begin(), buf.end(), f * int(frame));
            const size_t n = ring.write(buf);
            assert(n == frame);
        }
        ring.close();
    });
    std::vector<int> rx(frame);
    long long expect = 0;
    int frames = 0;
    while (true) {
        const size_t n = ring.read(rx);
        if (n != frame) break;
        for (int x : rx) assert(x == expect++);
        frames++;
    }
    prod.join();
    assert(frames == 50);
    // a closed ring refuses writes with a short count (the reference exits its thread on that)
    std::vector<int> one(10, 1);
    assert(ring.write(one) == 0);
    // table getters through the reference-named functions
    auto p = get_DAB_OFDM_params(1);
    auto d = get_dab_parameters(1);
    assert(p.nb_fft == 2048 && p.nb_data_carriers == 1536 && d.nb_frame_bits == 230400);
    std::vector<std::complex<float>> prs(p.nb_fft);
    get_DAB_PRS_reference(1, prs);
    std::vector<int> mapper(p.nb_data_carriers);
    get_DAB_mapper_ref(mapper, int(p.nb_fft));
    assert(prs[0] == std::complex<float>(0, 0) && std::abs(prs[1]) == 1.0f && mapper[0] == 255);
    bool threw = false;
    try { get_DAB_OFDM_params(2); } catch (const std::exception &) { threw = true; }
    assert(threw);
    // LRU cache as the GUI's texture cache uses it
    {
        LRU_Cache<uint32_t, std::unique_ptr<int>> cache;
        cache.set_max_size(2);
        cache.emplace(1u, std::make_unique<int>(10));
        cache.emplace(2u, std::make_unique<int>(20));
        assert(cache.find(1u) && **cache.find(1u) == 10);       // 1 is now the most recent
        cache.emplace(3u, std::make_unique<int>(30));           // evicts 2
        assert(cache.find(2u) == nullptr && cache.find(1u) && cache.find(3u) && cache.size() == 2);
        cache.set_max_size(1);
        assert(cache.size() == 1 && cache.find(3u));
    }
    // the database updater keeps the first value of a field and counts later disagreements
    {
        DAB_Database db;
        DAB_Database_Updater up(db);
        up.SetEnsembleLabel("One");
        up.SetEnsembleLabel("Two");
        up.SetEnsembleLabel("One");
        assert(db.ensemble.label == "One" && up.GetStatistics().nb_conflicts == 1);
        assert(escape_label("a[b]\\\x01") == "a\\x5Bb\\x5D\\x5C\\x01");
    }
    // the helpers the reference's formatters call (render_formatters.cpp:18-25); the UEP row comes through the C ABI
    {
        Subchannel eep;
        eep.length = 48; eep.eep_type = EEP_Type::TYPE_A; eep.eep_prot_level = 2;        // "EEP 3-A", 64 kbit/s
        assert(CalculateEEPBitrate(eep) == 64);
        eep.length = 42; eep.eep_type = EEP_Type::TYPE_B; eep.eep_prot_level = 1;        // "EEP 2-B", 64 kbit/s
        assert(CalculateEEPBitrate(eep) == 64);
        eep.length = 47;
        assert(CalculateEEPBitrate(eep) == 0);
        Subchannel uep;
        uep.is_uep = true; uep.uep_prot_index = 17;                                      // 64 kbit/s level 2: 58 CUs
        const auto d = GetUEPDescriptor(uep);
        assert(d.bitrate == 64 && d.protection_level == 2 && d.subchannel_size == 58);
        assert(mpeg_surround_from_config(1) == MPEG_Surround::SURROUND_51);
    }
    std::puts("host types ok");
    return 0;
}
