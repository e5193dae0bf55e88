// dab_host_demo -- drives the host-side mirror the way Radio_Block does
// (/root/reference/src/radio_block.cpp:11-49): OFDM_Demod -> ThreadedRingBuffer -> radio thread ->
// BasicRadio::Process, fed from a cf32 file in arbitrary chunks like OFDM_Demodulator_Sink::run
// (/root/reference/src/dab_module.cpp:20-28).  Writes the decoded FIBs / CRC flags / MSC bytes to files so a
// test can compare them with what was transmitted.  The multiplex is discovered from the FIC: <prefix>.db lists the
// database, <prefix>.aus holds every CRC-clean DAB+ access unit as records
// {u8 sub-channel id, u8 index, u8 units in super-frame, u16le length, bytes}.
//
//   dab_host_demo <iq.cf32> <out_prefix> [chunk_samples] [coarse_offset_cycles_per_sample] [bitrate start_cu level] [signal_l1.update_beta]
#include <algorithm>
#include <chrono>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "app_helpers/app_io_buffers.h"
#include "dab/database/dab_database_text.h"
#include "basic_radio/basic_radio.h"
#include "dab/database/dab_database_text.h"
#include "dab/constants/dab_parameters.h"
#include "ofdm/dab_mapper_ref.h"
#include "ofdm/dab_ofdm_params_ref.h"
#include "ofdm/dab_prs_ref.h"
#include "ofdm/ofdm_demodulator.h"

static constexpr float SAMPLING_RATE_HZ = 2.048e6f;   // the offsets are in cycles per sample (src/render_radio_block.cpp:202)

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s iq.cf32 out_prefix [chunk] [coarse] [bitrate start_cu level]\n", argv[0]);
        return 2;
    }
    const size_t chunk = argc > 3 ? size_t(std::atol(argv[3])) : 65536;
    const float coarse = argc > 4 ? float(std::atof(argv[4])) : 0.0f;
    constexpr int TRANSMISSION_MODE = 1;
    const auto ofdm_params = get_DAB_OFDM_params(TRANSMISSION_MODE);
    const auto dab_params = get_dab_parameters(TRANSMISSION_MODE);
    auto prs = std::vector<std::complex<float>>(ofdm_params.nb_fft);
    get_DAB_PRS_reference(TRANSMISSION_MODE, prs);
    auto mapper = std::vector<int>(ofdm_params.nb_data_carriers);
    get_DAB_mapper_ref(mapper, int(ofdm_params.nb_fft));

    std::shared_ptr<OFDM_Demod> demod;
    std::shared_ptr<BasicRadio> radio;
    try {
        demod = std::make_shared<OFDM_Demod>(ofdm_params, prs, mapper, 1);
        radio = std::make_shared<BasicRadio>(dab_params, 1);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "fatal: %s\n", e.what());
        return 3;
    }
    demod->SetCoarseFrequencyOffset(coarse);
    if (argc > 8) demod->GetConfig().signal_l1.update_beta = float(std::atof(argv[8]));   // the GUI's "L1 signal update beta"
    if (argc > 7) {
        dabgpu_subchannel sc{};
        sc.bitrate_kbps = std::atoi(argv[5]);
        sc.start_address = std::atoi(argv[6]);
        sc.protection_level = std::atoi(argv[7]);
        sc.eep_type = 0;
        const int n = sc.bitrate_kbps / 8;
        const int lens[5] = {0, 12 * n, 8 * n, 6 * n, 4 * n};
        sc.length = lens[sc.protection_level];
        if (radio->AddSubchannel(sc) < 0) { std::fprintf(stderr, "bad subchannel\n"); return 4; }
    }
    const std::string prefix = argv[2];
    std::ofstream f_fib(prefix + ".fib", std::ios::binary), f_crc(prefix + ".crc", std::ios::binary),
        f_msc(prefix + ".msc", std::ios::binary), f_aus(prefix + ".aus", std::ios::binary);
    // as Radio_Block::reset_radio does (/root/reference/src/radio_block.cpp:62): one hook per audio channel
    radio->On_Audio_Channel().Attach([&](subchannel_id_t id, Basic_Audio_Channel &channel) {
        if (auto *mp2 = dynamic_cast<Basic_DAB_Channel *>(&channel)) {
            // DAB (layer II) services: one record per MPEG frame, index/total = 0/1
            mp2->OnMP2Frame().Attach([&f_aus, id](tcb::span<const uint8_t> fr) {
                const uint8_t hdr[5] = {id, 0, 1, uint8_t(fr.size() & 0xFF), uint8_t(fr.size() >> 8)};
                f_aus.write(reinterpret_cast<const char *>(hdr), 5);
                f_aus.write(reinterpret_cast<const char *>(fr.data()), std::streamsize(fr.size()));
            });
            return;
        }
        auto *dabplus = dynamic_cast<Basic_DAB_Plus_Channel *>(&channel);
        if (!dabplus) return;
        dabplus->OnAccessUnit().Attach([&f_aus, id](int index, int total, tcb::span<const uint8_t> au) {
            const uint8_t hdr[5] = {id, uint8_t(index), uint8_t(total), uint8_t(au.size() & 0xFF), uint8_t(au.size() >> 8)};
            f_aus.write(reinterpret_cast<const char *>(hdr), 5);
            f_aus.write(reinterpret_cast<const char *>(au.data()), std::streamsize(au.size()));
        });
    });
    radio->On_FIC().Attach([&](tcb::span<const uint8_t> fib, tcb::span<const uint8_t> ok) {
        f_fib.write(reinterpret_cast<const char *>(fib.data()), std::streamsize(fib.size()));
        f_crc.write(reinterpret_cast<const char *>(ok.data()), std::streamsize(ok.size()));
    });
    radio->On_MSC_Frame().Attach([&](int, tcb::span<const uint8_t> bytes) {
        f_msc.write(reinterpret_cast<const char *>(bytes.data()), std::streamsize(bytes.size()));
    });

    auto ring = std::make_shared<ThreadedRingBuffer<viterbi_bit_t>>(size_t(dab_params.nb_frame_bits) * 2);
    demod->On_OFDM_Frame().Attach([ring](tcb::span<const viterbi_bit_t> buf) { ring->write(buf); });
    double t_radio = 0.0, t_ofdm = 0.0;                        // seconds spent inside the two Process calls
    std::thread radio_thread([&]() {
        auto data = std::vector<viterbi_bit_t>(size_t(dab_params.nb_frame_bits));
        while (true) {
            const size_t n = ring->read(data);
            if (n != data.size()) break;
            const auto t0 = std::chrono::steady_clock::now();
            radio->Process(data);
            t_radio += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
    });

    std::ifstream in(argv[1], std::ios::binary);
    if (!in) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 5; }
    // DAB_DEMO_PRELOAD=1: the whole file is read into memory first and only the processing is timed -- what the plugin's two
    // threads cost per frame when the samples are already there (SDR++ hands over blocks in memory), without this demo's
    // file reads in the OFDM thread
    const bool preload = std::getenv("DAB_DEMO_PRELOAD") != nullptr;
    std::vector<std::complex<float>> all;
    if (preload) {
        in.seekg(0, std::ios::end);
        all.resize(size_t(in.tellg()) / sizeof(all[0]));
        in.seekg(0);
        in.read(reinterpret_cast<char *>(all.data()), std::streamsize(all.size() * sizeof(all[0])));
    }
    // DAB_DEMO_CONSTELLATION=1: poll GetFrameDataVec() once per chunk, as the GUI's constellation view does
    const bool look = std::getenv("DAB_DEMO_CONSTELLATION") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<std::complex<float>> buf(chunk);
    if (preload) {
        for (size_t at = 0; at < all.size(); at += chunk) {
            const auto t0 = std::chrono::steady_clock::now();
            demod->Process(tcb::span<std::complex<float>>(all.data() + at, std::min(chunk, all.size() - at)));
            if (look) (void)demod->GetFrameDataVec();
            t_ofdm += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
    } else {
        while (in) {
            in.read(reinterpret_cast<char *>(buf.data()), std::streamsize(chunk * sizeof(buf[0])));
            const size_t got = size_t(in.gcount()) / sizeof(buf[0]);
            if (!got) break;
            demod->Process(tcb::span<std::complex<float>>(buf.data(), got));   // non-const span, as dab_module.cpp does
            if (look) (void)demod->GetFrameDataVec();
        }
    }
    ring->close();
    radio_thread.join();
    const double t_proc = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    {
        auto lock = std::scoped_lock(radio->GetMutex());
        const auto &db = radio->GetDatabase();
        std::FILE *f = std::fopen((prefix + ".db").c_str(), "w");
        if (f) {
            // the multiplex as the FIC describes it (the date/time line is left out: it moves from frame to frame)
            DAB_Misc_Info no_time;
            print_database(f, db, no_time);
            auto subs = db.subchannels;
            std::sort(subs.begin(), subs.end(), [](const Subchannel &a, const Subchannel &b) { return a.id < b.id; });
            for (const auto &sc : subs) {
                if (auto *mp2 = dynamic_cast<Basic_DAB_Channel *>(radio->Get_Audio_Channel(sc.id))) {
                    const auto &ap = mp2->GetAudioParams();
                    std::fprintf(f, "channel subchannel=%d mp2_frames=%d header_errors=%d rate=%u stereo=%d\n", sc.id,
                                 mp2->GetTotalFrames(), mp2->GetTotalHeaderErrors(), ap ? ap->sample_rate : 0u,
                                 ap ? int(ap->is_stereo) : 0);
                    continue;
                }
                auto *ch = dynamic_cast<Basic_DAB_Plus_Channel *>(radio->Get_Audio_Channel(sc.id));
                if (!ch) continue;
                const auto &h = ch->GetSuperFrameHeader();
                std::fprintf(f, "channel subchannel=%d superframes=%d aus=%d au_errors=%d rate=%u sbr=%d stereo=%d firecode_error=%d rs_error=%d\n",
                             sc.id, ch->GetTotalSuperFrames(), ch->GetTotalAccessUnits(), ch->GetTotalAccessUnitErrors(),
                             h.sampling_rate, int(h.is_spectral_band_replication), int(h.is_stereo), int(ch->IsFirecodeError()),
                             int(ch->IsRSError()));
            }
            std::fclose(f);
        }
    }
    {
        // what the constellation holds at the end: unit-power differential symbols when somebody looked, nothing otherwise
        double p = 0.0;
        const auto vec = look ? demod->GetFrameDataVec() : tcb::span<const std::complex<float>>();
        for (const auto &d : vec) p += double(std::norm(d));
        std::printf("constellation_mean_power=%.6g\n", vec.empty() ? 0.0 : p / double(vec.size()));
    }
    std::printf("processing_s=%.6f ofdm_process_s=%.6f radio_process_s=%.6f\n", t_proc, t_ofdm, t_radio);
    std::printf("state=%d frames_read=%d frames_desync=%d fine=%.6g net=%.6g level=%.4f fibs=%d fib_errors=%d\n",
                int(demod->GetState()), demod->GetTotalFramesRead(), demod->GetTotalFramesDesync(),
                demod->GetFineFrequencyOffset() * SAMPLING_RATE_HZ,
                demod->GetNetFrequencyOffset() * SAMPLING_RATE_HZ, demod->GetSignalAverage(),
                radio->GetTotalFIBs(), radio->GetTotalFIBErrors());
    return 0;
}
