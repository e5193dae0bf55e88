// dab_host_multi -- N plugin instances at once in ONE process on ONE device.  The plugin is a multi-instance module
// (/root/reference/src/main.cpp:12 MAX_INSTANCES -1, :22-30 one DABModule -- hence one Radio_Block -- per instance); each
// Radio_Block owns an OFDM_Demod and a BasicRadio (two libdabgpu contexts) and two threads (the sink's run loop,
// src/dab_module.cpp:20-28, and the radio thread, src/radio_block.cpp:33-44).  This program builds N such wirings, feeds each
// its own cf32 file in chunks from a thread of its own, all starting together, and writes per instance what it decoded:
// <prefix>.<i>.fib / .crc / .msc, and one line of counters.
//
//   dab_host_multi <out_prefix> <chunk_samples> <bitrate> <start_cu> <level> <iq0.cf32> [<iq1.cf32> ...]
//
// Test hooks (environment, "instance:n"): DAB_MULTI_FAIL_OFDM -- that instance's nth frame call into libdabgpu reports a
// device failure; DAB_MULTI_FAIL_RADIO -- its nth decode call does.  Process() must not throw on either
// (tests/test_host_multi.py).
#include <atomic>
#include <chrono>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "app_helpers/app_io_buffers.h"
#include "basic_radio/basic_radio.h"
#include "dab/constants/dab_parameters.h"
#include "ofdm/dab_mapper_ref.h"
#include "ofdm/dab_ofdm_params_ref.h"
#include "ofdm/dab_prs_ref.h"
#include "ofdm/ofdm_demodulator.h"

namespace {

struct Instance {
    std::shared_ptr<OFDM_Demod> demod;
    std::shared_ptr<BasicRadio> radio;
    std::shared_ptr<ThreadedRingBuffer<viterbi_bit_t>> ring;
    std::vector<std::complex<float>> iq;
    std::ofstream f_fib, f_crc, f_msc;
    std::thread feeder, radio_thread;
    double t_ofdm = 0.0, t_radio = 0.0;      // seconds inside the two Process calls
    int frames_out = 0;                      // frames the demodulator handed on
    bool threw = false;
};

bool hook(const char *name, int instance, int *nth) {
    const char *v = std::getenv(name);
    int i = -1, n = 0;
    if (!v || std::sscanf(v, "%d:%d", &i, &n) != 2 || i != instance || n <= 0) return false;
    *nth = n;
    return true;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 7) {
        std::fprintf(stderr, "usage: %s out_prefix chunk bitrate start_cu level iq0.cf32 [iq1.cf32 ...]\n", argv[0]);
        return 2;
    }
    const std::string prefix = argv[1];
    const size_t chunk = size_t(std::atol(argv[2]));
    constexpr int TRANSMISSION_MODE = 1;
    const auto ofdm_params = get_DAB_OFDM_params(TRANSMISSION_MODE);
    const auto dab_params = get_dab_parameters(TRANSMISSION_MODE);
    auto prs = std::vector<std::complex<float>>(ofdm_params.nb_fft);
    get_DAB_PRS_reference(TRANSMISSION_MODE, prs);
    auto mapper = std::vector<int>(ofdm_params.nb_data_carriers);
    get_DAB_mapper_ref(mapper, int(ofdm_params.nb_fft));
    dabgpu_subchannel sc{};
    sc.bitrate_kbps = std::atoi(argv[3]);
    sc.start_address = std::atoi(argv[4]);
    sc.protection_level = std::atoi(argv[5]);
    {
        const int n = sc.bitrate_kbps / 8;
        const int lens[5] = {0, 12 * n, 8 * n, 6 * n, 4 * n};
        if (sc.protection_level < 1 || sc.protection_level > 4) { std::fprintf(stderr, "bad level\n"); return 4; }
        sc.length = lens[sc.protection_level];
    }
    const int n_inst = argc - 6;
    std::vector<std::unique_ptr<Instance>> inst;
    for (int i = 0; i < n_inst; i++) {
        auto in = std::make_unique<Instance>();
        try {
            // as Radio_Block::Radio_Block does (/root/reference/src/radio_block.cpp:11-28)
            in->demod = std::make_shared<OFDM_Demod>(ofdm_params, prs, mapper, 1);
            in->radio = std::make_shared<BasicRadio>(dab_params, 1);
        } catch (const std::exception &e) {
            std::fprintf(stderr, "fatal: %s\n", e.what());
            return 3;
        }
        if (in->radio->AddSubchannel(sc) < 0) { std::fprintf(stderr, "bad subchannel\n"); return 4; }
        int nth = 0;
        if (hook("DAB_MULTI_FAIL_OFDM", i, &nth)) in->demod->TestFailDeviceCall(nth);
        if (hook("DAB_MULTI_FAIL_RADIO", i, &nth)) in->radio->TestFailDeviceCall(nth);
        const std::string p = prefix + "." + std::to_string(i);
        in->f_fib.open(p + ".fib", std::ios::binary);
        in->f_crc.open(p + ".crc", std::ios::binary);
        in->f_msc.open(p + ".msc", std::ios::binary);
        Instance *self = in.get();
        in->radio->On_FIC().Attach([self](tcb::span<const uint8_t> fib, tcb::span<const uint8_t> ok) {
            self->f_fib.write(reinterpret_cast<const char *>(fib.data()), std::streamsize(fib.size()));
            self->f_crc.write(reinterpret_cast<const char *>(ok.data()), std::streamsize(ok.size()));
        });
        in->radio->On_MSC_Frame().Attach([self](int, tcb::span<const uint8_t> bytes) {
            self->f_msc.write(reinterpret_cast<const char *>(bytes.data()), std::streamsize(bytes.size()));
        });
        in->ring = std::make_shared<ThreadedRingBuffer<viterbi_bit_t>>(size_t(dab_params.nb_frame_bits) * 2);
        auto ring = in->ring;
        in->demod->On_OFDM_Frame().Attach([ring, self](tcb::span<const viterbi_bit_t> buf) { self->frames_out++; ring->write(buf); });
        std::ifstream f(argv[6 + i], std::ios::binary);
        if (!f) { std::fprintf(stderr, "cannot open %s\n", argv[6 + i]); return 5; }
        f.seekg(0, std::ios::end);
        in->iq.resize(size_t(f.tellg()) / sizeof(in->iq[0]));
        f.seekg(0);
        f.read(reinterpret_cast<char *>(in->iq.data()), std::streamsize(in->iq.size() * sizeof(in->iq[0])));
        inst.push_back(std::move(in));
    }
    // every thread of every instance starts on one flag
    std::atomic<bool> go{false};
    for (auto &up : inst) {
        Instance *in = up.get();
        in->radio_thread = std::thread([in, &dab_params, &go]() {
            auto data = std::vector<viterbi_bit_t>(size_t(dab_params.nb_frame_bits));
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
            try {
                while (true) {
                    const size_t n = in->ring->read(data);
                    if (n != data.size()) break;
                    const auto t0 = std::chrono::steady_clock::now();
                    in->radio->Process(data);
                    in->t_radio += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                }
            } catch (...) { in->threw = true; }
        });
        in->feeder = std::thread([in, chunk, &go]() {
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
            try {
                for (size_t at = 0; at < in->iq.size(); at += chunk) {
                    const auto t0 = std::chrono::steady_clock::now();
                    in->demod->Process(tcb::span<std::complex<float>>(in->iq.data() + at, std::min(chunk, in->iq.size() - at)));
                    in->t_ofdm += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                }
            } catch (...) { in->threw = true; }
            in->ring->close();
        });
    }
    const auto t_begin = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for (auto &up : inst) { up->feeder.join(); up->radio_thread.join(); }
    const double t_proc = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    int rc = 0;
    for (int i = 0; i < n_inst; i++) {
        Instance &in = *inst[size_t(i)];
        std::printf("instance=%d threw=%d state=%d frames_read=%d frames_desync=%d frames_out=%d fibs=%d fib_errors=%d frames_lost=%d "
                    "ofdm_process_s=%.6f radio_process_s=%.6f\n", i, int(in.threw), int(in.demod->GetState()),
                    in.demod->GetTotalFramesRead(), in.demod->GetTotalFramesDesync(), in.frames_out, in.radio->GetTotalFIBs(),
                    in.radio->GetTotalFIBErrors(), in.radio->GetTotalFramesLost(), in.t_ofdm, in.t_radio);
        if (in.threw) rc = 6;
    }
    std::printf("instances=%d processing_s=%.6f\n", n_inst, t_proc);
    return rc;
}
