// replay.cpp -- test driver: feeds raw arrays (written by tests/test_upstream_dump.py from tests/golden/external_sample.npz)
// through dab_upstream_dump.h in the order and from the two threads the plugin calls it, so that the converter's output
// can be compared with the sample.   usage: replay <in_prefix> <out_prefix> <n_frames> <sub_id> <n_bytes> <first_cif> d0..d5
#include "dab_upstream_dump.h"

#include <cstdlib>
#include <fstream>
#include <iterator>
#include <thread>
#include <vector>

static std::vector<char> slurp(const std::string &p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv) {
    if (argc != 13) return 2;
    const std::string in = argv[1];
    const long n = std::atol(argv[3]);
    const int sub_id = std::atoi(argv[4]);
    const size_t n_bytes = size_t(std::atol(argv[5]));
    const long first_cif = std::atol(argv[6]);
    int desc[6];
    for (int i = 0; i < 6; i++) desc[i] = std::atoi(argv[7 + i]);
    const auto soft = slurp(in + ".soft.bin"), fib = slurp(in + ".fib.bin"), crc = slurp(in + ".crc.bin"), msc = slurp(in + ".msc.bin"),
               iq = slurp(in + ".iq.bin"), fo = slurp(in + ".fo.bin");
    if (!dab_upstream_dump::open(argv[2], n)) return 3;
    const size_t iq_frames = fo.size() / sizeof(float), iq_len = iq_frames ? iq.size() / sizeof(std::complex<float>) / iq_frames : 0;
    // the OFDM thread: samples (and the :25 hook, a no-op beside begin_radio_frame), a frame at a time
    std::thread ofdm([&] {
        for (long f = 0; f < n; f++) {
            if (size_t(f) < iq_frames)
                dab_upstream_dump::iq_frame(reinterpret_cast<const std::complex<float> *>(iq.data()) + size_t(f) * iq_len, iq_len,
                                            reinterpret_cast<const float *>(fo.data())[f]);
        }
    });
    // the radio thread: per frame its soft bits, the 12 FIBs, then per CIF the sub-channel's logical frame once its
    // de-interleaver has filled (it starts in the MIDDLE of a frame: CIF 15 is the last of frame 3)
    std::thread radio([&] {
        for (long f = 0; f < n; f++) {
            dab_upstream_dump::begin_radio_frame(reinterpret_cast<const int8_t *>(soft.data()) + size_t(f) * 230400, 230400);
            for (int i = 0; i < 12; i++)
                dab_upstream_dump::fib(reinterpret_cast<const uint8_t *>(fib.data()) + (size_t(f) * 12 + i) * 32, crc[size_t(f) * 12 + i] != 0);
            for (int c = 0; c < 4; c++) {
                const long t = 4 * f + c;
                if (t >= first_cif)
                    dab_upstream_dump::msc_logical_frame(sub_id, desc, reinterpret_cast<const uint8_t *>(msc.data()) + size_t(t) * n_bytes, n_bytes);
            }
        }
    });
    ofdm.join();
    radio.join();
    dab_upstream_dump::begin_radio_frame(reinterpret_cast<const int8_t *>(soft.data()), 230400);   // beyond max_frames: a no-op
    dab_upstream_dump::fib(reinterpret_cast<const uint8_t *>(fib.data()), true);
    dab_upstream_dump::close();
    return 0;
}
