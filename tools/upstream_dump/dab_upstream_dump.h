// dab_upstream_dump.h -- header-only dumper a maintainer drops into an UPSTREAM build of the plugin to record what crosses
// the two hand-over points of src/radio_block.cpp, as raw .bin files that tools/upstream_dump/to_npz.py turns into the
// tests/external/*.npz format.  One such file turns this repository's oracle -- and through
// it the HIP path -- from "parity unpinned" into a pass / fail line:
//     python tools/upstream_dump/to_npz.py /tmp/dabdump tests/external/offair.npz
//     python -m pytest tests/test_external_vectors.py -q            (CPU: the oracle)      ... -m gpu  (the HIP path)
//
// Where the calls go (upstream file:line as in /root/reference/src; nothing else of upstream changes):
//   src/radio_block.cpp:42   right before radio->Process(data):
//         dab_upstream_dump::begin_radio_frame(data.data(), data.size());
//     -- records the frame's 230400 soft bits (the payload On_OFDM_Frame delivered at :25, as the radio thread reads it
//     back from the ring) and opens the frame everything below is filed under; all on the radio thread, so soft bits,
//     FIBs and sub-channel bytes of a frame can never be misaligned.
//   (src/radio_block.cpp:25, builds that only want the OFDM side -- no begin_radio_frame anywhere: inside the
//     On_OFDM_Frame lambda   dab_upstream_dump::soft_frame(buf.data(), buf.size());)
//   the FIBs and sub-channel bytes are produced INSIDE radio->Process(data) (the DAB-Radio
//         sub-module): one line where a FIB has had its CRC checked --
//         dab_upstream_dump::fib(fib_bytes_32, crc_matches);               // 12 calls per frame, in FIB order
//     and one where a sub-channel's logical frame has left Viterbi + energy dispersal (once per CIF) --
//         dab_upstream_dump::msc_logical_frame(subchannel_id, desc6, bytes, n_bytes);
//     with desc6 = {start_address, length (CUs), is_uep, eep_type (0 = A, 1 = B), protection_level (EEP 1..4 / UEP 1..5),
//     bitrate_kbps} -- the Subchannel entity the GUI prints (src/render_formatters.cpp:9-25).  Against THIS repository's
//     host mirror both exist as observers (BasicRadio::On_FIC, On_MSC_Frame): tools/upstream_dump/wire_mirror_check.cpp.
//   optional, inside OFDM_Demod where the 76 symbols of a frame are complete:
//         dab_upstream_dump::iq_frame(first_prs_prefix_sample, 76 * 2552, net_freq_offset_cycles_per_sample);
//   dab_upstream_dump::open("/tmp/dabdump", max_frames) once before the first frame (Radio_Block's constructor);
//   close() at exit (also runs from a static destructor).  After max_frames frames every call is a no-op.
//
// Thread-safe (the OFDM thread and the radio thread both call in; one mutex), no dependency beyond the C++17 library.
// Files: <prefix>.soft.bin  int8  [frames][230400]      <prefix>.fib.bin  uint8 [frames][12][32]
//        <prefix>.crc.bin   uint8 [frames][12]          <prefix>.msc_<id>.bin  uint8 [rows][n_bytes]
//        <prefix>.msc_<id>.cif.bin  int32 [rows]: the CIF of the dump (4 * frame + 0..3) each row was completed by -- the
//            rows a sub-channel emits during one Process call are taken to be the LAST CIFs of that frame (a
//            de-interleaver that starts emitting mid-frame emits the later ones)
//        <prefix>.sub_<id>.txt  "start length is_uep eep_type level bitrate n_bytes"
//        <prefix>.iq.bin    cf32  [frames][n]           <prefix>.fo.bin   float32 [frames]     <prefix>.meta.txt
#pragma once
#include <complex>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>

namespace dab_upstream_dump {

struct State {
    std::mutex mu;
    std::string prefix;
    long max_frames = 0;
    long soft_frames = 0, fibs = 0, iq_frames = 0;
    long radio_frames = 0;               // begin_radio_frame calls so far (0: the :25 hook records the soft bits instead)
    size_t iq_len = 0;
    std::FILE *f_soft = nullptr, *f_fib = nullptr, *f_crc = nullptr, *f_iq = nullptr, *f_fo = nullptr;
    struct Sub {
        std::FILE *f = nullptr, *f_cif = nullptr;
        long rows = 0;
        size_t n_bytes = 0;
        int rows_this_frame = 0;         // rows written during the current Process call (their CIF indices are still open)
    };
    // the rows of the frame that just ended are the LAST `rows_this_frame` CIFs of it
    void end_frame_locked() {
        if (radio_frames == 0) return;
        for (auto &kv : subs) {
            Sub &sub = kv.second;
            for (int j = 0; j < sub.rows_this_frame && sub.f_cif; j++) {
                const int32_t cif = int32_t(4 * (radio_frames - 1) + (4 - sub.rows_this_frame) + j);
                std::fwrite(&cif, sizeof(cif), 1, sub.f_cif);
            }
            sub.rows_this_frame = 0;
        }
    }
    std::map<int, Sub> subs;
    bool is_open = false;
    ~State() { close_locked(); }
    void close_locked() {
        if (!is_open) return;
        end_frame_locked();
        for (std::FILE *f : {f_soft, f_fib, f_crc, f_iq, f_fo})
            if (f) std::fclose(f);
        f_soft = f_fib = f_crc = f_iq = f_fo = nullptr;
        for (auto &kv : subs) {
            if (kv.second.f) std::fclose(kv.second.f);
            if (kv.second.f_cif) std::fclose(kv.second.f_cif);
        }
        if (std::FILE *m = std::fopen((prefix + ".meta.txt").c_str(), "w")) {
            std::fprintf(m, "soft_frames %ld\nfibs %ld\niq_frames %ld\niq_len %zu\n", soft_frames, fibs, iq_frames, iq_len);
            std::fprintf(m, "radio_frames %ld\n", radio_frames);
            for (auto &kv : subs) std::fprintf(m, "msc_%d_rows %ld\n", kv.first, kv.second.rows);
            std::fclose(m);
        }
        subs.clear();
        is_open = false;
    }
};

inline State &state() {
    static State s;
    return s;
}

inline bool open(const char *prefix, long max_frames = 64) {
    State &s = state();
    std::lock_guard<std::mutex> lock(s.mu);
    s.close_locked();
    s.prefix = prefix;
    s.max_frames = max_frames;
    s.soft_frames = s.fibs = s.iq_frames = s.radio_frames = 0;
    s.iq_len = 0;
    s.f_soft = std::fopen((s.prefix + ".soft.bin").c_str(), "wb");
    s.f_fib = std::fopen((s.prefix + ".fib.bin").c_str(), "wb");
    s.f_crc = std::fopen((s.prefix + ".crc.bin").c_str(), "wb");
    s.is_open = s.f_soft && s.f_fib && s.f_crc;
    return s.is_open;
}

inline void close() {
    State &s = state();
    std::lock_guard<std::mutex> lock(s.mu);
    s.close_locked();
}

// src/radio_block.cpp:42, before radio->Process(data): the frame's soft bits as the radio thread got them from the ring;
// FIBs and sub-channel rows recorded until the next call belong to this frame
inline void begin_radio_frame(const int8_t *bits, size_t n) {
    State &s = state();
    std::lock_guard<std::mutex> lock(s.mu);
    if (!s.is_open || n != 230400) return;
    s.end_frame_locked();
    if (s.radio_frames >= s.max_frames) { s.radio_frames = s.max_frames + 1; return; }
    std::fwrite(bits, 1, n, s.f_soft);
    s.soft_frames++;
    s.radio_frames++;
}

// src/radio_block.cpp:25 -- the payload of On_OFDM_Frame: one frame's 230400 soft bits (viterbi_bit_t = int8).  Only for
// builds without begin_radio_frame (once that has been called the soft bits come from there and this is a no-op).
inline void soft_frame(const int8_t *bits, size_t n) {
    State &s = state();
    std::lock_guard<std::mutex> lock(s.mu);
    if (!s.is_open || s.radio_frames > 0 || s.soft_frames >= s.max_frames || n != 230400) return;
    std::fwrite(bits, 1, n, s.f_soft);
    s.soft_frames++;
}

// inside BasicRadio::Process (src/radio_block.cpp:42): one decoded FIB (30 bytes + CRC16, energy dispersal removed)
inline void fib(const uint8_t *fib32, bool crc_ok) {
    State &s = state();
    std::lock_guard<std::mutex> lock(s.mu);
    if (!s.is_open || s.fibs >= 12 * s.max_frames || s.radio_frames > s.max_frames) return;
    const uint8_t flag = crc_ok ? 1 : 0;
    std::fwrite(fib32, 1, 32, s.f_fib);
    std::fwrite(&flag, 1, 1, s.f_crc);
    s.fibs++;
}

// inside BasicRadio::Process: one logical frame of a followed sub-channel, after Viterbi + energy dispersal (a decoder
// that emits nothing while its 16-CIF de-interleaver fills simply starts later: rows carry their CIF index)
inline void msc_logical_frame(int subchannel_id, const int desc6[6], const uint8_t *bytes, size_t n_bytes) {
    State &s = state();
    std::lock_guard<std::mutex> lock(s.mu);
    if (!s.is_open || s.radio_frames == 0 || s.radio_frames > s.max_frames) return;
    State::Sub &sub = s.subs[subchannel_id];
    if (!sub.f) {
        const std::string base = s.prefix + ".msc_" + std::to_string(subchannel_id);
        sub.f = std::fopen((base + ".bin").c_str(), "wb");
        sub.f_cif = std::fopen((base + ".cif.bin").c_str(), "wb");
        sub.n_bytes = n_bytes;
        if (std::FILE *d = std::fopen((s.prefix + ".sub_" + std::to_string(subchannel_id) + ".txt").c_str(), "w")) {
            std::fprintf(d, "%d %d %d %d %d %d %zu\n", desc6[0], desc6[1], desc6[2], desc6[3], desc6[4], desc6[5], n_bytes);
            std::fclose(d);
        }
    }
    if (!sub.f || !sub.f_cif || n_bytes != sub.n_bytes || sub.rows_this_frame >= 4) return;
    std::fwrite(bytes, 1, n_bytes, sub.f);
    sub.rows++;
    sub.rows_this_frame++;
}

// optional: the samples OFDM_Demod demodulated (from the first sample it treats as PRS cyclic prefix) and the net
// frequency correction it applied to them, cycles per sample (GetNetFrequencyOffset, src/render_radio_block.cpp:204)
inline void iq_frame(const std::complex<float> *x, size_t n, float net_freq_offset) {
    State &s = state();
    std::lock_guard<std::mutex> lock(s.mu);
    if (!s.is_open || s.iq_frames >= s.max_frames || n < size_t(76) * 2552) return;
    if (!s.f_iq) {
        s.f_iq = std::fopen((s.prefix + ".iq.bin").c_str(), "wb");
        s.f_fo = std::fopen((s.prefix + ".fo.bin").c_str(), "wb");
        s.iq_len = n;
    }
    if (!s.f_iq || !s.f_fo || n != s.iq_len) return;
    std::fwrite(x, sizeof(*x), n, s.f_iq);
    std::fwrite(&net_freq_offset, sizeof(float), 1, s.f_fo);
    s.iq_frames++;
}

}  // namespace dab_upstream_dump
