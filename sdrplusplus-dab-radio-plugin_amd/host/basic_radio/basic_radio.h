// basic_radio/basic_radio.h -- host-side mirror of the reference's BasicRadio for the hot path:
// ctor (/root/reference/src/radio_block.cpp:60), Process(span<const viterbi_bit_t>) (:42), GetMutex()
// (src/render_radio_block.cpp:124), On_Audio_Channel() (src/radio_block.cpp:62).
//
// Process splits a frame into FIC and MSC (row A7) and runs rows A8..A12 on the GPU through libdabgpu:
// the FIC every frame, plus every subchannel registered with AddSubchannel().  What the reference does after
// that (FIG parsing -> database, DAB+ superframe, audio) is outside the hot path, so the
// decoded bytes are handed out through two observers instead.
#pragma once
#include <cstdint>
#include <mutex>
#include <vector>
#include "basic_radio/basic_audio_channel.h"
#include "dab/constants/dab_parameters.h"
#include "dabgpu.h"
#include "utility/observable.h"
#include "utility/span.h"
#include "viterbi_config.h"

class BasicRadio {
public:
    BasicRadio(const DAB_Parameters &params, size_t nb_threads = 0);
    ~BasicRadio();
    BasicRadio(const BasicRadio &) = delete;
    BasicRadio &operator=(const BasicRadio &) = delete;

    void Process(tcb::span<const viterbi_bit_t> buf);
    std::mutex &GetMutex() { return m_mutex; }
    Observable<subchannel_id_t, Basic_Audio_Channel &> &On_Audio_Channel() { return m_obs_audio_channel; }

    // ---- hot-path outputs (extensions) ----
    // 12 FIBs x 32 bytes and their CRC flags, once per frame
    Observable<tcb::span<const uint8_t>, tcb::span<const uint8_t>> &On_FIC() { return m_obs_fic; }
    // decoded logical frame of a registered subchannel: (index from AddSubchannel, bytes); 4 per frame
    Observable<int, tcb::span<const uint8_t>> &On_MSC_Frame() { return m_obs_msc; }
    // returns the subchannel index, or a negative dabgpu_status
    int AddSubchannel(const dabgpu_subchannel &sc);
    int GetTotalFIBs() const { return m_total_fibs; }
    int GetTotalFIBErrors() const { return m_total_fib_errors; }

private:
    struct Subchannel {
        dabgpu_subchannel desc;
        int nbytes;
        std::vector<int8_t> history[2];   // 15 CIFs of de-interleaver state, ping-pong
        int cur = 0;
        int cifs_seen = 0;
        std::vector<uint8_t> out;
    };
    const DAB_Parameters m_params;
    dabgpu_ctx *m_ctx;
    std::mutex m_mutex;
    std::vector<Subchannel> m_subchannels;
    std::vector<uint8_t> m_fib, m_crc;
    int m_total_fibs = 0, m_total_fib_errors = 0;
    Observable<subchannel_id_t, Basic_Audio_Channel &> m_obs_audio_channel;
    Observable<tcb::span<const uint8_t>, tcb::span<const uint8_t>> m_obs_fic;
    Observable<int, tcb::span<const uint8_t>> m_obs_msc;
};
