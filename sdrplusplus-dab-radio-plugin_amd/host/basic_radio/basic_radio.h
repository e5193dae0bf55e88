// basic_radio/basic_radio.h -- host-side mirror of the reference's BasicRadio for the hot path:
// ctor (/root/reference/src/radio_block.cpp:60), Process(span<const viterbi_bit_t>) (:42), GetMutex()
// (src/render_radio_block.cpp:124), On_Audio_Channel() (src/radio_block.cpp:62).
//
// Process splits a frame into FIC and MSC (row A7) and runs rows A8..A12 on the GPU through libdabgpu: the FIC and
// every registered sub-channel in ONE dabgpu_decode_stream_frames call per frame (one upload of the 230400 soft bits,
// three launches, one download, one synchronisation; the de-interleaver state never leaves the device).  A sub-channel the FIC announces in frame f is decoded from
// frame f+1 on.  The FIBs go through the FIG
// parser into the database; every audio component found there (DAB+ or DAB, EEP or UEP sub-channel)
// gets its sub-channel decoded and a Basic_DAB_Plus_Channel (8f-3) / Basic_DAB_Channel without being asked (On_Audio_Channel fires once per
// channel, as in the reference).  Audio decoding is outside the path: channels hand out access units.
#pragma once
#include <cstdint>
#include <mutex>
#include <vector>
#include <map>
#include <memory>
#include "basic_radio/basic_audio_channel.h"
#include "basic_radio/basic_dab_channel.h"
#include "basic_radio/basic_dab_plus_channel.h"
#include "basic_radio/basic_data_packet_channel.h"
#include "dab/constants/dab_parameters.h"
#include "dab/database/dab_database.h"
#include "dab/database/dab_database_updater.h"
#include "dab/fic/fic_parser.h"
#include "dabgpu.h"
#include "utility/gpu_buffers.h"
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
    // read under GetMutex(), as the GUI does (/root/reference/src/render_radio_block.cpp:124, 158-160, 239, 755)
    DAB_Database &GetDatabase() { return m_database; }
    const DAB_Database_Statistics &GetDatabaseStatistics() const { return m_updater.GetStatistics(); }
    const DAB_Misc_Info &GetMiscInfo() const { return m_updater.GetMiscInfo(); }     // render_radio_block.cpp:814
    Basic_Audio_Channel *Get_Audio_Channel(subchannel_id_t id) {
        auto it = m_channels.find(id);
        return it == m_channels.end() ? nullptr : it->second.get();
    }
    // packet-mode data sub-channels are not followed (render_radio_block.cpp:533): never a channel
    Basic_Data_Packet_Channel *Get_Data_Packet_Channel(subchannel_id_t) { return nullptr; }
    // sub-channels listed in the FIC that cannot be decoded here (invalid profile, does not fit the CIF)
    int GetTotalUnsupportedSubchannels() const { return m_total_unsupported; }
    void SetAutoChannels(bool v) { m_auto_channels = v; }

    // ---- hot-path outputs (extensions) ----
    // 12 FIBs x 32 bytes and their CRC flags, once per frame
    Observable<tcb::span<const uint8_t>, tcb::span<const uint8_t>> &On_FIC() { return m_obs_fic; }
    // decoded logical frame of a registered subchannel: (index from AddSubchannel, bytes); 4 per frame
    Observable<int, tcb::span<const uint8_t>> &On_MSC_Frame() { return m_obs_msc; }
    // returns the subchannel index, or a negative dabgpu_status
    int AddSubchannel(const dabgpu_subchannel &sc);
    int GetTotalFIBs() const { return m_total_fibs; }
    int GetTotalFIBErrors() const { return m_total_fib_errors; }
    // frames whose sub-channels could not be decoded (the per-frame GPU call failed; the FIC was decoded on its own)
    int GetTotalFramesLost() const { return m_total_frames_lost; }
    // test hook (host/demo/dab_host_multi.cpp): the nth decode call into libdabgpu from now on reports a device failure
    void TestFailDeviceCall(int nth) { (void)dabgpu_test_fail_frame_call(m_ctx, nth); }

private:
    struct Subchannel {
        dabgpu_subchannel desc;
        int nbytes;
        int cifs_seen = 0;
        PinnedBuffer<uint8_t> out;
        Basic_DAB_Plus_Channel *dab_plus = nullptr;   // set for sub-channels opened from the database
        Basic_DAB_Channel *dab = nullptr;
    };
    void update_channels_from_database();
    void process_fibs_locked();
    int add_subchannel_locked(const dabgpu_subchannel &sc);
    const DAB_Parameters m_params;
    dabgpu_ctx *m_ctx;
    std::mutex m_mutex;
    std::vector<Subchannel> m_subchannels;
    PinnedBuffer<uint8_t> m_fib, m_crc;
    PinnedBuffer<viterbi_bit_t> m_frame;                 // the frame's soft bits, staged for the single upload
    std::vector<dabgpu_subchannel> m_call_sc;            // argument arrays of the per-frame call
    std::vector<uint8_t *> m_call_out;
    int m_total_fibs = 0, m_total_fib_errors = 0;
    Observable<subchannel_id_t, Basic_Audio_Channel &> m_obs_audio_channel;
    Observable<tcb::span<const uint8_t>, tcb::span<const uint8_t>> m_obs_fic;
    Observable<int, tcb::span<const uint8_t>> m_obs_msc;
    DAB_Database m_database;
    DAB_Database_Updater m_updater{m_database};
    FIC_Parser m_fic_parser{m_updater};
    std::map<subchannel_id_t, std::unique_ptr<Basic_Audio_Channel>> m_channels;
    std::vector<subchannel_id_t> m_rejected;             // sub-channels we looked at and cannot open
    size_t m_seen_components = 0;
    bool m_pending_components = false;                   // a component whose sub-channel is not described yet
    bool m_auto_channels = true;
    int m_total_unsupported = 0;
    int m_total_frames_lost = 0;
};
