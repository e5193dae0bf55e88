// basic_radio/basic_dab_channel.h -- a DAB (MPEG-1/2 layer II) audio sub-channel after the channel decoder.
// A logical frame of such a sub-channel IS one MP2 audio frame (24 ms at 48 kHz; two logical frames at 24 kHz),
// so there is nothing to reassemble: the channel checks the MPEG header of every frame, keeps what the GUI shows
// through GetAudioParams() (/root/reference/src/render_radio_block.cpp:439-460) and hands the frames to whoever
// decodes MP2 (not part of the hot path; OnAudioData() never fires).  These services normally sit on UEP
// sub-channels.
#pragma once
#include <cstdint>
#include <optional>
#include "basic_radio/basic_audio_channel.h"

// what the GUI prints about a layer-II stream (/root/reference/src/render_radio_block.cpp:444-467): the names of the
// reference's MP2 decoder front end; only the frame header is parsed here
struct MP2_Audio_Decoder {
    enum class MPEG_Version : uint8_t { MPEG_1_0, MPEG_2_0, MPEG_2_5 };
    enum class MPEG_Layer : uint8_t { LAYER_I, LAYER_II, LAYER_III };
    struct AudioParams {
        MPEG_Version mpeg_version = MPEG_Version::MPEG_1_0;
        MPEG_Layer mpeg_layer = MPEG_Layer::LAYER_II;
        uint32_t sample_rate = 0;
        uint32_t bitrate_kbps = 0;
        bool is_stereo = false;
    };
};

class Basic_DAB_Channel : public Basic_Audio_Channel {
public:
    Basic_DAB_Channel(const Subchannel &subchannel, int bitrate_kbps)
        : m_subchannel(subchannel), m_lf_bytes(size_t(bitrate_kbps) * 3) {}
    AudioServiceType GetType() const override { return AudioServiceType::DAB; }
    void Process(tcb::span<const uint8_t> lf) {
        if (lf.size() != m_lf_bytes || lf.size() < 4) return;
        m_total_frames++;
        // ISO 11172-3 header: 12 sync bits, ID (1 = MPEG-1 48 kHz, 0 = MPEG-2 LSF 24 kHz in DAB), layer II = 10b,
        // protection, bit-rate index, sampling frequency (01b = 48/24 kHz), padding, private, mode (11b = mono)
        static const uint16_t BITRATES_V1[16] = {0, 32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384, 0};
        static const uint16_t BITRATES_V2[16] = {0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, 0};
        const bool sync = lf[0] == 0xFF && (lf[1] & 0xF0) == 0xF0;
        const bool layer2 = ((lf[1] >> 1) & 3) == 2;
        const bool fs_ok = ((lf[2] >> 2) & 3) == 1;
        if (!(sync && layer2 && fs_ok)) {
            m_total_header_errors++;
            m_is_error = true;
            return;
        }
        m_is_error = false;
        const bool v1 = (lf[1] & 0x08) != 0;
        MP2_Audio_Decoder::AudioParams p;
        p.mpeg_version = v1 ? MP2_Audio_Decoder::MPEG_Version::MPEG_1_0 : MP2_Audio_Decoder::MPEG_Version::MPEG_2_0;
        p.mpeg_layer = MP2_Audio_Decoder::MPEG_Layer::LAYER_II;
        p.sample_rate = v1 ? 48000u : 24000u;
        p.bitrate_kbps = (v1 ? BITRATES_V1 : BITRATES_V2)[lf[2] >> 4];
        p.is_stereo = ((lf[3] >> 6) & 3) != 3;
        m_params = p;
        if (m_controls.GetIsDecodeAudio()) m_obs_frame.Notify(lf);
    }
    const std::optional<MP2_Audio_Decoder::AudioParams> &GetAudioParams() const { return m_params; }
    bool GetIsError() const { return m_is_error; }
    // one MPEG layer II frame (as transmitted, header first)
    Observable<tcb::span<const uint8_t>> &OnMP2Frame() { return m_obs_frame; }
    int GetTotalFrames() const { return m_total_frames; }
    int GetTotalHeaderErrors() const { return m_total_header_errors; }
    const Subchannel &GetSubchannel() const { return m_subchannel; }

private:
    const Subchannel m_subchannel;
    const size_t m_lf_bytes;
    std::optional<MP2_Audio_Decoder::AudioParams> m_params;
    bool m_is_error = false;
    int m_total_frames = 0, m_total_header_errors = 0;
    Observable<tcb::span<const uint8_t>> m_obs_frame;
};
