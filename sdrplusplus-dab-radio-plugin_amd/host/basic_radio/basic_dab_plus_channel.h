// basic_radio/basic_dab_plus_channel.h -- a DAB+ audio sub-channel after the channel decoder: groups logical
// frames into audio super-frames, finds their alignment with the Fire code, corrects them with RS(120,110) and
// checks every access unit's CRC -- all on the GPU through dabgpu_dabplus_superframes -- and
// hands out the access units.  Mirrors what the GUI reads from the reference's Basic_DAB_Plus_Channel:
// GetSuperFrameHeader(), IsFirecodeError(), IsRSError(), IsAUError(), IsCodecError(), GetDynamicLabel()
// (/root/reference/src/render_radio_block.cpp:410-437).  AAC decoding itself is not part of the path, so
// IsCodecError() is always false and GetDynamicLabel() empty; the access units leave through OnAccessUnit().
#pragma once
#include <cstdint>
#include <string_view>
#include <vector>
#include "basic_radio/basic_audio_channel.h"
#include "dab/audio/aac_frame_processor.h"
#include "dabgpu.h"
#include "utility/gpu_buffers.h"

struct SuperFrameHeader {                  // TS 102 563 clause 5.2; fields as the GUI prints them
    uint32_t sampling_rate = 0;            // AAC core output rate, 0 until a super-frame was decoded
    bool is_stereo = false;
    bool is_spectral_band_replication = false;
    bool is_parametric_stereo = false;
    MPEG_Surround mpeg_surround = MPEG_Surround::NOT_USED;
    uint8_t nb_access_units = 0;
};

class Basic_DAB_Plus_Channel : public Basic_Audio_Channel {
public:
    Basic_DAB_Plus_Channel(dabgpu_ctx *ctx, const Subchannel &subchannel, int bitrate_kbps);
    // one decoded logical frame (bitrate*3 bytes) of the sub-channel
    void Process(tcb::span<const uint8_t> logical_frame);
    const SuperFrameHeader &GetSuperFrameHeader() const { return m_header; }
    bool IsFirecodeError() const { return m_firecode_error; }
    bool IsRSError() const { return m_rs_error; }
    bool IsAUError() const { return m_au_error; }
    bool IsCodecError() const { return false; }
    AudioServiceType GetType() const override { return AudioServiceType::DAB_PLUS; }
    // (index of the unit in its super-frame, units in the super-frame, bytes incl. CRC16); only CRC-clean units
    Observable<int, int, tcb::span<const uint8_t>> &OnAccessUnit() { return m_obs_au; }
    int GetTotalSuperFrames() const { return m_total_superframes; }
    int GetTotalAccessUnits() const { return m_total_aus; }
    int GetTotalAccessUnitErrors() const { return m_total_au_errors; }
    const Subchannel &GetSubchannel() const { return m_subchannel; }

private:
    dabgpu_ctx *m_ctx;
    const Subchannel m_subchannel;
    const int m_bitrate;
    const size_t m_lf_bytes;
    // (page-locked: the super-frame kernel reads the window and writes data + status in place, no copies)
    PinnedBuffer<uint8_t> m_window;        // up to 5 logical frames
    int m_frames_in_window = 0;
    bool m_synced = false;
    PinnedBuffer<uint8_t> m_data;          // corrected super-frame (110*s bytes)
    PinnedBuffer<dabgpu_superframe_status> m_status;
    SuperFrameHeader m_header;
    bool m_firecode_error = true, m_rs_error = false, m_au_error = false;
    int m_total_superframes = 0, m_total_aus = 0, m_total_au_errors = 0;
    Observable<int, int, tcb::span<const uint8_t>> m_obs_au;
};
