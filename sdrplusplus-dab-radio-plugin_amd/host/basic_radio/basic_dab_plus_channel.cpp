#include "basic_radio/basic_dab_plus_channel.h"

#include <cstring>

Basic_DAB_Plus_Channel::Basic_DAB_Plus_Channel(dabgpu_ctx *ctx, const Subchannel &subchannel, int bitrate_kbps)
    : m_ctx(ctx), m_subchannel(subchannel), m_bitrate(bitrate_kbps), m_lf_bytes(size_t(bitrate_kbps) * 3) {
    m_window.resize(5 * m_lf_bytes);
    m_data.resize(size_t(110) * (bitrate_kbps / 8));
    m_status.resize(1);
}

void Basic_DAB_Plus_Channel::Process(tcb::span<const uint8_t> lf) {
    if (lf.size() != m_lf_bytes) return;
    std::memcpy(m_window.data() + size_t(m_frames_in_window) * m_lf_bytes, lf.data(), m_lf_bytes);
    if (++m_frames_in_window < 5) return;
    // five logical frames: a super-frame if we are aligned; the Fire code (checked after RS correction) says so
    dabgpu_superframe_status &st = m_status[0];
    const int rc = dabgpu_dabplus_superframes(m_ctx, m_window.data(), m_window.size(), 1, m_bitrate, m_data.data(), &st);
    if (rc != DABGPU_OK || !st.firecode_ok) {
        // not aligned (or a lost super-frame): slide by one logical frame and try again with the next one
        m_firecode_error = true;
        m_synced = false;
        std::memmove(m_window.data(), m_window.data() + m_lf_bytes, 4 * m_lf_bytes);
        m_frames_in_window = 4;
        return;
    }
    m_synced = true;
    m_frames_in_window = 0;
    m_total_superframes++;
    m_firecode_error = false;
    m_rs_error = st.rs_uncorrectable != 0;
    // header byte 2: rfa | dac_rate | sbr_flag | aac_channel_mode | ps_flag | mpeg_surround_config(3)
    const uint8_t h = m_data[2];
    const bool dac_rate = (h >> 6) & 1, sbr = (h >> 5) & 1, stereo = (h >> 4) & 1, ps = (h >> 3) & 1;
    m_header.sampling_rate = dac_rate ? 48000 : 32000;
    m_header.is_spectral_band_replication = sbr;
    m_header.is_stereo = stereo;
    m_header.is_parametric_stereo = ps;
    m_header.mpeg_surround = mpeg_surround_from_config(h & 7);
    m_header.nb_access_units = uint8_t(st.num_aus);
    bool au_error = false;
    for (int a = 0; a < st.num_aus; a++) {
        m_total_aus++;
        if (!((st.au_crc_mask >> a) & 1)) {
            au_error = true;
            m_total_au_errors++;
            continue;
        }
        // the access unit handed on is the payload alone: its trailing CRC16 (already checked on the GPU) is dropped,
        // which is what an AAC decoder / ADTS muxer behind the observer expects
        const int b = st.au_start[a], e = st.au_start[a + 1] - 2;
        if (b < 0 || e <= b || size_t(e) + 2 > m_data.size()) continue;
        if (m_controls.GetIsDecodeAudio())
            m_obs_au.Notify(a, st.num_aus, tcb::span<const uint8_t>(m_data.data() + b, size_t(e - b)));
    }
    m_au_error = au_error;
}
