// Minimal stand-in so that radio_block.cpp's audio hook compiles (/root/reference/src/radio_block.cpp:62-79).
// Audio decoding (AAC/MP2) is outside the hot path: BasicRadio never creates one of these.
#pragma once
#include <cstdint>
#include "utility/observable.h"
#include "utility/span.h"

typedef uint8_t subchannel_id_t;

struct BasicAudioParams {
    uint32_t frequency = 48000;
    bool is_stereo = true;
    uint8_t bytes_per_sample = 2;
};

class Basic_Audio_Controls {
public:
    bool GetIsPlayAudio() const { return m_play; }
    void SetIsPlayAudio(bool v) { m_play = v; }
    void SetIsDecodeAudio(bool v) { m_decode = v; }
    void SetIsDecodeData(bool v) { m_data = v; }

private:
    bool m_play = false, m_decode = true, m_data = true;
};

class Basic_Audio_Channel {
public:
    Basic_Audio_Controls &GetControls() { return m_controls; }
    Observable<BasicAudioParams, tcb::span<const uint8_t>> &OnAudioData() { return m_obs; }

private:
    Basic_Audio_Controls m_controls;
    Observable<BasicAudioParams, tcb::span<const uint8_t>> m_obs;
};
