// basic_radio/basic_audio_channel.h -- the audio-channel interface radio_block.cpp hooks its audio pipeline to
// (/root/reference/src/radio_block.cpp:62-79: GetControls(), OnAudioData()).  Audio decoding (AAC/MP2) is outside
// the hot path, so OnAudioData() never fires here; what the channel does produce is listed
// in basic_dab_plus_channel.h.
#pragma once
#include <cstdint>
#include <string_view>
#include "basic_radio/basic_slideshow.h"
#include "dab/database/dab_database_entities.h"
#include "utility/observable.h"
#include "utility/span.h"

struct BasicAudioParams {
    uint32_t frequency = 48000;
    bool is_stereo = true;
    uint8_t bytes_per_sample = 2;
};

class Basic_Audio_Controls {
public:
    bool GetIsPlayAudio() const { return m_play; }
    bool GetIsDecodeAudio() const { return m_decode; }
    bool GetIsDecodeData() const { return m_data; }
    void SetIsPlayAudio(bool v) { m_play = v; }
    void SetIsDecodeAudio(bool v) { m_decode = v; }
    void SetIsDecodeData(bool v) { m_data = v; }
    void StopAll() { m_play = m_decode = m_data = false; }            // render_radio_block.cpp:392-394
    void RunAll() { m_play = m_decode = m_data = true; }

private:
    bool m_play = false, m_decode = true, m_data = true;
};

class Basic_Audio_Channel {
public:
    virtual ~Basic_Audio_Channel() = default;
    // which concrete channel this is (the GUI dispatches on it, /root/reference/src/render_radio_block.cpp:480-487)
    virtual AudioServiceType GetType() const = 0;
    // programme-associated text needs the audio decoder's PAD extraction: never set here
    std::string_view GetDynamicLabel() const { return {}; }
    Basic_Audio_Controls &GetControls() { return m_controls; }
    // slideshows come out of the audio decoder's PAD as well: always empty (render_radio_block.cpp:591)
    Basic_Slideshow_Manager &GetSlideshowManager() { return m_slideshows; }
    Observable<BasicAudioParams, tcb::span<const uint8_t>> &OnAudioData() { return m_obs_audio; }

protected:
    Basic_Audio_Controls m_controls;
    Basic_Slideshow_Manager m_slideshows;
    Observable<BasicAudioParams, tcb::span<const uint8_t>> m_obs_audio;
};
