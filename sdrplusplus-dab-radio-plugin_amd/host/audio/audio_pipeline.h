// Minimal stand-in for the reference's audio mixer types so radio_block.{h,cpp} and dab_module.h compile
// (/root/reference/src/radio_block.cpp:46,65-75; src/dab_module.h:36-55).  48 kHz audio mixing is outside the
// hot path; sources accept and drop frames.
#pragma once
#include <functional>
#include <memory>
#include <mutex>
#include <string_view>
#include <vector>
#include "utility/span.h"

template <class T>
struct Frame {
    T channels[2];
};

class AudioPipelineSource {
public:
    template <class T>
    void write(tcb::span<const Frame<T>> /*buf*/, float /*sample_rate*/, bool /*is_blocking*/) {}
};

class AudioPipelineSink {
public:
    using Callback = std::function<size_t(tcb::span<Frame<float>>, float)>;
    virtual ~AudioPipelineSink() {}
    virtual void set_callback(Callback callback) = 0;
    virtual std::string_view get_name() const = 0;
};

class AudioPipeline {
public:
    void clear_sources() {
        std::lock_guard<std::mutex> lock(m_mutex);
        m_sources.clear();
    }
    void add_source(std::shared_ptr<AudioPipelineSource> s) {
        std::lock_guard<std::mutex> lock(m_mutex);
        m_sources.push_back(std::move(s));
    }
    AudioPipelineSink *get_sink() { return m_sink.get(); }
    float &get_global_gain() { return m_global_gain; }                 // render_radio_block.cpp:839
    void set_sink(std::unique_ptr<AudioPipelineSink> sink) { m_sink = std::move(sink); }

private:
    std::mutex m_mutex;
    std::vector<std::shared_ptr<AudioPipelineSource>> m_sources;
    std::unique_ptr<AudioPipelineSink> m_sink;
    float m_global_gain = 1.0f;
};
