// ofdm/ofdm_demodulator.h -- host-side mirror of the reference's OFDM_Demod, backed by libdabgpu.
//
// Same names, argument meaning and error behaviour (void Process, no exceptions after construction) as the
// class the plugin uses: ctor (/root/reference/src/radio_block.cpp:22), Process (src/dab_module.cpp:24-25),
// On_OFDM_Frame (src/radio_block.cpp:25), and the getters the GUI reads
// (src/render_radio_block.cpp:96, 109, 192-207, 213-235).
//
// Row A1 of SURVEY.md section 8a lives here on the host: chunk reassembly and the null-symbol power-dip search.
// Everything per sample runs on the GPU, in ONE call per frame (dabgpu_ofdm_demod_stream_frame: one upload, one
// download, one synchronisation): the coarse-frequency / fine-time search on the phase reference symbol (section 8f-1)
// on every frame's PRS -- at acquisition it sets the coarse offset and rejects false locks
// (impulse_peak_threshold_db), afterwards it tracks timing drift and nudges the coarse offset
// (coarse_freq_slow_beta); the tap it aligns to is scored by impulse_peak_distance_probability --, rows A2..A6 with the
// stream's own offsets, the fine-frequency loop, the counters and the level average (signal_l1.update_beta).  The
// offsets live in a dabgpu_stream_state on the device; the getters below return the copy that came back with the last
// frame.  The frame and soft-bit buffers are page-locked (dabgpu_host_alloc).  Differential
// demodulation is insensitive to a constant timing offset inside the cyclic prefix, so the FFT windows are kept
// `TIMING_MARGIN` samples early.
//
// Threads (the GUI polls this object lock-free while the OFDM thread runs Process): every scalar a getter
// returns is a relaxed std::atomic written by the Process thread; Reset() may be called from any thread -- it only raises
// a flag, the Process thread starts over at the top of its next call (the buffers it would have to clear are that
// thread's); the knobs of GetConfig() are plain fields the GUI writes through ImGui's float* / bool* -- Process reads each
// of them ONCE per use with a relaxed atomic load, so a torn or half-updated configuration cannot be observed;
// GetFrameDataVec() is display-only data that the next frame overwrites (as upstream's: it may tear, nothing else).
// tests/test_host_sanitizers.py runs the plugin's wiring under ThreadSanitizer and ASan / UBSan with a third thread
// hammering all of this.
#pragma once
#include <atomic>
#include <complex>
#include <cstdint>
#include <vector>
#include "dabgpu.h"
#include "ofdm/ofdm_params.h"
#include "utility/gpu_buffers.h"
#include "utility/observable.h"
#include "utility/span.h"
#include "viterbi_config.h"

// (No OFDM_DEMOD_SAMPLING_RATE here: the plugin's GUI defines that name itself, /root/reference/src/render_radio_block.cpp:62,
// and must compile unchanged against this header -- tests/test_host_mirror.py.)

struct OFDM_Demod_Config {
    struct {
        float update_beta = 0.95f;
    } signal_l1;
    struct {
        float thresh_null_start = 0.35f;
        float thresh_null_end = 0.75f;
    } null_l1_search;
    struct {
        bool is_coarse_freq_correction = true;
        float fine_freq_update_beta = 0.9f;
        float max_coarse_freq_correction_norm = 0.1f;
        // once locked, a residual of k whole carriers found on a frame's PRS moves the coarse offset by this
        // fraction of k per frame (0 = keep the offset found at acquisition)
        float coarse_freq_slow_beta = 0.1f;
        float impulse_peak_threshold_db = 20.0f;
        // taps of the channel impulse response are scored |h|^2 w^2, w = 1 - (1 - p) |offset - expected| / 2552
        // (dabgpu_track_cfg; the GUI's slider at src/render_radio_block.cpp:225)
        float impulse_peak_distance_probability = 0.15f;
        // extension, OFF by default: false = the fine-frequency loop runs on the 76 cyclic-prefix correlations of the
        // frame, the reference's estimator (fine_freq_update_beta, src/render_radio_block.cpp:216); true = on this
        // library's decision-directed estimator (dabgpu_track_cfg.decision_directed: fourth powers of the differential
        // symbols, only the PRS's prefix read)
        bool is_decision_directed_fine_freq = false;
    } sync;
};

class OFDM_Demod {
public:
    enum class State {
        FINDING_NULL_POWER_DIP,
        READING_NULL_AND_PRS,
        RUNNING_COARSE_FREQ_SYNC,
        RUNNING_FINE_TIME_SYNC,
        READING_SYMBOLS
    };

    OFDM_Demod(const OFDM_Params &params, tcb::span<const std::complex<float>> prs_fft_ref,
               tcb::span<const int> carrier_mapper, int nb_desired_threads = 0);
    ~OFDM_Demod();
    OFDM_Demod(const OFDM_Demod &) = delete;
    OFDM_Demod &operator=(const OFDM_Demod &) = delete;

    void Process(tcb::span<const std::complex<float>> block);
    void Reset();

    OFDM_Params GetOFDMParams() const { return m_params; }
    State GetState() const { return m_state.load(std::memory_order_relaxed); }
    float GetFineFrequencyOffset() const { return m_freq_fine_offset.load(std::memory_order_relaxed); }
    float GetCoarseFrequencyOffset() const { return m_freq_coarse_offset.load(std::memory_order_relaxed); }
    float GetNetFrequencyOffset() const { return GetFineFrequencyOffset() + GetCoarseFrequencyOffset(); }
    float GetSignalAverage() const { return m_signal_l1_average.load(std::memory_order_relaxed); }
    int GetTotalFramesRead() const { return m_total_frames_read.load(std::memory_order_relaxed); }
    int GetTotalFramesDesync() const { return m_total_frames_desync.load(std::memory_order_relaxed); }
    OFDM_Demod_Config &GetConfig() { return m_cfg; }
    // The differential symbols of the last frame (the GUI's constellation, /root/reference/src/render_radio_block.cpp:109).  They
    // are 0.9 MB per frame that nothing but a display reads: the frame call brings them along only while somebody is looking --
    // every call here asks for the next FRAME_DATA_KEEP frames' worth (a GUI polls at its refresh rate; the first look after a
    // pause shows the last frame fetched, the next DAB frame, <= 96 ms later, is fresh).
    static constexpr int FRAME_DATA_KEEP = 32;
    tcb::span<const std::complex<float>> GetFrameDataVec() const {
        m_frame_data_wanted.store(FRAME_DATA_KEEP, std::memory_order_relaxed);
        return {m_frame_data_vec.data(), m_frame_data_vec.size()};
    }
    Observable<tcb::span<const viterbi_bit_t>> &On_OFDM_Frame() { return m_obs_on_ofdm_frame; }

    // extension: apply a known coarse offset (cycles/sample); it goes to the device-side state the demodulation reads
    // (with is_coarse_freq_correction on, the next acquisition overwrites it).  Process thread / before streaming only:
    // it talks to the context.
    void SetCoarseFrequencyOffset(float f) {
        m_freq_coarse_offset.store(f, std::memory_order_relaxed);
        (void)dabgpu_set_stream_offsets(m_ctx, 0, nullptr, &f);
    }
    // test hook (host/demo/dab_host_multi.cpp): the nth frame call into libdabgpu from now on reports a device failure
    void TestFailDeviceCall(int nth) { (void)dabgpu_test_fail_frame_call(m_ctx, nth); }
    int GetFineTimeOffset() const { return m_last_time_offset.load(std::memory_order_relaxed); }
    float GetImpulsePeakDb() const { return m_last_peak_db.load(std::memory_order_relaxed); }

private:
    static constexpr size_t L1_BLOCK = 64;         // power measured in blocks of 64 samples
    static constexpr size_t TIMING_MARGIN = 128;   // place FFT windows this many samples early (inside the CP)

    void push_sample_block(const std::complex<float> *x, size_t n);
    size_t take_locked(const std::complex<float> *x, size_t n);
    void demodulate_frame();
    void reset_now();                               // on the Process thread (or before it exists)

    const OFDM_Params m_params;
    OFDM_Demod_Config m_cfg;
    std::atomic<State> m_state;
    std::atomic<bool> m_reset_requested{false};
    dabgpu_ctx *m_ctx;
    // sync state
    std::atomic<float> m_signal_l1_average;
    bool m_in_null;
    size_t m_null_blocks;
    std::vector<std::complex<float>> m_history;     // last TIMING_MARGIN + block samples, to start a frame early
    PinnedBuffer<std::complex<float>> m_frame;      // 76 * 2552 samples being assembled (page-locked: uploaded every frame)
    size_t m_frame_fill;
    size_t m_skip;                                  // samples to drop before the next frame starts
    size_t m_next_skip;                             // null-symbol gap to the next frame incl. timing correction
    bool m_is_acquiring;                            // first frame after a null detection: coarse sync + lock check
    std::atomic<int> m_last_time_offset;
    std::atomic<float> m_last_peak_db;
    std::vector<std::complex<float>> m_carry;       // partial L1 block between Process calls
    // tracking
    std::atomic<float> m_freq_fine_offset, m_freq_coarse_offset;
    std::atomic<int> m_total_frames_read, m_total_frames_desync;
    int m_host_desyncs = 0;            // lost frames the device never saw (failed calls)
    int m_device_desyncs_seen = 0;
    // outputs
    PinnedBuffer<viterbi_bit_t> m_soft;             // page-locked: downloaded every frame
    PinnedBuffer<std::complex<float>> m_frame_data_vec;
    mutable std::atomic<int> m_frame_data_wanted{0};   // frames that still take the constellation along (GetFrameDataVec)
    Observable<tcb::span<const viterbi_bit_t>> m_obs_on_ofdm_frame;
};
