// ofdm/ofdm_demodulator.cpp -- see the header.  Host work is O(samples) additions for the power detector and
// memcpy into the frame buffer; every per-sample DSP operation of rows A2..A6 is inside libdabgpu.
#include "ofdm/ofdm_demodulator.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

#ifdef DABHOST_TIMING
#include <chrono>
#include <cstdio>
static double g_ofdm_t[3];
static long g_ofdm_n;
struct OfdmLap {
    int i;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    explicit OfdmLap(int k) : i(k) {}
    ~OfdmLap() { g_ofdm_t[i] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); }
};
#define OFDM_LAP(i) OfdmLap lap_##i(i)
#else
#define OFDM_LAP(i)
#endif

// one relaxed atomic read of a knob the GUI thread may be writing (GetConfig(): plain fields, see the header)
template <class T>
static inline T knob(const T &v) {
    T out;
    __atomic_load(&v, &out, __ATOMIC_RELAXED);
    return out;
}

OFDM_Demod::OFDM_Demod(const OFDM_Params &params, tcb::span<const std::complex<float>> prs_fft_ref,
                       tcb::span<const int> carrier_mapper, int /*nb_desired_threads*/)
    : m_params(params), m_state(State::FINDING_NULL_POWER_DIP), m_ctx(nullptr) {
    // the device tables are generated from the same constants; reject anything else loudly
    dabgpu_ofdm_params ref;
    if (dabgpu_get_ofdm_params(1, &ref) != DABGPU_OK || size_t(ref.nb_fft) != params.nb_fft ||
        size_t(ref.nb_data_carriers) != params.nb_data_carriers || prs_fft_ref.size() != params.nb_fft ||
        carrier_mapper.size() != params.nb_data_carriers)
        throw std::runtime_error("OFDM_Demod: only DAB transmission mode I is supported");
    std::vector<int32_t> mapper(params.nb_data_carriers);
    dabgpu_get_mapper_reference(mapper.data(), int(mapper.size()), int(params.nb_fft));
    for (size_t i = 0; i < mapper.size(); i++)
        if (mapper[i] != carrier_mapper[i]) throw std::runtime_error("OFDM_Demod: carrier mapper differs from Mode I");
    dabgpu_cfg cfg{};
    cfg.device = GetDabGpuDefaultDevice();
    cfg.max_frames = 1;
    cfg.transmission_mode = 1;
    int rc = dabgpu_create(&cfg, &m_ctx);
    if (rc == DABGPU_OK && (rc = dabgpu_streams_reset(m_ctx, 1)) != DABGPU_OK) {
        dabgpu_destroy(m_ctx);
        m_ctx = nullptr;
    }
    if (rc != DABGPU_OK) throw std::runtime_error(std::string("OFDM_Demod: ") + dabgpu_strerror(rc));
    m_frame.resize(params.nb_frame_symbols * params.nb_symbol_period);
    m_soft.resize(size_t(params.nb_frame_symbols - 1) * params.nb_data_carriers * 2);
    m_frame_data_vec.resize(size_t(params.nb_frame_symbols - 1) * params.nb_data_carriers);
    reset_now();
}

OFDM_Demod::~OFDM_Demod() {
#ifdef DABHOST_TIMING
    if (g_ofdm_n) std::fprintf(stderr, "OFDM_Demod per frame, us: copy into the frame buffer %.1f | frame call %.1f | On_OFDM_Frame observers %.1f (%ld frames)\n",
                               g_ofdm_t[0] / g_ofdm_n * 1e6, g_ofdm_t[1] / g_ofdm_n * 1e6, g_ofdm_t[2] / g_ofdm_n * 1e6, g_ofdm_n);
#endif
    dabgpu_destroy(m_ctx);
}

// Any thread (the GUI's "Reset" button, /root/reference/src/render_radio_block.cpp:95-97): what the getters show starts
// over at once; the buffers and the device-side state start over on the Process thread, at the top of its next call.
void OFDM_Demod::Reset() {
    m_reset_requested.store(true, std::memory_order_release);
    m_state = State::FINDING_NULL_POWER_DIP;
    m_freq_fine_offset = 0.0f;
    m_freq_coarse_offset = 0.0f;
    m_total_frames_read = 0;
    m_total_frames_desync = 0;
}

void OFDM_Demod::reset_now() {
    m_state = State::FINDING_NULL_POWER_DIP;
    m_signal_l1_average = 0.0f;
    m_in_null = false;
    m_null_blocks = 0;
    m_history.clear();
    m_frame_fill = 0;
    m_skip = 0;
    m_is_acquiring = true;
    m_last_time_offset = 0;
    m_last_peak_db = 0.0f;
    m_carry.clear();
    m_freq_fine_offset = 0.0f;
    m_freq_coarse_offset = 0.0f;
    m_total_frames_read = 0;
    m_total_frames_desync = 0;
    m_host_desyncs = 0;
    m_device_desyncs_seen = 0;
    (void)dabgpu_streams_reset(m_ctx, 1);                       // the device-side loop state starts over as well
}

void OFDM_Demod::Process(tcb::span<const std::complex<float>> block) {
    if (m_reset_requested.exchange(false, std::memory_order_acquire)) reset_now();
    const std::complex<float> *x = block.data();
    size_t n = block.size();
    // a partial L1 block carried over from the last call (chunks are arbitrary, dab_module.cpp:23-25) is completed first
    if (!m_carry.empty()) {
        const size_t need = std::min(n, L1_BLOCK - m_carry.size());
        m_carry.insert(m_carry.end(), x, x + need);
        x += need;
        n -= need;
        if (m_carry.size() < L1_BLOCK) return;
        push_sample_block(m_carry.data(), L1_BLOCK);
        m_carry.clear();
    }
    // While locked, samples go straight into the frame buffer in whole pieces (gap skipped, frame filled, frame
    // demodulated): 64 samples at a time was 3 072 small copies and calls per frame.  The power detector -- and with it the
    // 64-sample blocks and the carry -- only runs while searching for a null symbol.
    while (n > 0) {
        if (m_state.load(std::memory_order_relaxed) == State::READING_SYMBOLS) {
            const size_t took = take_locked(x, n);
            x += took;
            n -= took;
        } else if (n >= L1_BLOCK) {
            push_sample_block(x, L1_BLOCK);
            x += L1_BLOCK;
            n -= L1_BLOCK;
        } else {
            break;
        }
    }
    m_carry.assign(x, x + n);
}

// READING_SYMBOLS: consume up to n samples -- the rest of the gap before the frame, then as much of the frame as there is;
// a completed frame is demodulated and the gap to the next one set up.  Returns how many samples were consumed (> 0).
size_t OFDM_Demod::take_locked(const std::complex<float> *x, size_t n) {
    const size_t frame_len = m_frame.size();
    size_t i = 0;
    if (m_skip) {
        const size_t d = std::min(m_skip, n);
        m_skip -= d;
        i = d;
    }
    const size_t take = std::min(n - i, frame_len - m_frame_fill);
    {
        OFDM_LAP(0);
        std::memcpy(m_frame.data() + m_frame_fill, x + i, take * sizeof(*x));
    }
    m_frame_fill += take;
    if (m_frame_fill == frame_len) {
        demodulate_frame();
        // locked: the next PRS starts one null symbol (+/- the timing correction) after this frame's symbols
        m_frame_fill = 0;
        m_skip = m_next_skip;
    }
    return i + take;
}

void OFDM_Demod::push_sample_block(const std::complex<float> *x, size_t n) {
    if (m_state == State::READING_SYMBOLS) {
        for (size_t done = 0; done < n && m_state == State::READING_SYMBOLS;) done += take_locked(x + done, n - done);
        return;                                                // (a frame that lost lock drops the rest of its 64-sample block)
    }
    // ---- acquisition: null-symbol power dip on the L1 norm ----
    float l1 = 0.0f;
    for (size_t i = 0; i < n; i++) l1 += std::fabs(x[i].real()) + std::fabs(x[i].imag());
    l1 /= float(n);
    // keep TIMING_MARGIN + one block of history so the frame can start before the detection point
    m_history.insert(m_history.end(), x, x + n);
    const size_t keep = TIMING_MARGIN + 2 * L1_BLOCK;
    if (m_history.size() > keep) m_history.erase(m_history.begin(), m_history.end() - keep);

    if (!m_in_null) {
        const float avg = m_signal_l1_average;
        if (avg > 0.0f && l1 < knob(m_cfg.null_l1_search.thresh_null_start) * avg) {
            m_in_null = true;
            m_null_blocks = 1;
            m_state = State::READING_NULL_AND_PRS;
        } else {
            const float beta = knob(m_cfg.signal_l1.update_beta);
            m_signal_l1_average = (avg == 0.0f) ? l1 : beta * avg + (1.0f - beta) * l1;
        }
        return;
    }
    if (l1 > knob(m_cfg.null_l1_search.thresh_null_end) * m_signal_l1_average) {
        // the PRS began somewhere inside this block (or the previous one): start TIMING_MARGIN samples
        // before the START of this block
        m_in_null = false;
        const bool plausible = m_null_blocks * L1_BLOCK > m_params.nb_null_period / 2;
        if (!plausible) {
            m_state = State::FINDING_NULL_POWER_DIP;
            return;
        }
        const size_t back = std::min(m_history.size(), TIMING_MARGIN + n);
        std::memcpy(m_frame.data(), m_history.data() + (m_history.size() - back), back * sizeof(*x));
        m_frame_fill = back;
        m_skip = 0;
        m_is_acquiring = true;
        m_state = State::READING_SYMBOLS;
    } else {
        m_null_blocks++;
        if (m_null_blocks * L1_BLOCK > 2 * m_params.nb_null_period) {   // not a null symbol: signal vanished
            m_in_null = false;
            m_signal_l1_average = 0.0f;
            m_state = State::FINDING_NULL_POWER_DIP;
        }
    }
}

// One frame: ONE call into libdabgpu (dabgpu_ofdm_demod_stream_frame) -- the frame goes up once; synchronisation on
// the PRS (coarse frequency at acquisition, fine time always), the coarse-offset update, demodulation with the
// stream's own offsets, the fine-frequency loop, the counters and the level average all run on the device; soft bits,
// sync result and statistics come back in one download behind one synchronisation.  (Round 2 made three synchronous
// calls here: dabgpu_sync_prs, dabgpu_ofdm_demod_streams, dabgpu_get_stats.)
void OFDM_Demod::demodulate_frame() {
    m_next_skip = m_params.nb_null_period;
    dabgpu_track_cfg cfg;
    dabgpu_track_default_cfg(&cfg);
    // (every knob read once: the GUI thread may be moving a slider)
    auto unit = [](float v) { return std::min(1.0f, std::max(0.0f, v)); };
    cfg.fine_freq_update_beta = unit(knob(m_cfg.sync.fine_freq_update_beta));
    cfg.signal_update_beta = unit(knob(m_cfg.signal_l1.update_beta));
    cfg.thr_null_start = unit(knob(m_cfg.null_l1_search.thresh_null_start));
    cfg.min_peak_to_mean = std::pow(10.0f, 0.1f * knob(m_cfg.sync.impulse_peak_threshold_db));
    cfg.impulse_peak_distance_probability = unit(knob(m_cfg.sync.impulse_peak_distance_probability));
    cfg.coarse_freq_slow_beta = unit(knob(m_cfg.sync.coarse_freq_slow_beta));
    cfg.timing_margin = int(TIMING_MARGIN);
    cfg.decision_directed = knob(m_cfg.sync.is_decision_directed_fine_freq) ? 1 : 0;   // default: the reference's cyclic-prefix loop
    cfg.max_coarse_carriers = 0;
    if (knob(m_cfg.sync.is_coarse_freq_correction) && (m_is_acquiring || cfg.coarse_freq_slow_beta > 0.0f))
        cfg.max_coarse_carriers = std::max(0, std::min(1023, int(knob(m_cfg.sync.max_coarse_freq_correction_norm) * float(m_params.nb_fft))));
    m_state = m_is_acquiring ? State::RUNNING_COARSE_FREQ_SYNC : State::RUNNING_FINE_TIME_SYNC;
    dabgpu_frame_result res{};
#ifdef DABHOST_TIMING
    g_ofdm_n++;
    const auto t_call = std::chrono::steady_clock::now();
#endif
    // the constellation only while a display asks for it (GetFrameDataVec): 0.9 MB of download and a slower launch otherwise saved
    bool want_dqpsk = false;
    if (m_frame_data_wanted.load(std::memory_order_relaxed) > 0) {
        m_frame_data_wanted.fetch_sub(1, std::memory_order_relaxed);
        want_dqpsk = true;
    }
    const int rc = dabgpu_ofdm_demod_stream_frame(m_ctx, 0, reinterpret_cast<const float *>(m_frame.data()), m_is_acquiring ? 1 : 0,
                                                  &cfg, m_soft.data(),
                                                  want_dqpsk ? reinterpret_cast<float *>(m_frame_data_vec.data()) : nullptr, &res);
#ifdef DABHOST_TIMING
    g_ofdm_t[1] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call).count();
#endif
    if (rc != DABGPU_OK) {   // no exceptions on the streaming path: count it as a lost frame
        m_host_desyncs++;
        m_total_frames_desync.fetch_add(1, std::memory_order_relaxed);
        m_state = State::FINDING_NULL_POWER_DIP;
        m_in_null = false;
        return;
    }
    // a demodulated frame that still raised the device's desync count: its level fell under the null threshold
    const bool level_lost = (res.flags & 3) == 3 && res.stats.total_frames_desync > m_device_desyncs_seen;
    m_device_desyncs_seen = res.stats.total_frames_desync;
    m_last_time_offset = res.sync.time_offset;
    m_last_peak_db = 10.0f * std::log10(std::max(res.sync.peak_to_mean, 1e-9f));
    // the getters the GUI polls (src/render_radio_block.cpp:202-207): the device's values after this frame
    m_freq_fine_offset = res.stats.fine_freq_offset;
    m_freq_coarse_offset = res.stats.coarse_freq_offset;
    m_total_frames_read = res.stats.total_frames_read;
    m_total_frames_desync = res.stats.total_frames_desync + m_host_desyncs;
    if (res.stats.signal_average > 0.0f) m_signal_l1_average = res.stats.signal_average;       // live while locked
    if (!(res.flags & 1)) {                                   // not a PRS: false lock / signal lost -> search again
        m_state = State::FINDING_NULL_POWER_DIP;
        m_in_null = false;
        return;
    }
    m_is_acquiring = false;
    // keep the FFT windows TIMING_MARGIN samples inside the cyclic prefix: time_offset is how early this frame's
    // windows are; steer the start of the next frame
    const int err = res.sync.time_offset - int(TIMING_MARGIN);
    m_next_skip = size_t(std::max(0L, long(m_params.nb_null_period) + err));
    m_state = State::READING_SYMBOLS;
    if (!(res.flags & 2)) return;                             // windows outside the prefix: next frame at the corrected position
    if (level_lost) {                                         // the signal is gone
        m_state = State::FINDING_NULL_POWER_DIP;
        m_in_null = false;
        return;
    }
    OFDM_LAP(2);
    m_obs_on_ofdm_frame.Notify(tcb::span<const viterbi_bit_t>(m_soft.data(), m_soft.size()));
}
