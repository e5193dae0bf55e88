// mirror_threads.cpp -- TEST-ONLY: the plugin's thread structure around the host mirror, to be run under
// ThreadSanitizer and AddressSanitizer / UBSan on a machine without a GPU (the C ABI is tests/fake_abi/fake_dabgpu.cpp).
//   thread 1  feeds IQ chunks to OFDM_Demod::Process                     (/root/reference/src/dab_module.cpp:20-28)
//   thread 2  the radio thread inside Radio_Block: ring -> BasicRadio::Process  (src/radio_block.cpp:33-44)
//   thread 3  what the GUI does every frame (src/render_radio_block.cpp:89-236, 410-437, 754-840): polls every getter,
//             moves the sliders of GetConfig(), presses "Reset" on the demodulator and on the radio, walks the database
//             and the audio channels under GetMutex()
// With -DUSE_REFERENCE_RADIO_BLOCK the wiring is the reference's own src/radio_block.cpp, compiled unchanged from where it
// lies; without it (machines that do not have /root/reference) a local equivalent with the same three accessors.
//   usage: mirror_threads iq.cf32 chunk_samples
#include <atomic>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#ifdef USE_REFERENCE_RADIO_BLOCK
#include "radio_block.h"
#else
#include "app_helpers/app_io_buffers.h"
#include "basic_radio/basic_radio.h"
#include "dab/constants/dab_parameters.h"
#include "ofdm/dab_mapper_ref.h"
#include "ofdm/dab_ofdm_params_ref.h"
#include "ofdm/dab_prs_ref.h"
#include "ofdm/ofdm_demodulator.h"
// the same objects, threads and locks as Radio_Block, in the fewest lines that keep them
class Radio_Block {
public:
    Radio_Block(size_t, size_t) : m_dab(get_dab_parameters(1)) {
        const auto op = get_DAB_OFDM_params(1);
        std::vector<std::complex<float>> prs(op.nb_fft);
        get_DAB_PRS_reference(1, prs);
        std::vector<int> mapper(op.nb_data_carriers);
        get_DAB_mapper_ref(mapper, int(op.nb_fft));
        m_demod = std::make_shared<OFDM_Demod>(op, prs, mapper, 1);
        m_ring = std::make_shared<ThreadedRingBuffer<viterbi_bit_t>>(size_t(m_dab.nb_frame_bits) * 2);
        auto ring = m_ring;
        m_demod->On_OFDM_Frame().Attach([ring](tcb::span<const viterbi_bit_t> b) { ring->write(b); });
        reset_radio();
        m_thread = std::thread([this] {
            std::vector<viterbi_bit_t> data(size_t(m_dab.nb_frame_bits));
            while (m_ring->read(data) == data.size()) {
                std::shared_ptr<BasicRadio> radio;
                { std::lock_guard<std::mutex> l(m_mu); radio = m_radio; }
                if (radio) radio->Process(data);
            }
        });
    }
    ~Radio_Block() { m_ring->close(); m_thread.join(); }
    std::shared_ptr<OFDM_Demod> get_ofdm_demodulator() { return m_demod; }
    std::shared_ptr<BasicRadio> get_basic_radio() { std::lock_guard<std::mutex> l(m_mu); return m_radio; }
    void reset_radio() {
        auto radio = std::make_shared<BasicRadio>(m_dab, 1);
        std::lock_guard<std::mutex> l(m_mu);
        m_radio = radio;
    }
private:
    const DAB_Parameters m_dab;
    std::shared_ptr<OFDM_Demod> m_demod;
    std::shared_ptr<ThreadedRingBuffer<viterbi_bit_t>> m_ring;
    std::shared_ptr<BasicRadio> m_radio;
    std::mutex m_mu;
    std::thread m_thread;
};
#endif
#include "basic_radio/basic_dab_channel.h"
#include "basic_radio/basic_dab_plus_channel.h"

// the GUI writes the knobs through ImGui's float* / bool*: plain stores on ITS thread.  Here they are relaxed atomic
// stores, so that the sanitizer checks the mirror's side of the contract (one relaxed atomic read per use).
template <class T>
static void poke(T &knob, T v) { __atomic_store(&knob, &v, __ATOMIC_RELAXED); }

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const size_t chunk = size_t(std::atol(argv[2]));
    std::ifstream in(argv[1], std::ios::binary);
    if (!in) return 3;
    std::atomic<bool> done{false};
    std::atomic<long> polls{0}, fibs_seen{0}, channels_seen{0}, resets{0};
    long frames_before_reset = 0;
    {
        Radio_Block block(1, 1);
        auto demod = block.get_ofdm_demodulator();
        std::thread gui([&] {
            bool reset_demod = false, reset_radio = false;
            float sink = 0.0f;
            while (!done.load()) {
                polls++;
                // RenderOFDMState / RenderOFDMControls / the constellation tab
                const int read = demod->GetTotalFramesRead();
                sink += float(int(demod->GetState())) + demod->GetFineFrequencyOffset() + demod->GetCoarseFrequencyOffset() +
                        demod->GetNetFrequencyOffset() + demod->GetSignalAverage() + float(demod->GetTotalFramesDesync()) +
                        float(demod->GetFineTimeOffset()) + demod->GetImpulsePeakDb() + float(demod->GetOFDMParams().nb_fft);
                sink += float(demod->GetFrameDataVec().size());           // (display-only data: not dereferenced, see the header)
                auto &cfg = demod->GetConfig();
                const float wobble = float(polls.load() % 7) * 0.01f;
                poke(cfg.sync.fine_freq_update_beta, 0.86f + wobble);
                poke(cfg.sync.coarse_freq_slow_beta, 0.08f + wobble);
                poke(cfg.sync.impulse_peak_threshold_db, 18.0f + 10.0f * wobble);
                poke(cfg.sync.impulse_peak_distance_probability, 0.12f + wobble);
                poke(cfg.sync.max_coarse_freq_correction_norm, 0.09f + wobble * 0.1f);
                poke(cfg.sync.is_coarse_freq_correction, true);
                poke(cfg.signal_l1.update_beta, 0.93f + wobble * 0.5f);
                poke(cfg.null_l1_search.thresh_null_start, 0.33f + wobble * 0.5f);
                poke(cfg.null_l1_search.thresh_null_end, 0.72f + wobble * 0.5f);
                if (!reset_demod && read >= 4) {                           // the "Reset" button of the OFDM tab, mid-stream
                    frames_before_reset = read;
                    demod->Reset();
                    reset_demod = true;
                    resets++;
                }
                auto radio = block.get_basic_radio();
                if (radio) {
                    auto lock = std::scoped_lock(radio->GetMutex());       // render_radio_block.cpp:124
                    auto &db = radio->GetDatabase();
                    sink += float(db.services.size() + db.service_components.size() + db.subchannels.size());
                    sink += float(radio->GetDatabaseStatistics().nb_total) + float(radio->GetMiscInfo().cif_counter.GetTotalCount());
                    fibs_seen = radio->GetTotalFIBs() - radio->GetTotalFIBErrors();
                    long ch = 0;
                    for (auto &sub : db.subchannels) {
                        auto *channel = radio->Get_Audio_Channel(sub.id);
                        if (!channel) continue;
                        ch++;
                        auto &controls = channel->GetControls();
                        controls.SetIsPlayAudio(!controls.GetIsPlayAudio());
                        controls.SetIsDecodeData(controls.GetIsDecodeData());
                        if (auto *plus = dynamic_cast<Basic_DAB_Plus_Channel *>(channel)) {
                            const auto &h = plus->GetSuperFrameHeader();
                            sink += float(h.sampling_rate) + float(plus->IsFirecodeError()) + float(plus->IsRSError()) +
                                    float(plus->IsAUError()) + float(plus->IsCodecError()) + float(plus->GetDynamicLabel().size());
                        } else if (auto *mp2 = dynamic_cast<Basic_DAB_Channel *>(channel)) {
                            sink += float(mp2->GetAudioParams().has_value()) + float(mp2->GetIsError());
                        }
                        sink += float(channel->GetSlideshowManager().GetSlideshows().size());
                    }
                    if (ch > channels_seen.load()) channels_seen = ch;
                }
                if (!reset_radio && channels_seen.load() > 0 && read >= 2) {   // the "Reset" button of the DAB tab
                    block.reset_radio();
                    reset_radio = true;
                    resets++;
                }
                std::this_thread::yield();
            }
            if (sink == 12345.678f) std::puts("");
        });
        std::vector<std::complex<float>> buf(chunk);
        while (in) {
            in.read(reinterpret_cast<char *>(buf.data()), std::streamsize(chunk * sizeof(buf[0])));
            const size_t got = size_t(in.gcount()) / sizeof(buf[0]);
            if (!got) break;
            demod->Process(tcb::span<std::complex<float>>(buf.data(), got));
        }
        // let the radio thread drain the ring before the GUI thread stops looking
        for (int i = 0; i < 200 && fibs_seen.load() == 0; i++) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
        done = true;
        gui.join();
        std::printf("frames_read_after_reset=%d frames_before_reset=%ld desync=%d state=%d\n", demod->GetTotalFramesRead(),
                    frames_before_reset, demod->GetTotalFramesDesync(), int(demod->GetState()));
    }
    std::printf("polls=%ld good_fibs_last_radio=%ld channels_seen=%ld resets=%ld\nok\n", polls.load(), fibs_seen.load(),
                channels_seen.load(), resets.load());
    return 0;
}
