#include "basic_radio/basic_radio.h"

#include "dab/constants/subchannel_protection_tables.h"

#include <cstring>
#include <stdexcept>
#include <string>
#ifdef DABHOST_TIMING
#include <chrono>
#include <cstdio>
static double g_t[6];
static long g_n;
struct Tick { std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
              void lap(int i) { auto n = std::chrono::steady_clock::now(); g_t[i] += std::chrono::duration<double>(n - t).count(); t = n; } };
#define LAP(i) tick.lap(i)
#else
#define LAP(i)
#endif

BasicRadio::BasicRadio(const DAB_Parameters &params, size_t /*nb_threads*/) : m_params(params), m_ctx(nullptr) {
    dabgpu_cfg cfg{};
    cfg.device = GetDabGpuDefaultDevice();
    cfg.max_frames = 1;
    cfg.transmission_mode = 1;
    const int rc = dabgpu_create(&cfg, &m_ctx);
    if (rc != DABGPU_OK) throw std::runtime_error(std::string("BasicRadio: ") + dabgpu_strerror(rc));
    m_fib.resize(size_t(params.nb_fibs) * 32);
    m_crc.resize(size_t(params.nb_fibs));
    m_frame.resize(size_t(params.nb_frame_bits));
}

BasicRadio::~BasicRadio() {
#ifdef DABHOST_TIMING
    if (g_n) std::fprintf(stderr, "BasicRadio::Process per frame, us: set-up + copy %.1f | decode call %.1f | FIBs + FIG parser %.1f | sub-channel observers + channels %.1f | channel update %.1f (%ld frames)\n",
                          g_t[0] / g_n * 1e6, g_t[1] / g_n * 1e6, g_t[2] / g_n * 1e6, g_t[3] / g_n * 1e6, g_t[4] / g_n * 1e6, g_n);
#endif
    dabgpu_destroy(m_ctx);
}

int BasicRadio::AddSubchannel(const dabgpu_subchannel &sc) {
    std::lock_guard<std::mutex> lock(m_mutex);
    return add_subchannel_locked(sc);
}

// A sub-channel joins the per-frame call only if that call will accept it: inside the CIF and clear of every
// sub-channel already in the call (dabgpu_decode_frames_dev rejects a whole call whose sub-channels overlap -- one
// bad FIG 0/1 must not be able to silence the FIC and every other service).
int BasicRadio::add_subchannel_locked(const dabgpu_subchannel &sc) {
    const int nbytes = dabgpu_subchannel_bytes(&sc);
    if (nbytes < 0) return nbytes;
    if (sc.start_address < 0 || sc.length <= 0 || sc.start_address + sc.length > 864) return DABGPU_ERR_ARG;
    for (const Subchannel &o : m_subchannels)
        if (sc.start_address < o.desc.start_address + o.desc.length && o.desc.start_address < sc.start_address + sc.length)
            return DABGPU_ERR_ARG;
    Subchannel s;
    s.desc = sc;
    s.nbytes = nbytes;
    s.out.resize(size_t(m_params.nb_cifs) * nbytes);
    m_subchannels.push_back(std::move(s));
    return int(m_subchannels.size()) - 1;
}

void BasicRadio::Process(tcb::span<const viterbi_bit_t> buf) {
    if (buf.size() != size_t(m_params.nb_frame_bits)) return;   // the reference drops short frames the same way
    std::lock_guard<std::mutex> lock(m_mutex);
#ifdef DABHOST_TIMING
    Tick tick;
    g_n++;
#endif
    // A7..A12 in one call: the frame goes up once, the FIC and every registered sub-channel are decoded from that
    // copy, FIBs / CRC flags / logical frames / de-interleaver state come back together
    const size_t n_sub = m_subchannels.size();
    m_call_sc.resize(n_sub);
    m_call_out.resize(n_sub);
    for (size_t i = 0; i < n_sub; i++) {
        m_call_sc[i] = m_subchannels[i].desc;
        m_call_out[i] = m_subchannels[i].out.data();
    }
    std::memcpy(m_frame.data(), buf.data(), buf.size());
    LAP(0);
    // (the time de-interleaver state of every sub-channel stays on the device between frames)
    const int rc = dabgpu_decode_stream_frames(m_ctx, m_frame.data(), m_frame.size(), 1, m_fib.data(), m_crc.data(),
                                               m_call_sc.data(), int(n_sub), m_call_out.data());
    if (rc != DABGPU_OK) {
        // No exceptions on the streaming path.  The sub-channels' frame is lost (their de-interleaver rings on the device
        // were dropped by the failed call, so they start over), but the FIC must keep flowing: it is what repairs a
        // bad configuration.
        m_total_frames_lost++;
        for (Subchannel &s : m_subchannels) s.cifs_seen = 0;
        if (dabgpu_fic_decode(m_ctx, m_frame.data(), m_frame.size(), 1, m_fib.data(), m_crc.data()) != DABGPU_OK) return;
        process_fibs_locked();
        if (m_auto_channels) update_channels_from_database();
        return;
    }
    LAP(1);
    process_fibs_locked();
    LAP(2);
    for (size_t i = 0; i < n_sub; i++) {
        Subchannel &s = m_subchannels[i];
        for (int c = 0; c < m_params.nb_cifs; c++) {
            // the de-interleaver needs 16 CIFs before its first complete logical frame
            if (++s.cifs_seen < 16) continue;
            const tcb::span<const uint8_t> lf(s.out.data() + size_t(c) * s.nbytes, size_t(s.nbytes));
            m_obs_msc.Notify(int(i), lf);
            if (s.dab_plus) s.dab_plus->Process(lf);
            if (s.dab) s.dab->Process(lf);
        }
    }
    LAP(3);
    // sub-channels the FIC has announced by now join the call from the next frame on
    if (m_auto_channels) update_channels_from_database();
    LAP(4);
}

// FIB counters, the On_FIC observers and the FIG parser (mutex held)
void BasicRadio::process_fibs_locked() {
    for (uint8_t ok : m_crc) {
        m_total_fibs++;
        if (!ok) m_total_fib_errors++;
    }
    m_obs_fic.Notify(tcb::span<const uint8_t>(m_fib.data(), m_fib.size()),
                     tcb::span<const uint8_t>(m_crc.data(), m_crc.size()));
    for (size_t i = 0; i < m_crc.size(); i++)
        if (m_crc[i]) m_fic_parser.ProcessFIB(tcb::span<const uint8_t>(m_fib.data() + 32 * i, 32));
}

// Open every audio component whose sub-channel the FIC has described (called with the mutex held).
void BasicRadio::update_channels_from_database() {
    if (m_database.service_components.size() == m_seen_components && !m_pending_components) return;
    m_pending_components = false;
    for (const auto &comp : m_database.service_components) {
        if (comp.transport_mode != TransportMode::STREAM_MODE_AUDIO) continue;
        const bool is_dab_plus = comp.audio_service_type == AudioServiceType::DAB_PLUS;
        const bool is_dab = comp.audio_service_type == AudioServiceType::DAB;
        if (!is_dab_plus && !is_dab) continue;
        if (m_channels.count(comp.subchannel_id)) continue;
        bool rejected = false;
        for (auto id : m_rejected) rejected |= (id == comp.subchannel_id);
        if (rejected) continue;
        ::Subchannel *sub = nullptr;
        for (auto &s : m_database.subchannels)
            if (s.id == comp.subchannel_id) sub = &s;
        if (!sub) {                                          // FIG 0/1 for it has not arrived yet
            m_pending_components = true;
            continue;
        }
        dabgpu_subchannel sc{};
        bool ok = false;
        if (sub->is_uep) {
            // short form: bit rate, level and size come from the protection profile table
            ok = dabgpu_uep_subchannel(sub->uep_prot_index, sub->start_address, &sc) == DABGPU_OK;
            if (ok) sub->length = uint16_t(sc.length);
        } else if (sub->eep_prot_level <= 3 && sub->length > 0) {
            const bool type_b = sub->eep_type == EEP_Type::TYPE_B;
            sc.start_address = sub->start_address;
            sc.length = sub->length;
            sc.protection_level = sub->eep_prot_level + 1;       // the ABI counts levels 1..4
            sc.eep_type = type_b ? 1 : 0;
            sc.bitrate_kbps = int(CalculateEEPBitrate(*sub));
            ok = sc.bitrate_kbps > 0;
        }
        int idx = -1;
        if (ok) {
            // a sub-channel registered by hand (AddSubchannel) that the FIC now announces with the same range and
            // profile is already open: the audio channel attaches to it instead of claiming the range twice
            for (size_t k = 0; k < m_subchannels.size() && idx < 0; k++) {
                const dabgpu_subchannel &d = m_subchannels[k].desc;
                if (d.start_address == sc.start_address && d.length == sc.length && d.bitrate_kbps == sc.bitrate_kbps &&
                    d.is_uep == sc.is_uep && d.eep_type == sc.eep_type && d.protection_level == sc.protection_level && !m_subchannels[k].dab_plus &&
                    !m_subchannels[k].dab)
                    idx = int(k);
            }
            if (idx < 0) idx = add_subchannel_locked(sc);
        }
        if (idx < 0) {
            m_rejected.push_back(sub->id);
            m_total_unsupported++;
            continue;
        }
        Basic_Audio_Channel *ref = nullptr;
        if (is_dab_plus) {
            auto ch = std::make_unique<Basic_DAB_Plus_Channel>(m_ctx, *sub, sc.bitrate_kbps);
            m_subchannels[size_t(idx)].dab_plus = ch.get();
            ref = ch.get();
            m_channels.emplace(sub->id, std::move(ch));
        } else {
            auto ch = std::make_unique<Basic_DAB_Channel>(*sub, sc.bitrate_kbps);
            m_subchannels[size_t(idx)].dab = ch.get();
            ref = ch.get();
            m_channels.emplace(sub->id, std::move(ch));
        }
        m_obs_audio_channel.Notify(sub->id, *ref);
    }
    m_seen_components = m_database.service_components.size();
}
