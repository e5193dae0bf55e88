#include "basic_radio/basic_radio.h"

#include <stdexcept>
#include <string>

BasicRadio::BasicRadio(const DAB_Parameters &params, size_t /*nb_threads*/) : m_params(params), m_ctx(nullptr) {
    dabgpu_cfg cfg{0, 1, 1, 0};
    const int rc = dabgpu_create(&cfg, &m_ctx);
    if (rc != DABGPU_OK) throw std::runtime_error(std::string("BasicRadio: ") + dabgpu_strerror(rc));
    m_fib.resize(size_t(params.nb_fibs) * 32);
    m_crc.resize(size_t(params.nb_fibs));
}

BasicRadio::~BasicRadio() { dabgpu_destroy(m_ctx); }

int BasicRadio::AddSubchannel(const dabgpu_subchannel &sc) {
    const int nbytes = dabgpu_subchannel_bytes(&sc);
    if (nbytes < 0) return nbytes;
    std::lock_guard<std::mutex> lock(m_mutex);
    Subchannel s;
    s.desc = sc;
    s.nbytes = nbytes;
    for (auto &h : s.history) h.assign(size_t(15) * sc.length * 64, 0);
    s.out.resize(size_t(m_params.nb_cifs) * nbytes);
    m_subchannels.push_back(std::move(s));
    return int(m_subchannels.size()) - 1;
}

void BasicRadio::Process(tcb::span<const viterbi_bit_t> buf) {
    if (buf.size() != size_t(m_params.nb_frame_bits)) return;   // the reference drops short frames the same way
    std::lock_guard<std::mutex> lock(m_mutex);
    // A7: first nb_fic_bits are the FIC, the rest is 4 CIFs
    if (dabgpu_fic_decode(m_ctx, buf.data(), buf.size(), 1, m_fib.data(), m_crc.data()) == DABGPU_OK) {
        for (uint8_t ok : m_crc) {
            m_total_fibs++;
            if (!ok) m_total_fib_errors++;
        }
        m_obs_fic.Notify(tcb::span<const uint8_t>(m_fib.data(), m_fib.size()),
                         tcb::span<const uint8_t>(m_crc.data(), m_crc.size()));
    }
    for (size_t i = 0; i < m_subchannels.size(); i++) {
        Subchannel &s = m_subchannels[i];
        const int rc = dabgpu_msc_decode(m_ctx, &s.desc, buf.data(), buf.size(), 1, 1, s.history[s.cur].data(),
                                         s.history[s.cur ^ 1].data(), s.out.data());
        if (rc != DABGPU_OK) continue;
        s.cur ^= 1;
        for (int c = 0; c < m_params.nb_cifs; c++) {
            // the de-interleaver needs 16 CIFs before its first complete logical frame
            if (++s.cifs_seen < 16) continue;
            m_obs_msc.Notify(int(i), tcb::span<const uint8_t>(s.out.data() + size_t(c) * s.nbytes, size_t(s.nbytes)));
        }
    }
}
