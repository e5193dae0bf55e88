// fake_dabgpu.cpp -- TEST-ONLY stand-in for libdabgpu: the entry points of include/dabgpu.h that the host mirror calls
// (sdrplusplus-dab-radio-plugin_amd/host/), implemented on the CPU by calling the oracle.  It exists for ONE purpose:
// letting the mirror's threads (OFDM thread, radio thread, a polling GUI thread) run under ThreadSanitizer and
// AddressSanitizer / UBSan on a machine without a GPU (tests/test_host_sanitizers.py; GPU sanitizers are not available on
// the pool).  It is never shipped, never built by __graft_entry__.build(), never on the product path; the product
// library has no CPU fallback.  Behaviour follows the documentation in dabgpu.h closely enough for the mirror's state
// machine to lock, decode FIBs and open channels; it is not a parity reference for anything.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

#include "dabgpu.h"
extern "C" {
#include "dab_oracle.h"
}

namespace {
constexpr int NB_FRAME_BITS = 230400, NB_FIC_BITS = 9216, NB_CIF_BITS = 55296, NB_SYM = 2552, NB_FRAME_SYMBOLS = 76;
struct Ring {
    std::vector<int8_t> rows;      // [15][bits], oldest first
    bool live = false;
};
struct Profile {
    std::vector<uint8_t> mask;
    int nsteps = 0, cu = 0, n_kept = 0;
};
int profile_of(const dabgpu_subchannel &sc, Profile &p) {
    if (sc.is_uep) {
        for (int i = 0; i < 64; i++) {
            int br, lv, cu;
            if (oracle_uep_profile(i, &br, &lv, &cu) == 0 && br == sc.bitrate_kbps && lv == sc.protection_level) {
                if (cu != sc.length) return DABGPU_ERR_PROFILE;
                p.mask.assign(size_t(4) * (24 * br + 6), 0);
                int size_cu;
                oracle_uep_puncture_mask(i, p.mask.data(), &p.nsteps, &p.n_kept, &size_cu);
                p.cu = cu;
                return DABGPU_OK;
            }
        }
        return DABGPU_ERR_PROFILE;
    }
    if (sc.bitrate_kbps <= 0 || sc.bitrate_kbps > 1824 || sc.protection_level < 1 || sc.protection_level > 4) return DABGPU_ERR_PROFILE;
    p.mask.assign(size_t(4) * (24 * sc.bitrate_kbps + 6), 0);
    p.n_kept = oracle_eep_puncture_mask(sc.eep_type, sc.protection_level, sc.bitrate_kbps, p.mask.data(), &p.nsteps, &p.cu);
    if (p.n_kept < 0 || p.cu != sc.length) return DABGPU_ERR_PROFILE;
    return DABGPU_OK;
}
}  // namespace

struct dabgpu_ctx {
    std::vector<dabgpu_stream_state> states;
    std::map<std::pair<int, int>, Ring> rings;
    int test_fail_in = 0;          // dabgpu_test_fail_frame_call: the library's hook, same meaning
};

extern "C" {

int dabgpu_abi_version(void) { return DABGPU_ABI_VERSION; }
const char *dabgpu_strerror(int status) { return status == 0 ? "ok" : "fake libdabgpu error"; }

int dabgpu_get_ofdm_params(int mode, dabgpu_ofdm_params *out) {
    if (!out) return DABGPU_ERR_ARG;
    if (mode != 1) return DABGPU_ERR_PROFILE;
    *out = dabgpu_ofdm_params{76, 2552, 2656, 2048, 504, 1536, 1000, 196608};
    return DABGPU_OK;
}
int dabgpu_get_dab_params(int mode, dabgpu_dab_params *out) {
    if (!out) return DABGPU_ERR_ARG;
    if (mode != 1) return DABGPU_ERR_PROFILE;
    *out = dabgpu_dab_params{230400, 75, 3, 72, 3072, 9216, 221184, 12, 4, 256, 2304, 3, 55296};
    return DABGPU_OK;
}
int dabgpu_get_prs_reference(int mode, float *out, int nb_fft) {
    if (!out || nb_fft != 2048 || mode != 1) return DABGPU_ERR_ARG;
    oracle_get_prs(out);
    return DABGPU_OK;
}
int dabgpu_get_mapper_reference(int32_t *out, int n, int nb_fft) {
    if (!out || n != 1536 || nb_fft != 2048) return DABGPU_ERR_ARG;
    oracle_get_mapper(out);
    return DABGPU_OK;
}
int dabgpu_create(const dabgpu_cfg *cfg, dabgpu_ctx **out) {
    if (!cfg || !out) return DABGPU_ERR_ARG;
    *out = new dabgpu_ctx();
    return DABGPU_OK;
}
void dabgpu_destroy(dabgpu_ctx *ctx) { delete ctx; }
int dabgpu_test_fail_frame_call(dabgpu_ctx *ctx, int nth) {
    if (!ctx || nth < 0) return DABGPU_ERR_ARG;
    ctx->test_fail_in = nth;
    return DABGPU_OK;
}
void *dabgpu_host_alloc(size_t bytes) { return std::calloc(1, (bytes + 63) & ~size_t(63)); }
void dabgpu_host_free(void *p) { std::free(p); }

int dabgpu_streams_reset(dabgpu_ctx *ctx, int n) {
    if (!ctx || n < 0) return DABGPU_ERR_ARG;
    ctx->states.assign(size_t(n), dabgpu_stream_state{});
    return DABGPU_OK;
}
int dabgpu_set_stream_offsets(dabgpu_ctx *ctx, int i, const float *fine, const float *coarse) {
    if (!ctx || i < 0 || size_t(i) >= ctx->states.size()) return DABGPU_ERR_ARG;
    if (fine) ctx->states[size_t(i)].fine_freq_offset = *fine;
    if (coarse) ctx->states[size_t(i)].coarse_freq_offset = *coarse;
    return DABGPU_OK;
}
void dabgpu_track_default_cfg(dabgpu_track_cfg *c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->fine_freq_update_beta = 0.9f;
    c->signal_update_beta = 0.95f;
    c->thr_null_start = 0.35f;
    c->min_peak_to_mean = 100.0f;
    c->impulse_peak_distance_probability = 0.15f;
    c->first_path_rel = 0.25f;
    c->drift_beta = 0.5f;
    c->coarse_freq_slow_beta = 0.1f;
    c->timing_margin = 64;
    c->max_coarse_carriers = 204;
    c->dd_gate = 2.5f;
}

int dabgpu_ofdm_demod_stream_frame(dabgpu_ctx *ctx, int idx, const float *iq, int acquiring, const dabgpu_track_cfg *cfg,
                                   int8_t *soft, float *dqpsk, dabgpu_frame_result *res) {
    if (!ctx || !iq || !soft || !res || !cfg || idx < 0 || size_t(idx) >= ctx->states.size()) return DABGPU_ERR_ARG;
    if (ctx->test_fail_in > 0 && --ctx->test_fail_in == 0) return DABGPU_ERR_HIP;
    dabgpu_stream_state &st = ctx->states[size_t(idx)];
    const std::complex<float> *x = reinterpret_cast<const std::complex<float> *>(iq);
    const float TWO_PI = 6.283185307179586f;
    if (acquiring) {                                          // the fine offset from this PRS's own cyclic prefix
        std::complex<double> acc = 0;
        for (int i = 64; i < 440; i++) acc += std::conj(std::complex<double>(x[i])) * std::complex<double>(x[i + 2048]);
        st.fine_freq_offset = -float(std::arg(acc)) / (TWO_PI * 2048.0f);
    }
    int32_t k = 0, toff = 0;
    float p2m = 0, cp2m = 0;
    oracle_sync_prs_ex(iq, st.fine_freq_offset + st.coarse_freq_offset, cfg->max_coarse_carriers, cfg->timing_margin,
                       cfg->impulse_peak_distance_probability, cfg->first_path_rel, &k, &toff, &p2m, &cp2m);
    if (cfg->max_coarse_carriers > 0 && k != 0)
        st.coarse_freq_offset -= (acquiring ? 1.0f : cfg->coarse_freq_slow_beta) * float(k) / 2048.0f;
    res->sync = dabgpu_sync_result{k, toff, p2m, cp2m};
    res->flags = (p2m >= cfg->min_peak_to_mean ? 1 : 0) | ((toff >= 0 && toff <= 488) ? 2 : 0);
    res->reserved = 0;
    if (res->flags == 3) {
        std::vector<float> cyc(2 * NB_FRAME_SYMBOLS);
        oracle_ofdm_demod_frame(iq, st.fine_freq_offset + st.coarse_freq_offset, soft, nullptr, cyc.data(), dqpsk);
        double ang = 0;
        for (int l = 0; l < NB_FRAME_SYMBOLS; l++) ang += std::atan2(double(cyc[2 * l + 1]), double(cyc[2 * l]));
        const float err = float(ang / NB_FRAME_SYMBOLS) / (TWO_PI * 2048.0f);
        float l1 = 0;
        for (int i = 0; i < 4096; i++) l1 += std::fabs(x[i].real()) + std::fabs(x[i].imag());
        l1 /= 4096.0f;
        const bool level_lost = st.signal_average > 0 && l1 < cfg->thr_null_start * st.signal_average;
        if (!level_lost) {
            float f = st.fine_freq_offset - cfg->fine_freq_update_beta * err;
            const float half = 0.5f / 2048.0f;
            if (f > half) f -= 2 * half;
            if (f < -half) f += 2 * half;
            st.fine_freq_offset = f;
            st.total_frames_read++;
            st.signal_average = st.signal_average > 0 ? cfg->signal_update_beta * st.signal_average + (1 - cfg->signal_update_beta) * l1 : l1;
        } else {
            st.total_frames_desync++;
        }
        st.last_fine_error = err;
    } else {
        std::memset(soft, 0, NB_FRAME_BITS);
        st.total_frames_desync++;
    }
    st.last_time_offset = toff;
    st.last_peak_to_mean = p2m;
    dabgpu_stats &o = res->stats;
    std::memset(&o, 0, sizeof(o));
    o.state = st.total_frames_read > 0 ? 4 : 0;
    o.fine_freq_offset = st.fine_freq_offset;
    o.coarse_freq_offset = st.coarse_freq_offset;
    o.net_freq_offset = st.fine_freq_offset + st.coarse_freq_offset;
    o.signal_average = st.signal_average;
    o.total_frames_read = st.total_frames_read;
    o.total_frames_desync = st.total_frames_desync;
    o.last_fine_error = st.last_fine_error;
    o.last_time_offset = toff;
    o.last_peak_to_mean = p2m;
    return DABGPU_OK;
}

int dabgpu_fic_decode(dabgpu_ctx *ctx, const int8_t *soft, size_t stride, int n, uint8_t *fib, uint8_t *crc) {
    if (!ctx || !soft || !fib || !crc || n < 0) return DABGPU_ERR_ARG;
    for (int f = 0; f < n; f++) oracle_fic_decode(soft + size_t(f) * stride, fib + size_t(f) * 384, crc + size_t(f) * 12);
    return DABGPU_OK;
}

int dabgpu_subchannel_bytes(const dabgpu_subchannel *sc) {
    if (!sc) return DABGPU_ERR_ARG;
    Profile p;
    const int rc = profile_of(*sc, p);
    if (rc) return rc;
    if (sc->start_address < 0 || sc->start_address + sc->length > 864) return DABGPU_ERR_ARG;
    return (p.nsteps - 6) / 8;
}
int dabgpu_uep_subchannel(int index, int start, dabgpu_subchannel *out) {
    int br, lv, cu;
    if (!out) return DABGPU_ERR_ARG;
    if (oracle_uep_profile(index, &br, &lv, &cu) != 0) return DABGPU_ERR_PROFILE;
    if (start < 0 || start + cu > 864) return DABGPU_ERR_ARG;
    *out = dabgpu_subchannel{start, cu, 1, 0, lv, br};
    return DABGPU_OK;
}

int dabgpu_decode_stream_reset(dabgpu_ctx *ctx) {
    if (!ctx) return DABGPU_ERR_ARG;
    ctx->rings.clear();
    return DABGPU_OK;
}
int dabgpu_decode_stream_frames(dabgpu_ctx *ctx, const int8_t *soft, size_t stride, int n_frames, uint8_t *fib, uint8_t *crc,
                                const dabgpu_subchannel *sc, int nsc, uint8_t *const *out) {
    if (!ctx || !soft || !fib || !crc || n_frames < 0 || nsc < 0 || (nsc > 0 && (!sc || !out))) return DABGPU_ERR_ARG;
    std::vector<Profile> prof(size_t(nsc), Profile{});
    for (int i = 0; i < nsc; i++) {
        const int rc = profile_of(sc[i], prof[size_t(i)]);
        if (rc) return rc;
    }
    if (ctx->test_fail_in > 0 && --ctx->test_fail_in == 0) {     // as the library: a failed call drops every ring
        ctx->rings.clear();
        return DABGPU_ERR_HIP;
    }
    for (auto &kv : ctx->rings) kv.second.live = false;
    for (int i = 0; i < nsc; i++) {
        Ring &r = ctx->rings[{sc[i].start_address, sc[i].length}];
        const int bits = sc[i].length * 64;
        if (r.rows.empty()) r.rows.assign(size_t(15) * bits, 0);
        r.live = true;
    }
    for (auto it = ctx->rings.begin(); it != ctx->rings.end();) it = it->second.live ? std::next(it) : ctx->rings.erase(it);
    for (int f = 0; f < n_frames; f++) {
        const int8_t *fr = soft + size_t(f) * stride;
        oracle_fic_decode(fr, fib + size_t(f) * 384, crc + size_t(f) * 12);
        for (int i = 0; i < nsc; i++) {
            Ring &r = ctx->rings[{sc[i].start_address, sc[i].length}];
            const int bits = sc[i].length * 64, nbytes = (prof[size_t(i)].nsteps - 6) / 8;
            std::vector<int8_t> deint(size_t(bits), 0);
            for (int c = 0; c < 4; c++) {
                const int8_t *cur = fr + NB_FIC_BITS + size_t(c) * NB_CIF_BITS + size_t(sc[i].start_address) * 64;
                const int8_t *rows[16];
                for (int k = 0; k < 15; k++) rows[k] = r.rows.data() + size_t(k) * bits;
                rows[15] = cur;
                oracle_time_deinterleave(rows, bits, deint.data());
                oracle_msc_decode_lf(deint.data(), prof[size_t(i)].mask.data(), prof[size_t(i)].nsteps,
                                     out[i] + (size_t(f) * 4 + c) * nbytes);
                std::memmove(r.rows.data(), r.rows.data() + bits, size_t(14) * bits);
                std::memcpy(r.rows.data() + size_t(14) * bits, cur, size_t(bits));
            }
        }
    }
    return DABGPU_OK;
}

int dabgpu_dabplus_superframes(dabgpu_ctx *ctx, const uint8_t *in, size_t in_stride, int n, int bitrate, uint8_t *out,
                               dabgpu_superframe_status *status) {
    if (!ctx || !in || !out || !status || n < 0 || bitrate <= 0 || bitrate % 8) return DABGPU_ERR_ARG;
    const int s = bitrate / 8;
    std::vector<uint8_t> sf(size_t(120) * s);
    for (int f = 0; f < n; f++) {
        std::memcpy(sf.data(), in + size_t(f) * in_stride, sf.size());
        int32_t st5[5], au[8];
        oracle_dabplus_superframe(sf.data(), s, st5, au);
        std::memcpy(out + size_t(f) * 110 * s, sf.data(), size_t(110) * s);
        dabgpu_superframe_status &o = status[f];
        std::memset(&o, 0, sizeof(o));
        o.firecode_ok = st5[0]; o.rs_corrected = st5[1]; o.rs_uncorrectable = st5[2]; o.num_aus = st5[3]; o.au_crc_mask = st5[4];
        std::memcpy(o.au_start, au, sizeof(au));
    }
    return DABGPU_OK;
}

}  // extern "C"
