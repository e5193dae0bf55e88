// dabgpu_api.hip -- the C ABI of libdabgpu (include/dabgpu.h): context, tables, the OFDM front end, synchronisation,
// acquisition and tracking entry points.  (Channel decoder: dabgpu_decode_api.hip; frame buffers: dabgpu_placement.hip;
// host-fed ring: dabgpu_pipeline.hip.)  No CPU fallback: without a gfx950 device dabgpu_create fails with
// DABGPU_ERR_NODEVICE and every compute entry point needs a context.
#include "dabgpu_ctx.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <utility>

using namespace dab;
using namespace dabapi;

namespace {

// allocations of dabgpu_host_alloc: coherent page-locked memory (known_coherent_host); process-wide, any thread
std::mutex g_host_mutex;
std::vector<std::pair<const char *, size_t>> g_host_ranges;

template <class T>
int upload(T **dst, const std::vector<T> &src) {
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(dst), src.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return DABGPU_OK;
}

int build_device_code(DeviceCode &dc) {
    std::vector<uint16_t> pos;
    pos.reserve(dc.prof.n_punct);
    for (size_t i = 0; i < dc.prof.mask.size(); i++)
        if (dc.prof.mask[i]) pos.push_back(uint16_t(i));
    if (int(pos.size()) != dc.prof.n_punct) return DABGPU_ERR_PROFILE;
    int rc = DABGPU_OK;
    if (dabk::viterbi_fits(dc.prof.nsteps)) {                  // (only the wave-per-codeword kernels read this table)
        // ... and behind it, for codewords of 96 k + 6 steps, where each 96-step chunk's punctured bits begin (CodeTables)
        const int n_punct = dc.prof.n_punct, nsteps = dc.prof.nsteps;
        if (nsteps >= 102 && (nsteps - 6) % 96 == 0) {
            const int k = (nsteps - 6) / 96;
            pos.resize(size_t(dabk::code_chunk_table_offset(n_punct)), 0);
            for (int c = 0; c <= k; c++) {
                int below = 0;
                while (below < n_punct && int(pos[size_t(below)]) < 4 * 96 * c) below++;
                pos.push_back(uint16_t(below));
            }
            pos.push_back(uint16_t(n_punct));
        }
        if ((rc = upload(&dc.d_mother_pos, pos))) return rc;
    }
    std::vector<int32_t> pidx(dc.prof.mask.size(), -1);
    for (size_t i = 0, j = 0; i < dc.prof.mask.size(); i++)
        if (dc.prof.mask[i]) pidx[i] = int32_t(j++);
    if ((rc = upload(&dc.d_punct_idx, pidx))) return rc;
    {
        std::vector<int32_t> desc, tiles;
        dabk::build_lane_fused_tables(dc.prof.mask.data(), dc.prof.nsteps, desc, tiles);
        if ((rc = upload(&dc.d_fused_desc, desc))) return rc;
        if ((rc = upload(&dc.d_fused_tiles, tiles))) return rc;
    }
    return upload(&dc.d_prbs, dab::make_prbs_bytes((dc.prof.nsteps - 6 + 7) / 8));
}
}  // namespace

namespace dabapi {

void free_device_code(DeviceCode &dc) {
    if (dc.d_mother_pos) (void)hipFree(dc.d_mother_pos);
    if (dc.d_prbs) (void)hipFree(dc.d_prbs);
    if (dc.d_punct_idx) (void)hipFree(dc.d_punct_idx);
    if (dc.d_fused_desc) (void)hipFree(dc.d_fused_desc);
    if (dc.d_fused_tiles) (void)hipFree(dc.d_fused_tiles);
    dc.d_fused_desc = dc.d_fused_tiles = nullptr;
    dc.d_punct_idx = nullptr;
    dc.d_mother_pos = nullptr;
    dc.d_prbs = nullptr;
}

int get_code(dabgpu_ctx *ctx, dab::PunctureProfile &&prof, DeviceCode **out) {
    auto it = ctx->codes.find(prof.mask);
    if (it == ctx->codes.end()) {
        auto dc = std::make_unique<DeviceCode>();
        dc->prof = std::move(prof);
        int rc = build_device_code(*dc);
        if (rc) { free_device_code(*dc); return rc; }
        it = ctx->codes.emplace(dc->prof.mask, std::move(dc)).first;
    }
    *out = it->second.get();
    return DABGPU_OK;
}

int stage(dabgpu_ctx *ctx, int slot, size_t bytes, void **out) {
    if (ctx->stage_bytes[slot] < bytes) {
        if (ctx->d_stage[slot]) (void)hipFree(ctx->d_stage[slot]);
        ctx->d_stage[slot] = nullptr;
        ctx->stage_bytes[slot] = 0;
        if (hipMalloc(&ctx->d_stage[slot], bytes) != hipSuccess) return DABGPU_ERR_NOMEM;
        ctx->stage_bytes[slot] = bytes;
    }
    *out = ctx->d_stage[slot];
    return DABGPU_OK;
}

}  // namespace dabapi

namespace {

// How a launch's frames are cut into runs of consecutive symbols (one run = one wavefront; 12 resident per CU).  A cut
// costs one more transform and one more symbol read (the run's differential reference), so cuts are made only where they
// buy balance: as many whole frames as fill the resident wave slots an integer number of times go first, uncut; the
// frames behind them -- which alone would leave most slots idle for the length of a frame -- are cut into `parts`.
// Cost model, in symbol transforms per wave slot: rounds x (symbols per item + 1).
struct RunPlan {
    int uncut_frames, parts;
};
RunPlan plan_runs(const dabgpu_ctx *ctx, int n_frames, int total_syms) {
    if (ctx->ofdm_parts_override > 0 && ctx->ofdm_parts_override <= total_syms) return RunPlan{0, ctx->ofdm_parts_override};
    const long slots = long(ctx->wave_slots);
    auto uniform = [&](long frames, int *best_p) {
        long best_cost = -1;
        *best_p = 1;
        for (int p = 1; p <= total_syms && frames > 0; p++) {
            const long rounds = (frames * p + slots - 1) / slots;
            const long cost = rounds * ((total_syms + p - 1) / p + 1);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; *best_p = p; }
        }
        return best_cost < 0 ? 0 : best_cost;
    };
    int p_all = 1, p_tail = 1;
    const long cost_all = uniform(n_frames, &p_all);
    const long whole = long(n_frames) / slots * slots;
    const long cost_mixed = whole / slots * (total_syms + 1) + uniform(long(n_frames) - whole, &p_tail);
    if (whole > 0 && cost_mixed < cost_all) return RunPlan{int(whole), p_tail};
    return RunPlan{0, p_all};
}

}  // namespace

extern "C" {

int dabgpu_abi_version(void) { return DABGPU_ABI_VERSION; }

const char *dabgpu_strerror(int status) {
    switch (status) {
    case DABGPU_OK: return "ok";
    case DABGPU_ERR_ARG: return "invalid argument";
    case DABGPU_ERR_HIP: return "HIP runtime error";
    case DABGPU_ERR_NOMEM: return "out of memory";
    case DABGPU_ERR_NODEVICE: return "no gfx950 device available (libdabgpu has no CPU fallback)";
    case DABGPU_ERR_PROFILE: return "unsupported transmission mode or protection profile";
    case DABGPU_ERR_CAPACITY: return "request exceeds context capacity";
    default: return "unknown status";
    }
}

int dabgpu_get_ofdm_params(int mode, dabgpu_ofdm_params *out) {
    if (!out) return DABGPU_ERR_ARG;
    if (mode != 1) return DABGPU_ERR_PROFILE;
    out->nb_frame_symbols = NB_FRAME_SYMBOLS;
    out->nb_symbol_period = NB_SYM_PERIOD;
    out->nb_null_period = NB_NULL_PERIOD;
    out->nb_fft = NB_FFT;
    out->nb_cyclic_prefix = NB_CP;
    out->nb_data_carriers = NB_CARRIERS;
    out->freq_carrier_spacing = 1000;
    out->nb_frame_samples = NB_FRAME_SAMPLES;
    return DABGPU_OK;
}

int dabgpu_get_dab_params(int mode, dabgpu_dab_params *out) {
    if (!out) return DABGPU_ERR_ARG;
    if (mode != 1) return DABGPU_ERR_PROFILE;
    out->nb_frame_bits = NB_FRAME_BITS;
    out->nb_symbols = NB_DATA_SYMBOLS;
    out->nb_fic_symbols = NB_FIC_SYMBOLS;
    out->nb_msc_symbols = NB_DATA_SYMBOLS - NB_FIC_SYMBOLS;
    out->nb_sym_bits = NB_SYM_BITS;
    out->nb_fic_bits = NB_FIC_BITS;
    out->nb_msc_bits = NB_FRAME_BITS - NB_FIC_BITS;
    out->nb_fibs = NB_FIBS;
    out->nb_cifs = NB_CIFS;
    out->nb_fib_bits = 256;
    out->nb_fib_cif_bits = NB_FIC_GROUP_BITS;
    out->nb_fibs_per_cif = NB_FIBS / NB_CIFS;
    out->nb_cif_bits = NB_CIF_BITS;
    return DABGPU_OK;
}

int dabgpu_get_prs_reference(int mode, float *out, int nb_fft) {
    if (!out || nb_fft != NB_FFT) return DABGPU_ERR_ARG;
    if (mode != 1) return DABGPU_ERR_PROFILE;
    static const float RE[4] = {1.f, 0.f, -1.f, 0.f}, IM[4] = {0.f, 1.f, 0.f, -1.f};
    const std::vector<int8_t> q = make_prs_quarter_turns();
    for (int b = 0; b < NB_FFT; b++) {
        out[2 * b] = q[b] < 0 ? 0.f : RE[q[b]];
        out[2 * b + 1] = q[b] < 0 ? 0.f : IM[q[b]];
    }
    return DABGPU_OK;
}

int dabgpu_get_mapper_reference(int32_t *out, int nb_data_carriers, int nb_fft) {
    if (!out || nb_data_carriers != NB_CARRIERS || nb_fft != NB_FFT) return DABGPU_ERR_ARG;
    const std::vector<int32_t> m = make_mapper();
    if (int(m.size()) != NB_CARRIERS) return DABGPU_ERR_PROFILE;
    std::memcpy(out, m.data(), sizeof(int32_t) * NB_CARRIERS);
    return DABGPU_OK;
}

int dabgpu_create(const dabgpu_cfg *cfg, dabgpu_ctx **out) {
    if (!cfg || !out) return DABGPU_ERR_ARG;
    *out = nullptr;
    if (cfg->transmission_mode != 1) return DABGPU_ERR_PROFILE;
    constexpr int KNOWN_FLAGS = DABGPU_FLAG_VITERBI_WAVE | DABGPU_FLAG_VITERBI_LANE | DABGPU_FLAG_LANE_UNFUSED | DABGPU_FLAG_TEST_ONE_DOMAIN;
    if ((cfg->flags & ~KNOWN_FLAGS) || ((cfg->flags & DABGPU_FLAG_VITERBI_WAVE) && (cfg->flags & DABGPU_FLAG_VITERBI_LANE)))
        return DABGPU_ERR_ARG;
    if (cfg->ofdm_symbol_runs < 0 || cfg->ofdm_symbol_runs > NB_DATA_SYMBOLS) return DABGPU_ERR_ARG;
    if (cfg->reserved[0] || cfg->reserved[1] || cfg->reserved[2]) return DABGPU_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DABGPU_ERR_NODEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return DABGPU_ERR_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return DABGPU_ERR_HIP;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return DABGPU_ERR_NODEVICE;
    dabgpu_ctx *ctx = new (std::nothrow) dabgpu_ctx();
    if (!ctx) return DABGPU_ERR_NOMEM;
    ctx->device = cfg->device;
    DeviceGuard guard(ctx);
    ctx->max_frames = cfg->max_frames;
    ctx->ofdm_parts_override = cfg->ofdm_symbol_runs;
    ctx->lane_mode = (cfg->flags & DABGPU_FLAG_VITERBI_LANE) ? 1 : (cfg->flags & DABGPU_FLAG_VITERBI_WAVE) ? 0 : -1;
    ctx->lane_unfused = (cfg->flags & DABGPU_FLAG_LANE_UNFUSED) != 0;
    ctx->test_one_domain = (cfg->flags & DABGPU_FLAG_TEST_ONE_DOMAIN) != 0;
    ctx->wave_slots = prop.multiProcessorCount > 0 ? prop.multiProcessorCount * 12 : 3072;   // 3 workgroups x 4 waves per CU
    int rc = DABGPU_OK;
    do {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { rc = DABGPU_ERR_HIP; break; }
        if (dabk::init_viterbi_kernel_attributes() != hipSuccess || dabk::init_lane_kernel_attributes() != hipSuccess) {
            rc = DABGPU_ERR_HIP;
            break;
        }
        std::vector<float2> tw(NB_FFT + 64 + 512);
        for (int m = 0; m < NB_FFT; m++) {
            const double a = -2.0 * M_PI * double(m) / double(NB_FFT);
            tw[m] = make_float2(float(std::cos(a)), float(std::sin(a)));
        }
        // ... and the same values again in the order the synchronisation's block FFT reads them (fft_common.hpp TWC8_OFF / TWC64_OFF)
        for (int r = 0; r < 8; r++) {
            for (int k = 0; k < 8; k++) tw[NB_FFT + r * 8 + k] = tw[32 * r * k];
            for (int k = 0; k < 64; k++) tw[NB_FFT + 64 + r * 64 + k] = tw[4 * r * k];
        }
        if ((rc = upload(&ctx->d_twiddle, tw))) break;
        const std::vector<int32_t> mapper = make_mapper();
        std::vector<uint16_t> bins(NB_CARRIERS);
        for (int n = 0; n < NB_CARRIERS; n++) bins[n] = uint16_t(carrier_bin(mapper[n]));
        if ((rc = upload(&ctx->d_bin_of_n, bins))) break;
        // wave kernel: lane v ends each symbol holding bins v + 64*m; carrier register j <-> m = j (j<12) or j+8;
        // lane 0 register 0 holds bin 768 instead of DC
        std::vector<int> n_of_bin(NB_FFT, -1);
        for (int n = 0; n < NB_CARRIERS; n++) n_of_bin[bins[n]] = n;
        // layout [12][64] dwords: dword (jj, v) = n(2jj, v) | n(2jj+1, v) << 16
        std::vector<uint16_t> nvj(24 * 64);
        bool ok = true;
        for (int j = 0; j < 24; j++)
            for (int v = 0; v < 64; v++) {
                int bin = v + 64 * (j < 12 ? j : j + 8);
                if (j == 0 && v == 0) bin = 768;
                if (n_of_bin[bin] < 0) ok = false;
                nvj[((j >> 1) * 64 + v) * 2 + (j & 1)] = uint16_t(n_of_bin[bin] < 0 ? 0 : n_of_bin[bin]);
            }
        if (!ok) { rc = DABGPU_ERR_PROFILE; break; }
        if ((rc = upload(&ctx->d_n_of_vj, nvj))) break;
        {   // synchronisation tables: PRS quarter turns and the adjacent-carrier pair list
            const std::vector<int8_t> qt = make_prs_quarter_turns();
            std::vector<uint16_t> pairs;
            for (int b = 0; b + 1 < NB_FFT; b++)
                if (qt[b] >= 0 && qt[b + 1] >= 0) pairs.push_back(uint16_t(b | (((qt[b + 1] - qt[b]) & 3) << 11)));
            ctx->n_sync_pairs = int(pairs.size());
            if ((rc = upload(&ctx->d_prs_qt, qt))) break;
            if ((rc = upload(&ctx->d_sync_pairs, pairs))) break;
            // spectrum of S[b] = j^s on the pair bins: a 2048-point radix-2 FFT in double on the host (once per context)
            std::vector<double> fr(NB_FFT, 0.0), fi(NB_FFT, 0.0);
            {
                static const double SR[4] = {1, 0, -1, 0}, SI[4] = {0, 1, 0, -1};
                for (uint16_t pr : pairs) {
                    int b = pr & 2047, rev = 0;
                    for (int bit = 0; bit < 11; bit++) rev |= ((b >> bit) & 1) << (10 - bit);      // bit-reversed input order
                    fr[rev] = SR[pr >> 11];
                    fi[rev] = SI[pr >> 11];
                }
                for (int len = 2; len <= NB_FFT; len <<= 1) {
                    const double ang = -2.0 * M_PI / double(len);
                    for (int i = 0; i < NB_FFT; i += len)
                        for (int j = 0; j < len / 2; j++) {
                            const double wr = std::cos(ang * j), wi = std::sin(ang * j);
                            const int p0 = i + j, p1 = i + j + len / 2;
                            const double tr = fr[p1] * wr - fi[p1] * wi, ti = fr[p1] * wi + fi[p1] * wr;
                            fr[p1] = fr[p0] - tr; fi[p1] = fi[p0] - ti;
                            fr[p0] += tr; fi[p0] += ti;
                        }
                }
            }
            std::vector<float2> fs(NB_FFT);
            for (int m = 0; m < NB_FFT; m++) fs[m] = make_float2(float(fr[m]), float(fi[m]));
            if ((rc = upload(&ctx->d_sync_fs, fs))) break;
        }
        ctx->fic.prof = make_fic_profile();
        if (ctx->fic.prof.nsteps != NB_FIC_STEPS || ctx->fic.prof.n_punct != NB_FIC_GROUP_BITS) { rc = DABGPU_ERR_PROFILE; break; }
        if ((rc = build_device_code(ctx->fic))) break;
    } while (0);
    if (rc) { dabgpu_destroy(ctx); return rc; }
    *out = ctx;
    return DABGPU_OK;
}

void dabgpu_destroy(dabgpu_ctx *ctx) {
    if (!ctx) return;
    {
    DeviceGuard guard(ctx);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_states) (void)hipFree(ctx->d_states);
    for (auto &h : ctx->sub_history) { (void)hipFree(h.ring[0]); (void)hipFree(h.ring[1]); }
    if (ctx->h_bounce) (void)hipHostFree(ctx->h_bounce);
    if (ctx->d_twiddle) (void)hipFree(ctx->d_twiddle);
    if (ctx->d_bin_of_n) (void)hipFree(ctx->d_bin_of_n);
    if (ctx->d_n_of_vj) (void)hipFree(ctx->d_n_of_vj);
    if (ctx->d_prs_qt) (void)hipFree(ctx->d_prs_qt);
    if (ctx->d_sync_pairs) (void)hipFree(ctx->d_sync_pairs);
    if (ctx->d_sync_fs) (void)hipFree(ctx->d_sync_fs);
    free_device_code(ctx->fic);
    for (auto &kv : ctx->codes) free_device_code(*kv.second);
    for (void *p : ctx->d_stage) if (p) (void)hipFree(p);
    if (ctx->d_lane_scratch) (void)hipFree(ctx->d_lane_scratch);
    if (ctx->d_acq_scratch) (void)hipFree(ctx->d_acq_scratch);
    for (void *p : ctx->keep_tables) (void)hipFree(p);
    for (Timer &t : ctx->timers)
        for (int i = 0; i < TIMER_RING; i++) {
            if (t.start[i]) (void)hipEventDestroy(t.start[i]);
            if (t.stop[i]) (void)hipEventDestroy(t.stop[i]);
            for (hipEvent_t e : t.mid[i])
                if (e) (void)hipEventDestroy(e);
        }
    pipeline_destroy(ctx);
    arena_destroy(ctx);
    if (ctx->ev_states) (void)hipEventDestroy(ctx->ev_states);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

void *dabgpu_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (bytes == 0) return nullptr;
    // coherent page-locked memory, asked for by name (the runtime's default can be switched by its environment)
    if (hipHostMalloc(&p, bytes, hipHostMallocCoherent) == hipSuccess) {
        std::lock_guard<std::mutex> lock(g_host_mutex);
        g_host_ranges.emplace_back(static_cast<const char *>(p), bytes);
        return p;
    }
    (void)hipGetLastError();
    // (not registered: the one-frame calls then end in a stream synchronisation instead of the watched word)
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void dabgpu_host_free(void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lock(g_host_mutex);
        for (size_t i = 0; i < g_host_ranges.size(); i++)
            if (g_host_ranges[i].first == p) { g_host_ranges[i] = g_host_ranges.back(); g_host_ranges.pop_back(); break; }
    }
    (void)hipHostFree(p);
}

int dabgpu_test_fail_frame_call(dabgpu_ctx *ctx, int nth) {
    if (!ctx || nth < 0) return DABGPU_ERR_ARG;
    ctx->test_fail_in = nth;
    return DABGPU_OK;
}

int dabgpu_sync(dabgpu_ctx *ctx) {
    if (!ctx) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DABGPU_OK;
}

void *dabgpu_stream(dabgpu_ctx *ctx) { return ctx ? reinterpret_cast<void *>(ctx->stream) : nullptr; }

int dabgpu_set_timing(dabgpu_ctx *ctx, int enable) {
    if (!ctx) return DABGPU_ERR_ARG;
    ctx->timing = enable != 0;
    for (Timer &t : ctx->timers) t.recorded = 0;               // a new measurement starts
    return DABGPU_OK;
}

int dabgpu_last_kernel_ms(dabgpu_ctx *ctx, int which, float *ms) {
    if (!ctx || !ms || which < 0 || which > 3) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    Timer &t = ctx->timers[which];
    if (t.recorded == 0) return DABGPU_ERR_ARG;
    const int i = int((t.recorded - 1) % TIMER_RING);
    HIP_TRY(hipEventSynchronize(t.stop[i]));
    HIP_TRY(hipEventElapsedTime(ms, t.start[i], t.stop[i]));
    return DABGPU_OK;
}

int dabgpu_mean_kernel_ms(dabgpu_ctx *ctx, int which, float *mean_ms, int *launches) {
    if (!ctx || !mean_ms || which < 0 || which > 6) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    // 4 / 5 / 6: the parts of slot 2's grouped lane decode (forward pass | traceback | history copy), where it recorded them
    const int part = which >= 4 ? which - 4 : -1;
    Timer &t = ctx->timers[part >= 0 ? 2 : which];
    const int n = int(std::min<long>(t.recorded, TIMER_RING));
    double sum = 0.0;
    int used = 0;
    for (int k = 0; k < n; k++) {
        const int i = int((t.recorded - 1 - k) % TIMER_RING);
        if (part >= 0 && !t.has_mid[i]) continue;
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(t.stop[i]));
        hipEvent_t a = part <= 0 ? t.start[i] : t.mid[i][part - 1], b = part < 0 || part == 2 ? t.stop[i] : t.mid[i][part];
        HIP_TRY(hipEventElapsedTime(&ms, a, b));
        sum += double(ms);
        used++;
    }
    if (used == 0) return DABGPU_ERR_ARG;
    *mean_ms = float(sum / used);
    if (launches) *launches = used;
    return DABGPU_OK;
}

// ---------------------------------------------------------------------------- OFDM
static int check_iq(const void *d_iq, size_t frame_stride, int n_frames) {
    if (!d_iq || n_frames < 0) return DABGPU_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d_iq) & 15u) || (frame_stride & 1u)) return DABGPU_ERR_ARG;
    if (n_frames > 1 && frame_stride < size_t(NB_FRAME_SYMBOLS) * NB_SYM_PERIOD) return DABGPU_ERR_ARG;
    return DABGPU_OK;
}

int dabgpu_ofdm_demod_frames_dev(dabgpu_ctx *ctx, const void *d_iq, size_t frame_stride, int n_frames,
                                 const float *d_freq_offset, int8_t *d_soft, void *d_cyc, void *d_dqpsk,
                                 void *stream) {
    if (!ctx || !d_soft) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    int rc = check_iq(d_iq, frame_stride, n_frames);
    if (rc) return rc;
    if (reinterpret_cast<uintptr_t>(d_soft) & 15u) return DABGPU_ERR_ARG;
    if (n_frames == 0) return DABGPU_OK;
    hipStream_t s = pick_stream(ctx, stream);
    dabk::OfdmTables tab{ctx->d_twiddle, ctx->d_bin_of_n, ctx->d_n_of_vj};
    dabk::OfdmArgs a{};
    a.iq = static_cast<const float2 *>(d_iq);
    a.frame_stride = frame_stride;
    a.freq_offset = d_freq_offset;
    a.n_frames = n_frames;
    a.soft = d_soft;
    a.cyc = static_cast<float2 *>(d_cyc);
    a.dqpsk = static_cast<float2 *>(d_dqpsk);
    a.keep = ctx->d_keep;
    ScopedTimer tm(ctx, 0, s);
    const RunPlan plan = plan_runs(ctx, n_frames, NB_DATA_SYMBOLS);
    a.uncut_frames = plan.uncut_frames;
    HIP_TRY(dabk::launch_ofdm_demod(tab, a, plan.parts, s));
    return DABGPU_OK;
}

int dabgpu_ofdm_demod_frames_dd_dev(dabgpu_ctx *ctx, const void *d_iq, size_t frame_stride, int n_frames,
                                    const float *d_freq_offset, int8_t *d_soft, void *d_dd4, void *stream) {
    if (!ctx || !d_soft || !d_dd4) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    int rc = check_iq(d_iq, frame_stride, n_frames);
    if (rc) return rc;
    if (reinterpret_cast<uintptr_t>(d_soft) & 15u) return DABGPU_ERR_ARG;
    if (n_frames == 0) return DABGPU_OK;
    hipStream_t s = pick_stream(ctx, stream);
    dabk::OfdmTables tab{ctx->d_twiddle, ctx->d_bin_of_n, ctx->d_n_of_vj};
    dabk::OfdmArgs a{};
    a.iq = static_cast<const float2 *>(d_iq);
    a.frame_stride = frame_stride;
    a.freq_offset = d_freq_offset;
    a.n_frames = n_frames;
    a.soft = d_soft;
    a.dd4 = static_cast<float2 *>(d_dd4);
    a.keep = ctx->d_keep;
    ScopedTimer tm(ctx, 0, s);
    const RunPlan plan = plan_runs(ctx, n_frames, NB_DATA_SYMBOLS);
    a.uncut_frames = plan.uncut_frames;
    HIP_TRY(dabk::launch_ofdm_demod(tab, a, plan.parts, s));
    return DABGPU_OK;
}

int dabgpu_mover_frames_dev(dabgpu_ctx *ctx, const void *d_iq, size_t frame_stride, int n_frames, int8_t *d_soft,
                            int with_prefixes, void *stream) {
    if (!ctx || !d_soft) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    int rc = check_iq(d_iq, frame_stride, n_frames);
    if (rc) return rc;
    if (reinterpret_cast<uintptr_t>(d_soft) & 15u) return DABGPU_ERR_ARG;
    if (n_frames == 0) return DABGPU_OK;
    const RunPlan plan = plan_runs(ctx, n_frames, NB_DATA_SYMBOLS);
    HIP_TRY(dabk::launch_geometry_mover(static_cast<const float2 *>(d_iq), frame_stride, n_frames, d_soft, plan.uncut_frames,
                                        plan.parts, with_prefixes != 0, pick_stream(ctx, stream)));
    return DABGPU_OK;
}

int dabgpu_ofdm_set_soft_selection(dabgpu_ctx *ctx, const dabgpu_bit_range *ranges, int n_ranges) {
    if (!ctx || n_ranges < 0 || (n_ranges > 0 && !ranges)) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (n_ranges == 0) { ctx->d_keep = nullptr; ctx->keep_ranges.clear(); ctx->keep_symbols = NB_DATA_SYMBOLS; return DABGPU_OK; }
    constexpr int CHUNKS_PER_SYMBOL = NB_SYM_BITS / 16;          // 192 = 3 words
    std::vector<unsigned long long> words(size_t(NB_DATA_SYMBOLS) * 3, 0ull);
    for (int r = 0; r < n_ranges; r++) {
        const int first = ranges[r].first, count = ranges[r].count;
        if (first < 0 || count < 0 || (first & 15) || (count & 15) || first > NB_FRAME_BITS - count) return DABGPU_ERR_ARG;
        for (int c = first / 16; c < (first + count) / 16; c++) {
            const int sym = c / CHUNKS_PER_SYMBOL, k = c % CHUNKS_PER_SYMBOL;
            words[size_t(sym) * 3 + (k >> 6)] |= 1ull << (k & 63);
        }
    }
    // kernels already launched keep reading the table they were given: a new selection gets a new table
    if (ctx->keep_tables.size() >= 256) {
        HIP_TRY(hipDeviceSynchronize());
        for (void *p : ctx->keep_tables) (void)hipFree(p);
        ctx->keep_tables.clear();
        ctx->d_keep = nullptr;
    }
    void *d = nullptr;
    if (hipMalloc(&d, words.size() * sizeof(words[0])) != hipSuccess) return DABGPU_ERR_NOMEM;
    if (hipMemcpy(d, words.data(), words.size() * sizeof(words[0]), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return DABGPU_ERR_HIP;
    }
    ctx->keep_tables.push_back(d);
    ctx->d_keep = static_cast<const unsigned long long *>(d);
    ctx->keep_symbols = 0;
    for (int l = 0; l < NB_DATA_SYMBOLS; l++)
        if (words[size_t(l) * 3] | words[size_t(l) * 3 + 1] | words[size_t(l) * 3 + 2]) ctx->keep_symbols++;
    // the same selection as merged byte runs, for the host-pointer call's copy-back
    ctx->keep_ranges.clear();
    for (int c = 0; c < NB_FRAME_BITS / 16; c++) {
        if (!(words[size_t(c / CHUNKS_PER_SYMBOL) * 3 + ((c % CHUNKS_PER_SYMBOL) >> 6)] >> ((c % CHUNKS_PER_SYMBOL) & 63) & 1ull)) continue;
        if (!ctx->keep_ranges.empty() && ctx->keep_ranges.back().first + ctx->keep_ranges.back().count == 16 * c)
            ctx->keep_ranges.back().count += 16;
        else
            ctx->keep_ranges.push_back(dabgpu_bit_range{16 * c, 16});
    }
    return DABGPU_OK;
}

int dabgpu_fft_symbols_dev(dabgpu_ctx *ctx, const void *d_iq, size_t frame_stride, int n_frames,
                           const float *d_freq_offset, void *d_spectra, void *stream) {
    if (!ctx || !d_spectra) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    int rc = check_iq(d_iq, frame_stride, n_frames);
    if (rc) return rc;
    if (n_frames == 0) return DABGPU_OK;
    hipStream_t s = pick_stream(ctx, stream);
    dabk::OfdmTables tab{ctx->d_twiddle, ctx->d_bin_of_n, ctx->d_n_of_vj};
    dabk::OfdmArgs a{};
    a.iq = static_cast<const float2 *>(d_iq);
    a.frame_stride = frame_stride;
    a.freq_offset = d_freq_offset;
    a.n_frames = n_frames;
    a.spectra = static_cast<float2 *>(d_spectra);
    ScopedTimer tm(ctx, 3, s);
    const RunPlan plan = plan_runs(ctx, n_frames, NB_FRAME_SYMBOLS);
    a.uncut_frames = plan.uncut_frames;
    HIP_TRY(dabk::launch_fft_symbols(tab, a, plan.parts, s));
    return DABGPU_OK;
}

// host-pointer variants: stage through device buffers on the context stream
static size_t iq_span(size_t frame_stride, int n_frames) {
    return (size_t(n_frames - 1) * frame_stride + size_t(NB_FRAME_SYMBOLS) * NB_SYM_PERIOD) * sizeof(float2);
}

int dabgpu_ofdm_demod_frames(dabgpu_ctx *ctx, const float *iq, size_t frame_stride, int n_frames,
                             const float *freq_offset, int8_t *soft, float *cyc, float *dqpsk) {
    if (!ctx || !iq || !soft || n_frames < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (n_frames == 0) return DABGPU_OK;
    void *d_iq, *d_fo = nullptr, *d_soft, *d_cyc = nullptr, *d_dq = nullptr;
    int rc;
    const size_t nb_iq = iq_span(frame_stride, n_frames);
    const size_t nb_soft = size_t(n_frames) * NB_FRAME_BITS;
    const size_t nb_cyc = size_t(n_frames) * NB_FRAME_SYMBOLS * sizeof(float2);
    const size_t nb_dq = size_t(n_frames) * NB_DATA_SYMBOLS * NB_CARRIERS * sizeof(float2);
    if ((rc = stage(ctx, 0, nb_iq, &d_iq))) return rc;
    if ((rc = stage(ctx, 1, nb_soft, &d_soft))) return rc;
    if (freq_offset && (rc = stage(ctx, 2, sizeof(float) * n_frames, &d_fo))) return rc;
    if (cyc && (rc = stage(ctx, 3, nb_cyc, &d_cyc))) return rc;
    if (dqpsk && (rc = stage(ctx, 4, nb_dq, &d_dq))) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(hipMemcpyAsync(d_iq, iq, nb_iq, hipMemcpyHostToDevice, s));
    if (freq_offset) HIP_TRY(hipMemcpyAsync(d_fo, freq_offset, sizeof(float) * n_frames, hipMemcpyHostToDevice, s));
    rc = dabgpu_ofdm_demod_frames_dev(ctx, d_iq, frame_stride, n_frames, static_cast<const float *>(d_fo),
                                      static_cast<int8_t *>(d_soft), d_cyc, d_dq, s);
    if (rc) return rc;
    if (ctx->d_keep && !dqpsk) {
        // a selection is active: the kernel wrote only the selected runs of the staging buffer, and only those go
        // back -- the rest of the caller's `soft` stays as it was (one strided copy per run, over all frames)
        for (const dabgpu_bit_range &r : ctx->keep_ranges)
            HIP_TRY(hipMemcpy2DAsync(soft + r.first, NB_FRAME_BITS, static_cast<const int8_t *>(d_soft) + r.first, NB_FRAME_BITS,
                                     size_t(r.count), size_t(n_frames), hipMemcpyDeviceToHost, s));
    } else {
        HIP_TRY(hipMemcpyAsync(soft, d_soft, nb_soft, hipMemcpyDeviceToHost, s));
    }
    if (cyc) HIP_TRY(hipMemcpyAsync(cyc, d_cyc, nb_cyc, hipMemcpyDeviceToHost, s));
    if (dqpsk) HIP_TRY(hipMemcpyAsync(dqpsk, d_dq, nb_dq, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return DABGPU_OK;
}

// ---------------------------------------------------------------------------- closed-loop stream call
static_assert(sizeof(dabgpu_stream_state) == 64 && sizeof(dabk::StreamState) == 64, "stream state layout");
static_assert(offsetof(dabgpu_stream_state, next_frame_start) == offsetof(dabk::StreamState, next_frame_start) &&
              offsetof(dabgpu_stream_state, drift) == offsetof(dabk::StreamState, drift), "stream state layout");

// The stream states are read and written by launches on whatever stream the caller passed: remember the most recent
// one, so that the host-side accessors can wait for exactly that work.
}  // extern "C"
namespace dabapi {
bool known_coherent_host(const void *host, size_t bytes) {
    const char *p = static_cast<const char *>(host);
    std::lock_guard<std::mutex> lock(g_host_mutex);
    for (const auto &r : g_host_ranges)
        if (p >= r.first && p + bytes <= r.first + r.second) return true;
    return false;
}

int ensure_bounce(dabgpu_ctx *ctx, size_t bytes) {
    if (ctx->h_bounce_bytes >= bytes + 64) return DABGPU_OK;
    if (ctx->h_bounce) (void)hipHostFree(ctx->h_bounce);
    ctx->h_bounce = nullptr;
    ctx->h_bounce_bytes = 0;
    const size_t want = std::max<size_t>(bytes + 64, 256);
    if (hipHostMalloc(&ctx->h_bounce, want, hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();
        ctx->h_bounce = nullptr;
        return DABGPU_ERR_NOMEM;
    }
    ctx->h_bounce_bytes = want;
    std::memset(ctx->h_bounce, 0, want);
    return DABGPU_OK;
}

void *device_alias_of_pinned(const void *host) {
    hipPointerAttribute_t at{};
    if (!host || hipPointerGetAttributes(&at, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return at.type == hipMemoryTypeHost ? at.devicePointer : nullptr;
}

int note_state_use(dabgpu_ctx *ctx, hipStream_t s) {
    if (!ctx->ev_states && hipEventCreateWithFlags(&ctx->ev_states, hipEventDisableTiming) != hipSuccess) return DABGPU_ERR_HIP;
    HIP_TRY(hipEventRecord(ctx->ev_states, s));
    ctx->ev_states_pending = true;
    return DABGPU_OK;
}
int wait_state_use(dabgpu_ctx *ctx) {
    if (ctx->ev_states_pending) {
        HIP_TRY(hipEventSynchronize(ctx->ev_states));
        ctx->ev_states_pending = false;
    }
    return DABGPU_OK;
}
}  // namespace dabapi
extern "C" {

int dabgpu_streams_reset(dabgpu_ctx *ctx, int n_streams) {
    if (!ctx || n_streams < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    int wrc = wait_state_use(ctx);
    if (wrc) return wrc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (n_streams > ctx->n_states) {
        if (ctx->d_states) (void)hipFree(ctx->d_states);
        ctx->d_states = nullptr;
        ctx->n_states = 0;
        if (hipMalloc(reinterpret_cast<void **>(&ctx->d_states), sizeof(dabk::StreamState) * size_t(n_streams)) != hipSuccess)
            return DABGPU_ERR_NOMEM;
    }
    ctx->n_states = n_streams;
    if (n_streams > 0) HIP_TRY(hipMemset(ctx->d_states, 0, sizeof(dabk::StreamState) * size_t(n_streams)));
    return DABGPU_OK;
}

dabgpu_stream_state *dabgpu_stream_states(dabgpu_ctx *ctx) {
    return ctx ? reinterpret_cast<dabgpu_stream_state *>(ctx->d_states) : nullptr;
}

int dabgpu_set_stream_offsets(dabgpu_ctx *ctx, int stream_index, const float *fine, const float *coarse) {
    if (!ctx || stream_index < 0 || stream_index >= ctx->n_states) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    int wrc = wait_state_use(ctx);
    if (wrc) return wrc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    dabk::StreamState *st = ctx->d_states + stream_index;
    if (fine) HIP_TRY(hipMemcpy(&st->fine_freq_offset, fine, sizeof(float), hipMemcpyHostToDevice));
    if (coarse) HIP_TRY(hipMemcpy(&st->coarse_freq_offset, coarse, sizeof(float), hipMemcpyHostToDevice));
    // a re-seeded stream pulls in again: the decision-directed loop's gate forgets the branch it was holding
    const int32_t pulling_in[2] = {dabk::DD_NO_BRANCH, dabk::DD_NO_BRANCH};
    static_assert(offsetof(dabk::StreamState, dd_pending) == offsetof(dabk::StreamState, dd_branch) + 4, "branch, pending adjacent");
    if (fine || coarse) HIP_TRY(hipMemcpy(&st->dd_branch, pulling_in, sizeof(pulling_in), hipMemcpyHostToDevice));
    return DABGPU_OK;
}

}  // extern "C"
namespace dabapi {
void stats_of(const dabk::StreamState &st, dabgpu_stats *out) {
    // READING_SYMBOLS / FINDING_NULL_POWER_DIP (a tracked stream that lost every frame of a call is searching again)
    out->state = (st.total_frames_read > 0 && !(st.tracking == 0 && st.next_frame_start != 0.0)) ? 4 : 0;
    out->fine_freq_offset = st.fine_freq_offset;
    out->coarse_freq_offset = st.coarse_freq_offset;
    out->net_freq_offset = st.fine_freq_offset + st.coarse_freq_offset;
    out->signal_average = st.signal_average;
    out->total_frames_read = st.total_frames_read;
    out->total_frames_desync = st.total_frames_desync;
    out->last_fine_error = st.last_fine_error;
    out->tracking = st.tracking;
    out->last_time_offset = st.last_time_offset;
    out->next_frame_start = st.next_frame_start;
    out->drift = st.drift;
    out->last_peak_to_mean = st.last_peak_to_mean;
    out->loop_gated = st.loop_gated;
    out->reserved = 0;
}
}  // namespace dabapi
extern "C" {

int dabgpu_get_stats(dabgpu_ctx *ctx, int stream_index, dabgpu_stats *out) {
    if (!ctx || !out || stream_index < 0 || stream_index >= ctx->n_states) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    int wrc = wait_state_use(ctx);
    if (wrc) return wrc;
    dabk::StreamState st;
    HIP_TRY(hipMemcpyAsync(&st, ctx->d_states + stream_index, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    stats_of(st, out);
    return DABGPU_OK;
}

int dabgpu_set_stream_loop(dabgpu_ctx *ctx, float signal_update_beta, float thr_null_start, int decision_directed) {
    if (!ctx || !(signal_update_beta >= 0.f && signal_update_beta <= 1.f) || !(thr_null_start >= 0.f && thr_null_start <= 1.f))
        return DABGPU_ERR_ARG;
    ctx->signal_beta = signal_update_beta;
    ctx->thr_null_start = thr_null_start;
    ctx->loop_dd = decision_directed != 0;
    return DABGPU_OK;
}

int dabgpu_set_loop_gate(dabgpu_ctx *ctx, float dd_gate) {
    if (!ctx || !(dd_gate >= 0.f && dd_gate <= 1000.f)) return DABGPU_ERR_ARG;
    ctx->dd_gate = dd_gate;
    return DABGPU_OK;
}

int dabgpu_ofdm_demod_streams_dev(dabgpu_ctx *ctx, const void *d_iq, size_t frame_stride, int n_streams,
                                  int frames_per_stream, float fine_freq_update_beta, int8_t *d_soft, void *d_cyc,
                                  void *d_dqpsk, void *stream) {
    if (!ctx || !d_soft || n_streams < 0 || frames_per_stream < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (n_streams > ctx->n_states) return DABGPU_ERR_CAPACITY;   // dabgpu_streams_reset first
    if (!(fine_freq_update_beta >= 0.f && fine_freq_update_beta <= 1.f)) return DABGPU_ERR_ARG;
    if (size_t(n_streams) * size_t(frames_per_stream) > size_t(0x7fffffff) / NB_DATA_SYMBOLS) return DABGPU_ERR_ARG;
    const int n_frames = n_streams * frames_per_stream;
    int rc = check_iq(d_iq, frame_stride, n_frames);
    if (rc) return rc;
    if (reinterpret_cast<uintptr_t>(d_soft) & 15u) return DABGPU_ERR_ARG;
    if (n_frames == 0) return DABGPU_OK;
    hipStream_t s = pick_stream(ctx, stream);
    // No correlation output asked for: the loop's input stays in the library's scratch -- the correlations, or, once the
    // caller has switched the loop to decision-directed (dabgpu_set_stream_loop), the fourth-power sums of the
    // differential symbols: then the cyclic prefixes are not read at all, 17 % fewer bytes for an HBM-bound kernel.
    const bool dd = d_cyc == nullptr && ctx->loop_dd;
    void *d_dd = nullptr;
    if (dd && (rc = stage(ctx, 6, size_t(n_frames) * NB_FRAME_SYMBOLS * sizeof(float2), &d_dd))) return rc;
    if (!dd && !d_cyc && (rc = stage(ctx, 6, size_t(n_frames) * NB_FRAME_SYMBOLS * sizeof(float2), &d_cyc))) return rc;
    dabk::OfdmTables tab{ctx->d_twiddle, ctx->d_bin_of_n, ctx->d_n_of_vj};
    dabk::OfdmArgs a{};
    a.iq = static_cast<const float2 *>(d_iq);
    a.frame_stride = frame_stride;
    a.n_frames = n_frames;
    a.soft = d_soft;
    a.cyc = static_cast<float2 *>(d_cyc);
    a.dd4 = static_cast<float2 *>(d_dd);
    a.dqpsk = static_cast<float2 *>(d_dqpsk);
    a.keep = ctx->d_keep;
    a.state = ctx->d_states;
    a.frames_per_stream = frames_per_stream;
    {
        ScopedTimer tm(ctx, 0, s);
        const RunPlan plan = plan_runs(ctx, n_frames, NB_DATA_SYMBOLS);
        a.uncut_frames = plan.uncut_frames;
        HIP_TRY(dabk::launch_ofdm_demod(tab, a, plan.parts, s));
    }
    HIP_TRY(dabk::launch_stream_update(ctx->d_states, dd ? a.dd4 : a.cyc, a.iq, frame_stride, n_streams, frames_per_stream,
                                       fine_freq_update_beta, ctx->thr_null_start, ctx->signal_beta, dd ? 1 : 0, ctx->dd_gate,
                                       256 * ((a.keep && !a.dqpsk) ? ctx->keep_symbols : NB_DATA_SYMBOLS), s));
    return note_state_use(ctx, s);
}

int dabgpu_ofdm_demod_streams(dabgpu_ctx *ctx, const float *iq, size_t frame_stride, int n_streams,
                              int frames_per_stream, float fine_freq_update_beta, int8_t *soft, float *cyc,
                              float *dqpsk) {
    if (!ctx || !iq || !soft || n_streams < 0 || frames_per_stream < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (size_t(n_streams) * size_t(frames_per_stream) > size_t(0x7fffffff) / NB_DATA_SYMBOLS) return DABGPU_ERR_ARG;
    const int n_frames = n_streams * frames_per_stream;
    if (n_frames == 0) return DABGPU_OK;
    void *d_iq, *d_soft, *d_cyc = nullptr, *d_dq = nullptr;
    int rc;
    const size_t nb_iq = iq_span(frame_stride, n_frames);
    const size_t nb_soft = size_t(n_frames) * NB_FRAME_BITS;
    const size_t nb_cyc = size_t(n_frames) * NB_FRAME_SYMBOLS * sizeof(float2);
    const size_t nb_dq = size_t(n_frames) * NB_DATA_SYMBOLS * NB_CARRIERS * sizeof(float2);
    if ((rc = stage(ctx, 0, nb_iq, &d_iq))) return rc;
    if ((rc = stage(ctx, 1, nb_soft, &d_soft))) return rc;
    if ((rc = stage(ctx, 3, nb_cyc, &d_cyc))) return rc;
    if (dqpsk && (rc = stage(ctx, 4, nb_dq, &d_dq))) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(hipMemcpyAsync(d_iq, iq, nb_iq, hipMemcpyHostToDevice, s));
    rc = dabgpu_ofdm_demod_streams_dev(ctx, d_iq, frame_stride, n_streams, frames_per_stream, fine_freq_update_beta,
                                       static_cast<int8_t *>(d_soft), d_cyc, d_dq, s);
    if (rc) return rc;
    if (ctx->d_keep && !dqpsk) {
        for (const dabgpu_bit_range &r : ctx->keep_ranges)
            HIP_TRY(hipMemcpy2DAsync(soft + r.first, NB_FRAME_BITS, static_cast<const int8_t *>(d_soft) + r.first, NB_FRAME_BITS,
                                     size_t(r.count), size_t(n_frames), hipMemcpyDeviceToHost, s));
    } else {
        HIP_TRY(hipMemcpyAsync(soft, d_soft, nb_soft, hipMemcpyDeviceToHost, s));
    }
    if (cyc) HIP_TRY(hipMemcpyAsync(cyc, d_cyc, nb_cyc, hipMemcpyDeviceToHost, s));
    if (dqpsk) HIP_TRY(hipMemcpyAsync(dqpsk, d_dq, nb_dq, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return DABGPU_OK;
}

int dabgpu_fft_symbols(dabgpu_ctx *ctx, const float *iq, size_t frame_stride, int n_frames,
                       const float *freq_offset, float *spectra) {
    if (!ctx || !iq || !spectra || n_frames < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (n_frames == 0) return DABGPU_OK;
    void *d_iq, *d_fo = nullptr, *d_sp;
    int rc;
    const size_t nb_iq = iq_span(frame_stride, n_frames);
    const size_t nb_sp = size_t(n_frames) * NB_FRAME_SYMBOLS * NB_FFT * sizeof(float2);
    if ((rc = stage(ctx, 0, nb_iq, &d_iq))) return rc;
    if ((rc = stage(ctx, 4, nb_sp, &d_sp))) return rc;
    if (freq_offset && (rc = stage(ctx, 2, sizeof(float) * n_frames, &d_fo))) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(hipMemcpyAsync(d_iq, iq, nb_iq, hipMemcpyHostToDevice, s));
    if (freq_offset) HIP_TRY(hipMemcpyAsync(d_fo, freq_offset, sizeof(float) * n_frames, hipMemcpyHostToDevice, s));
    rc = dabgpu_fft_symbols_dev(ctx, d_iq, frame_stride, n_frames, static_cast<const float *>(d_fo), d_sp, s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(spectra, d_sp, nb_sp, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return DABGPU_OK;
}

// ---------------------------------------------------------------------------- PRS sync
static_assert(sizeof(dabgpu_sync_result) == sizeof(dabk::SyncResult), "ABI struct mirrors the kernel's");

int dabgpu_sync_prs_dev(dabgpu_ctx *ctx, const void *d_iq, size_t frame_stride, int n_frames,
                        const float *d_freq_offset, int max_coarse, dabgpu_sync_result *d_out, void *stream) {
    if (!ctx || !d_iq || !d_out || n_frames < 0 || max_coarse < 0 || max_coarse > 1023) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if ((reinterpret_cast<uintptr_t>(d_iq) & 15u) || (frame_stride & 1u)) return DABGPU_ERR_ARG;
    if (n_frames > 1 && frame_stride < size_t(NB_SYM_PERIOD)) return DABGPU_ERR_ARG;
    if (n_frames == 0) return DABGPU_OK;
    hipStream_t s = pick_stream(ctx, stream);
    dabk::SyncTables tab{ctx->d_twiddle, ctx->d_prs_qt, ctx->d_sync_pairs, ctx->n_sync_pairs, ctx->d_sync_fs};
    HIP_TRY(dabk::launch_prs_sync(tab, static_cast<const float2 *>(d_iq), frame_stride, n_frames, d_freq_offset,
                                  max_coarse, reinterpret_cast<dabk::SyncResult *>(d_out), s));
    return DABGPU_OK;
}

int dabgpu_sync_prs(dabgpu_ctx *ctx, const float *iq, size_t frame_stride, int n_frames, const float *freq_offset,
                    int max_coarse, dabgpu_sync_result *out) {
    if (!ctx || !iq || !out || n_frames < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (n_frames == 0) return DABGPU_OK;
    void *d_iq, *d_fo = nullptr, *d_out;
    int rc;
    const size_t nb_iq = (size_t(n_frames - 1) * frame_stride + NB_SYM_PERIOD) * sizeof(float2);
    if ((rc = stage(ctx, 0, nb_iq, &d_iq))) return rc;
    if ((rc = stage(ctx, 3, sizeof(dabgpu_sync_result) * n_frames, &d_out))) return rc;
    if (freq_offset && (rc = stage(ctx, 2, sizeof(float) * n_frames, &d_fo))) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(hipMemcpyAsync(d_iq, iq, nb_iq, hipMemcpyHostToDevice, s));
    if (freq_offset) HIP_TRY(hipMemcpyAsync(d_fo, freq_offset, sizeof(float) * n_frames, hipMemcpyHostToDevice, s));
    rc = dabgpu_sync_prs_dev(ctx, d_iq, frame_stride, n_frames, static_cast<const float *>(d_fo), max_coarse,
                             static_cast<dabgpu_sync_result *>(d_out), s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, d_out, sizeof(dabgpu_sync_result) * n_frames, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return DABGPU_OK;
}

// ---------------------------------------------------------------------------- acquisition
void dabgpu_acquire_default_cfg(dabgpu_acquire_cfg *cfg) {
    if (!cfg) return;
    cfg->thr_null_start = 0.35f;
    cfg->thr_null_end = 0.75f;
    cfg->min_null_blocks = 30;
    cfg->max_coarse_carriers = 200;
    cfg->min_peak_to_mean = 30.0f;
    cfg->timing_margin = 64;
    cfg->impulse_peak_distance_probability = 0.15f;
    cfg->first_path_rel = 0.25f;
    cfg->level_chunk_blocks = 256;
    cfg->reserved = 0;
}

static bool peak_rule_ok(float distance_prob, float first_path_rel) {
    return distance_prob >= 0.f && distance_prob <= 1.f && first_path_rel >= 0.f && first_path_rel <= 1.f;
}

// scratch + argument block of the acquisition kernels (dabgpu_acquire_dev, auto-acquisition of the tracked call)
static int acquire_args(dabgpu_ctx *ctx, const void *d_iq, size_t stream_stride, int n_streams, int64_t n_samples,
                        const dabgpu_acquire_cfg &c, int max_frames, dabgpu_acquired_frame *d_out, int32_t *d_counts,
                        hipStream_t s, dabk::AcquireArgs &a) {
    const size_t need = dabk::acquire_scratch_bytes(n_streams, n_samples, max_frames);
    if (ctx->acq_scratch_bytes < need) {
        HIP_TRY(hipStreamSynchronize(s));
        if (ctx->d_acq_scratch) (void)hipFree(ctx->d_acq_scratch);
        ctx->d_acq_scratch = nullptr;
        ctx->acq_scratch_bytes = 0;
        if (hipMalloc(&ctx->d_acq_scratch, need) != hipSuccess) return DABGPU_ERR_NOMEM;
        ctx->acq_scratch_bytes = need;
    }
    a.iq = static_cast<const float2 *>(d_iq);
    a.stream_stride = stream_stride;
    a.n_streams = n_streams;
    a.n_samples = n_samples;
    a.thr_start = c.thr_null_start;
    a.thr_end = c.thr_null_end;
    a.level_chunk = c.level_chunk_blocks;
    a.min_blocks = c.min_null_blocks;
    a.max_coarse = c.max_coarse_carriers;
    a.min_peak_to_mean = c.min_peak_to_mean;
    a.margin = c.timing_margin;
    a.rule.distance_prob = c.impulse_peak_distance_probability;
    a.rule.expected = 0;
    a.rule.first_path_rel = c.first_path_rel;
    a.max_out = max_frames;
    a.l1 = static_cast<float *>(ctx->d_acq_scratch);
    const size_t l1_bytes = (size_t(n_streams) * size_t(n_samples / 64) * sizeof(float) + 255) & ~size_t(255);
    a.cands = reinterpret_cast<int64_t *>(static_cast<char *>(ctx->d_acq_scratch) + l1_bytes);
    a.out = reinterpret_cast<dabk::AcquiredFrame *>(d_out);
    a.counts = d_counts;
    return DABGPU_OK;
}

int dabgpu_acquire_dev(dabgpu_ctx *ctx, const void *d_iq, size_t stream_stride, int n_streams, int64_t n_samples,
                       const dabgpu_acquire_cfg *cfg, int max_frames, dabgpu_acquired_frame *d_out, int32_t *d_counts,
                       void *stream) {
    static_assert(sizeof(dabgpu_acquired_frame) == 32 && sizeof(dabk::AcquiredFrame) == 32, "acquired-frame layout");
    if (!ctx || !d_iq || !d_out || !d_counts || n_streams < 0 || max_frames <= 0 || n_samples < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (reinterpret_cast<uintptr_t>(d_iq) & 7u) return DABGPU_ERR_ARG;
    if (n_streams > 1 && stream_stride < size_t(n_samples)) return DABGPU_ERR_ARG;
    dabgpu_acquire_cfg c;
    if (cfg) c = *cfg; else dabgpu_acquire_default_cfg(&c);
    if (c.max_coarse_carriers < 0 || c.max_coarse_carriers > 1023 || c.min_null_blocks < 1 || c.timing_margin < 0 ||
        c.timing_margin > NB_CP || !(c.thr_null_start > 0.f) || !(c.thr_null_end >= c.thr_null_start) ||
        !peak_rule_ok(c.impulse_peak_distance_probability, c.first_path_rel) ||
        (c.level_chunk_blocks != 0 && (c.level_chunk_blocks < 64 || c.level_chunk_blocks > 16384 ||
                                       (c.level_chunk_blocks & (c.level_chunk_blocks - 1)))))
        return DABGPU_ERR_ARG;
    if (n_streams == 0) return DABGPU_OK;
    hipStream_t s = pick_stream(ctx, stream);
    int rc2;
    if (n_samples < 64) {                                      // nothing to search: no frames anywhere
        HIP_TRY(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * n_streams, s));
        HIP_TRY(hipMemsetAsync(d_out, 0, sizeof(dabgpu_acquired_frame) * size_t(n_streams) * max_frames, s));
        return DABGPU_OK;
    }
    dabk::AcquireArgs a{};
    if ((rc2 = acquire_args(ctx, d_iq, stream_stride, n_streams, n_samples, c, max_frames, d_out, d_counts, s, a))) return rc2;
    dabk::SyncTables tab{ctx->d_twiddle, ctx->d_prs_qt, ctx->d_sync_pairs, ctx->n_sync_pairs, ctx->d_sync_fs};
    HIP_TRY(dabk::launch_acquire(tab, a, s));
    return DABGPU_OK;
}

int dabgpu_acquire(dabgpu_ctx *ctx, const float *iq, size_t stream_stride, int n_streams, int64_t n_samples,
                   const dabgpu_acquire_cfg *cfg, int max_frames, dabgpu_acquired_frame *out, int32_t *counts) {
    if (!ctx || !iq || !out || !counts || n_streams < 0 || max_frames <= 0 || n_samples < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (n_streams == 0) return DABGPU_OK;
    if (n_streams > 1 && stream_stride < size_t(n_samples)) return DABGPU_ERR_ARG;
    // a capture is as large as the caller makes it: its device copy is allocated for the call, not kept
    const size_t nb_iq = (size_t(n_streams - 1) * stream_stride + size_t(n_samples)) * sizeof(float2);
    const size_t nb_out = sizeof(dabgpu_acquired_frame) * size_t(n_streams) * max_frames;
    void *d_iq = nullptr, *d_out = nullptr, *d_cnt = nullptr;
    hipStream_t s = ctx->stream;
    int rc = DABGPU_OK;
    if (hipMalloc(&d_iq, std::max<size_t>(nb_iq, 16)) != hipSuccess || hipMalloc(&d_out, nb_out) != hipSuccess ||
        hipMalloc(&d_cnt, sizeof(int32_t) * n_streams) != hipSuccess)
        rc = DABGPU_ERR_NOMEM;
    if (!rc && hipMemcpyAsync(d_iq, iq, nb_iq, hipMemcpyHostToDevice, s) != hipSuccess) rc = DABGPU_ERR_HIP;
    if (!rc)
        rc = dabgpu_acquire_dev(ctx, d_iq, stream_stride, n_streams, n_samples, cfg, max_frames,
                                static_cast<dabgpu_acquired_frame *>(d_out), static_cast<int32_t *>(d_cnt), s);
    if (!rc && (hipMemcpyAsync(out, d_out, nb_out, hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipMemcpyAsync(counts, d_cnt, sizeof(int32_t) * n_streams, hipMemcpyDeviceToHost, s) != hipSuccess))
        rc = DABGPU_ERR_HIP;
    if (hipStreamSynchronize(s) != hipSuccess && !rc) rc = DABGPU_ERR_HIP;
    if (d_iq) (void)hipFree(d_iq);
    if (d_out) (void)hipFree(d_out);
    if (d_cnt) (void)hipFree(d_cnt);
    return rc;
}

int dabgpu_ofdm_demod_acquired_dev(dabgpu_ctx *ctx, const void *d_iq, size_t stream_stride, int n_streams,
                                   int max_frames, const dabgpu_acquired_frame *d_frames, int8_t *d_soft, void *d_cyc,
                                   void *d_dqpsk, void *stream) {
    if (!ctx || !d_iq || !d_frames || !d_soft || n_streams < 0 || max_frames <= 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if ((reinterpret_cast<uintptr_t>(d_iq) & 7u) || (reinterpret_cast<uintptr_t>(d_soft) & 15u)) return DABGPU_ERR_ARG;
    if (n_streams == 0) return DABGPU_OK;
    if (size_t(n_streams) * size_t(max_frames) > size_t(0x7fffffff) / NB_DATA_SYMBOLS) return DABGPU_ERR_ARG;
    hipStream_t s = pick_stream(ctx, stream);
    dabk::OfdmTables tab{ctx->d_twiddle, ctx->d_bin_of_n, ctx->d_n_of_vj};
    dabk::OfdmArgs a{};
    a.iq = static_cast<const float2 *>(d_iq);
    a.frame_stride = stream_stride;
    a.n_frames = n_streams * max_frames;
    a.soft = d_soft;
    a.cyc = static_cast<float2 *>(d_cyc);
    a.dqpsk = static_cast<float2 *>(d_dqpsk);
    a.acq = reinterpret_cast<const dabk::AcquiredFrame *>(d_frames);
    a.acq_per_stream = max_frames;
    a.keep = ctx->d_keep;
    ScopedTimer tm(ctx, 0, s);
    const RunPlan plan = plan_runs(ctx, a.n_frames, NB_DATA_SYMBOLS);
    a.uncut_frames = plan.uncut_frames;
    HIP_TRY(dabk::launch_ofdm_demod(tab, a, plan.parts, s));
    return DABGPU_OK;
}

// ---------------------------------------------------------------------------- timing tracking
void dabgpu_track_default_cfg(dabgpu_track_cfg *cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->fine_freq_update_beta = 0.9f;
    cfg->signal_update_beta = 0.95f;
    cfg->thr_null_start = 0.35f;
    cfg->min_peak_to_mean = 100.0f;
    cfg->impulse_peak_distance_probability = 0.15f;
    cfg->first_path_rel = 0.25f;
    cfg->drift_beta = 0.5f;
    cfg->coarse_freq_slow_beta = 0.1f;
    cfg->timing_margin = 64;
    cfg->max_coarse_carriers = 204;
    cfg->decision_directed = 0;         // the reference's estimator (cyclic-prefix correlations); 1 = this library's own, opt-in
    cfg->auto_acquire = 0;
    cfg->dd_gate = 2.5f;
}

static int track_cfg(const dabgpu_track_cfg *cfg, dabgpu_track_cfg &c) {
    if (cfg) c = *cfg; else dabgpu_track_default_cfg(&c);
    auto unit = [](float v) { return v >= 0.f && v <= 1.f; };
    if (!unit(c.fine_freq_update_beta) || !unit(c.signal_update_beta) || !unit(c.thr_null_start) || !unit(c.drift_beta) ||
        !unit(c.coarse_freq_slow_beta) || !peak_rule_ok(c.impulse_peak_distance_probability, c.first_path_rel) ||
        !(c.min_peak_to_mean >= 0.f) || c.timing_margin < 0 || c.timing_margin > NB_CP || c.max_coarse_carriers < 0 ||
        c.max_coarse_carriers > 1023 || !(c.dd_gate >= 0.f && c.dd_gate <= 1000.f) || c.reserved != 0)
        return DABGPU_ERR_ARG;
    return DABGPU_OK;
}

int dabgpu_track_start_dev(dabgpu_ctx *ctx, const dabgpu_acquired_frame *d_frames, const int32_t *d_counts, int n_streams,
                           int max_frames, int64_t advance, int only_lost, void *stream) {
    if (!ctx || !d_frames || !d_counts || n_streams < 0 || max_frames <= 0 || advance < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (n_streams > ctx->n_states) return DABGPU_ERR_CAPACITY;   // dabgpu_streams_reset first
    if (n_streams == 0) return DABGPU_OK;
    hipStream_t s = pick_stream(ctx, stream);
    HIP_TRY(dabk::launch_track_start(ctx->d_states, reinterpret_cast<const dabk::AcquiredFrame *>(d_frames), d_counts, n_streams,
                                     max_frames, advance, only_lost ? 1 : 0, s));
    if (only_lost) {
        // (tracking = 2 marks "started in this call" for the tracked call's own use; a stand-alone start has no update
        // launch behind it: turn the marks into 1 here)
        dabk::TrackUpdateArgs u{};
        u.state = ctx->d_states;
        u.n_streams = n_streams;
        u.max_out = 1;
        u.fixed_start = 0;
        u.settle_only = 1;
        HIP_TRY(dabk::launch_track_update(u, s));
    }
    return note_state_use(ctx, s);
}

// What a tracked call launches (tracked_launches): where the streams lie, where the results go, and the riders the
// one-frame call adds.  Value-initialised: everything optional is off.
struct TrackedCall {
    dabk::StreamState *states = nullptr;             // the streams' tracking state (device)
    const void *d_iq = nullptr;                      // [n_streams][stream_stride] cf32
    size_t stream_stride = 0;
    int n_streams = 0;
    int64_t n_samples = 0;                           // samples per stream in this call
    int max_frames = 1;                              // output rows per stream
    int64_t advance = 0;                             // samples the streams move on by after the call
    int fixed_start = 0;                             // the frame starts at sample 0 of its stream (one-frame call)
    int acquiring = 0;                               // ... and is the first after a null detection (coarse search, lock check)
    int8_t *d_soft = nullptr;
    void *d_cyc = nullptr, *d_dd4 = nullptr;         // the fine loop's input: cyclic-prefix correlations, or fourth-power sums
    void *d_dqpsk = nullptr;
    dabgpu_acquired_frame *d_frames = nullptr;
    dabgpu_sync_result *d_sync = nullptr;
    int32_t *d_counts = nullptr;
    const dabk::AcquireArgs *auto_acq = nullptr;     // streams that are not tracking are acquired in the same call
    // riders of the one-frame call: the frame's upload inside the synchronisation launch, the download inside the update's
    const void *upload_from = nullptr;
    size_t upload_bytes = 0;
    const dabk::CopyPiece *down = nullptr;           // n_down (<= 3) pieces
    int n_down = 0;
    dabk::StreamState *state_out = nullptr;          // the new state, written to page-locked memory by the updating workgroup
    bool note_states = true;                         // record the state event behind the call (off: the call synchronises itself)
};

// the three launches of a tracked call on `s`: PRS synchronisation at the predicted positions, demodulation of the frames
// where they lie, state update
static int tracked_launches(dabgpu_ctx *ctx, const TrackedCall &k, const dabgpu_track_cfg &c, hipStream_t s) {
    dabk::StreamState *const states = k.states;
    const void *const d_iq = k.d_iq;
    const size_t stream_stride = k.stream_stride;
    const int n_streams = k.n_streams, max_frames = k.max_frames, fixed_start = k.fixed_start;
    const int64_t n_samples = k.n_samples, advance = k.advance;
    const dabk::AcquireArgs *const auto_acq = k.auto_acq;
    int32_t *const d_counts = k.d_counts;
    dabk::SyncTables stab{ctx->d_twiddle, ctx->d_prs_qt, ctx->d_sync_pairs, ctx->n_sync_pairs, ctx->d_sync_fs};
    dabk::TrackArgs t{};
    t.state = states;
    t.iq = static_cast<const float2 *>(d_iq);
    t.stream_stride = stream_stride;
    t.n_streams = n_streams;
    t.n_samples = n_samples;
    t.max_out = max_frames;
    t.margin = c.timing_margin;
    t.min_peak_to_mean = c.min_peak_to_mean;
    t.rule.distance_prob = c.impulse_peak_distance_probability;
    t.rule.first_path_rel = c.first_path_rel;
    t.fixed_start = fixed_start;
    t.max_coarse = fixed_start ? c.max_coarse_carriers : 0;
    t.acquiring = k.acquiring;
    t.coarse_slow_beta = c.coarse_freq_slow_beta;
    t.out = reinterpret_cast<dabk::AcquiredFrame *>(k.d_frames);
    t.sync_out = reinterpret_cast<dabk::SyncResult *>(k.d_sync);
    if (k.upload_from) {
        // the one-frame call: the frame's upload rides in this launch, and the synchronisation reads its PRS straight
        // from the caller's page-locked buffer meanwhile (TrackArgs::copy_*)
        t.sync_iq = static_cast<const float2 *>(k.upload_from);
        t.copy_dst = static_cast<uint4 *>(const_cast<void *>(d_iq));
        t.copy_src = static_cast<const uint4 *>(k.upload_from);
        t.copy_n16 = unsigned(k.upload_bytes >> 4);
    }
    HIP_TRY(dabk::launch_track_sync(stab, t, s));
    // streams that are not tracking: acquired here (their rows of d_frames / d_counts; the pass above left them empty)
    if (auto_acq) HIP_TRY(dabk::launch_acquire(stab, *auto_acq, s));
    dabk::OfdmTables tab{ctx->d_twiddle, ctx->d_bin_of_n, ctx->d_n_of_vj};
    dabk::OfdmArgs a{};
    a.iq = static_cast<const float2 *>(d_iq);
    a.frame_stride = stream_stride;
    a.n_frames = n_streams * max_frames;
    a.soft = k.d_soft;
    a.cyc = static_cast<float2 *>(k.d_cyc);
    a.dd4 = static_cast<float2 *>(k.d_dd4);
    a.dqpsk = static_cast<float2 *>(k.d_dqpsk);
    a.acq = reinterpret_cast<const dabk::AcquiredFrame *>(k.d_frames);
    a.acq_per_stream = max_frames;
    a.keep = ctx->d_keep;
    {
        ScopedTimer tm(ctx, 0, s);
        const RunPlan plan = plan_runs(ctx, a.n_frames, NB_DATA_SYMBOLS);
        a.uncut_frames = plan.uncut_frames;
        HIP_TRY(dabk::launch_ofdm_demod(tab, a, plan.parts, s));
    }
    dabk::TrackUpdateArgs u{};
    u.state = states;
    u.frames = t.out;
    u.cyc = a.cyc ? a.cyc : a.dd4;
    u.dd = a.cyc ? 0 : 1;
    u.iq = t.iq;
    u.stream_stride = stream_stride;
    u.n_streams = n_streams;
    u.n_samples = n_samples;
    u.max_out = max_frames;
    u.advance = advance;
    u.fine_beta = c.fine_freq_update_beta;
    u.drift_beta = c.drift_beta;
    u.signal_beta = c.signal_update_beta;
    u.thr_null_start = c.thr_null_start;
    u.fixed_start = fixed_start;
    u.counts = d_counts;
    u.dd_gate = c.dd_gate;
    u.dd_terms_per_frame = 256 * ((a.keep && !a.dqpsk) ? ctx->keep_symbols : NB_DATA_SYMBOLS);
    for (int i = 0; i < k.n_down && i < 3; i++) u.down[i] = k.down[i];
    u.state_out = k.state_out;
    // ... and their tracking starts from what the acquisition found (marked 2; the update launch makes it 1)
    if (auto_acq)
        HIP_TRY(dabk::launch_track_start(states, t.out, d_counts, n_streams, max_frames, advance, 1, s));
    HIP_TRY(dabk::launch_track_update(u, s));
    return k.note_states ? note_state_use(ctx, s) : DABGPU_OK;
}

int dabgpu_ofdm_demod_tracked_dev(dabgpu_ctx *ctx, const void *d_iq, size_t stream_stride, int n_streams,
                                  int64_t n_samples, int max_frames, int64_t advance, const dabgpu_track_cfg *cfg,
                                  int8_t *d_soft, void *d_cyc, void *d_dqpsk, dabgpu_acquired_frame *d_frames,
                                  int32_t *d_counts, void *stream) {
    if (!ctx || !d_iq || !d_soft || !d_frames || !d_counts || n_streams < 0 || max_frames <= 0 || n_samples < 0 || advance < 0)
        return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if ((reinterpret_cast<uintptr_t>(d_iq) & 7u) || (reinterpret_cast<uintptr_t>(d_soft) & 15u)) return DABGPU_ERR_ARG;
    if (n_streams > 1 && stream_stride < size_t(n_samples)) return DABGPU_ERR_ARG;
    if (n_streams > ctx->n_states) return DABGPU_ERR_CAPACITY;   // dabgpu_streams_reset first
    if (size_t(n_streams) * size_t(max_frames) > size_t(0x7fffffff) / NB_DATA_SYMBOLS) return DABGPU_ERR_ARG;
    dabgpu_track_cfg c;
    int rc = track_cfg(cfg, c);
    if (rc) return rc;
    if (n_streams == 0) return DABGPU_OK;
    hipStream_t s = pick_stream(ctx, stream);
    // (no correlation output asked for: the loop's input stays in the library's scratch -- the cyclic-prefix correlations by
    // default, as the reference's loop; cfg.decision_directed: the fourth-power sums, the cyclic prefixes are not read -- see
    // the stream call; acquisition leaves the fine offset well inside that estimator's range)
    void *d_dd = nullptr;
    if (!d_cyc && (rc = stage(ctx, 6, size_t(n_streams) * max_frames * NB_FRAME_SYMBOLS * sizeof(float2), c.decision_directed ? &d_dd : &d_cyc)))
        return rc;
    dabk::AcquireArgs acq{};
    if (c.auto_acquire && n_samples >= 64) {
        dabgpu_acquire_cfg ac;
        dabgpu_acquire_default_cfg(&ac);
        ac.thr_null_start = c.thr_null_start;
        ac.max_coarse_carriers = c.max_coarse_carriers;
        ac.timing_margin = c.timing_margin;
        ac.impulse_peak_distance_probability = c.impulse_peak_distance_probability;
        ac.first_path_rel = c.first_path_rel;
        if ((rc = acquire_args(ctx, d_iq, stream_stride, n_streams, n_samples, ac, max_frames, d_frames, d_counts, s, acq))) return rc;
        acq.skip_tracked = ctx->d_states;
    }
    TrackedCall k;
    k.states = ctx->d_states;
    k.d_iq = d_iq;
    k.stream_stride = stream_stride;
    k.n_streams = n_streams;
    k.n_samples = n_samples;
    k.max_frames = max_frames;
    k.advance = advance;
    k.d_soft = d_soft;
    k.d_cyc = d_cyc;
    k.d_dd4 = d_dd;
    k.d_dqpsk = d_dqpsk;
    k.d_frames = d_frames;
    k.d_counts = d_counts;
    k.auto_acq = (c.auto_acquire && n_samples >= 64) ? &acq : nullptr;
    return tracked_launches(ctx, k, c, s);
}

int dabgpu_ofdm_demod_stream_frame(dabgpu_ctx *ctx, int stream_index, const float *iq, int acquiring,
                                   const dabgpu_track_cfg *cfg, int8_t *soft, float *dqpsk, dabgpu_frame_result *result) {
    if (!ctx || !iq || !soft || !result || stream_index < 0 || stream_index >= ctx->n_states) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    dabgpu_track_cfg c;
    int rc = track_cfg(cfg, c);
    if (rc) return rc;
    constexpr size_t nb_iq = size_t(NB_FRAME_SYMBOLS) * NB_SYM_PERIOD * sizeof(float2);
    constexpr size_t nb_dq = size_t(NB_DATA_SYMBOLS) * NB_CARRIERS * sizeof(float2);
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    // one result block: soft bits | acquired frame | sync result | state  (| constellation, in a buffer of its own)
    const size_t off_fr = al(NB_FRAME_BITS), off_sy = off_fr + al(sizeof(dabgpu_acquired_frame)),
                 off_st = off_sy + al(sizeof(dabgpu_sync_result)), nb_res = off_st + al(sizeof(dabk::StreamState));
    void *d_iq, *d_res, *d_cyc = nullptr, *d_dd = nullptr, *d_dq = nullptr;
    if ((rc = stage(ctx, 0, nb_iq, &d_iq))) return rc;
    if ((rc = stage(ctx, 1, nb_res, &d_res))) return rc;
    // the fine loop's input: the 76 cyclic-prefix correlations (the reference's estimator, the default), or -- opt-in,
    // cfg->decision_directed -- the fourth-power sums
    if ((rc = stage(ctx, 6, NB_FRAME_SYMBOLS * sizeof(float2), c.decision_directed ? &d_dd : &d_cyc))) return rc;
    if (dqpsk && (rc = stage(ctx, 4, nb_dq, &d_dq))) return rc;
    if ((rc = ensure_bounce(ctx, nb_res))) return rc;
    if ((rc = wait_state_use(ctx))) return rc;
    if (injected_failure(ctx)) return DABGPU_ERR_HIP;            // (test hook: a device call that fails before any launch)
    hipStream_t s = ctx->stream;
    char *res = static_cast<char *>(d_res);
    // one upload: by a kernel when the frame lies in page-locked memory the device can address (the host mirror's does)
    static_assert(nb_iq % 16 == 0 && NB_FRAME_BITS % 16 == 0, "whole 16-byte words");
    // (every query first: once the upload is enqueued the host only enqueues, and stays ahead of the device)
    void *h_dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&h_dev, ctx->h_bounce, 0));
    void *iq_alias = device_alias_of_pinned(iq), *soft_alias = device_alias_of_pinned(soft);
    if ((reinterpret_cast<uintptr_t>(iq_alias) | reinterpret_cast<uintptr_t>(soft_alias)) & 15) iq_alias = soft_alias = nullptr;
    if (!iq_alias) HIP_TRY(hipMemcpyAsync(d_iq, iq, nb_iq, hipMemcpyHostToDevice, s));
    dabk::StreamState *st = ctx->d_states + stream_index;
    // Page-locked buffers (the host mirror's): the 1.55 MB upload is the longest single piece of the call (37 us), and the
    // PRS synchronisation (20 us) only needs the frame's first symbol -- so the upload rides in the synchronisation's own
    // launch: its extra workgroups copy the frame while the first one reads its 20 kB straight from the caller's buffer.
    // The call ends in its own synchronisation, so no state event is recorded (an event record between two launches
    // cost 5.6 us of idle device).  Measured and rejected on the way (profiles/r05_frame_path.md): the synchronisation
    // on a second stream (the cross-stream event: 11 us of idle device), soft bits written by the demodulation launch
    // straight into the caller's buffer (the launch 3.4 us slower, the copy behind it only 1.6 us shorter).
    // ... and the download rides in the LAST launch (the state update's): soft bits (straight into the caller's buffer when
    // that is page-locked too: no copy by the CPU afterwards), frame and sync records; the updating workgroup writes the
    // new state to the landing area itself.  Three launches per call: upload + synchronisation, demodulation, update + download.
    static_assert(sizeof(dabk::StreamState) % 16 == 0, "the state goes out in 16-byte words");
    char *hd = static_cast<char *>(h_dev);
    // ... and the constellation, when a display asks for it and its buffer is coherent page-locked memory (the host mirror's
    // is): 0.9 MB more in the same launch instead of a copy-engine transfer and a sleep behind it
    void *dq_alias = (dqpsk && known_coherent_host(dqpsk, nb_dq)) ? device_alias_of_pinned(dqpsk) : nullptr;
    if (reinterpret_cast<uintptr_t>(dq_alias) & 15) dq_alias = nullptr;
    const dabk::CopyPiece down[3] = {{soft_alias ? soft_alias : static_cast<void *>(hd), d_res, size_t(NB_FRAME_BITS)},
                                     {hd + off_fr, res + off_fr, off_st - off_fr},
                                     {dq_alias, d_dq, dq_alias ? nb_dq : 0}};
    TrackedCall k;
    k.states = st;
    k.d_iq = d_iq;
    k.stream_stride = nb_iq / sizeof(float2);
    k.n_streams = 1;
    k.n_samples = int64_t(nb_iq / sizeof(float2));
    k.fixed_start = 1;
    k.acquiring = acquiring ? 1 : 0;
    k.d_soft = reinterpret_cast<int8_t *>(res);
    k.d_cyc = d_cyc;
    k.d_dd4 = d_dd;
    k.d_dqpsk = d_dq;
    k.d_frames = reinterpret_cast<dabgpu_acquired_frame *>(res + off_fr);
    k.d_sync = reinterpret_cast<dabgpu_sync_result *>(res + off_sy);
    k.upload_from = iq_alias;
    k.upload_bytes = iq_alias ? nb_iq : 0;
    k.down = down;
    k.n_down = 3;
    k.state_out = reinterpret_cast<dabk::StreamState *>(hd + off_st);
    k.note_states = false;
    if ((rc = tracked_launches(ctx, k, c, s))) return rc;
    if (dqpsk && !dq_alias) {
        HIP_TRY(hipMemcpyAsync(dqpsk, d_dq, nb_dq, hipMemcpyDeviceToHost, s));           // (the constellation into any other memory:
        HIP_TRY(hipStreamSynchronize(s));                                                // a copy-engine transfer ends the usual way)
    } else {
        // one synchronisation: the word behind the landing area's payload (the area is at least nb_res + 64 bytes)
        const size_t off_flag = ctx->h_bounce_bytes - 64;
        // (the soft bits may have gone straight into the caller's buffer: the word is watched only when that buffer is coherent)
        if ((rc = wait_for_signal(s, reinterpret_cast<volatile unsigned long long *>(static_cast<char *>(ctx->h_bounce) + off_flag),
                                  reinterpret_cast<unsigned long long *>(hd + off_flag), ++ctx->signal_seq, false,
                                  !soft_alias || known_coherent_host(soft, NB_FRAME_BITS))))
            return rc;
    }
    ctx->ev_states_pending = false;
    const char *hb = static_cast<const char *>(ctx->h_bounce);
    if (!soft_alias) std::memcpy(soft, hb, NB_FRAME_BITS);
    dabgpu_acquired_frame fr;
    std::memcpy(&fr, hb + off_fr, sizeof(fr));
    std::memcpy(&result->sync, hb + off_sy, sizeof(result->sync));
    dabk::StreamState hs;
    std::memcpy(&hs, hb + off_st, sizeof(hs));
    result->flags = fr.flags;
    result->reserved = 0;
    stats_of(hs, &result->stats);
    return DABGPU_OK;
}

}  // extern "C"
