// dabgpu_api.hip -- the C ABI of libdabgpu (include/dabgpu.h): context, tables,
// argument checking and kernel launches.  No CPU fallback: without a gfx950 device
// dabgpu_create fails with DABGPU_ERR_NODEVICE and every compute entry point needs a
// context.
#include "../../include/dabgpu.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <vector>

#include "dab_tables.hpp"
#include "kernels.hpp"

using namespace dab;

namespace {

struct DeviceCode {
    dab::PunctureProfile prof;
    uint16_t *d_mother_pos = nullptr;
    uint8_t *d_prbs = nullptr;
    int32_t *d_punct_idx = nullptr;      // [4*nsteps] punctured index of each mother bit, -1 = erased (lane kernels)
    int32_t *d_fused_desc = nullptr, *d_fused_tiles = nullptr;   // fused lane forward pass (build_lane_fused_tables)
    dabk::LaneTables lane_tables() const { return dabk::LaneTables{d_punct_idx, d_fused_desc, d_fused_tiles}; }
    dabk::CodeTables tables(bool descramble) const {
        return dabk::CodeTables{d_mother_pos, prof.n_punct, prof.nsteps, descramble ? d_prbs : nullptr};
    }
};

// hipEvent pairs around the most recent launches of one kernel family (a ring: the launches are asynchronous, so
// the pairs can only be read back after the stream has caught up)
constexpr int TIMER_RING = 32;
struct Timer {
    hipEvent_t start[TIMER_RING] = {}, stop[TIMER_RING] = {};
    long recorded = 0;                   // launches timed since timing was switched on
};

}  // namespace

struct dabgpu_ctx {
    int device = 0;
    int max_frames = 0;
    hipStream_t stream = nullptr;
    float2 *d_twiddle = nullptr;
    uint16_t *d_bin_of_n = nullptr;
    uint16_t *d_n_of_vj = nullptr;
    int8_t *d_prs_qt = nullptr;
    uint16_t *d_sync_pairs = nullptr;
    int n_sync_pairs = 0;
    float2 *d_sync_fs = nullptr;         // FFT of the PRS's adjacent-carrier differential (coarse search by FFT)
    DeviceCode fic;
    std::map<std::vector<uint8_t>, std::unique_ptr<DeviceCode>> codes;   // keyed by puncture mask
    // staging for the host-pointer entry points
    // slots 0..5: staging of the host-pointer entry points; slot 6: the stream call's own cyclic-prefix correlations
    // (it runs on a caller's stream, so it must not share a slot with calls that run on the context stream)
    void *d_stage[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t stage_bytes[7] = {0, 0, 0, 0, 0, 0, 0};
    bool timing = false;
    Timer timers[4];
    int ofdm_parts_override = 0;
    const unsigned long long *d_keep = nullptr;          // current soft-bit selection table ([75][3] words) or nullptr
    std::vector<void *> keep_tables;                     // every table handed to a kernel so far (freed on destroy)
    void *d_acq_scratch = nullptr;       // block norms + candidates of dabgpu_acquire
    size_t acq_scratch_bytes = 0;
    void *d_lane_scratch = nullptr;      // work buffers of the codeword-per-lane Viterbi
    size_t lane_scratch_bytes = 0;
    int lane_mode = -1;                  // 1 = DABGPU_FLAG_VITERBI_LANE, 0 = DABGPU_FLAG_VITERBI_WAVE, -1 by batch size
    bool lane_unfused = false;           // DABGPU_FLAG_LANE_UNFUSED
    int wave_slots = 3072;       // resident OFDM wavefronts: 12 per CU
    std::vector<dabgpu_bit_range> keep_ranges;           // the current selection, merged, for the host-pointer copy-back
    dabk::StreamState *d_states = nullptr;               // per-stream tracking state (dabgpu_streams_reset)
    int n_states = 0;
    hipEvent_t ev_states = nullptr;      // recorded behind the last launch that reads or writes d_states, on ITS stream
    bool ev_states_pending = false;
    float thr_null_start = 0.35f;        // desync threshold of the stream call (null_l1_search.thresh_null_start)
    float signal_beta = 0.95f;           // signal_l1.update_beta of the stream call
    bool loop_dd = false;                // stream call without a correlation buffer: decision-directed fine loop
    // dabgpu_decode_stream_frames: de-interleaver rings of the stream's sub-channels, kept on the device between calls
    struct SubHistory {
        int start_address, length;
        size_t bytes;
        int8_t *ring[2];
        int cur;
        bool live;                       // decoded in the current call (a ring that misses a frame is stale: dropped)
    };
    std::vector<SubHistory> sub_history;
    void *h_bounce = nullptr;            // page-locked landing area of that call's single download
    size_t h_bounce_bytes = 0;
    // buffers handed out by dabgpu_alloc_frame_buffers_placed: virtual ranges mapped over physical chunks
    struct Mapped {
        void *va;
        size_t bytes, chunk;             // reserved = mapped bytes; scratch while mapping
        std::vector<hipMemGenericAllocationHandle_t> handles;
        std::vector<size_t> sizes;       // bytes of each handle's mapping, in address order
        // every mapping is undone on its own extents (an unmap spanning several mappings is not something the
        // virtual-memory API promises), then the physical memory and the address range go back
        void release() {
            size_t off = 0;
            for (size_t k = 0; k < handles.size(); k++) {
                (void)hipMemUnmap(static_cast<char *>(va) + off, sizes[k]);
                (void)hipMemRelease(handles[k]);
                off += sizes[k];
            }
            if (va) (void)hipMemAddressFree(va, bytes);
            (void)hipGetLastError();
            handles.clear();
            sizes.clear();
            va = nullptr;
        }
    };
    std::vector<Mapped> mapped;
};

namespace {

// Makes the context's device current for the duration of an entry point and restores the caller's afterwards:
// allocations, copies and launches of a context must never land on whatever device the calling thread last used.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const dabgpu_ctx *ctx) {
        if (ctx && hipGetDevice(&prev) == hipSuccess && prev != ctx->device) switched = hipSetDevice(ctx->device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

#define HIP_TRY(expr)                               \
    do {                                            \
        hipError_t e_ = (expr);                     \
        if (e_ != hipSuccess) return DABGPU_ERR_HIP; \
    } while (0)

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

struct ScopedTimer {
    dabgpu_ctx *ctx;
    int which;
    hipStream_t s;
    ScopedTimer(dabgpu_ctx *c, int w, hipStream_t st) : ctx(c), which(w), s(st) {
        if (ctx->timing) {
            Timer &t = ctx->timers[which];
            const int i = int(t.recorded % TIMER_RING);
            if (!t.start[i]) { (void)hipEventCreate(&t.start[i]); (void)hipEventCreate(&t.stop[i]); }
            (void)hipEventRecord(t.start[i], s);
        }
    }
    ~ScopedTimer() {
        if (ctx->timing) {
            Timer &t = ctx->timers[which];
            (void)hipEventRecord(t.stop[int(t.recorded % TIMER_RING)], s);
            t.recorded++;
        }
    }
};

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

constexpr size_t PLACE_MIN_BYTES = size_t(256) << 20;      // below this the HBM domains do not matter

// The codeword-per-lane Viterbi pays once a launch has enough codewords to give every SIMD a wave (one wave =
// 64 codewords; its single-wave latency equals the wave-per-codeword kernels' time at ~24k codewords).
constexpr int LANE_MIN_CODEWORDS = 24576;

// returns true and a scratch descriptor when the lane kernels should take this launch
// (`force`: the codeword is too long for the wave-per-codeword kernels' LDS slab -- the lane kernels keep their
// survivors in HBM and take any length)
bool use_lane(dabgpu_ctx *ctx, int nsteps, int n_codewords, hipStream_t s, dabk::LaneScratch *sc, int *rc, bool force = false) {
    *rc = DABGPU_OK;
    if (!dabk::lane_supported(nsteps)) return false;
    if (!force && (ctx->lane_mode == 0 || (ctx->lane_mode < 0 && n_codewords < LANE_MIN_CODEWORDS))) return false;
    const size_t need = dabk::lane_scratch_bytes(nsteps, n_codewords);
    if (ctx->lane_scratch_bytes < need) {
        // growing the buffer must not race with work still using the old one
        if (hipStreamSynchronize(s) != hipSuccess) { *rc = DABGPU_ERR_HIP; return false; }
        if (ctx->d_lane_scratch) (void)hipFree(ctx->d_lane_scratch);
        ctx->d_lane_scratch = nullptr;
        ctx->lane_scratch_bytes = 0;
        if (hipMalloc(&ctx->d_lane_scratch, need) != hipSuccess) {
            ctx->d_lane_scratch = nullptr;
            if (ctx->lane_mode > 0 || force) *rc = DABGPU_ERR_NOMEM;
            return false;                                     // fall back to the wave kernels
        }
        ctx->lane_scratch_bytes = need;
    }
    sc->base = ctx->d_lane_scratch;
    sc->bytes = ctx->lane_scratch_bytes;
    sc->unfused = ctx->lane_unfused;
    return true;
}

hipStream_t pick_stream(dabgpu_ctx *ctx, void *stream) {
    return stream ? reinterpret_cast<hipStream_t>(stream) : ctx->stream;
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
    constexpr int KNOWN_FLAGS = DABGPU_FLAG_VITERBI_WAVE | DABGPU_FLAG_VITERBI_LANE | DABGPU_FLAG_LANE_UNFUSED;
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
    ctx->wave_slots = prop.multiProcessorCount > 0 ? prop.multiProcessorCount * 12 : 3072;   // 3 workgroups x 4 waves per CU
    int rc = DABGPU_OK;
    do {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { rc = DABGPU_ERR_HIP; break; }
        if (dabk::init_viterbi_kernel_attributes() != hipSuccess || dabk::init_lane_kernel_attributes() != hipSuccess) {
            rc = DABGPU_ERR_HIP;
            break;
        }
        std::vector<float2> tw(NB_FFT);
        for (int m = 0; m < NB_FFT; m++) {
            const double a = -2.0 * M_PI * double(m) / double(NB_FFT);
            tw[m] = make_float2(float(std::cos(a)), float(std::sin(a)));
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

static bool release_mapped(dabgpu_ctx *ctx, void *p);

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
        }
    while (!ctx->mapped.empty()) (void)release_mapped(ctx, ctx->mapped.back().va);
    if (ctx->ev_states) (void)hipEventDestroy(ctx->ev_states);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

void *dabgpu_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void dabgpu_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

// releases a range handed out by the placed allocator; false when `p` is not one of them
static bool release_mapped(dabgpu_ctx *ctx, void *p) {
    for (size_t i = 0; i < ctx->mapped.size(); i++) {
        dabgpu_ctx::Mapped &m = ctx->mapped[i];
        if (m.va != p) continue;
        m.release();
        ctx->mapped.erase(ctx->mapped.begin() + long(i));
        return true;
    }
    return false;
}

int dabgpu_free_frame_buffers(dabgpu_ctx *ctx, void *d_iq, int8_t *d_soft) {
    if (!ctx) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    HIP_TRY(hipDeviceSynchronize());
    if (d_iq && !release_mapped(ctx, d_iq)) HIP_TRY(hipFree(d_iq));
    if (d_soft && !release_mapped(ctx, d_soft)) HIP_TRY(hipFree(d_soft));
    return DABGPU_OK;
}

// ---- domain-aware placement (dabgpu_alloc_frame_buffers_placed) ----
namespace {
struct Chunks {
    struct Item {
        hipMemGenericAllocationHandle_t h;
        size_t bytes, off;                                   // offset inside the probe mapping
        bool in_probe = true;                                // still mapped for probing
        char *own = nullptr;                                 // a spacer: its own little address range (the runtime books
                                                             // reserved ADDRESS SPACE against free device memory, so the
                                                             // probe range is never reserved larger than what fills it)
    };
    std::vector<Item> items;
    char *va = nullptr;                                      // probe mapping
    size_t reserved = 0, mapped = 0;                         // mapped: bytes handed out of the range so far
    // every chunk leaves its probe mapping on its own extents.  (The address ranges themselves are all given back
    // together at the very end, after whatever the caller builds from the chunks has its own ranges: this runtime was
    // seen to look a new range's addresses up in a range freed moments before.)
    bool unmap_all() {
        bool ok = true;
        for (auto &x : items) {
            if (!x.in_probe) continue;
            ok = hipMemUnmap(x.own ? x.own : va + x.off, x.bytes) == hipSuccess && ok;
            x.in_probe = false;
        }
        return ok;
    }
    // one more chunk of `bytes` outside the probe range, mapped and accessible; -1 when the device has no more to give
    int add_spacer(size_t bytes, const hipMemAllocationProp &prop, const hipMemAccessDesc &acc) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, bytes, &prop, 0) != hipSuccess) { (void)hipGetLastError(); return -1; }
        void *p = nullptr;
        if (hipMemAddressReserve(&p, bytes, 0, nullptr, 0) != hipSuccess) { (void)hipMemRelease(h); (void)hipGetLastError(); return -1; }
        if (hipMemMap(p, bytes, 0, h, 0) != hipSuccess || hipMemSetAccess(p, bytes, &acc, 1) != hipSuccess) {
            (void)hipMemUnmap(p, bytes);
            (void)hipMemAddressFree(p, bytes);
            (void)hipMemRelease(h);
            (void)hipGetLastError();
            return -1;
        }
        Item it{h, bytes, 0};
        it.own = static_cast<char *>(p);
        items.push_back(it);
        return int(items.size()) - 1;
    }
    ~Chunks() {
        (void)unmap_all();
        for (auto &x : items) if (x.h) (void)hipMemRelease(x.h);                  // the chunks nobody took
        for (auto &x : items) if (x.own) (void)hipMemAddressFree(x.own, x.bytes);
        if (va) (void)hipMemAddressFree(va, reserved);
        (void)hipGetLastError();
    }
    char *at(int i) const { const Item &x = items[size_t(i)]; return x.own ? x.own : va + x.off; }
};

// time of a mover launch that reads [in, in + in_b) and writes [out, out + out_b): min of two after a warm-up
float mover_ms(const void *in, size_t in_b, void *out, size_t out_b, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        if (hipEventRecord(e0, s) != hipSuccess || dabk::launch_placement_probe(in, in_b, out, out_b, s) != hipSuccess ||
            hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)
            return -1.f;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.f;
        if (rep > 0) best = std::min(best, ms);
    }
    return best;
}
bool same_domain_as(const Chunks &c, int ref, const std::vector<int> &idx, hipStream_t s, hipEvent_t e0, hipEvent_t e1,
                    std::vector<int> &same, std::vector<int> &other);
// ... reading chunk a and writing (a sixth of its size of) chunk b
float pair_ms(const Chunks &c, int a, int b, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const size_t in_b = c.items[size_t(a)].bytes;
    return mover_ms(c.at(a), in_b, c.at(b), std::min(c.items[size_t(b)].bytes, in_b / 6), s, e0, e1);
}

// Every chunk's domain (0, 1, 2 in order of first appearance) in two passes: against chunk 0, then against the first
// chunk that differed.  Returns the number of domains seen, or -1.
// Is chunk x in the HBM domain of chunk r?  Reading r and writing x against reading r and writing r itself (the
// same-domain time by construction), the two taken side by side.  "Elsewhere" decides where a buffer goes, and a single
// slow reference measurement is enough to fake it (seen in 3 of 40 fresh processes): it has to repeat twice more.
// Returns 1 same, 0 elsewhere, -1 error.
int in_domain_of(const Chunks &c, int r, int x, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    for (int round = 0; round < 3; round++) {
        const float t = pair_ms(c, r, x, s, e0, e1), self = pair_ms(c, r, r, s, e0, e1);
        if (t < 0.f || self < 0.f) return -1;
        if (t >= 0.97f * self) return 1;
    }
    return 0;
}

// The mover's time moves with the memory side's clocks, which take tens of milliseconds to settle on a device that was
// idle (a fresh process on a fresh box): chunk 0 against itself is timed until two consecutive rounds agree within 1 %
// (at most ~60 ms) before any comparison is made.
void settle_clocks(const Chunks &c, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    float last = 0.f;
    for (int round = 0; round < 40; round++) {
        float t = 0.f;
        for (int k = 0; k < 2; k++) t = pair_ms(c, 0, 0, s, e0, e1);
        if (t <= 0.f) return;
        if (round >= 4 && fabsf(t - last) <= 0.01f * t) return;
        last = t;
    }
}

int classify_chunks(const Chunks &c, hipStream_t s, hipEvent_t e0, hipEvent_t e1, std::vector<int> &dom) {
    const int n = int(c.items.size());
    dom.assign(size_t(n), 0);
    settle_clocks(c, s, e0, e1);
    std::vector<int> rest, a_set, others;
    for (int i = 1; i < n; i++) rest.push_back(i);
    if (!same_domain_as(c, 0, rest, s, e0, e1, a_set, others)) return -1;
    if (others.empty()) return 1;
    const int r2 = others[0];
    std::vector<int> rest2(others.begin() + 1, others.end()), b_set, c_set;
    if (!same_domain_as(c, r2, rest2, s, e0, e1, b_set, c_set)) return -1;
    b_set.push_back(r2);
    for (int i : b_set) dom[size_t(i)] = 1;
    for (int i : c_set) dom[size_t(i)] = 2;
    return c_set.empty() ? 2 : 3;
}

// Chunks whose pairing with `ref` is slow share its domain.  The times of `idx` fall into two groups ~10 % apart; with
// no gap, the pairing of ref with itself (the same-domain time by construction) says which group everything is in.
bool same_domain_as(const Chunks &c, int ref, const std::vector<int> &idx, hipStream_t s, hipEvent_t e0, hipEvent_t e1,
                    std::vector<int> &same, std::vector<int> &other) {
    if (idx.empty()) return true;
    std::vector<float> t(idx.size());
    float lo = 1e30f, hi = 0.f;
    for (size_t k = 0; k < idx.size(); k++) {
        if ((t[k] = pair_ms(c, ref, idx[k], s, e0, e1)) < 0.f) return false;
        lo = std::min(lo, t[k]);
        hi = std::max(hi, t[k]);
    }
    if (hi > 1.04f * lo) {
        const float thr = 0.5f * (lo + hi);
        for (size_t k = 0; k < idx.size(); k++) (t[k] > thr ? same : other).push_back(idx[k]);
        return true;
    }
    // (noise only ever makes a measurement slower: the reference time is the fastest of three, or everything would
    // look "elsewhere" after one slow one)
    float self = 1e30f;
    for (int k = 0; k < 3; k++) {
        const float t1 = pair_ms(c, ref, ref, s, e0, e1);
        if (t1 < 0.f) return false;
        self = std::min(self, t1);
    }
    (lo > 0.97f * self ? same : other) = idx;
    return true;
}
}  // namespace

// (A placement whose own check -- pair_over_same_domain -- says the two buffers do NOT lie apart is reported as such and
// kept.  Throwing it away and placing again was tried: every free-then-reserve of address ranges gives this runtime
// another chance to answer a look-up in the new range with the freed one ("Sub buffer memory end cannot be greater than
// base_end", then a crash inside hipMemMap: once in ~40 allocations on a device another process also uses), and a
// suboptimal placement costs 4 %, a crash the process.  Call these allocators once, at start-up.)
int dabgpu_alloc_frame_buffers_placed(dabgpu_ctx *ctx, int n_frames, size_t frame_stride, void **d_iq, int8_t **d_soft,
                                      dabgpu_placement_report *report) {
    if (!ctx || !d_iq || !d_soft || n_frames <= 0) return DABGPU_ERR_ARG;
    if (frame_stride < size_t(NB_FRAME_SAMPLES) || (frame_stride & 1u)) return DABGPU_ERR_ARG;
    if (size_t(n_frames) > size_t(0x7fffffff) / NB_DATA_SYMBOLS) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    *d_iq = nullptr;
    *d_soft = nullptr;
    dabgpu_placement_report rep;
    std::memset(&rep, 0, sizeof(rep));
    const size_t iq_bytes = size_t(n_frames) * frame_stride * sizeof(float2);
    const size_t soft_bytes = size_t(n_frames) * NB_FRAME_BITS;
    // IQ in 1 GiB chunks; the soft bits in 256 MiB chunks: each is written beside only ~1.7 GiB of samples, so where
    // the IQ buffer has to change domain only one or two of them cannot avoid both of its neighbours' domains
    const size_t CH = size_t(1) << 30, CS = size_t(256) << 20;
    auto plain = [&]() -> int {
        if (hipMalloc(d_iq, iq_bytes) != hipSuccess) return DABGPU_ERR_NOMEM;
        if (hipMalloc(reinterpret_cast<void **>(d_soft), soft_bytes) != hipSuccess) {
            (void)hipFree(*d_iq);
            *d_iq = nullptr;
            return DABGPU_ERR_NOMEM;
        }
        rep.method = 0;
        rep.setup_peak_bytes = iq_bytes + soft_bytes;
        if (report) *report = rep;
        return DABGPU_OK;
    };
    const int n_iq = int((iq_bytes + CH - 1) / CH), n_soft = int((soft_bytes + CS - 1) / CS);
    // (nothing is gained below a few GB, and rounding to whole chunks would cost too much)
    if (iq_bytes < 4 * CH || n_iq > 60 || n_soft > 20) return plain();
    // what may be held during set-up: 1.2 x the buffers; a third of the spare as whole IQ-size chunks, the rest small
    const size_t budget = size_t(1.2 * double(iq_bytes + soft_bytes));
    const size_t need = size_t(n_iq) * CH + size_t(n_soft) * CS;
    const size_t spare = budget > need ? budget - need : 0;
    int n_big = n_iq + int(spare / 3 / CH);
    int n_small = n_soft + int((spare - size_t(n_big - n_iq) * CH) / CS);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    while (n_big > n_iq && size_t(n_big) * CH + size_t(n_small) * CS > free_b - free_b / 16) n_big--;
    while (n_small > n_soft && size_t(n_big) * CH + size_t(n_small) * CS > free_b - free_b / 16) n_small--;
    if (size_t(n_big) * CH + size_t(n_small) * CS > free_b - free_b / 16) return plain();
    int n_total = n_big + n_small;
    if (n_total > 70) return plain();
    const int n_budget = n_total;                            // chunks of the 1.2 x budget; spacers (below) come after them
    hipStream_t s = ctx->stream;
    HIP_TRY(hipStreamSynchronize(s));

    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ctx->device;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    constexpr int MAX_SPACERS = 100;
    Chunks c;
    c.reserved = size_t(n_big) * CH + size_t(n_small) * CS;
    void *va = nullptr;
    if (hipMemAddressReserve(&va, c.reserved, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return plain(); }
    c.va = static_cast<char *>(va);
    bool ok = true;
    for (int i = 0; i < n_total && ok; i++) {
        const size_t bytes = i < n_big ? CH : CS;
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, bytes, &prop, 0) != hipSuccess) { ok = false; break; }
        if (hipMemMap(c.va + c.mapped, bytes, 0, h, 0) != hipSuccess) { (void)hipMemRelease(h); ok = false; break; }
        c.items.push_back(Chunks::Item{h, bytes, c.mapped});
        c.mapped += bytes;
    }
    if (!ok || hipMemSetAccess(c.va, c.mapped, &acc, 1) != hipSuccess) { (void)hipGetLastError(); return plain(); }
    rep.n_chunks = n_total;
    rep.chunk_bytes = CH;
    rep.setup_peak_bytes = c.mapped;

    // ---- which domain is every chunk in? ----
    hipEvent_t e0 = nullptr, e1 = nullptr, ec0 = nullptr, ec1 = nullptr;
    int rc = DABGPU_OK;
    std::vector<int> dom(size_t(n_total), 0);
    int n_dom = 1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&ec0) != hipSuccess ||
        hipEventCreate(&ec1) != hipSuccess || dabk::launch_fill_noise(c.va, c.mapped, s) != hipSuccess)
        rc = DABGPU_ERR_HIP;
    if (!rc) {
        (void)hipEventRecord(ec0, s);
        n_dom = classify_chunks(c, s, e0, e1, dom);
        if (n_dom < 0) { rc = DABGPU_ERR_HIP; n_dom = 1; }
        (void)hipEventRecord(ec1, s);
        (void)hipEventSynchronize(ec1);
        (void)hipEventElapsedTime(&rep.classify_ms, ec0, ec1);
    }
    // ---- the budget may not hold what a clean placement needs: one domain with room for all the samples AND enough
    //      memory elsewhere for the soft bits (a fresh device hands out its memory in address order, and a domain's
    //      address ranges are tens of GB long: the whole budget can lie in one of them).  Then 1 GiB spacers are taken and
    //      classified one by one until it does; they take part in the selection below like any other chunk, and the ones
    //      nobody takes go back at the end of this call (what is held at the peak is reported). ----
    int iq_domain = -1;                                        // the domain that can carry the samples alone, if any
    if (!rc) {
        size_t bytes_in[3] = {0, 0, 0};
        for (int i = 0; i < n_total; i++) bytes_in[dom[size_t(i)]] += c.items[size_t(i)].bytes;
        auto clean = [&]() {
            int best = -1;
            const size_t all = bytes_in[0] + bytes_in[1] + bytes_in[2];
            for (int d = 0; d < 3; d++)
                if (bytes_in[d] >= iq_bytes && all - bytes_in[d] >= soft_bytes && (best < 0 || bytes_in[d] > bytes_in[best])) best = d;
            return best;
        };
        iq_domain = clean();
        int ref[3] = {-1, -1, -1};
        size_t spacer_bytes = 0;
        for (int i = n_total - 1; i >= 0; i--) ref[dom[size_t(i)]] = i;          // a reference chunk per domain seen: its first one
        for (int k = 0; k < MAX_SPACERS && iq_domain < 0; k++) {
            size_t fb = 0, tb = 0;
            if (hipMemGetInfo(&fb, &tb) != hipSuccess || fb < 2 * CH + tb / 16) break;          // leave the device some air
            const int x = c.add_spacer(CH, prop, acc);
            if (x < 0) break;
            n_total++;
            dom.push_back(0);
            spacer_bytes += CH;
            // same domain as a known one?  (reading the reference and writing the spacer is ~10 % slower then)
            int d = -1;
            for (int q = 0; q < 3 && d < 0; q++) {
                if (ref[q] < 0) continue;
                const int same = in_domain_of(c, ref[q], x, s, e0, e1);
                if (same < 0) { rc = DABGPU_ERR_HIP; break; }
                if (same) d = q;
            }
            if (rc) break;
            if (d < 0) {                                           // a domain not seen before
                d = ref[0] < 0 ? 0 : ref[1] < 0 ? 1 : ref[2] < 0 ? 2 : 0;   // (a fourth cannot happen; stay consistent)
                if (ref[d] < 0) { ref[d] = x; n_dom++; }
            }
            dom[size_t(x)] = d;
            bytes_in[d] += CH;
            iq_domain = clean();
        }
        (void)hipGetLastError();
        rep.setup_peak_bytes = c.mapped + spacer_bytes;
        rep.n_chunks = n_total;
    }
    rep.n_domains = n_dom;
    // (small chunks in lower case; spacers, 1 GiB each, follow the budget's chunks)
    for (int i = 0; i < n_total && i < 71; i++) rep.domains[i] = char((i < n_big || i >= n_budget ? 'A' : 'a') + dom[size_t(i)]);

    // ---- IQ: the domain with the most bytes first (1 GiB chunks, then 256 MiB ones); the soft bits, piece by piece:
    //      a chunk whose domain the samples read beside that piece are not in ----
    std::vector<int> iq_sel, soft_sel;
    std::vector<char> used(size_t(n_total), 0);
    size_t iq_mapped = 0, soft_mapped = 0;
    if (!rc) {
        size_t bytes_in[3] = {0, 0, 0};
        for (int i = 0; i < n_total; i++) bytes_in[dom[size_t(i)]] += c.items[size_t(i)].bytes;
        int order[3] = {0, 1, 2};
        std::sort(order, order + 3, [&](int x, int y) {
            if ((x == iq_domain) != (y == iq_domain)) return x == iq_domain;
            return bytes_in[x] != bytes_in[y] ? bytes_in[x] > bytes_in[y] : x < y;
        });
        auto is_big = [&](int i) { return i < n_big || i >= n_budget; };
        // leave the other domains what the soft bits need of them whenever the first domain can carry the samples alone
        for (int k = 0; k < 3 && iq_mapped < iq_bytes; k++)
            for (int pass = 0; pass < 2 && iq_mapped < iq_bytes; pass++)          // pass 0: 1 GiB chunks, pass 1: 256 MiB ones
                for (int i = 0; i < n_total && iq_mapped < iq_bytes; i++) {
                    if (is_big(i) != (pass == 0) || used[size_t(i)] || dom[size_t(i)] != order[k]) continue;
                    if (pass == 0 && iq_bytes - iq_mapped < CH && bytes_in[order[k]] > 0) {
                        // less than a whole big chunk is missing: small chunks of this domain first, if there are enough
                        size_t small_left = 0;
                        for (int j = n_big; j < n_budget; j++) if (!used[size_t(j)] && dom[size_t(j)] == order[k]) small_left += CS;
                        if (small_left >= iq_bytes - iq_mapped) break;
                    }
                    iq_sel.push_back(i);
                    used[size_t(i)] = 1;
                    iq_mapped += c.items[size_t(i)].bytes;
                }
        // domain of the samples at byte offset x of the IQ buffer
        std::vector<size_t> iq_end;
        { size_t e = 0; for (int i : iq_sel) { e += c.items[size_t(i)].bytes; iq_end.push_back(e); } }
        auto iq_bytes_by_domain = [&](double lo, double hi, double w[3]) {
            w[0] = w[1] = w[2] = 0.0;
            size_t begin = 0;
            for (size_t k = 0; k < iq_sel.size(); k++) {
                const double a0 = std::max(lo, double(begin)), a1 = std::min(hi, double(iq_end[k]));
                if (a1 > a0) w[dom[size_t(iq_sel[k])]] += a1 - a0;
                begin = iq_end[k];
            }
        };
        const double iq_per_soft = double(frame_stride * sizeof(float2)) / double(NB_FRAME_BITS);
        const double slack = 1.5 * double(CH);                // samples of the ~1000 frames in flight
        double shared = 0.0;
        while (soft_mapped < soft_bytes) {
            int best = -1;
            double best_cost = 0.0;
            for (int i = 0; i < n_total; i++) {
                if (used[size_t(i)]) continue;
                const size_t sz = c.items[size_t(i)].bytes;
                double w[3];
                iq_bytes_by_domain(double(soft_mapped) * iq_per_soft - slack, double(std::min(soft_bytes, soft_mapped + sz)) * iq_per_soft + slack, w);
                const double tot = w[0] + w[1] + w[2];
                const double cost = tot > 0.0 ? w[dom[size_t(i)]] / tot : 0.0;
                // the least overlap wins; between equals a small chunk (a big one is kept for where it is needed)
                if (best < 0 || cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && sz < c.items[size_t(best)].bytes)) { best = i; best_cost = cost; }
            }
            if (best < 0) { rc = DABGPU_ERR_NOMEM; break; }
            soft_sel.push_back(best);
            used[size_t(best)] = 1;
            const size_t sz = std::min(c.items[size_t(best)].bytes, soft_bytes - soft_mapped);
            shared += best_cost * double(sz);
            soft_mapped += c.items[size_t(best)].bytes;
        }
        rep.conflicts = int(1000.0 * shared / double(soft_bytes) + 0.5);   // per mille of the soft bits written beside same-domain reads
        rep.iq_chunks = int(iq_sel.size());
        rep.soft_chunks = int(soft_sel.size());
        for (size_t k = 0; k < iq_sel.size() && k < 71; k++) rep.iq_map[k] = char((iq_sel[k] < n_big ? 'A' : 'a') + dom[size_t(iq_sel[k])]);
        for (size_t k = 0; k < soft_sel.size() && k < 23; k++)
            rep.soft_map[k] = char((soft_sel[k] < n_big || soft_sel[k] >= n_budget ? 'A' : 'a') + dom[size_t(soft_sel[k])]);
    }
    // ---- final mappings; the chunks nobody took go back ----
    dabgpu_ctx::Mapped m_iq{nullptr, iq_mapped, 0, {}}, m_soft{nullptr, soft_mapped, 0, {}};
    if (!rc) {
        (void)hipStreamSynchronize(s);
        if (!c.unmap_all()) rc = DABGPU_ERR_HIP;
    }
    auto map_over = [&](dabgpu_ctx::Mapped &m, const std::vector<int> &sel) -> int {
        if (hipMemAddressReserve(&m.va, m.bytes, 0, nullptr, 0) != hipSuccess) return DABGPU_ERR_NOMEM;
        size_t off = 0;
        for (size_t k = 0; k < sel.size(); k++) {
            Chunks::Item &it = c.items[size_t(sel[k])];
            if (hipMemMap(static_cast<char *>(m.va) + off, it.bytes, 0, it.h, 0) != hipSuccess) return DABGPU_ERR_HIP;
            m.handles.push_back(it.h);
            m.sizes.push_back(it.bytes);
            it.h = nullptr;                                    // owned by the mapping from here on
            off += it.bytes;
            m.chunk = off;                                     // bytes mapped so far
        }
        return hipMemSetAccess(m.va, m.bytes, &acc, 1) == hipSuccess ? DABGPU_OK : DABGPU_ERR_HIP;
    };
    if (!rc) rc = map_over(m_iq, iq_sel);
    if (!rc) rc = map_over(m_soft, soft_sel);
    if (rc) {
        for (dabgpu_ctx::Mapped *m : {&m_iq, &m_soft}) m->release();
    } else {
        ctx->mapped.push_back(m_iq);
        ctx->mapped.push_back(m_soft);
        *d_iq = m_iq.va;
        *d_soft = static_cast<int8_t *>(m_soft.va);
        rep.method = 1;
        // one timed front-end launch on the pair (the IQ buffer still holds the classification's noise)
        void *d_fo = nullptr, *d_cyc = nullptr;
        const size_t fo_bytes = sizeof(float) * size_t(n_frames), cyc_bytes = size_t(n_frames) * NB_FRAME_SYMBOLS * sizeof(float2);
        if (hipMalloc(&d_fo, fo_bytes) == hipSuccess && hipMalloc(&d_cyc, cyc_bytes) == hipSuccess &&
            hipMemsetAsync(d_fo, 0, fo_bytes, s) == hipSuccess) {
            const bool was_timing = ctx->timing;
            ctx->timing = false;
            const unsigned long long *keep = ctx->d_keep;
            ctx->d_keep = nullptr;
            int prc = DABGPU_OK;
            for (int r = 0; r < 3 && !prc; r++) {
                if (r == 1) (void)hipEventRecord(e0, s);
                prc = dabgpu_ofdm_demod_frames_dev(ctx, static_cast<char *>(*d_iq) + size_t(NB_NULL_PERIOD) * sizeof(float2), frame_stride,
                                                   n_frames, static_cast<const float *>(d_fo), *d_soft, d_cyc, nullptr, s);
            }
            float ms = 0.f;
            if (!prc && hipEventRecord(e1, s) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
                hipEventElapsedTime(&ms, e0, e1) == hipSuccess)
                rep.front_end_ms = 0.5f * ms;
            ctx->timing = was_timing;
            ctx->d_keep = keep;
        }
        (void)hipStreamSynchronize(s);
        if (d_fo) (void)hipFree(d_fo);
        if (d_cyc) (void)hipFree(d_cyc);
        (void)hipGetLastError();
        // check of the result, independent of the classification: the mover reads the first GiB of the samples and
        // writes (a) the start of the soft-bit buffer, (b) into the samples' own buffer two GiB further on
        if (iq_mapped >= 4 * CH) {
            const size_t out_b = std::min(soft_mapped, CH / 6);
            char *iq0 = static_cast<char *>(*d_iq);
            float ta = 1e30f, tb = 1e30f;
            for (int r = 0; r < 3; r++) {                          // alternated: drift hits both alike
                const float a = mover_ms(iq0, CH, *d_soft, out_b, s, e0, e1);
                const float b = mover_ms(iq0, CH, iq0 + 2 * CH, out_b, s, e0, e1);
                if (a > 0.f) ta = std::min(ta, a);
                if (b > 0.f) tb = std::min(tb, b);
            }
            if (ta < 1e29f && tb < 1e29f) rep.pair_over_same_domain = ta / tb;
            // (the mover left noise-like words at the start of the soft-bit buffer and in the samples: both are the
            // caller's to fill; the classification's noise is what was there before)
        }
    }
    for (hipEvent_t e : {e0, e1, ec0, ec1}) if (e) (void)hipEventDestroy(e);
    if (report) *report = rep;
    return rc;                                               // (~Chunks releases the chunks nobody took)
}

// A buffer that a launch WRITES while it reads [ref, ref + ref_bytes), both walked front to back in step: physical
// chunks (1.2 x bytes at most), their domains among themselves, the domain of every GiB of `ref` against one
// representative chunk per domain, then every piece of the new buffer over a chunk whose domain the part of `ref` read
// beside it is not in.  probe_ms: [0] mover time on the result (first GiB of ref -> start of the buffer), [1] per mille of
// the buffer left beside same-domain reads, [2] ms spent classifying.
static int alloc_apart(dabgpu_ctx *ctx, size_t bytes, const void *ref, size_t ref_bytes, void **out, float *probe_ms) {
    if (probe_ms) probe_ms[0] = probe_ms[1] = probe_ms[2] = 0.f;
    *out = nullptr;
    auto plain = [&]() { return hipMalloc(out, std::max<size_t>(bytes, 16)) == hipSuccess ? DABGPU_OK : DABGPU_ERR_NOMEM; };
    if (!ref || ref_bytes < PLACE_MIN_BYTES || bytes < PLACE_MIN_BYTES) return plain();
    const size_t CH = bytes >= (size_t(8) << 30) ? size_t(1) << 30 : size_t(256) << 20;
    const int n_need = int((bytes + CH - 1) / CH);
    int n_total = std::max(n_need + 1, int(1.2 * double(bytes) / double(CH)));
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    while (n_total > n_need && size_t(n_total) * CH > free_b - free_b / 16) n_total--;
    if (n_total > 70 || size_t(n_total) * CH > free_b - free_b / 16) return plain();
    hipStream_t s = ctx->stream;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ctx->device;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    constexpr int MAX_SPACERS = 100;                        // (see dabgpu_alloc_frame_buffers_placed: the one-domain case)
    Chunks c;
    c.reserved = size_t(n_total) * CH;
    void *va = nullptr;
    if (hipMemAddressReserve(&va, c.reserved, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return plain(); }
    c.va = static_cast<char *>(va);
    bool ok = true;
    for (int i = 0; i < n_total && ok; i++) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, CH, &prop, 0) != hipSuccess) { ok = false; break; }
        if (hipMemMap(c.va + c.mapped, CH, 0, h, 0) != hipSuccess) { (void)hipMemRelease(h); ok = false; break; }
        c.items.push_back(Chunks::Item{h, CH, c.mapped});
        c.mapped += CH;
    }
    if (!ok || hipMemSetAccess(c.va, c.mapped, &acc, 1) != hipSuccess) { (void)hipGetLastError(); return plain(); }
    hipEvent_t e0 = nullptr, e1 = nullptr, ec0 = nullptr, ec1 = nullptr;
    int rc = DABGPU_OK;
    std::vector<int> dom, sel;
    double shared = 0.0;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&ec0) != hipSuccess ||
        hipEventCreate(&ec1) != hipSuccess || dabk::launch_fill_noise(c.va, c.mapped, s) != hipSuccess)
        rc = DABGPU_ERR_HIP;
    if (!rc) {
        (void)hipEventRecord(ec0, s);
        int n_dom = classify_chunks(c, s, e0, e1, dom);
        if (n_dom < 0) { rc = DABGPU_ERR_HIP; n_dom = 1; }
        // the domain of every piece of ref: the representative it is slow against (none: a domain no chunk is in)
        const size_t RP = size_t(1) << 30;
        const int n_ref = int((ref_bytes + RP - 1) / RP);
        std::vector<int> ref_dom(size_t(n_ref), -1), repr(3, -1);
        for (int i = n_total - 1; i >= 0 && !rc; i--) repr[size_t(dom[size_t(i)])] = i;
        for (int r = 0; r < n_ref && !rc; r++) {
            const char *in = static_cast<const char *>(ref) + size_t(r) * RP;
            const size_t in_b = std::min(RP, ref_bytes - size_t(r) * RP);
            if (in_b < (size_t(64) << 20)) { ref_dom[size_t(r)] = r > 0 ? ref_dom[size_t(r) - 1] : -1; continue; }
            float t[3] = {0.f, 0.f, 0.f}, lo = 1e30f, hi = 0.f;
            for (int d = 0; d < n_dom; d++) {
                t[d] = mover_ms(in, in_b, c.at(repr[size_t(d)]), std::min(CH, in_b / 6), s, e0, e1);
                if (t[d] < 0.f) { rc = DABGPU_ERR_HIP; break; }
                lo = std::min(lo, t[d]);
                hi = std::max(hi, t[d]);
            }
            if (rc) break;
            if (n_dom > 1 && hi > 1.04f * lo) {
                for (int d = 0; d < n_dom; d++) if (t[d] == hi) ref_dom[size_t(r)] = d;
            } else if (n_dom == 1) {
                // one domain among the chunks: is ref in it?  chunk 0 against itself is the same-domain time
                const float self = pair_ms(c, 0, 0, s, e0, e1) * float(double(in_b) / double(CH));
                if (t[0] > 0.97f * self) ref_dom[size_t(r)] = 0;
            }
        }
        // every piece of the buffer: the chunk least beside its own domain.  Where the 1.2 x budget leaves more than a
        // twentieth of the buffer beside same-domain reads (all of it, when budget and reference lie in one domain),
        // 1 GiB-class spacers are taken four at a time, placed among the known domains (or given a new one), and the
        // selection is repeated; the ones nobody takes go back when this call returns.
        const double ratio = double(ref_bytes) / double(bytes);
        int spacers = 0;
      select_again:
        std::vector<char> used(size_t(n_total), 0);
        sel.clear();
        shared = 0.0;
        for (int m = 0; m < n_need && !rc; m++) {
            const double lo_b = double(m) * double(CH) * ratio - 1.5 * double(RP), hi_b = double(m + 1) * double(CH) * ratio + 1.5 * double(RP);
            double w[3] = {0.0, 0.0, 0.0}, tot = 0.0;
            for (int r = 0; r < n_ref; r++) {
                const double a0 = std::max(lo_b, double(r) * double(RP)), a1 = std::min(hi_b, std::min(double(ref_bytes), double(r + 1) * double(RP)));
                if (a1 <= a0) continue;
                tot += a1 - a0;
                if (ref_dom[size_t(r)] >= 0) w[ref_dom[size_t(r)]] += a1 - a0;
            }
            int best = -1;
            for (int i = 0; i < n_total; i++) {
                if (used[size_t(i)]) continue;
                if (best < 0 || w[dom[size_t(i)]] < w[dom[size_t(best)]] - 1e-9) best = i;
            }
            sel.push_back(best);
            used[size_t(best)] = 1;
            if (tot > 0.0) shared += w[dom[size_t(best)]] / tot;
        }
        if (!rc && shared > 0.05 * double(n_need) && spacers < MAX_SPACERS) {
            int added = 0;
            for (int k = 0; k < 4 && spacers < MAX_SPACERS; k++) {
                size_t fb = 0, tb = 0;
                if (hipMemGetInfo(&fb, &tb) != hipSuccess || fb < 2 * CH + tb / 16) break;
                const int x = c.add_spacer(CH, prop, acc);
                if (x < 0) break;
                n_total++;
                dom.push_back(0);
                spacers++;
                added++;
                int d = -1;
                for (int q = 0; q < n_dom && d < 0; q++) {
                    const int same = in_domain_of(c, repr[size_t(q)], x, s, e0, e1);
                    if (same < 0) { rc = DABGPU_ERR_HIP; break; }
                    if (same) d = q;
                }
                if (rc) break;
                if (d < 0) {
                    if (n_dom < 3) { d = n_dom++; repr[size_t(d)] = x; }
                    else d = 0;
                    // a domain no chunk was in before: are pieces of the reference in it?  (those that matched none)
                    for (int r = 0; r < n_ref && d == n_dom - 1 && repr[size_t(d)] == x; r++) {
                        if (ref_dom[size_t(r)] >= 0) continue;
                        const size_t in_b = std::min(RP, ref_bytes - size_t(r) * RP);
                        if (in_b < (size_t(64) << 20)) { ref_dom[size_t(r)] = r > 0 ? ref_dom[size_t(r) - 1] : -1; continue; }
                        const char *in = static_cast<const char *>(ref) + size_t(r) * RP;
                        const float tn = mover_ms(in, in_b, c.at(x), std::min(CH, in_b / 6), s, e0, e1);
                        const float to = mover_ms(in, in_b, c.at(repr[0]), std::min(CH, in_b / 6), s, e0, e1);   // (not its domain: it matched none)
                        if (tn < 0.f || to < 0.f) { rc = DABGPU_ERR_HIP; break; }
                        if (tn > 1.04f * to) ref_dom[size_t(r)] = d;
                    }
                }
                dom[size_t(x)] = d;
            }
            (void)hipGetLastError();
            if (!rc && added > 0) goto select_again;
        }
        (void)hipEventRecord(ec1, s);
        (void)hipEventSynchronize(ec1);
        float cms = 0.f;
        (void)hipEventElapsedTime(&cms, ec0, ec1);
        if (probe_ms) { probe_ms[1] = float(1000.0 * shared / double(n_need)); probe_ms[2] = cms; }
    }
    dabgpu_ctx::Mapped m{nullptr, CH * size_t(n_need), 0, {}};
    if (!rc) {
        (void)hipStreamSynchronize(s);
        if (!c.unmap_all()) rc = DABGPU_ERR_HIP;
    }
    if (!rc && hipMemAddressReserve(&m.va, m.bytes, 0, nullptr, 0) != hipSuccess) rc = DABGPU_ERR_NOMEM;
    for (size_t k = 0; k < sel.size() && !rc; k++) {
        Chunks::Item &it = c.items[size_t(sel[k])];
        if (hipMemMap(static_cast<char *>(m.va) + CH * k, CH, 0, it.h, 0) != hipSuccess) { rc = DABGPU_ERR_HIP; break; }
        m.handles.push_back(it.h);
        m.sizes.push_back(CH);
        it.h = nullptr;
        m.chunk = CH * (k + 1);
    }
    if (!rc && hipMemSetAccess(m.va, m.bytes, &acc, 1) != hipSuccess) rc = DABGPU_ERR_HIP;
    if (rc) {
        m.release();
    } else {
        ctx->mapped.push_back(m);
        *out = m.va;
        if (probe_ms) {
            const size_t in_b = std::min(ref_bytes, size_t(1) << 30);
            probe_ms[0] = mover_ms(ref, in_b, m.va, std::min(m.bytes, in_b / 6), s, e0, e1);
        }
    }
    for (hipEvent_t e : {e0, e1, ec0, ec1}) if (e) (void)hipEventDestroy(e);
    return rc;
}

int dabgpu_alloc_frame_buffers(dabgpu_ctx *ctx, int n_frames, size_t frame_stride, int candidates, void **d_iq,
                               int8_t **d_soft, float *probe_ms, int *kept) {
    if (!ctx || !d_iq || !d_soft || n_frames <= 0 || candidates < 1 || candidates > 8) return DABGPU_ERR_ARG;
    if (frame_stride < size_t(NB_FRAME_SAMPLES) || (frame_stride & 1u)) return DABGPU_ERR_ARG;
    if (size_t(n_frames) > size_t(0x7fffffff) / NB_DATA_SYMBOLS) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    *d_iq = nullptr;
    *d_soft = nullptr;
    const size_t iq_bytes = size_t(n_frames) * frame_stride * sizeof(float2);
    const size_t soft_bytes = size_t(n_frames) * NB_FRAME_BITS;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    int K = candidates;
    while (K > 1 && double(K) * double(iq_bytes + soft_bytes) > 0.6 * double(free_b)) K--;
    hipStream_t s = ctx->stream;
    std::vector<void *> iq(size_t(K), nullptr), soft(size_t(K), nullptr);
    void *d_fo = nullptr, *d_cyc = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<float> table(size_t(K) * K, 0.f);
    int rc = DABGPU_OK, bi = 0, bj = 0;
    do {
        for (int k = 0; k < K && !rc; k++)
            if (hipMalloc(&iq[k], iq_bytes) != hipSuccess || hipMalloc(&soft[k], soft_bytes) != hipSuccess) rc = DABGPU_ERR_NOMEM;
        if (rc) break;
        if (K == 1) break;
        const size_t fo_bytes = sizeof(float) * size_t(n_frames), cyc_bytes = size_t(n_frames) * NB_FRAME_SYMBOLS * sizeof(float2);
        if (hipMalloc(&d_fo, fo_bytes) != hipSuccess || hipMalloc(&d_cyc, cyc_bytes) != hipSuccess) { rc = DABGPU_ERR_NOMEM; break; }
        if (hipMemsetAsync(d_fo, 0, fo_bytes, s) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
            rc = DABGPU_ERR_HIP;
            break;
        }
        for (int k = 0; k < K && !rc; k++)
            if (dabk::launch_fill_noise(iq[k], iq_bytes, s) != hipSuccess) rc = DABGPU_ERR_HIP;
        if (rc) break;
        const bool was_timing = ctx->timing;
        ctx->timing = false;                                   // the probe launches are not the caller's measurements
        const unsigned long long *keep = ctx->d_keep;
        ctx->d_keep = nullptr;                                 // whole frames, whatever selection is active
        float best = -1.f;
        for (int i = 0; i < K && !rc; i++)
            for (int j = 0; j < K && !rc; j++) {
                for (int rep = 0; rep < 3 && !rc; rep++) {
                    if (rep == 1 && hipEventRecord(e0, s) != hipSuccess) rc = DABGPU_ERR_HIP;
                    if (!rc)
                        rc = dabgpu_ofdm_demod_frames_dev(ctx, static_cast<char *>(iq[i]) + size_t(NB_NULL_PERIOD) * sizeof(float2),
                                                          frame_stride, n_frames, static_cast<const float *>(d_fo),
                                                          static_cast<int8_t *>(soft[j]), d_cyc, nullptr, s);
                }
                float ms = 0.f;
                if (!rc && (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                            hipEventElapsedTime(&ms, e0, e1) != hipSuccess))
                    rc = DABGPU_ERR_HIP;
                table[size_t(i) * K + j] = ms * 0.5f;
                if (!rc && (best < 0.f || ms < best)) { best = ms; bi = i; bj = j; }
            }
        ctx->timing = was_timing;
        ctx->d_keep = keep;
    } while (0);
    (void)hipStreamSynchronize(s);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d_fo) (void)hipFree(d_fo);
    if (d_cyc) (void)hipFree(d_cyc);
    for (int k = 0; k < K; k++) {
        if (iq[k] && (rc || k != bi)) (void)hipFree(iq[k]);
        if (soft[k] && (rc || k != bj)) (void)hipFree(soft[k]);
    }
    if (rc) return rc;
    *d_iq = iq[bi];
    *d_soft = static_cast<int8_t *>(soft[bj]);
    if (probe_ms)
        for (int i = 0; i < candidates; i++)
            for (int j = 0; j < candidates; j++)
                probe_ms[size_t(i) * candidates + j] = (i < K && j < K) ? table[size_t(i) * K + j] : 0.f;
    if (kept) { kept[0] = bi; kept[1] = bj; }
    return DABGPU_OK;
}

int dabgpu_device_alloc_apart(dabgpu_ctx *ctx, size_t bytes, const void *d_other, size_t other_bytes, void **d_out,
                              float *probe_ms) {
    if (!ctx || !d_out || bytes == 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return alloc_apart(ctx, bytes, d_other, other_bytes, d_out, probe_ms);
}

int dabgpu_device_free(dabgpu_ctx *ctx, void *d_ptr) {
    if (!ctx) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    HIP_TRY(hipDeviceSynchronize());
    if (d_ptr && !release_mapped(ctx, d_ptr)) HIP_TRY(hipFree(d_ptr));
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
    if (!ctx || !mean_ms || which < 0 || which > 3) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    Timer &t = ctx->timers[which];
    const int n = int(std::min<long>(t.recorded, TIMER_RING));
    if (n == 0) return DABGPU_ERR_ARG;
    double sum = 0.0;
    for (int k = 0; k < n; k++) {
        const int i = int((t.recorded - 1 - k) % TIMER_RING);
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(t.stop[i]));
        HIP_TRY(hipEventElapsedTime(&ms, t.start[i], t.stop[i]));
        sum += double(ms);
    }
    *mean_ms = float(sum / n);
    if (launches) *launches = n;
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

int dabgpu_ofdm_set_soft_selection(dabgpu_ctx *ctx, const dabgpu_bit_range *ranges, int n_ranges) {
    if (!ctx || n_ranges < 0 || (n_ranges > 0 && !ranges)) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (n_ranges == 0) { ctx->d_keep = nullptr; ctx->keep_ranges.clear(); return DABGPU_OK; }
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
static int note_state_use(dabgpu_ctx *ctx, hipStream_t s) {
    if (!ctx->ev_states && hipEventCreateWithFlags(&ctx->ev_states, hipEventDisableTiming) != hipSuccess) return DABGPU_ERR_HIP;
    HIP_TRY(hipEventRecord(ctx->ev_states, s));
    ctx->ev_states_pending = true;
    return DABGPU_OK;
}
static int wait_state_use(dabgpu_ctx *ctx) {
    if (ctx->ev_states_pending) {
        HIP_TRY(hipEventSynchronize(ctx->ev_states));
        ctx->ev_states_pending = false;
    }
    return DABGPU_OK;
}

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
    return DABGPU_OK;
}

static void stats_of(const dabk::StreamState &st, dabgpu_stats *out) {
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
}

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
                                       fine_freq_update_beta, ctx->thr_null_start, ctx->signal_beta, dd ? 1 : 0, s));
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
    cfg->decision_directed = 1;
    cfg->auto_acquire = 0;
}

static int track_cfg(const dabgpu_track_cfg *cfg, dabgpu_track_cfg &c) {
    if (cfg) c = *cfg; else dabgpu_track_default_cfg(&c);
    auto unit = [](float v) { return v >= 0.f && v <= 1.f; };
    if (!unit(c.fine_freq_update_beta) || !unit(c.signal_update_beta) || !unit(c.thr_null_start) || !unit(c.drift_beta) ||
        !unit(c.coarse_freq_slow_beta) || !peak_rule_ok(c.impulse_peak_distance_probability, c.first_path_rel) ||
        !(c.min_peak_to_mean >= 0.f) || c.timing_margin < 0 || c.timing_margin > NB_CP || c.max_coarse_carriers < 0 ||
        c.max_coarse_carriers > 1023)
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

// the three launches of a tracked call on `s`: PRS synchronisation at the predicted positions, demodulation of the frames
// where they lie, state update
static int tracked_launches(dabgpu_ctx *ctx, dabk::StreamState *states, const void *d_iq, size_t stream_stride, int n_streams,
                            int64_t n_samples, int max_frames, int64_t advance, const dabgpu_track_cfg &c, int fixed_start,
                            int acquiring, int8_t *d_soft, void *d_cyc, void *d_dqpsk, dabgpu_acquired_frame *d_frames,
                            dabgpu_sync_result *d_sync, int32_t *d_counts, hipStream_t s, const dabk::AcquireArgs *auto_acq = nullptr,
                            void *d_dd4 = nullptr) {
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
    t.acquiring = acquiring;
    t.coarse_slow_beta = c.coarse_freq_slow_beta;
    t.out = reinterpret_cast<dabk::AcquiredFrame *>(d_frames);
    t.sync_out = reinterpret_cast<dabk::SyncResult *>(d_sync);
    HIP_TRY(dabk::launch_track_sync(stab, t, s));
    // streams that are not tracking: acquired here (their rows of d_frames / d_counts; the pass above left them empty)
    if (auto_acq) HIP_TRY(dabk::launch_acquire(stab, *auto_acq, s));
    dabk::OfdmTables tab{ctx->d_twiddle, ctx->d_bin_of_n, ctx->d_n_of_vj};
    dabk::OfdmArgs a{};
    a.iq = static_cast<const float2 *>(d_iq);
    a.frame_stride = stream_stride;
    a.n_frames = n_streams * max_frames;
    a.soft = d_soft;
    a.cyc = static_cast<float2 *>(d_cyc);
    a.dd4 = static_cast<float2 *>(d_dd4);
    a.dqpsk = static_cast<float2 *>(d_dqpsk);
    a.acq = reinterpret_cast<const dabk::AcquiredFrame *>(d_frames);
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
    // ... and their tracking starts from what the acquisition found (marked 2; the update launch makes it 1)
    if (auto_acq)
        HIP_TRY(dabk::launch_track_start(states, t.out, d_counts, n_streams, max_frames, advance, 1, s));
    HIP_TRY(dabk::launch_track_update(u, s));
    return note_state_use(ctx, s);
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
    // (no correlation output asked for: by default a decision-directed loop, the cyclic prefixes are not read -- see
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
    return tracked_launches(ctx, ctx->d_states, d_iq, stream_stride, n_streams, n_samples, max_frames, advance, c, 0, 0, d_soft,
                            d_cyc, d_dqpsk, d_frames, nullptr, d_counts, s, (c.auto_acquire && n_samples >= 64) ? &acq : nullptr, d_dd);
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
    void *d_iq, *d_res, *d_cyc, *d_dq = nullptr;
    if ((rc = stage(ctx, 0, nb_iq, &d_iq))) return rc;
    if ((rc = stage(ctx, 1, nb_res, &d_res))) return rc;
    if ((rc = stage(ctx, 6, NB_FRAME_SYMBOLS * sizeof(float2), &d_cyc))) return rc;
    if (dqpsk && (rc = stage(ctx, 4, nb_dq, &d_dq))) return rc;
    if (ctx->h_bounce_bytes < nb_res) {
        if (ctx->h_bounce) (void)hipHostFree(ctx->h_bounce);
        ctx->h_bounce = nullptr;
        ctx->h_bounce_bytes = 0;
        if (hipHostMalloc(&ctx->h_bounce, nb_res, hipHostMallocDefault) != hipSuccess) return DABGPU_ERR_NOMEM;
        ctx->h_bounce_bytes = nb_res;
    }
    if ((rc = wait_state_use(ctx))) return rc;
    hipStream_t s = ctx->stream;
    char *res = static_cast<char *>(d_res);
    HIP_TRY(hipMemcpyAsync(d_iq, iq, nb_iq, hipMemcpyHostToDevice, s));                    // one upload
    dabk::StreamState *st = ctx->d_states + stream_index;
    rc = tracked_launches(ctx, st, d_iq, nb_iq / sizeof(float2), 1, int64_t(nb_iq / sizeof(float2)), 1, 0, c, 1, acquiring ? 1 : 0,
                          reinterpret_cast<int8_t *>(res), d_cyc, d_dq, reinterpret_cast<dabgpu_acquired_frame *>(res + off_fr),
                          reinterpret_cast<dabgpu_sync_result *>(res + off_sy), nullptr, s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(res + off_st, st, sizeof(dabk::StreamState), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(ctx->h_bounce, d_res, nb_res, hipMemcpyDeviceToHost, s));        // one download
    if (dqpsk) HIP_TRY(hipMemcpyAsync(dqpsk, d_dq, nb_dq, hipMemcpyDeviceToHost, s));       // (+ the constellation, when asked for)
    HIP_TRY(hipStreamSynchronize(s));                                                    // one synchronisation
    ctx->ev_states_pending = false;
    const char *hb = static_cast<const char *>(ctx->h_bounce);
    std::memcpy(soft, hb, NB_FRAME_BITS);
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

// ---------------------------------------------------------------------------- FIC
int dabgpu_fic_decode_dev(dabgpu_ctx *ctx, const int8_t *d_soft, size_t soft_stride, int n_frames,
                          uint8_t *d_fib, uint8_t *d_crc_ok, void *stream) {
    if (!ctx || !d_soft || !d_fib || !d_crc_ok || n_frames < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (n_frames > 1 && soft_stride < size_t(NB_FIC_BITS)) return DABGPU_ERR_ARG;
    if (n_frames == 0) return DABGPU_OK;
    hipStream_t s = pick_stream(ctx, stream);
    ScopedTimer tm(ctx, 1, s);
    dabk::LaneScratch lsc{};
    int lrc;
    if (use_lane(ctx, ctx->fic.prof.nsteps, n_frames * NB_FIC_GROUPS, s, &lsc, &lrc)) {
        HIP_TRY(dabk::launch_fic_decode_lane(ctx->fic.tables(true), ctx->fic.lane_tables(), d_soft, soft_stride, n_frames,
                                             lsc, d_fib, d_crc_ok, s));
        return DABGPU_OK;
    }
    if (lrc) return lrc;
    HIP_TRY(dabk::launch_fic_decode(ctx->fic.tables(true), d_soft, soft_stride, n_frames, d_fib, d_crc_ok, s));
    return DABGPU_OK;
}

int dabgpu_fic_decode(dabgpu_ctx *ctx, const int8_t *soft, size_t soft_stride, int n_frames, uint8_t *fib,
                      uint8_t *crc_ok) {
    if (!ctx || !soft || !fib || !crc_ok || n_frames < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (n_frames == 0) return DABGPU_OK;
    void *d_soft, *d_fib, *d_ok;
    int rc;
    const size_t nb_soft = size_t(n_frames - 1) * soft_stride + NB_FIC_BITS;
    if ((rc = stage(ctx, 1, nb_soft, &d_soft))) return rc;
    if ((rc = stage(ctx, 3, size_t(n_frames) * NB_FIBS * 32, &d_fib))) return rc;
    if ((rc = stage(ctx, 2, size_t(n_frames) * NB_FIBS, &d_ok))) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(hipMemcpyAsync(d_soft, soft, nb_soft, hipMemcpyHostToDevice, s));
    rc = dabgpu_fic_decode_dev(ctx, static_cast<const int8_t *>(d_soft), soft_stride, n_frames,
                               static_cast<uint8_t *>(d_fib), static_cast<uint8_t *>(d_ok), s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(fib, d_fib, size_t(n_frames) * NB_FIBS * 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(crc_ok, d_ok, size_t(n_frames) * NB_FIBS, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return DABGPU_OK;
}

// ---------------------------------------------------------------------------- MSC
static int subchannel_profile(const dabgpu_subchannel *sc, dab::PunctureProfile &prof) {
    if (!sc) return DABGPU_ERR_ARG;
    int size_cu = 0;
    if (sc->is_uep) {
        if (!make_uep_profile(uep_table_index(sc->bitrate_kbps, sc->protection_level), prof, size_cu)) return DABGPU_ERR_PROFILE;
    } else if (!make_eep_profile(sc->eep_type, sc->protection_level, sc->bitrate_kbps, prof, size_cu)) {
        return DABGPU_ERR_PROFILE;
    }
    if (size_cu != sc->length) return DABGPU_ERR_PROFILE;
    if (sc->start_address < 0 || sc->start_address + sc->length > 864) return DABGPU_ERR_ARG;
    return DABGPU_OK;
}

int dabgpu_soft_selection(const dabgpu_subchannel *subchannels, int n_subchannels, int with_fic,
                          dabgpu_bit_range *out, int max_out) {
    if (n_subchannels < 0 || (n_subchannels > 0 && !subchannels) || max_out < 0 || (max_out > 0 && !out)) return DABGPU_ERR_ARG;
    int n = 0;
    auto put = [&](int first, int count) {
        if (n < max_out) { out[n].first = first; out[n].count = count; }
        n++;
    };
    if (with_fic) put(0, NB_FIC_BITS);
    for (int i = 0; i < n_subchannels; i++) {
        dab::PunctureProfile prof;
        const int rc = subchannel_profile(&subchannels[i], prof);
        if (rc) return rc;
        for (int c = 0; c < NB_CIFS; c++)
            put(NB_FIC_BITS + c * NB_CIF_BITS + subchannels[i].start_address * 64, subchannels[i].length * 64);
    }
    return n;
}

int dabgpu_uep_subchannel(int table_index, int start_address, dabgpu_subchannel *out) {
    if (!out) return DABGPU_ERR_ARG;
    if (table_index < 0 || table_index >= 64) return DABGPU_ERR_PROFILE;
    const UepProfileRow &r = UEP_TABLE[table_index];
    if (start_address < 0 || start_address + r.size > 864) return DABGPU_ERR_ARG;
    out->start_address = start_address;
    out->length = r.size;
    out->is_uep = 1;
    out->eep_type = 0;
    out->protection_level = r.level;
    out->bitrate_kbps = r.bitrate;
    return DABGPU_OK;
}

int dabgpu_subchannel_bytes(const dabgpu_subchannel *sc) {
    dab::PunctureProfile prof;
    int rc = subchannel_profile(sc, prof);
    if (rc) return rc;
    return (prof.nsteps - 6) / 8;
}

int dabgpu_msc_decode_dev(dabgpu_ctx *ctx, const dabgpu_subchannel *sc, const int8_t *d_soft, size_t soft_stride,
                          int n_streams, int frames_per_stream, const int8_t *d_history_in,
                          int8_t *d_history_out, uint8_t *d_out, void *stream) {
    if (!ctx || !d_soft || !d_out || n_streams < 0 || frames_per_stream < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (d_history_in && d_history_in == d_history_out) return DABGPU_ERR_ARG;
    if (soft_stride < size_t(NB_FRAME_BITS) && size_t(n_streams) * frames_per_stream > 1) return DABGPU_ERR_ARG;
    dab::PunctureProfile prof;
    int rc = subchannel_profile(sc, prof);
    if (rc) return rc;
    if (n_streams == 0 || frames_per_stream == 0) return DABGPU_OK;
    const bool too_long = !dabk::viterbi_fits(prof.nsteps);   // above ~800 kbit/s: only the lane kernels hold it
    if (too_long && !dabk::lane_supported(prof.nsteps)) return DABGPU_ERR_CAPACITY;
    DeviceCode *dc = nullptr;
    if ((rc = get_code(ctx, std::move(prof), &dc))) return rc;
    hipStream_t s = pick_stream(ctx, stream);
    dabk::MscArgs a{};
    a.soft = d_soft;
    a.soft_stride = soft_stride;
    a.n_streams = n_streams;
    a.frames_per_stream = frames_per_stream;
    a.start_bit = sc->start_address * CU_BITS;
    a.nbits = sc->length * CU_BITS;
    a.hist_in = d_history_in;
    a.hist_out = d_history_out;
    a.out = d_out;
    ScopedTimer tm(ctx, 2, s);
    dabk::LaneScratch lsc{};
    int lrc;
    if (use_lane(ctx, dc->prof.nsteps, n_streams * frames_per_stream * NB_CIFS, s, &lsc, &lrc, too_long)) {
        HIP_TRY(dabk::launch_msc_decode_lane(dc->tables(true), dc->lane_tables(), a, lsc, s));
        HIP_TRY(dabk::launch_msc_history(a, s));
        return DABGPU_OK;
    }
    if (lrc) return lrc;
    if (too_long) return DABGPU_ERR_CAPACITY;
    HIP_TRY(dabk::launch_msc_decode(dc->tables(true), a, s));
    return DABGPU_OK;
}

int dabgpu_msc_decode(dabgpu_ctx *ctx, const dabgpu_subchannel *sc, const int8_t *soft, size_t soft_stride,
                      int n_streams, int frames_per_stream, const int8_t *history_in, int8_t *history_out,
                      uint8_t *out) {
    if (!ctx || !soft || !out || n_streams < 0 || frames_per_stream < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    const int nbytes = dabgpu_subchannel_bytes(sc);
    if (nbytes < 0) return nbytes;
    const size_t nframes = size_t(n_streams) * frames_per_stream;
    if (nframes == 0) return DABGPU_OK;
    const size_t nb_soft = (nframes - 1) * soft_stride + NB_FRAME_BITS;
    const size_t nb_hist = size_t(n_streams) * 15 * sc->length * CU_BITS;
    const size_t nb_out = nframes * NB_CIFS * nbytes;
    void *d_soft, *d_hi = nullptr, *d_ho = nullptr, *d_out;
    int rc;
    if ((rc = stage(ctx, 1, nb_soft, &d_soft))) return rc;
    if ((rc = stage(ctx, 3, nb_out, &d_out))) return rc;
    if (history_in && (rc = stage(ctx, 4, nb_hist, &d_hi))) return rc;
    if (history_out && (rc = stage(ctx, 5, nb_hist, &d_ho))) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(hipMemcpyAsync(d_soft, soft, nb_soft, hipMemcpyHostToDevice, s));
    if (history_in) HIP_TRY(hipMemcpyAsync(d_hi, history_in, nb_hist, hipMemcpyHostToDevice, s));
    rc = dabgpu_msc_decode_dev(ctx, sc, static_cast<const int8_t *>(d_soft), soft_stride, n_streams,
                               frames_per_stream, static_cast<const int8_t *>(d_hi), static_cast<int8_t *>(d_ho),
                               static_cast<uint8_t *>(d_out), s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, d_out, nb_out, hipMemcpyDeviceToHost, s));
    if (history_out) HIP_TRY(hipMemcpyAsync(history_out, d_ho, nb_hist, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return DABGPU_OK;
}

// The FIC (d_fib != nullptr) and/or several sub-channels in one grouped lane launch.  Returns 0 when everything was
// enqueued, 1 when the grouped path does not apply (caller falls back to one call per part), < 0 on errors.
static int decode_grouped(dabgpu_ctx *ctx, uint8_t *d_fib, uint8_t *d_crc_ok, const dabgpu_subchannel *sc, int n_subchannels,
                          const int8_t *d_soft, size_t soft_stride, int n_streams, int frames_per_stream,
                          const int8_t *const *d_history_in, int8_t *const *d_history_out, uint8_t *const *d_out,
                          void *stream) {
    const int n_items = n_subchannels + (d_fib ? 1 : 0);
    if (n_items < 2 || ctx->lane_mode == 0 || ctx->lane_unfused || !d_soft || n_streams <= 0 || frames_per_stream <= 0 ||
        soft_stride < size_t(NB_FRAME_BITS))
        return 1;
    const long total_cw = long(n_items) * n_streams * frames_per_stream * NB_CIFS;
    if (ctx->lane_mode < 0 && total_cw < LANE_MIN_CODEWORDS) return 1;
    std::vector<dabk::LaneGroupItem> items;
    if (d_fib) {
        dabk::LaneGroupItem it{};
        it.code = ctx->fic.tables(true);
        it.tables = ctx->fic.lane_tables();
        it.args.soft = d_soft;
        it.args.soft_stride = soft_stride;
        it.args.n_streams = n_streams;
        it.args.frames_per_stream = frames_per_stream;
        it.args.out = d_fib;
        it.is_fic = true;
        it.crc_ok = d_crc_ok;
        if (((reinterpret_cast<uintptr_t>(d_soft) | soft_stride) & 15) || (reinterpret_cast<uintptr_t>(d_fib) & 3)) return 1;
        items.push_back(it);
    }
    for (int i = 0; i < n_subchannels; i++) {
        dab::PunctureProfile prof;
        int rc = subchannel_profile(&sc[i], prof);
        if (rc) return rc;
        DeviceCode *dc = nullptr;
        if ((rc = get_code(ctx, std::move(prof), &dc))) return rc;
        dabk::LaneGroupItem it{};
        it.code = dc->tables(true);
        it.tables = dc->lane_tables();
        it.args.soft = d_soft;
        it.args.soft_stride = soft_stride;
        it.args.n_streams = n_streams;
        it.args.frames_per_stream = frames_per_stream;
        it.args.start_bit = sc[i].start_address * CU_BITS;
        it.args.nbits = sc[i].length * CU_BITS;
        it.args.hist_in = d_history_in ? d_history_in[i] : nullptr;
        it.args.hist_out = d_history_out ? d_history_out[i] : nullptr;
        it.args.out = d_out[i];
        if (it.args.hist_in && it.args.hist_in == it.args.hist_out) return DABGPU_ERR_ARG;
        if (!dabk::lane_supported(dc->prof.nsteps) || !dabk::lane_group_fusable(it.args)) return 1;
        items.push_back(it);
    }
    hipStream_t s = pick_stream(ctx, stream);
    const size_t need = dabk::lane_group_scratch_bytes(items.data(), n_items);
    if (ctx->lane_scratch_bytes < need) {
        HIP_TRY(hipStreamSynchronize(s));
        if (ctx->d_lane_scratch) (void)hipFree(ctx->d_lane_scratch);
        ctx->d_lane_scratch = nullptr;
        ctx->lane_scratch_bytes = 0;
        if (hipMalloc(&ctx->d_lane_scratch, need) != hipSuccess) {
            ctx->d_lane_scratch = nullptr;
            return 1;
        }
        ctx->lane_scratch_bytes = need;
    }
    ScopedTimer tm(ctx, 2, s);
    dabk::LaneScratch lsc{ctx->d_lane_scratch, ctx->lane_scratch_bytes};
    HIP_TRY(dabk::launch_lane_group(items.data(), n_items, lsc, s));
    for (const dabk::LaneGroupItem &it : items)
        if (!it.is_fic) HIP_TRY(dabk::launch_msc_history(it.args, s));
    return 0;
}

// Sub-channels that do not go through the grouped lane launch.  Small batches (each sub-channel below the lane
// kernels' threshold: the plugin's one frame at a time) go through ONE launch of the wave-per-codeword kernel and one
// for the history rings; anything else is decoded sub-channel by sub-channel.
static int decode_subchannels(dabgpu_ctx *ctx, const dabgpu_subchannel *sc, int n_subchannels, const int8_t *d_soft,
                              size_t soft_stride, int n_streams, int frames_per_stream, const int8_t *const *d_history_in,
                              int8_t *const *d_history_out, uint8_t *const *d_out, void *stream, uint8_t *d_fib = nullptr,
                              uint8_t *d_crc_ok = nullptr) {
    const long cw_each = long(n_streams) * frames_per_stream * NB_CIFS;
    bool group = n_subchannels >= 2 && ctx->lane_mode <= 0 && (ctx->lane_mode == 0 || cw_each < LANE_MIN_CODEWORDS) &&
                 d_soft && n_streams > 0 && frames_per_stream > 0;
    std::vector<dabk::WaveGroupItem> items;
    for (int i = 0; group && i < n_subchannels; i++) {
        dab::PunctureProfile prof;
        int rc = subchannel_profile(&sc[i], prof);
        if (rc) return rc;
        if (!dabk::wave_group_supported(prof.nsteps)) { group = false; break; }
        DeviceCode *dc = nullptr;
        if ((rc = get_code(ctx, std::move(prof), &dc))) return rc;
        dabk::WaveGroupItem it{};
        it.code = dc->tables(true);
        it.args.soft = d_soft;
        it.args.soft_stride = soft_stride;
        it.args.n_streams = n_streams;
        it.args.frames_per_stream = frames_per_stream;
        it.args.start_bit = sc[i].start_address * CU_BITS;
        it.args.nbits = sc[i].length * CU_BITS;
        it.args.hist_in = d_history_in ? d_history_in[i] : nullptr;
        it.args.hist_out = d_history_out ? d_history_out[i] : nullptr;
        it.args.out = d_out[i];
        if (it.args.hist_in && it.args.hist_in == it.args.hist_out) return DABGPU_ERR_ARG;
        items.push_back(it);
    }
    if (group) {
        hipStream_t s = pick_stream(ctx, stream);
        ScopedTimer tm(ctx, 2, s);
        // a small batch's FIC rides along: its four codewords per frame are shorter than any sub-channel's, a launch
        // of their own would only queue up in front
        dabk::WaveFicItem fic{ctx->fic.tables(true), d_soft, soft_stride, n_streams * frames_per_stream, d_fib, d_crc_ok};
        HIP_TRY(dabk::launch_msc_decode_group(items.data(), int(items.size()), s, d_fib ? &fic : nullptr));
        return DABGPU_OK;
    }
    if (d_fib) {
        const int rc = dabgpu_fic_decode_dev(ctx, d_soft, soft_stride, n_streams * frames_per_stream, d_fib, d_crc_ok, stream);
        if (rc) return rc;
    }
    for (int i = 0; i < n_subchannels; i++) {
        const int rc = dabgpu_msc_decode_dev(ctx, &sc[i], d_soft, soft_stride, n_streams, frames_per_stream,
                                             d_history_in ? d_history_in[i] : nullptr,
                                             d_history_out ? d_history_out[i] : nullptr, d_out[i], stream);
        if (rc) return rc;
    }
    return DABGPU_OK;
}

int dabgpu_msc_decode_multi_dev(dabgpu_ctx *ctx, const dabgpu_subchannel *sc, int n_subchannels,
                                const int8_t *d_soft, size_t soft_stride, int n_streams, int frames_per_stream,
                                const int8_t *const *d_history_in, int8_t *const *d_history_out,
                                uint8_t *const *d_out, void *stream) {
    if (!ctx || !sc || !d_out || n_subchannels < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    // validate everything before enqueueing anything: profiles, bounds, no overlap inside the CIF
    std::vector<char> used(864, 0);
    for (int i = 0; i < n_subchannels; i++) {
        dab::PunctureProfile prof;
        const int rc = subchannel_profile(&sc[i], prof);
        if (rc) return rc;
        if (!d_out[i]) return DABGPU_ERR_ARG;
        for (int cu = sc[i].start_address; cu < sc[i].start_address + sc[i].length; cu++) {
            if (used[cu]) return DABGPU_ERR_ARG;
            used[cu] = 1;
        }
    }
    {
        const int g = decode_grouped(ctx, nullptr, nullptr, sc, n_subchannels, d_soft, soft_stride, n_streams,
                                     frames_per_stream, d_history_in, d_history_out, d_out, stream);
        if (g <= 0) return g;                                  // done (0) or a real error (< 0); 1 = not applicable
    }
    return decode_subchannels(ctx, sc, n_subchannels, d_soft, soft_stride, n_streams, frames_per_stream, d_history_in,
                              d_history_out, d_out, stream);
}

int dabgpu_decode_frames_dev(dabgpu_ctx *ctx, const int8_t *d_soft, size_t soft_stride, int n_streams,
                             int frames_per_stream, uint8_t *d_fib, uint8_t *d_crc_ok, const dabgpu_subchannel *sc,
                             int n_subchannels, const int8_t *const *d_history_in, int8_t *const *d_history_out,
                             uint8_t *const *d_out, void *stream) {
    if (!ctx || !d_soft || !d_fib || !d_crc_ok || n_streams < 0 || frames_per_stream < 0 || n_subchannels < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (n_subchannels > 0 && (!sc || !d_out)) return DABGPU_ERR_ARG;
    if (soft_stride < size_t(NB_FRAME_BITS) && size_t(n_streams) * frames_per_stream > 1) return DABGPU_ERR_ARG;
    std::vector<char> used(864, 0);
    for (int i = 0; i < n_subchannels; i++) {
        dab::PunctureProfile prof;
        const int rc = subchannel_profile(&sc[i], prof);
        if (rc) return rc;
        if (!d_out[i]) return DABGPU_ERR_ARG;
        for (int cu = sc[i].start_address; cu < sc[i].start_address + sc[i].length; cu++) {
            if (used[cu]) return DABGPU_ERR_ARG;
            used[cu] = 1;
        }
    }
    if (n_streams == 0 || frames_per_stream == 0) return DABGPU_OK;
    const int g = decode_grouped(ctx, d_fib, d_crc_ok, sc, n_subchannels, d_soft, soft_stride, n_streams, frames_per_stream,
                                 d_history_in, d_history_out, d_out, stream);
    if (g <= 0) return g;
    // (the FIC goes into the sub-channels' grouped wave launch when there is one, else it gets its own)
    return decode_subchannels(ctx, sc, n_subchannels, d_soft, soft_stride, n_streams, frames_per_stream, d_history_in,
                              d_history_out, d_out, stream, d_fib, d_crc_ok);
}

int dabgpu_decode_frames(dabgpu_ctx *ctx, const int8_t *soft, size_t soft_stride, int n_streams, int frames_per_stream,
                         uint8_t *fib, uint8_t *crc_ok, const dabgpu_subchannel *sc, int n_subchannels,
                         const int8_t *const *history_in, int8_t *const *history_out, uint8_t *const *out) {
    if (!ctx || !soft || !fib || !crc_ok || n_streams < 0 || frames_per_stream < 0 || n_subchannels < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (n_subchannels > 0 && (!sc || !out)) return DABGPU_ERR_ARG;
    const size_t nframes = size_t(n_streams) * frames_per_stream;
    if (nframes == 0) return DABGPU_OK;
    if (soft_stride < size_t(NB_FRAME_BITS) && nframes > 1) return DABGPU_ERR_ARG;
    // layout of the result and history staging buffers: [fib | crc | out_0 | out_1 ...], [hist_0 | hist_1 ...]
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    std::vector<size_t> out_off(n_subchannels), out_bytes(n_subchannels), hist_off(n_subchannels), hist_bytes(n_subchannels);
    const size_t nb_fib = nframes * NB_FIBS * 32, nb_crc = nframes * NB_FIBS;
    size_t res_total = al(nb_fib) + al(nb_crc), hist_total = 0;
    for (int i = 0; i < n_subchannels; i++) {
        const int nbytes = dabgpu_subchannel_bytes(&sc[i]);
        if (nbytes < 0) return nbytes;
        if (!out[i]) return DABGPU_ERR_ARG;
        out_off[i] = res_total;
        out_bytes[i] = nframes * NB_CIFS * size_t(nbytes);
        res_total += al(out_bytes[i]);
        hist_off[i] = hist_total;
        hist_bytes[i] = size_t(n_streams) * 15 * sc[i].length * CU_BITS;
        hist_total += al(hist_bytes[i]);
    }
    const size_t nb_soft = (nframes - 1) * soft_stride + NB_FRAME_BITS;
    void *d_soft, *d_res, *d_hi = nullptr, *d_ho = nullptr;
    int rc;
    if ((rc = stage(ctx, 1, nb_soft, &d_soft))) return rc;
    if ((rc = stage(ctx, 3, res_total, &d_res))) return rc;
    if (hist_total && history_in && (rc = stage(ctx, 4, hist_total, &d_hi))) return rc;
    if (hist_total && history_out && (rc = stage(ctx, 5, hist_total, &d_ho))) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(hipMemcpyAsync(d_soft, soft, nb_soft, hipMemcpyHostToDevice, s));            // the frames go up once
    std::vector<const int8_t *> p_hi(n_subchannels, nullptr);
    std::vector<int8_t *> p_ho(n_subchannels, nullptr);
    std::vector<uint8_t *> p_out(n_subchannels, nullptr);
    char *res = static_cast<char *>(d_res);
    for (int i = 0; i < n_subchannels; i++) {
        p_out[i] = reinterpret_cast<uint8_t *>(res + out_off[i]);
        if (history_in && history_in[i]) {
            p_hi[i] = reinterpret_cast<const int8_t *>(static_cast<char *>(d_hi) + hist_off[i]);
            HIP_TRY(hipMemcpyAsync(const_cast<int8_t *>(p_hi[i]), history_in[i], hist_bytes[i], hipMemcpyHostToDevice, s));
        }
        if (history_out && history_out[i]) p_ho[i] = reinterpret_cast<int8_t *>(static_cast<char *>(d_ho) + hist_off[i]);
    }
    uint8_t *d_fib = reinterpret_cast<uint8_t *>(res), *d_crc = reinterpret_cast<uint8_t *>(res + al(nb_fib));
    rc = dabgpu_decode_frames_dev(ctx, static_cast<const int8_t *>(d_soft), soft_stride, n_streams, frames_per_stream, d_fib,
                                  d_crc, sc, n_subchannels, n_subchannels ? p_hi.data() : nullptr,
                                  n_subchannels ? p_ho.data() : nullptr, n_subchannels ? p_out.data() : nullptr, s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(fib, d_fib, nb_fib, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(crc_ok, d_crc, nb_crc, hipMemcpyDeviceToHost, s));
    for (int i = 0; i < n_subchannels; i++) {
        HIP_TRY(hipMemcpyAsync(out[i], p_out[i], out_bytes[i], hipMemcpyDeviceToHost, s));
        if (p_ho[i]) HIP_TRY(hipMemcpyAsync(history_out[i], p_ho[i], hist_bytes[i], hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));                                                  // one synchronisation
    return DABGPU_OK;
}

int dabgpu_decode_stream_reset(dabgpu_ctx *ctx) {
    if (!ctx) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (auto &h : ctx->sub_history) { (void)hipFree(h.ring[0]); (void)hipFree(h.ring[1]); }
    ctx->sub_history.clear();
    return DABGPU_OK;
}

static int decode_stream_frames_body(dabgpu_ctx *ctx, const int8_t *soft, size_t soft_stride, int n_frames, uint8_t *fib,
                                     uint8_t *crc_ok, const dabgpu_subchannel *sc, int n_subchannels, uint8_t *const *out);

int dabgpu_decode_stream_frames(dabgpu_ctx *ctx, const int8_t *soft, size_t soft_stride, int n_frames, uint8_t *fib,
                                uint8_t *crc_ok, const dabgpu_subchannel *sc, int n_subchannels, uint8_t *const *out) {
    if (!ctx || !soft || !fib || !crc_ok || n_frames < 0 || n_subchannels < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (n_subchannels > 0 && (!sc || !out)) return DABGPU_ERR_ARG;
    if (n_frames == 0) return DABGPU_OK;
    if (soft_stride < size_t(NB_FRAME_BITS) && n_frames > 1) return DABGPU_ERR_ARG;
    const int rc = decode_stream_frames_body(ctx, soft, soft_stride, n_frames, fib, crc_ok, sc, n_subchannels, out);
    if (rc != DABGPU_OK) {
        // A call that failed part-way leaves rings that have missed this frame (and `live` marks on some of them): no
        // ring continues the stream any more.  All of them go; the next call starts every sub-channel from erasures.
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipGetLastError();
        for (auto &h : ctx->sub_history) { (void)hipFree(h.ring[0]); (void)hipFree(h.ring[1]); }
        ctx->sub_history.clear();
    }
    return rc;
}

static int decode_stream_frames_body(dabgpu_ctx *ctx, const int8_t *soft, size_t soft_stride, int n_frames, uint8_t *fib,
                                     uint8_t *crc_ok, const dabgpu_subchannel *sc, int n_subchannels, uint8_t *const *out) {
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t nb_fib = size_t(n_frames) * NB_FIBS * 32, nb_crc = size_t(n_frames) * NB_FIBS;
    std::vector<size_t> out_off(n_subchannels), out_bytes(n_subchannels);
    size_t res_total = al(nb_fib) + al(nb_crc);
    std::vector<const int8_t *> p_hi(n_subchannels, nullptr);
    std::vector<int8_t *> p_ho(n_subchannels, nullptr);
    std::vector<int> hist_index(n_subchannels, -1);
    hipStream_t s = ctx->stream;
    for (int i = 0; i < n_subchannels; i++) {
        const int nbytes = dabgpu_subchannel_bytes(&sc[i]);
        if (nbytes < 0) return nbytes;
        if (!out[i]) return DABGPU_ERR_ARG;
        out_off[i] = res_total;
        out_bytes[i] = size_t(n_frames) * NB_CIFS * size_t(nbytes);
        res_total += al(out_bytes[i]);
        // the sub-channel's ring from the call before, or a new (erased) one
        for (size_t k = 0; k < ctx->sub_history.size(); k++)
            if (ctx->sub_history[k].start_address == sc[i].start_address && ctx->sub_history[k].length == sc[i].length) hist_index[i] = int(k);
        if (hist_index[i] < 0) {
            dabgpu_ctx::SubHistory h{};
            h.start_address = sc[i].start_address;
            h.length = sc[i].length;
            h.bytes = size_t(15) * sc[i].length * CU_BITS;
            if (hipMalloc(reinterpret_cast<void **>(&h.ring[0]), h.bytes) != hipSuccess) return DABGPU_ERR_NOMEM;
            if (hipMalloc(reinterpret_cast<void **>(&h.ring[1]), h.bytes) != hipSuccess) { (void)hipFree(h.ring[0]); return DABGPU_ERR_NOMEM; }
            hist_index[i] = int(ctx->sub_history.size());
            ctx->sub_history.push_back(h);
            HIP_TRY(hipMemsetAsync(h.ring[0], 0, h.bytes, s));
        }
        dabgpu_ctx::SubHistory &h = ctx->sub_history[size_t(hist_index[i])];
        h.live = true;
        p_hi[i] = h.ring[h.cur];
        p_ho[i] = h.ring[h.cur ^ 1];
    }
    const size_t nb_soft = size_t(n_frames - 1) * soft_stride + NB_FRAME_BITS;
    void *d_soft, *d_res;
    int rc;
    if ((rc = stage(ctx, 1, nb_soft, &d_soft))) return rc;
    if ((rc = stage(ctx, 3, res_total, &d_res))) return rc;
    if (ctx->h_bounce_bytes < res_total) {
        if (ctx->h_bounce) (void)hipHostFree(ctx->h_bounce);
        ctx->h_bounce = nullptr;
        ctx->h_bounce_bytes = 0;
        if (hipHostMalloc(&ctx->h_bounce, res_total, hipHostMallocDefault) != hipSuccess) return DABGPU_ERR_NOMEM;
        ctx->h_bounce_bytes = res_total;
    }
    HIP_TRY(hipMemcpyAsync(d_soft, soft, nb_soft, hipMemcpyHostToDevice, s));            // one upload
    char *res = static_cast<char *>(d_res);
    std::vector<uint8_t *> p_out(n_subchannels, nullptr);
    for (int i = 0; i < n_subchannels; i++) p_out[i] = reinterpret_cast<uint8_t *>(res + out_off[i]);
    rc = dabgpu_decode_frames_dev(ctx, static_cast<const int8_t *>(d_soft), soft_stride, 1, n_frames,
                                  reinterpret_cast<uint8_t *>(res), reinterpret_cast<uint8_t *>(res + al(nb_fib)), sc, n_subchannels,
                                  n_subchannels ? p_hi.data() : nullptr, n_subchannels ? p_ho.data() : nullptr,
                                  n_subchannels ? p_out.data() : nullptr, s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->h_bounce, d_res, res_total, hipMemcpyDeviceToHost, s));     // one download
    HIP_TRY(hipStreamSynchronize(s));                                                   // one synchronisation
    const char *hb = static_cast<const char *>(ctx->h_bounce);
    std::memcpy(fib, hb, nb_fib);
    std::memcpy(crc_ok, hb + al(nb_fib), nb_crc);
    for (int i = 0; i < n_subchannels; i++) {
        std::memcpy(out[i], hb + out_off[i], out_bytes[i]);
        ctx->sub_history[size_t(hist_index[i])].cur ^= 1;
    }
    // a sub-channel left out of this call has missed a frame: its ring no longer continues the stream, and a later
    // call starts it from erasures again (this also bounds the list over any number of reconfigurations)
    size_t kept = 0;
    for (auto &h : ctx->sub_history) {
        if (h.live) { h.live = false; ctx->sub_history[kept++] = h; }
        else { (void)hipFree(h.ring[0]); (void)hipFree(h.ring[1]); }
    }
    ctx->sub_history.resize(kept);
    return DABGPU_OK;
}

// ---------------------------------------------------------------------------- DAB+ super-frame
static_assert(sizeof(dabgpu_superframe_status) == sizeof(dabk::SuperframeStatus), "ABI struct mirrors the kernel's");

int dabgpu_dabplus_superframes_dev(dabgpu_ctx *ctx, const uint8_t *d_in, size_t in_stride, int n_superframes,
                                   int bitrate_kbps, uint8_t *d_out, dabgpu_superframe_status *d_status,
                                   void *stream) {
    if (!ctx || !d_in || !d_out || !d_status || n_superframes < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (bitrate_kbps < 8 || bitrate_kbps % 8 || bitrate_kbps > 512) return DABGPU_ERR_PROFILE;
    const int s = bitrate_kbps / 8;
    if (n_superframes > 1 && in_stride < size_t(120) * s) return DABGPU_ERR_ARG;
    if (n_superframes == 0) return DABGPU_OK;
    HIP_TRY(dabk::launch_dabplus_superframes(d_in, in_stride, n_superframes, s, d_out,
                                             reinterpret_cast<dabk::SuperframeStatus *>(d_status),
                                             pick_stream(ctx, stream)));
    return DABGPU_OK;
}

int dabgpu_dabplus_superframes(dabgpu_ctx *ctx, const uint8_t *in, size_t in_stride, int n_superframes,
                               int bitrate_kbps, uint8_t *out, dabgpu_superframe_status *status) {
    if (!ctx || !in || !out || !status || n_superframes < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (bitrate_kbps < 8 || bitrate_kbps % 8 || bitrate_kbps > 512) return DABGPU_ERR_PROFILE;
    if (n_superframes == 0) return DABGPU_OK;
    const int s = bitrate_kbps / 8;
    const size_t nb_in = size_t(n_superframes - 1) * in_stride + size_t(120) * s;
    const size_t nb_out = size_t(n_superframes) * 110 * s;
    void *d_in, *d_out, *d_st;
    int rc;
    if ((rc = stage(ctx, 1, nb_in, &d_in))) return rc;
    if ((rc = stage(ctx, 3, nb_out, &d_out))) return rc;
    if ((rc = stage(ctx, 2, sizeof(dabgpu_superframe_status) * n_superframes, &d_st))) return rc;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(d_in, in, nb_in, hipMemcpyHostToDevice, st));
    rc = dabgpu_dabplus_superframes_dev(ctx, static_cast<const uint8_t *>(d_in), in_stride, n_superframes,
                                        bitrate_kbps, static_cast<uint8_t *>(d_out),
                                        static_cast<dabgpu_superframe_status *>(d_st), st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, d_out, nb_out, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(status, d_st, sizeof(dabgpu_superframe_status) * n_superframes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DABGPU_OK;
}

// ---------------------------------------------------------------------------- plain Viterbi
int dabgpu_viterbi_dev(dabgpu_ctx *ctx, const int8_t *d_punct, int n_codewords, const uint8_t *mask, int nsteps,
                       uint8_t *d_out_bytes, void *stream) {
    if (!ctx || !d_punct || !mask || !d_out_bytes || n_codewords < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (nsteps < 14 || ((nsteps - 6) & 7)) return DABGPU_ERR_ARG;
    const bool too_long = !dabk::viterbi_fits(nsteps);
    if (too_long && !dabk::lane_supported(nsteps)) return DABGPU_ERR_CAPACITY;
    dab::PunctureProfile prof;
    prof.mask.assign(mask, mask + 4 * size_t(nsteps));
    for (uint8_t &f : prof.mask) f = f ? 1 : 0;
    finish_profile(prof);
    if (n_codewords == 0) return DABGPU_OK;
    DeviceCode *dc = nullptr;
    int rc = get_code(ctx, std::move(prof), &dc);
    if (rc) return rc;
    hipStream_t s = pick_stream(ctx, stream);
    dabk::LaneScratch lsc{};
    int lrc;
    if (use_lane(ctx, dc->prof.nsteps, n_codewords, s, &lsc, &lrc, too_long)) {
        HIP_TRY(dabk::launch_viterbi_plain_lane(dc->tables(false), dc->lane_tables(), d_punct, n_codewords, lsc,
                                                d_out_bytes, s));
        return DABGPU_OK;
    }
    if (lrc) return lrc;
    if (too_long) return DABGPU_ERR_CAPACITY;
    HIP_TRY(dabk::launch_viterbi_plain(dc->tables(false), d_punct, n_codewords, d_out_bytes, s));
    return DABGPU_OK;
}

int dabgpu_viterbi(dabgpu_ctx *ctx, const int8_t *punct, int n_codewords, const uint8_t *mask, int nsteps,
                   uint8_t *out_bytes) {
    if (!ctx || !punct || !mask || !out_bytes || n_codewords < 0) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    if (nsteps < 14 || ((nsteps - 6) & 7)) return DABGPU_ERR_ARG;
    if (n_codewords == 0) return DABGPU_OK;
    size_t n_punct = 0;
    for (int i = 0; i < 4 * nsteps; i++) n_punct += mask[i] ? 1 : 0;
    const size_t nb_in = size_t(n_codewords) * n_punct, nb_out = size_t(n_codewords) * ((nsteps - 6) / 8);
    void *d_in, *d_out;
    int rc;
    if ((rc = stage(ctx, 1, nb_in ? nb_in : 1, &d_in))) return rc;
    if ((rc = stage(ctx, 3, nb_out, &d_out))) return rc;
    hipStream_t s = ctx->stream;
    if (nb_in) HIP_TRY(hipMemcpyAsync(d_in, punct, nb_in, hipMemcpyHostToDevice, s));
    rc = dabgpu_viterbi_dev(ctx, static_cast<const int8_t *>(d_in), n_codewords, mask, nsteps,
                            static_cast<uint8_t *>(d_out), s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out_bytes, d_out, nb_out, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return DABGPU_OK;
}

}  // extern "C"
