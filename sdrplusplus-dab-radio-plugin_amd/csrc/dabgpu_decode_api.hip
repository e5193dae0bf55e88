// dabgpu_decode_api.hip -- the channel-decoder entry points of the C ABI (include/dabgpu.h): FIC, MSC sub-channels,
// whole frames, the one-stream host call, DAB+ super-frames, plain Viterbi.
#include "dabgpu_ctx.hpp"

#include <algorithm>
#include <cstring>

using namespace dab;
using namespace dabapi;

namespace {

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
}  // namespace

extern "C" {

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

// the sub-channel's code tables through the context's descriptor cache (see dabgpu_ctx::code_by_descriptor); the same
// checks and status codes as subchannel_profile
static int lookup_code(dabgpu_ctx *ctx, const dabgpu_subchannel *sc, DeviceCode **out) {
    if (!sc) return DABGPU_ERR_ARG;
    const uint64_t key = (uint64_t(sc->is_uep != 0) << 63) | (uint64_t(uint32_t(sc->eep_type) & 0xFu) << 56) |
                         (uint64_t(uint32_t(sc->protection_level) & 0xFFu) << 48) | (uint64_t(uint32_t(sc->bitrate_kbps) & 0xFFFFFFu) << 16) |
                         uint64_t(uint32_t(sc->length) & 0xFFFFu);
    // (the key holds the fields masked: only a descriptor whose fields fit their masks may use -- or fill -- the cache)
    const bool in_range = sc->eep_type >= 0 && sc->eep_type <= 15 && sc->protection_level >= 0 && sc->protection_level <= 255 &&
                          sc->bitrate_kbps >= 0 && sc->bitrate_kbps <= 0xFFFFFF && sc->length >= 0 && sc->length <= 0xFFFF;
    auto it = in_range ? ctx->code_by_descriptor.find(key) : ctx->code_by_descriptor.end();
    if (it == ctx->code_by_descriptor.end()) {
        dab::PunctureProfile prof;
        int rc = subchannel_profile(sc, prof);
        if (rc) return rc;
        // a length no decoder holds gets no device tables (they would stay allocated for the context's lifetime)
        if (!dabk::viterbi_fits(prof.nsteps) && !dabk::lane_supported(prof.nsteps)) return DABGPU_ERR_CAPACITY;
        DeviceCode *dc = nullptr;
        if ((rc = get_code(ctx, std::move(prof), &dc))) return rc;
        // (ctx->codes only ever grows until dabgpu_destroy frees it: the pointers kept here stay valid for the context's life)
        if (in_range) ctx->code_by_descriptor[key] = dc;
        *out = dc;
        return DABGPU_OK;
    }
    if (sc->start_address < 0 || sc->start_address + sc->length > 864) return DABGPU_ERR_ARG;
    *out = it->second;
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
    DeviceCode *dc = nullptr;
    int rc = lookup_code(ctx, sc, &dc);
    if (rc) return rc;
    if (n_streams == 0 || frames_per_stream == 0) return DABGPU_OK;
    const bool too_long = !dabk::viterbi_fits(dc->prof.nsteps);   // above ~800 kbit/s: only the lane kernels hold it
    if (too_long && !dabk::lane_supported(dc->prof.nsteps)) return DABGPU_ERR_CAPACITY;
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
        DeviceCode *dc = nullptr;
        const int rc = lookup_code(ctx, &sc[i], &dc);
        if (rc) return rc;
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
    HIP_TRY(dabk::launch_lane_group(items.data(), n_items, lsc, s, tm.mids()));
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
    bool group = n_subchannels >= 1 && n_subchannels + (d_fib ? 1 : 0) >= 2 && ctx->lane_mode <= 0 &&
                 (ctx->lane_mode == 0 || cw_each < LANE_MIN_CODEWORDS) && d_soft && n_streams > 0 && frames_per_stream > 0;
    std::vector<dabk::WaveGroupItem> items;
    for (int i = 0; group && i < n_subchannels; i++) {
        DeviceCode *dc = nullptr;
        const int rc = lookup_code(ctx, &sc[i], &dc);
        if (rc) return rc;
        if (!dabk::wave_group_supported(dc->prof.nsteps)) { group = false; break; }
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
    // the FIC with ONE sub-channel: together only when every codeword is resident at once (the launch then takes as long
    // as the sub-channel alone: 135 -> 101 us per call up to 256 frames); queued up in rounds, two launches are faster
    if (group && n_subchannels == 1 && !dabk::wave_group_one_round(std::max(items[0].code.nsteps, ctx->fic.prof.nsteps), 2 * cw_each))
        group = false;
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
        DeviceCode *dc = nullptr;
        const int rc = lookup_code(ctx, &sc[i], &dc);
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
        DeviceCode *dc = nullptr;
        const int rc = lookup_code(ctx, &sc[i], &dc);
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
        DeviceCode *dc = nullptr;
        const int lrc = lookup_code(ctx, &sc[i], &dc);
        if (lrc) return lrc;
        const int nbytes = (dc->prof.nsteps - 6) / 8;
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
    // argument errors are refused before any ring is touched: the kept state survives them
    for (int i = 0; i < n_subchannels; i++) {
        DeviceCode *dc = nullptr;
        const int lrc = lookup_code(ctx, &sc[i], &dc);
        if (lrc) return lrc;
        if (!out[i]) return DABGPU_ERR_ARG;
    }
    if (!subchannels_disjoint(sc, n_subchannels)) return DABGPU_ERR_ARG;
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
        DeviceCode *dc = nullptr;
        const int lrc = lookup_code(ctx, &sc[i], &dc);
        if (lrc) return lrc;
        const int nbytes = (dc->prof.nsteps - 6) / 8;
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
    if ((rc = ensure_bounce(ctx, res_total))) return rc;
    if (injected_failure(ctx)) return DABGPU_ERR_HIP;            // (test hook: the caller's failure path drops every ring)
    // one upload (by a kernel when the soft bits lie in page-locked memory the device can address).  One frame with a
    // handful of sub-channels -- the plugin's call -- sends only what will be read: the FIC and the sub-channels' ranges
    // of the four CIFs (21.5 kB of the 230 kB for one 64 kbit/s service)
    // (every query first: once the upload is enqueued the host only enqueues, and stays ahead of the device)
    void *h_dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&h_dev, ctx->h_bounce, 0));
    void *soft_alias = (nb_soft & 15) ? nullptr : device_alias_of_pinned(soft);
    if (soft_alias && !(reinterpret_cast<uintptr_t>(soft_alias) & 15)) {
        std::vector<dabk::CopyPiece> up;
        if (n_frames == 1 && 1 + NB_CIFS * n_subchannels <= dabk::copy_pieces_max()) {
            char *d = static_cast<char *>(d_soft);
            const char *h = static_cast<const char *>(soft_alias);
            up.push_back(dabk::CopyPiece{d, h, size_t(NB_FIC_BITS)});
            for (int i = 0; i < n_subchannels; i++)
                for (int c = 0; c < NB_CIFS; c++) {
                    const size_t off = size_t(NB_FIC_BITS) + size_t(c) * NB_CIF_BITS + size_t(sc[i].start_address) * CU_BITS;
                    up.push_back(dabk::CopyPiece{d + off, h + off, size_t(sc[i].length) * CU_BITS});
                }
        } else {
            up.push_back(dabk::CopyPiece{d_soft, soft_alias, nb_soft});
        }
        HIP_TRY(dabk::launch_copy_pieces(up.data(), int(up.size()), s));
    } else {
        HIP_TRY(hipMemcpyAsync(d_soft, soft, nb_soft, hipMemcpyHostToDevice, s));
    }
    // the results -- a few hundred bytes per frame -- are written by the decoder's kernels straight into the page-locked
    // landing area (no copy behind them); one synchronisation
    (void)d_res;
    char *res = static_cast<char *>(h_dev);
    std::vector<uint8_t *> p_out(n_subchannels, nullptr);
    for (int i = 0; i < n_subchannels; i++) p_out[i] = reinterpret_cast<uint8_t *>(res + out_off[i]);
    rc = dabgpu_decode_frames_dev(ctx, static_cast<const int8_t *>(d_soft), soft_stride, 1, n_frames,
                                  reinterpret_cast<uint8_t *>(res), reinterpret_cast<uint8_t *>(res + al(nb_fib)), sc, n_subchannels,
                                  n_subchannels ? p_hi.data() : nullptr, n_subchannels ? p_ho.data() : nullptr,
                                  n_subchannels ? p_out.data() : nullptr, s);
    if (rc) return rc;
    {   // one synchronisation: the word behind the landing area's payload
        const size_t off_flag = ctx->h_bounce_bytes - 64;
        if ((rc = wait_for_signal(s, reinterpret_cast<volatile unsigned long long *>(static_cast<char *>(ctx->h_bounce) + off_flag),
                                  reinterpret_cast<unsigned long long *>(res + off_flag), ++ctx->signal_seq)))
            return rc;
    }
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
    // A few super-frames in page-locked buffers (the host mirror's channels hand over ONE, ~1-2 kB in, ~1 kB out): the kernel
    // reads and writes the caller's buffers themselves -- one launch and the watched word instead of three copy-engine
    // transfers and a sleep (~60 -> ~15 us per call; three DAB+ services make 2.4 such calls per frame).
    if (n_superframes <= 8) {
        void *a_in = device_alias_of_pinned(in), *a_out = device_alias_of_pinned(out), *a_st = device_alias_of_pinned(status);
        if (a_in && a_out && a_st && !(reinterpret_cast<uintptr_t>(a_st) & 7)) {
            const int brc = ensure_bounce(ctx, 0);
            if (brc) return brc;
            // the kernel writes the caller's buffers: the word is watched only when they are known to be coherent
            const bool coherent = known_coherent_host(out, nb_out) && known_coherent_host(status, sizeof(dabgpu_superframe_status) * size_t(n_superframes));
            void *h_dev = nullptr;
            HIP_TRY(hipHostGetDevicePointer(&h_dev, ctx->h_bounce, 0));
            hipStream_t st = ctx->stream;
            const size_t off_flag = ctx->h_bounce_bytes - 64;
            volatile unsigned long long *flag_host = reinterpret_cast<volatile unsigned long long *>(static_cast<char *>(ctx->h_bounce) + off_flag);
            unsigned long long *flag_dev = reinterpret_cast<unsigned long long *>(static_cast<char *>(h_dev) + off_flag);
            const unsigned long long seq = ++ctx->signal_seq;
            if (n_superframes == 1) {
                // ONE super-frame (the host mirror's call) is ONE workgroup: it stores the watched word itself, behind its results
                HIP_TRY(dabk::launch_dabplus_superframes(static_cast<const uint8_t *>(a_in), in_stride, 1, s, static_cast<uint8_t *>(a_out),
                                                         reinterpret_cast<dabk::SuperframeStatus *>(a_st), st, flag_dev, seq));
                return wait_for_signal(st, flag_host, flag_dev, seq, true, coherent);
            }
            const int rc0 = dabgpu_dabplus_superframes_dev(ctx, static_cast<const uint8_t *>(a_in), in_stride, n_superframes, bitrate_kbps,
                                                           static_cast<uint8_t *>(a_out), static_cast<dabgpu_superframe_status *>(a_st), st);
            if (rc0) return rc0;
            return wait_for_signal(st, flag_host, flag_dev, seq, false, coherent);
        }
    }
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