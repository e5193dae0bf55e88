// dabgpu_pipeline.hip -- the host-fed ring (dabgpu_pipe_*): dabgpu_ofdm_demod_frames + dabgpu_decode_frames for a caller
// whose samples start in host memory (files, a network), without the synchronous calls' serial upload -> kernels ->
// download.  The reference runs the same two stages on two threads with a 2-frame ring between them
// (/root/reference/src/radio_block.cpp:23-44); here the ring is `slots` device staging sets, and three engines work
// at once: the upload of batch k+1 (copy stream), the kernels of batch k (the context stream), the download of batch
// k-1 (a second copy stream).  Ordering is by events; the host only ever blocks in dabgpu_pipe_wait (and in a submit
// that finds every slot still busy).
#include "dabgpu_ctx.hpp"

#include <algorithm>
#include <cstring>
#include <new>

using namespace dab;
using namespace dabapi;

namespace dabapi {

struct Pipeline {
    struct Slot {
        void *d_iq = nullptr, *d_soft = nullptr, *d_res = nullptr, *d_fo = nullptr;
        size_t res_bytes = 0;
        hipEvent_t up = nullptr, done = nullptr;     // upload finished; every download of the batch finished
        hipEvent_t comp = nullptr;                   // kernels finished
        int64_t ticket = -1;                         // the batch that last used the slot
        bool busy = false;                           // ... and has not been waited for
    };
    std::vector<Slot> slots;
    hipStream_t s_up = nullptr, s_down = nullptr;
    int max_frames = 0;
    size_t frame_stride = 0;
    int64_t next_ticket = 0;
    // de-interleaver rings of the sub-channels the ring decodes, on the device, double-buffered: [n_streams][15][bits]
    struct Ring {
        int start_address, length, n_streams;
        size_t bytes;
        int8_t *buf[2];
        int cur;
        bool live;
    };
    std::vector<Ring> rings;
};

static void free_rings(Pipeline *p) {
    for (auto &r : p->rings) { (void)hipFree(r.buf[0]); (void)hipFree(r.buf[1]); }
    p->rings.clear();
}

void pipeline_destroy(dabgpu_ctx *ctx) {
    Pipeline *p = ctx->pipe;
    if (!p) return;
    if (p->s_up) (void)hipStreamSynchronize(p->s_up);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (p->s_down) (void)hipStreamSynchronize(p->s_down);
    for (auto &s : p->slots) {
        if (s.d_iq) (void)hipFree(s.d_iq);
        if (s.d_soft) (void)hipFree(s.d_soft);
        if (s.d_res) (void)hipFree(s.d_res);
        if (s.d_fo) (void)hipFree(s.d_fo);
        for (hipEvent_t e : {s.up, s.comp, s.done}) if (e) (void)hipEventDestroy(e);
    }
    free_rings(p);
    if (p->s_up) (void)hipStreamDestroy(p->s_up);
    if (p->s_down) (void)hipStreamDestroy(p->s_down);
    (void)hipGetLastError();
    delete p;
    ctx->pipe = nullptr;
}

}  // namespace dabapi

extern "C" {

int dabgpu_pipe_open(dabgpu_ctx *ctx, int slots, int max_frames, size_t frame_stride) {
    if (!ctx || slots < 2 || slots > 8 || max_frames <= 0) return DABGPU_ERR_ARG;
    if (frame_stride < size_t(NB_FRAME_SYMBOLS) * NB_SYM_PERIOD || (frame_stride & 1u)) return DABGPU_ERR_ARG;
    if (size_t(max_frames) > size_t(0x7fffffff) / NB_DATA_SYMBOLS) return DABGPU_ERR_ARG;
    if (ctx->pipe) return DABGPU_ERR_ARG;                       // one ring per context
    DeviceGuard guard(ctx);
    Pipeline *p = new (std::nothrow) Pipeline();
    if (!p) return DABGPU_ERR_NOMEM;
    ctx->pipe = p;
    p->max_frames = max_frames;
    p->frame_stride = frame_stride;
    p->slots.resize(size_t(slots));
    int rc = DABGPU_OK;
    if (hipStreamCreateWithFlags(&p->s_up, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&p->s_down, hipStreamNonBlocking) != hipSuccess)
        rc = DABGPU_ERR_HIP;
    const size_t nb_iq = size_t(max_frames) * frame_stride * sizeof(float2);
    for (auto &s : p->slots) {
        if (rc) break;
        if (hipMalloc(&s.d_iq, nb_iq) != hipSuccess || hipMalloc(&s.d_soft, size_t(max_frames) * NB_FRAME_BITS) != hipSuccess ||
            hipMalloc(&s.d_fo, sizeof(float) * size_t(max_frames)) != hipSuccess) { rc = DABGPU_ERR_NOMEM; break; }
        if (hipEventCreateWithFlags(&s.up, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.comp, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) { rc = DABGPU_ERR_HIP; break; }
    }
    if (rc) { (void)hipGetLastError(); pipeline_destroy(ctx); }
    return rc;
}

int dabgpu_pipe_close(dabgpu_ctx *ctx) {
    if (!ctx) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    pipeline_destroy(ctx);                                      // (waits for everything in flight)
    return DABGPU_OK;
}

int dabgpu_pipe_reset(dabgpu_ctx *ctx) {
    if (!ctx || !ctx->pipe) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    free_rings(ctx->pipe);
    return DABGPU_OK;
}

static int submit_body(dabgpu_ctx *ctx, const float *iq, int n_streams, int frames_per_stream, const float *freq_offset,
                       float fine_freq_update_beta, const dabgpu_subchannel *sc, int n_subchannels, int8_t *soft, uint8_t *fib,
                       uint8_t *crc_ok, uint8_t *const *out, int64_t *ticket);

int dabgpu_pipe_submit(dabgpu_ctx *ctx, const float *iq, int n_streams, int frames_per_stream, const float *freq_offset,
                       float fine_freq_update_beta, const dabgpu_subchannel *sc, int n_subchannels, int8_t *soft, uint8_t *fib,
                       uint8_t *crc_ok, uint8_t *const *out, int64_t *ticket) {
    if (!ctx || !ctx->pipe || !iq || !fib || !crc_ok || !ticket || n_streams <= 0 || frames_per_stream <= 0 || n_subchannels < 0)
        return DABGPU_ERR_ARG;
    if (n_subchannels > 0 && (!sc || !out)) return DABGPU_ERR_ARG;
    Pipeline *p = ctx->pipe;
    if (size_t(n_streams) * size_t(frames_per_stream) > size_t(p->max_frames)) return DABGPU_ERR_CAPACITY;
    if (!freq_offset) {
        if (n_streams > ctx->n_states) return DABGPU_ERR_CAPACITY;              // closed loop: dabgpu_streams_reset first
        if (!(fine_freq_update_beta >= 0.f && fine_freq_update_beta <= 1.f)) return DABGPU_ERR_ARG;
    }
    // argument errors are refused here, before a slot or a ring is touched: the ring's state survives them
    for (int i = 0; i < n_subchannels; i++) {
        const int nbytes = dabgpu_subchannel_bytes(&sc[i]);
        if (nbytes < 0) return nbytes;
        if (!out[i]) return DABGPU_ERR_ARG;
    }
    if (!subchannels_disjoint(sc, n_subchannels)) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    const int rc = submit_body(ctx, iq, n_streams, frames_per_stream, freq_offset, fine_freq_update_beta, sc, n_subchannels, soft, fib,
                               crc_ok, out, ticket);
    if (rc != DABGPU_OK) {
        // A runtime call failed part-way: a ring may exist that was never zeroed, copies into the caller's buffers may be
        // in flight behind a slot that is not marked busy, and every ring has missed this batch.  Everything enqueued so
        // far is waited for (the caller's buffers are safe to reuse when this returns) and every ring goes: the next
        // submit starts its sub-channels from erasures.
        (void)hipStreamSynchronize(p->s_up);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(p->s_down);
        (void)hipGetLastError();
        free_rings(p);
    }
    return rc;
}

static int submit_body(dabgpu_ctx *ctx, const float *iq, int n_streams, int frames_per_stream, const float *freq_offset,
                       float fine_freq_update_beta, const dabgpu_subchannel *sc, int n_subchannels, int8_t *soft, uint8_t *fib,
                       uint8_t *crc_ok, uint8_t *const *out, int64_t *ticket) {
    Pipeline *p = ctx->pipe;
    const int n_frames = n_streams * frames_per_stream;
    // layout of the slot's result block: [fib | crc | out_0 | out_1 ...]
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t nb_fib = size_t(n_frames) * NB_FIBS * 32, nb_crc = size_t(n_frames) * NB_FIBS;
    const size_t nsc = size_t(n_subchannels);
    std::vector<size_t> out_off(nsc), out_bytes(nsc);
    size_t res_total = al(nb_fib) + al(nb_crc);
    for (int i = 0; i < n_subchannels; i++) {
        const int nbytes = dabgpu_subchannel_bytes(&sc[i]);
        if (nbytes < 0) return nbytes;
        if (!out[i]) return DABGPU_ERR_ARG;
        out_off[size_t(i)] = res_total;
        out_bytes[size_t(i)] = size_t(n_frames) * NB_CIFS * size_t(nbytes);
        res_total += al(out_bytes[size_t(i)]);
    }
    Pipeline::Slot &s = p->slots[size_t(p->next_ticket % int64_t(p->slots.size()))];
    // the slot's previous batch must have left it (its caller may not have waited for it yet)
    if (s.busy) { HIP_TRY(hipEventSynchronize(s.done)); s.busy = false; }
    if (s.res_bytes < res_total) {
        if (s.d_res) (void)hipFree(s.d_res);
        s.d_res = nullptr;
        s.res_bytes = 0;
        if (hipMalloc(&s.d_res, res_total) != hipSuccess) return DABGPU_ERR_NOMEM;
        s.res_bytes = res_total;
    }
    // the rings of this batch's sub-channels (a sub-channel seen for the first time, with another stream count, or left
    // out of the batch before starts from erasures: a ring that misses a batch no longer continues its streams)
    hipStream_t sc_stream = ctx->stream;
    std::vector<const int8_t *> p_hi(size_t(n_subchannels), nullptr);
    std::vector<int8_t *> p_ho(size_t(n_subchannels), nullptr);
    std::vector<uint8_t *> p_out(size_t(n_subchannels), nullptr);
    std::vector<int> ring_of(size_t(n_subchannels), -1);
    for (int i = 0; i < n_subchannels; i++) {
        for (size_t k = 0; k < p->rings.size(); k++)
            if (p->rings[k].start_address == sc[i].start_address && p->rings[k].length == sc[i].length && p->rings[k].n_streams == n_streams)
                ring_of[size_t(i)] = int(k);
        if (ring_of[size_t(i)] < 0) {
            Pipeline::Ring r{};
            r.start_address = sc[i].start_address;
            r.length = sc[i].length;
            r.n_streams = n_streams;
            r.bytes = size_t(n_streams) * 15 * size_t(sc[i].length) * CU_BITS;
            if (hipMalloc(reinterpret_cast<void **>(&r.buf[0]), r.bytes) != hipSuccess) return DABGPU_ERR_NOMEM;
            if (hipMalloc(reinterpret_cast<void **>(&r.buf[1]), r.bytes) != hipSuccess) { (void)hipFree(r.buf[0]); return DABGPU_ERR_NOMEM; }
            ring_of[size_t(i)] = int(p->rings.size());
            p->rings.push_back(r);
            HIP_TRY(hipMemsetAsync(r.buf[0], 0, r.bytes, sc_stream));
        }
        Pipeline::Ring &r = p->rings[size_t(ring_of[size_t(i)])];
        r.live = true;
        p_hi[size_t(i)] = r.buf[r.cur];
        p_ho[size_t(i)] = r.buf[r.cur ^ 1];
        p_out[size_t(i)] = static_cast<uint8_t *>(s.d_res) + out_off[size_t(i)];
    }
    // ---- upload (copy stream) ----
    const size_t nb_iq = (size_t(n_frames - 1) * p->frame_stride + size_t(NB_FRAME_SYMBOLS) * NB_SYM_PERIOD) * sizeof(float2);
    HIP_TRY(hipMemcpyAsync(s.d_iq, iq, nb_iq, hipMemcpyHostToDevice, p->s_up));
    if (freq_offset) HIP_TRY(hipMemcpyAsync(s.d_fo, freq_offset, sizeof(float) * size_t(n_frames), hipMemcpyHostToDevice, p->s_up));
    HIP_TRY(hipEventRecord(s.up, p->s_up));
    // ---- kernels (the context stream: everything a context computes stays in one order) ----
    HIP_TRY(hipStreamWaitEvent(sc_stream, s.up, 0));
    int rc;
    if (freq_offset)
        rc = dabgpu_ofdm_demod_frames_dev(ctx, s.d_iq, p->frame_stride, n_frames, static_cast<const float *>(s.d_fo),
                                          static_cast<int8_t *>(s.d_soft), nullptr, nullptr, sc_stream);
    else
        rc = dabgpu_ofdm_demod_streams_dev(ctx, s.d_iq, p->frame_stride, n_streams, frames_per_stream, fine_freq_update_beta,
                                           static_cast<int8_t *>(s.d_soft), nullptr, nullptr, sc_stream);
    uint8_t *d_fib = static_cast<uint8_t *>(s.d_res), *d_crc = d_fib + al(nb_fib);
    if (!rc)
        rc = dabgpu_decode_frames_dev(ctx, static_cast<const int8_t *>(s.d_soft), NB_FRAME_BITS, n_streams, frames_per_stream, d_fib,
                                      d_crc, sc, n_subchannels, n_subchannels ? p_hi.data() : nullptr,
                                      n_subchannels ? p_ho.data() : nullptr, n_subchannels ? p_out.data() : nullptr, sc_stream);
    if (rc) return rc;                                          // (the caller of this body drops the rings)
    HIP_TRY(hipEventRecord(s.comp, sc_stream));
    for (int i = 0; i < n_subchannels; i++) p->rings[size_t(ring_of[size_t(i)])].cur ^= 1;
    {
        size_t kept = 0;
        bool dropped = false;
        for (auto &r : p->rings) {
            if (r.live) { r.live = false; p->rings[kept++] = r; }
            else { if (!dropped) { (void)hipStreamSynchronize(sc_stream); dropped = true; } (void)hipFree(r.buf[0]); (void)hipFree(r.buf[1]); }
        }
        p->rings.resize(kept);
    }
    // ---- download (second copy stream) ----
    HIP_TRY(hipStreamWaitEvent(p->s_down, s.comp, 0));
    if (soft) {
        if (ctx->d_keep) {
            for (const dabgpu_bit_range &r : ctx->keep_ranges)
                HIP_TRY(hipMemcpy2DAsync(soft + r.first, NB_FRAME_BITS, static_cast<const int8_t *>(s.d_soft) + r.first, NB_FRAME_BITS,
                                         size_t(r.count), size_t(n_frames), hipMemcpyDeviceToHost, p->s_down));
        } else {
            HIP_TRY(hipMemcpyAsync(soft, s.d_soft, size_t(n_frames) * NB_FRAME_BITS, hipMemcpyDeviceToHost, p->s_down));
        }
    }
    HIP_TRY(hipMemcpyAsync(fib, d_fib, nb_fib, hipMemcpyDeviceToHost, p->s_down));
    HIP_TRY(hipMemcpyAsync(crc_ok, d_crc, nb_crc, hipMemcpyDeviceToHost, p->s_down));
    for (int i = 0; i < n_subchannels; i++)
        HIP_TRY(hipMemcpyAsync(out[i], p_out[size_t(i)], out_bytes[size_t(i)], hipMemcpyDeviceToHost, p->s_down));
    HIP_TRY(hipEventRecord(s.done, p->s_down));
    s.ticket = p->next_ticket;
    s.busy = true;
    *ticket = p->next_ticket++;
    return DABGPU_OK;
}

int dabgpu_pipe_wait(dabgpu_ctx *ctx, int64_t ticket) {
    if (!ctx || !ctx->pipe) return DABGPU_ERR_ARG;
    Pipeline *p = ctx->pipe;
    if (ticket < 0 || ticket >= p->next_ticket) return DABGPU_ERR_ARG;
    DeviceGuard guard(ctx);
    Pipeline::Slot &s = p->slots[size_t(ticket % int64_t(p->slots.size()))];
    if (s.ticket != ticket) return DABGPU_OK;                   // a later batch took the slot: this one left it long ago
    if (s.busy) { HIP_TRY(hipEventSynchronize(s.done)); s.busy = false; }
    return DABGPU_OK;
}

}  // extern "C"
