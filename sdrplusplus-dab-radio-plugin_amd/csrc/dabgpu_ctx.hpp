// dabgpu_ctx.hpp -- the context behind the opaque dabgpu_ctx handle and the helpers the translation units of the C ABI
// share (dabgpu_api.hip: context + front end, dabgpu_decode_api.hip: channel decoder, dabgpu_placement.hip: frame
// buffers, dabgpu_pipeline.hip: the host-fed ring).  Internal to libdabgpu.
#pragma once
#include "../../include/dabgpu.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <map>
#include <memory>
#include <vector>

#include "dab_tables.hpp"
#include "kernels.hpp"

namespace dabapi {

struct DeviceCode {
    dab::PunctureProfile prof;
    uint16_t *d_mother_pos = nullptr;
    uint8_t *d_prbs = nullptr;
    int32_t *d_punct_idx = nullptr;      // [4*nsteps] punctured index of each mother bit, -1 = erased (lane kernels)
    int32_t *d_fused_desc = nullptr, *d_fused_tiles = nullptr;   // fused lane forward pass (build_lane_fused_tables)
    dabk::LaneTables lane_tables() const { return dabk::LaneTables{d_punct_idx, d_fused_desc, d_fused_tiles}; }
    dabk::CodeTables tables(bool descramble) const {
        const bool chunked = d_mother_pos && prof.nsteps >= 102 && (prof.nsteps - 6) % 96 == 0;
        return dabk::CodeTables{d_mother_pos, prof.n_punct, prof.nsteps, descramble ? d_prbs : nullptr,
                                chunked ? d_mother_pos + dabk::code_chunk_table_offset(prof.n_punct) : nullptr};
    }
};

// hipEvent pairs around the most recent launches of one kernel family (a ring: the launches are asynchronous, so
// the pairs can only be read back after the stream has caught up)
constexpr int TIMER_RING = 32;
struct Timer {
    hipEvent_t start[TIMER_RING] = {}, stop[TIMER_RING] = {};
    // the grouped lane decode is two kernels (+ the history copy) inside one timed call: two more events split it into
    // forward pass | traceback | history (dabgpu_mean_kernel_ms 4 / 5 / 6 read slot 2 through them)
    hipEvent_t mid[TIMER_RING][2] = {};
    bool has_mid[TIMER_RING] = {};
    long recorded = 0;                   // launches timed since timing was switched on
};

struct Pipeline;                         // dabgpu_pipeline.hip

// The domain-aware frame buffers of a context live inside TWO address ranges, both reserved by the first such allocation
// and given back by dabgpu_destroy only (dabgpu_placement.hip): `probe` is where physical chunks are mapped while their
// HBM domains are found, `pair` is where the pair handed to the caller is mapped.  (Two, not one: access rights are only
// ever set on a span that begins at the base of a reservation -- the one use of the virtual-memory API that was never
// seen to fail on this runtime.)
struct Arena {
    char *probe = nullptr, *pair = nullptr;
    size_t probe_bytes = 0, pair_bytes = 0;
    struct Piece {
        size_t off, bytes;               // offset inside `pair`, mapped bytes
        hipMemGenericAllocationHandle_t h;
    };
    std::vector<Piece> pieces;           // live mappings of the pair handed out (empty: the arena is idle)
    void *d_iq = nullptr, *d_soft = nullptr;
};

}  // namespace dabapi

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
    dabapi::DeviceCode fic;
    std::map<std::vector<uint8_t>, std::unique_ptr<dabapi::DeviceCode>> codes;   // keyed by puncture mask
    // ... and found again by the sub-channel's descriptor (form, profile, level, bit rate, size) without building the
    // 4 x (24 x bitrate + 6) flags of its puncture mask first: the plugin's per-frame decode call built it three times
    // per sub-channel (validation, grouping, code look-up), ~10 us of host time in front of and between its launches
    std::map<uint64_t, dabapi::DeviceCode *> code_by_descriptor;
    // slots 0..5: staging of the host-pointer entry points; slot 6: the stream / tracked / frame calls' own loop input
    // (correlations or decision-directed sums).  One caller stream at a time per context (dabgpu.h, conventions).
    void *d_stage[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t stage_bytes[7] = {0, 0, 0, 0, 0, 0, 0};
    bool timing = false;
    dabapi::Timer timers[4];
    int ofdm_parts_override = 0;
    const unsigned long long *d_keep = nullptr;          // current soft-bit selection table ([75][3] words) or nullptr
    std::vector<void *> keep_tables;                     // every table handed to a kernel so far (freed on destroy)
    void *d_acq_scratch = nullptr;       // block norms + candidates of dabgpu_acquire
    size_t acq_scratch_bytes = 0;
    void *d_lane_scratch = nullptr;      // work buffers of the codeword-per-lane Viterbi
    size_t lane_scratch_bytes = 0;
    int lane_mode = -1;                  // 1 = DABGPU_FLAG_VITERBI_LANE, 0 = DABGPU_FLAG_VITERBI_WAVE, -1 by batch size
    bool lane_unfused = false;           // DABGPU_FLAG_LANE_UNFUSED
    bool test_one_domain = false;        // DABGPU_FLAG_TEST_ONE_DOMAIN (dabgpu_placement.hip)
    int test_fail_in = 0;                // dabgpu_test_fail_frame_call: one-frame calls left until one reports DABGPU_ERR_HIP
    int wave_slots = 3072;               // resident OFDM wavefronts: 12 per CU
    std::vector<dabgpu_bit_range> keep_ranges;           // the current selection, merged, for the host-pointer copy-back
    dabk::StreamState *d_states = nullptr;               // per-stream tracking state (dabgpu_streams_reset)
    int n_states = 0;
    hipEvent_t ev_states = nullptr;      // recorded behind the last launch that reads or writes d_states, on ITS stream
    bool ev_states_pending = false;
    float thr_null_start = 0.35f;        // desync threshold of the stream call (null_l1_search.thresh_null_start)
    float signal_beta = 0.95f;           // signal_l1.update_beta of the stream call
    bool loop_dd = false;                // stream call without a correlation buffer: decision-directed fine loop
    float dd_gate = 2.5f;                // decision-directed loop: quality gate (dabgpu_set_loop_gate, dabk::dd_loop_error)
    int keep_symbols = 75;               // data symbols per frame the current selection demodulates (75: no selection)
    // dabgpu_decode_stream_frames: de-interleaver rings of the stream's sub-channels, kept on the device between calls
    struct SubHistory {
        int start_address, length;
        size_t bytes;
        int8_t *ring[2];
        int cur;
        bool live;                       // decoded in the current call (a ring that misses a frame is stale: dropped)
    };
    std::vector<SubHistory> sub_history;
    void *h_bounce = nullptr;            // page-locked landing area of that call's single download (+ 64 bytes: the
    size_t h_bounce_bytes = 0;           // word wait_for_signal watches sits behind the payload)
    unsigned long long signal_seq = 0;   // number of the last one-frame call that ended through wait_for_signal
    dabapi::Arena arena;                 // dabgpu_alloc_frame_buffers(DABGPU_PLACE_DOMAINS)
    dabapi::Pipeline *pipe = nullptr;    // dabgpu_pipe_open
};

namespace dabapi {

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

struct ScopedTimer {
    dabgpu_ctx *ctx;
    int which;
    hipStream_t s;
    ScopedTimer(dabgpu_ctx *c, int w, hipStream_t st) : ctx(c), which(w), s(st) {
        if (ctx->timing) {
            Timer &t = ctx->timers[which];
            const int i = int(t.recorded % TIMER_RING);
            if (!t.start[i]) { (void)hipEventCreate(&t.start[i]); (void)hipEventCreate(&t.stop[i]); }
            t.has_mid[i] = false;
            (void)hipEventRecord(t.start[i], s);
        }
    }
    // the two events a launch sequence records between its kernels (nullptr while timing is off)
    hipEvent_t *mids() {
        if (!ctx->timing) return nullptr;
        Timer &t = ctx->timers[which];
        const int i = int(t.recorded % TIMER_RING);
        for (hipEvent_t &e : t.mid[i])
            if (!e && hipEventCreate(&e) != hipSuccess) return nullptr;
        t.has_mid[i] = true;
        return t.mid[i];
    }
    ~ScopedTimer() {
        if (ctx->timing) {
            Timer &t = ctx->timers[which];
            (void)hipEventRecord(t.stop[int(t.recorded % TIMER_RING)], s);
            t.recorded++;
        }
    }
};

// The end of a one-frame call (dabgpu_ofdm_demod_stream_frame, dabgpu_decode_stream_frames): ~55 us of device work are in
// flight and the caller can do nothing until they are done.  hipStreamSynchronize puts the thread to sleep and pays the
// wake-up; instead a one-thread launch behind the call's last kernel stores the call's number into a word of the page-locked
// landing area and the host watches that word -- for at most SIGNAL_SPIN_US (no frame call takes that long on an idle
// device; a core is never held longer), then it sleeps on the stream after all.
// `flag_host` / `flag_dev`: the two addresses of the word; `seq`: this call's number.
// (already_signalled: the last kernel of the call stores the word itself -- a single-workgroup launch can)
// results_coherent: the kernels of these calls write their results straight into page-locked HOST memory, and
// a watched word orders nothing for memory the device does not write coherently: HIP guarantees host visibility of kernel
// writes to hipHostRegister'ed or non-coherent allocations only at a stream / event synchronisation.  The word is watched
// only when every such buffer is KNOWN to be coherent -- the context's own landing area and buffers from dabgpu_host_alloc
// (known_coherent_host) -- and the call ends in hipStreamSynchronize otherwise.
constexpr long SIGNAL_SPIN_US = 200;
inline int wait_for_signal(hipStream_t s, volatile unsigned long long *flag_host, unsigned long long *flag_dev, unsigned long long seq,
                           bool already_signalled = false, bool results_coherent = true) {
    if (!results_coherent) return hipStreamSynchronize(s) == hipSuccess ? DABGPU_OK : DABGPU_ERR_HIP;
    if (!already_signalled && dabk::launch_signal(flag_dev, seq, s) != hipSuccess) return DABGPU_ERR_HIP;
    const auto give_up = std::chrono::steady_clock::now() + std::chrono::microseconds(SIGNAL_SPIN_US);
    for (;;) {
        for (int i = 0; i < 64; i++) {
            if (*flag_host == seq) {
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                return DABGPU_OK;
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
        }
        if (std::chrono::steady_clock::now() >= give_up) break;
    }
    return hipStreamSynchronize(s) == hipSuccess ? DABGPU_OK : DABGPU_ERR_HIP;
}

// dabgpu_test_fail_frame_call: true exactly once, on the armed one-frame call
inline bool injected_failure(dabgpu_ctx *ctx) {
    return ctx->test_fail_in > 0 && --ctx->test_fail_in == 0;
}

inline hipStream_t pick_stream(dabgpu_ctx *ctx, void *stream) {
    return stream ? reinterpret_cast<hipStream_t>(stream) : ctx->stream;
}

// defined in dabgpu_api.hip
int stage(dabgpu_ctx *ctx, int slot, size_t bytes, void **out);
int get_code(dabgpu_ctx *ctx, dab::PunctureProfile &&prof, DeviceCode **out);
void free_device_code(DeviceCode &dc);
// the address the device reaches page-locked host memory under (hipHostMalloc / hipHostRegister: dabgpu_host_alloc), or
// nullptr for any other pointer
void *device_alias_of_pinned(const void *host);
// [host, host + bytes) lies inside one allocation of dabgpu_host_alloc (coherent page-locked memory: a kernel's writes to it
// are visible to a host that watched a word the kernel stored afterwards)
bool known_coherent_host(const void *host, size_t bytes);
// the context's page-locked landing area (coherent), at least `bytes` + the 64 bytes of the watched word, zeroed when (re)allocated
int ensure_bounce(dabgpu_ctx *ctx, size_t bytes);
int note_state_use(dabgpu_ctx *ctx, hipStream_t s);
int wait_state_use(dabgpu_ctx *ctx);
void stats_of(const dabk::StreamState &st, dabgpu_stats *out);
// defined in dabgpu_placement.hip / dabgpu_pipeline.hip: what dabgpu_destroy calls
void arena_destroy(dabgpu_ctx *ctx);
void pipeline_destroy(dabgpu_ctx *ctx);

// No capacity unit of the CIF used twice (and none outside it): the state-keeping entry points (dabgpu_decode_stream_frames,
// dabgpu_pipe_submit) key their de-interleaver rings by (start, size), so a sub-channel listed twice would share one ring --
// both entries reading and writing the same pair of buffers, the ring flipped twice -- and the NEXT call would continue from
// a stale buffer.  Checked before any ring is touched: a refused call leaves the kept state as it was.
inline bool subchannels_disjoint(const dabgpu_subchannel *sc, int n) {
    char used[864] = {};
    for (int i = 0; i < n; i++) {
        if (sc[i].start_address < 0 || sc[i].length <= 0 || sc[i].start_address + sc[i].length > 864) return false;
        for (int cu = sc[i].start_address; cu < sc[i].start_address + sc[i].length; cu++) {
            if (used[cu]) return false;
            used[cu] = 1;
        }
    }
    return true;
}

// The codeword-per-lane Viterbi pays once a launch has enough codewords to give every SIMD a wave (one wave =
// 64 codewords; its single-wave latency equals the wave-per-codeword kernels' time at ~24k codewords).
constexpr int LANE_MIN_CODEWORDS = 24576;

}  // namespace dabapi
