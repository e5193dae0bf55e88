// kernels.hpp -- launch interfaces of the hand-written gfx950 kernels (internal to libdabgpu).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>
#include <cstddef>
#include <cstdint>

namespace dabk {

// ---- OFDM front end (ofdm_kernels.hip) -------------------------------------
struct OfdmTables {
    const float2 *twiddle;     // [2048] exp(-2*pi*i*m/2048)
    const uint16_t *bin_of_n;  // [1536] FFT bin of data index n (mapper folded with carrier->bin)
    const uint16_t *n_of_vj;   // [12][64][2] data index of carrier registers 2jj, 2jj+1 of lane v (wave kernel)
};

// Per-stream tracking state kept in device memory between launches (== dabgpu_stream_state, 64 bytes): the
// frequency offsets OFDM_Demod shows through GetFineFrequencyOffset / GetCoarseFrequencyOffset
// (/root/reference/src/render_radio_block.cpp:202-207), its frame counters, and -- what RUNNING_FINE_TIME_SYNC
// maintains frame by frame (:196) -- where the stream's next frame starts.
struct StreamState {
    float fine_freq_offset;    // cycles/sample, within +-0.5 carrier
    float coarse_freq_offset;  // cycles/sample, whole carriers
    float signal_average;      // mean |re|+|im| of the stream's most recent frame
    float last_fine_error;     // residual the most recent update measured, cycles/sample
    int32_t total_frames_read;
    int32_t total_frames_desync;
    int32_t tracking;          // 1: next_frame_start is valid (set by track_start, cleared when every frame of a call is lost)
    int32_t last_time_offset;  // impulse-response peak position of the most recent frame relative to its predicted one, samples
    double next_frame_start;   // first sample (PRS prefix minus margin) of the next frame, relative to the NEXT capture
    float drift;               // samples per frame by which the frame period differs from 196608
    float last_peak_to_mean;
    // decision-directed fine loop (dd_loop_error below)
    int32_t loop_gated;        // calls whose fourth-power estimate was not used as it came (quality gate, branch hold)
    int32_t dd_branch;         // branch k of the most recent accepted estimate (residual = e_dd + k / (4 2552))
    int32_t dd_pending;        // a one-step departure from branch 0 seen once (DD_NO_BRANCH: none)
    int32_t reserved;
};

struct OfdmArgs {
    const float2 *iq;          // frame f at iq + f*frame_stride (first PRS sample)
    size_t frame_stride;       // complex samples
    const float *freq_offset;  // [n_frames] or nullptr
    int n_frames;
    int8_t *soft;              // [n_frames][230400]
    float2 *cyc;               // [n_frames][76] or nullptr
    // decision-directed frequency error (no cyclic prefix is read): [n_frames][76]; the sum of a frame's entries 1..75 =
    // sum over its data symbols and a fixed subset of 256 carriers (the ones nearest the centre) of (X_l conj X_{l-1})^4 (a run's total in the entry
    // of its last symbol, zeros in its others); entry 0 is never written.  Ignored when cyc is set.
    float2 *dd4 = nullptr;
    float2 *dqpsk;             // [n_frames][75][1536] or nullptr
    float2 *spectra;           // FFT-only mode: [n_frames][76][2048]
    // acquired mode (acq != nullptr): frame f belongs to stream f / acq_per_stream, which starts at
    // iq + stream*frame_stride; its first sample and frequency correction come from acq[f]; frames whose
    // flags are not 3 produce all-zero (erased) soft bits
    const struct AcquiredFrame *acq = nullptr;
    int acq_per_stream = 1;
    // soft-bit selection: [75][3] words, bit k of symbol s = write its 16-byte chunk k; nullptr = write everything
    const unsigned long long *keep = nullptr;
    // stream mode (state != nullptr): frame f belongs to stream f / frames_per_stream and is corrected by that
    // stream's fine + coarse offset; freq_offset is ignored
    const StreamState *state = nullptr;
    int frames_per_stream = 0;
    // the first `uncut_frames` frames of the launch are ONE run each (no symbol is transformed twice); only the frames
    // behind them -- the ones that would otherwise leave most of the chip idle at the end of the launch -- are cut
    // into `parts` runs
    int uncut_frames = 0;
};

// fused A2..A6.  Frames behind the first a.uncut_frames are cut into `parts` contiguous runs of data symbols (1..75);
// a run re-reads the symbol before it as differential reference.
hipError_t launch_ofdm_demod(const OfdmTables &t, const OfdmArgs &a, int parts, hipStream_t s);
// A2+A3 only; parts in 1..76.
hipError_t launch_fft_symbols(const OfdmTables &t, const OfdmArgs &a, int parts, hipStream_t s);
// Fine-frequency loop and counters of the stream call, after the demodulation launch on the same stream:
// per stream, fine -= beta * mean(arg cyc[frame][symbol]) / (2*pi*2048), wrapped to +-half a carrier; frame count;
// L1 level of the stream's last frame (first 4096 samples) into the running average, counted as a desync when it
// is below thr_null_start times the average so far.
// noise-like floats in [0.5, 1) with random signs: timing probes must not run on zeros (they move ~6 % faster)
hipError_t launch_fill_noise(void *p, size_t bytes, hipStream_t s);
// Copies done by a kernel instead of the copy engine, for page-locked host memory the device can address: behind (or in
// front of) the kernels of a one-frame call such a copy starts at once, where an engine copy costs ~8 us of hand-over on
// either side.  Up to three pieces per launch; sizes multiples of 16, pointers 16-byte aligned (hipErrorInvalidValue
// otherwise); a piece of 0 bytes is skipped.
struct CopyPiece {
    void *dst;
    const void *src;
    size_t bytes;
};
hipError_t launch_copy_pieces(const CopyPiece *pieces, int n, hipStream_t s);   // n <= copy_pieces_max()
// one thread stores `value` to `flag` (page-locked host memory, through its device alias): enqueued behind a call's last
// launch it tells a host that is watching the word that everything before it on the stream is done and visible
hipError_t launch_signal(unsigned long long *flag, unsigned long long value, hipStream_t s);
int copy_pieces_max();
// reads `in` and writes `out` at the same time (streaming, both whole): the launch is slower when the two buffers
// share an HBM domain -- what the placement helpers time
hipError_t launch_placement_probe(const void *in, size_t in_bytes, void *out, size_t out_bytes, hipStream_t s);
// the fused front end's loads and stores without its arithmetic, same run structure and occupancy (dabgpu_mover_frames_dev)
hipError_t launch_geometry_mover(const float2 *iq, size_t frame_stride, int n_frames, int8_t *soft, int uncut_frames, int parts,
                                 bool prefixes, hipStream_t s);
// dd != 0: `cyc` holds the front end's dd4 output instead; the error is angle(-sum of all entries l >= 1) / (4 * 2 pi * 2552)
// (dd_gate, dd_terms_per_frame: see dd_loop_error / TrackUpdateArgs)
hipError_t launch_stream_update(StreamState *state, const float2 *cyc, const float2 *iq, size_t frame_stride,
                                int n_streams, int frames_per_stream, float beta, float thr_null_start, float signal_beta,
                                int dd, float dd_gate, int dd_terms_per_frame, hipStream_t s);

// ---- synchronisation on the PRS (sync_kernels.hip) -----------------------------
struct SyncTables {
    const float2 *twiddle;     // [2048 + 576]: natural order, then the block FFT's compact tables (fft_common.hpp TW_TOTAL)
    const int8_t *prs_qt;      // [2048] quarter turns of the PRS per bin, -1 = not a carrier
    const uint16_t *pairs;     // [n_pairs] adjacent carrier pairs: bin | ((qt[bin+1]-qt[bin])&3) << 11
    int n_pairs;
    const float2 *pair_spectrum;   // [2048] FFT of S[b] = R[b+1] conj R[b] on carrier pairs (0 elsewhere): coarse search by FFT
};
struct SyncResult {            // == dabgpu_sync_result
    int32_t coarse_carriers;
    int32_t time_offset;
    float peak_to_mean;
    float coarse_peak_to_mean;
};
hipError_t launch_prs_sync(const SyncTables &t, const float2 *iq, size_t frame_stride, int n_frames,
                           const float *freq_offset, int max_coarse, SyncResult *out, hipStream_t s);

// Which tap of the channel impulse response a frame is aligned to (impulse_peak_distance_probability,
// /root/reference/src/render_radio_block.cpp:225):
//   score[n] = |h[n]|^2 * w(n)^2,  w(n) = 1 - (1 - distance_prob) * |t(n) - expected| / 2552,  t(n) the signed offset
//   peak = first maximum of the score;  peak_to_mean uses the unweighted |h[peak]|^2
//   first_path_rel > 0: the earliest tap within a cyclic prefix (504 samples) before the peak whose power is at least
//   max(first_path_rel * |h[peak]|^2, 16 * mean) replaces it -- the window is aligned to the first significant
//   path, so that a stronger late echo still falls inside the prefix.  0 = the reference's rule (weighted maximum).
struct PeakRule {
    float distance_prob = 1.0f;
    int expected = 0;
    float first_path_rel = 0.0f;
};

// ---- acquisition on unaligned captures (sync_kernels.hip) ----------------------
struct AcquiredFrame {         // == dabgpu_acquired_frame (32 bytes)
    int64_t start;             // first sample the demodulator treats as PRS cyclic prefix, relative to the stream
    float freq_offset;         // fine_offset - coarse_carriers/2048: what the demodulator applies
    int32_t coarse_carriers;
    float fine_offset;
    float peak_to_mean;
    float coarse_peak_to_mean;
    int32_t flags;             // bit 0 locked, bit 1 whole frame inside the capture
};
struct AcquireArgs {
    const float2 *iq;          // stream s at iq + s*stream_stride
    size_t stream_stride;      // complex samples
    int n_streams;
    int64_t n_samples;         // per stream
    float thr_start, thr_end;  // null-symbol dip thresholds relative to the (local) mean block L1
    int level_chunk;           // blocks per chunk of the local level estimate (power of two, 64..16384); 0 = capture mean
    const double *chunk_mean;  // scratch [n_streams][ceil(nb / level_chunk)], set by launch_acquire
    int min_blocks;            // shortest dip (64-sample blocks) accepted as a null symbol
    int max_coarse;            // carriers
    float min_peak_to_mean;
    int margin;                // samples the FFT windows are kept inside the cyclic prefix
    PeakRule rule;
    int max_out;               // frames per stream
    float *l1;                 // scratch [n_streams][n_samples/64]
    int64_t *cands;            // scratch [n_streams][max_out]
    AcquiredFrame *out;        // [n_streams][max_out]; entries >= counts[s] get flags 0, start -1
    int32_t *counts;           // [n_streams]
    // auto-acquisition inside a tracked call: streams whose state says tracking == 1 are left alone entirely (nothing
    // read, nothing written -- their rows of `out` and `counts` belong to the tracking pass)
    const StreamState *skip_tracked = nullptr;
};
size_t acquire_scratch_bytes(int n_streams, int64_t n_samples, int max_out);
hipError_t launch_acquire(const SyncTables &t, const AcquireArgs &a, hipStream_t s);

// ---- per-stream timing tracking (sync_kernels.hip) -------------------------------
// Batch mode (fixed_start = 0): frame slot i of stream s is predicted from the stream's state at
//   p_i = next_frame_start + (j0 + i) * (196608 + drift),  j0 = frames that would start before the capture (missed),
// synchronised on its PRS at llrint(p_i) (impulse response with the state's net frequency offset applied, PeakRule)
// and written as an AcquiredFrame for the demodulator; slots whose frame (+512 samples) does not fit the capture, and
// every slot of a stream that is not tracking, get flags 0 / start -1.
// Frame mode (fixed_start = 1; the host mirror's one frame per call): the frame is the stream's first 76*2552 samples
// as the host assembled it (max_out = 1, start stays 0); max_coarse > 0 also runs the whole-carrier search:
// acquiring != 0 SETS the state's coarse offset from it (the search then runs with the fine offset alone), otherwise a
// residual of k carriers moves it by coarse_slow_beta * k; bit 1 of flags = the window lies inside the prefix
// (0 <= time offset <= 488).
struct TrackArgs {
    StreamState *state;
    const float2 *iq;          // stream s at iq + s*stream_stride
    size_t stream_stride;
    int n_streams;
    int64_t n_samples;
    int max_out;
    int margin;
    float min_peak_to_mean;
    PeakRule rule;             // expected is set to `margin` by the launcher
    int fixed_start;
    int max_coarse;
    int acquiring;
    float coarse_slow_beta;
    AcquiredFrame *out;        // [n_streams][max_out]
    SyncResult *sync_out;      // [n_streams][max_out] or nullptr
    // an upload riding in the same launch (the one-frame call): extra workgroups copy copy_n16 16-byte words from copy_src
    // (page-locked host memory) to copy_dst while the first ones synchronise on `sync_iq` -- the PRS read straight from
    // the host buffer -- so the 20 us of the synchronisation disappear inside the 37 us of the upload without a second
    // stream (a cross-stream event costs ~11 us on this runtime: profiles/r05_frame_path.md)
    const float2 *sync_iq = nullptr;     // where the synchronisation reads its samples (nullptr: `iq`)
    uint4 *copy_dst = nullptr;
    const uint4 *copy_src = nullptr;
    unsigned copy_n16 = 0;
    int copy_blocks = 0;                 // set by the launcher
};
hipError_t launch_track_sync(const SyncTables &t, const TrackArgs &a, hipStream_t s);

// After the demodulation of those frames, one workgroup per stream:
//   fine loop   fine -= beta * mean(arg cyc) / (2 pi 2048) over the locked frames, wrapped to +-half a carrier
//   timing      residuals r_i = start_i - p_i of the n locked frames; with n >= 2 a least-squares line
//               r = alpha + slope*i, drift += drift_beta * min(1, n/4) * slope; with n = 1: alpha = r, slope = 0 and
//               (one slot per call only) drift += drift_beta/8 * r;
//               next_frame_start = p(count) + alpha + slope*count - advance
//               (count = frame slots inside the capture, p(count) with the OLD drift)
//   counters    total_frames_read += locked, total_frames_desync += missed + unlocked; no locked frame at all in a
//               call that had frames: tracking = 0
//   level       L1 mean of the first 4096 samples of the last locked frame into the running average
//               (signal_beta * average + (1 - signal_beta) * level); a level below thr_null_start * average counts one
//               more desync and leaves the average alone
//   counts[s]   = count
// The decision-directed fine-frequency error of a call (restated by oracle.dd_loop_error).
//   (sx, sy)  sum of u^4 over the call's differential symbols (unit magnitude each), n_terms of them
//   e_cp      mean angle of the call's PRS cyclic-prefix correlations / (2 pi 2048): the same residual, coarser, but
//             unambiguous within half a carrier
// The fourth-power estimate e_dd = angle(-sum) / (4 2 pi 2552) repeats every 1 / (4 2552) cycles per sample (0.2 carriers);
// e_cp picks its branch k.  Two things can make that wrong, and both are gated:
//   quality   |sum| is compared with what n_terms random unit phasors add up to (sqrt(n_terms)): below `gate` times that
//             (2.5 by default; 0 = off) the sum carries no usable phase -- single frames below ~3 dB SNR, an empty selection
//             -- and the call takes e_cp alone: unbiased, coarser.  (At s = |sum| / sqrt(n_terms) the estimate's spread
//             is 0.0225 / s carriers; a single PRS prefix has ~0.008-0.012 at 3-0 dB: the two cross near s = 2-3, and a
//             pure-noise sum exceeds 2.5 once in 500 calls.  profiles/r04_loop_gate.txt has the table: 8 was too
//             cautious -- at 3 dB it threw away an estimate twice as good as the prefix's.)
//   branch    a stream that was locked (previous branch 0: residual inside +-0.1 carriers) and now asks for branch +-1
//             has, far more often than a real 200 Hz jump between two calls, a PRS prefix hit by a fade or an impulse:
//             the departure must be seen on two consecutive calls before it is believed; the first time the branch is
//             held at 0.  (While a stream pulls in -- previous branch not 0, or its first call -- everything is believed.)
// Either gate counts in state.loop_gated.
constexpr int32_t DD_NO_BRANCH = 0x7fffffff;
__host__ __device__ inline float dd_loop_error(double sx, double sy, float e_cp, double n_terms, float gate, bool first_call,
                                               StreamState &st) {
    const double mag2 = sx * sx + sy * sy;
    if (!(n_terms > 0.0) || mag2 < double(gate) * double(gate) * n_terms) {
        st.loop_gated += 1;
        st.dd_pending = DD_NO_BRANCH;
        st.dd_branch = int32_t(rintf(e_cp * (4.0f * 2552.0f)));
        return e_cp;
    }
    const float e_dd = float(atan2(-sy, -sx) / (4.0 * 6.283185307179586 * 2552.0));
    int32_t k = int32_t(rintf((e_cp - e_dd) * (4.0f * 2552.0f)));
    if (!first_call && st.dd_branch == 0 && (k == 1 || k == -1) && st.dd_pending != k) {
        st.dd_pending = k;
        st.loop_gated += 1;
        k = 0;
    } else {
        st.dd_pending = DD_NO_BRANCH;
    }
    st.dd_branch = k;
    return e_dd + float(k) * (1.0f / (4.0f * 2552.0f));
}

struct TrackUpdateArgs {
    StreamState *state;
    const AcquiredFrame *frames;
    const float2 *cyc;         // [n_streams*max_out][76]
    int dd = 0;                // cyc holds dd4 sums (see OfdmArgs::dd4)
    const float2 *iq;
    size_t stream_stride;
    int n_streams;
    int64_t n_samples;
    int max_out;
    int64_t advance;           // samples between the first sample of this capture and of the next one
    float fine_beta, drift_beta, signal_beta, thr_null_start;
    int fixed_start;
    int32_t *counts;           // [n_streams] or nullptr
    int settle_only = 0;       // only turn "started in this call" marks (tracking == 2) into 1
    float dd_gate = 2.5f;      // dd_loop_error's quality gate
    int dd_terms_per_frame = 19200;   // unit terms one frame adds to the sums (256 carriers x the symbols that are demodulated)
    // the one-frame call's download riding in this launch: workgroups behind the n_streams updating ones copy `down` (what
    // the launches before this one produced: soft bits, frame and sync records); the updating workgroup writes the new state
    // to `state_out` as well (page-locked host memory) -- one launch instead of two at the end of the call
    CopyPiece down[3] = {{nullptr, nullptr, 0}, {nullptr, nullptr, 0}, {nullptr, nullptr, 0}};   // (the third: the constellation, when asked for)
    StreamState *state_out = nullptr;
    int copy_blocks = 0;       // set by the launcher
};
hipError_t launch_track_update(const TrackUpdateArgs &a, hipStream_t s);
// Start tracking from an acquisition result (dabgpu_acquire_dev on the same capture): per stream, a least-squares line
// through the starts of the locked frames against their frame number round((start - first)/196608) gives the drift
// (0 with fewer than 4 locked frames); next_frame_start = start_last + 196608 + drift - advance; fine offset = mean of
// the locked frames', coarse = the last locked frame's; tracking = 1 (0 when no frame locked).
// only_lost != 0: streams that are tracking keep their state; the others start with tracking = 2 ("started in this
// call"), which the track_update launch behind it turns into 1 without touching anything else.
hipError_t launch_track_start(StreamState *state, const AcquiredFrame *frames, const int32_t *counts, int n_streams,
                              int max_out, int64_t advance, int only_lost, hipStream_t s);

// ---- DAB+ audio super-frame (dabplus_kernels.hip) ------------------------------
struct SuperframeStatus {      // == dabgpu_superframe_status
    int32_t firecode_ok;
    int32_t rs_corrected;
    int32_t rs_uncorrectable;
    int32_t num_aus;
    int32_t au_crc_mask;
    int32_t au_start[8];
    int32_t reserved[3];
};
// `done_flag` (optional, ONE super-frame only): the launch's single workgroup stores `done_seq` there behind its results --
// the word a host thread watches (dabgpu_ctx.hpp wait_for_signal) -- instead of a launch of its own behind the kernel
hipError_t launch_dabplus_superframes(const uint8_t *in, size_t in_stride, int n_superframes, int s, uint8_t *out,
                                      SuperframeStatus *status, hipStream_t stream, unsigned long long *done_flag = nullptr,
                                      unsigned long long done_seq = 0);

// ---- channel decoder (viterbi_kernels.hip) ---------------------------------
struct CodeTables {
    const uint16_t *mother_pos;  // [n_punct] mother-bit position of punctured bit i
    int n_punct;
    int nsteps;                  // trellis steps (info bits + 6)
    const uint8_t *prbs_bytes;   // [(nsteps-6)/8] energy-dispersal bytes, or nullptr = no descramble
    // codewords of 96 k + 6 steps (the rot kernel's): [k + 2] -- chunk_first[c] = punctured bits whose mother position lies
    // below step 96 c (c <= k), chunk_first[k + 1] = n_punct: the punctured bits of the 96-step chunk c are
    // [chunk_first[c], chunk_first[c + 1]), the last entry's the six tail steps'.  nullptr for other lengths.
    const uint16_t *chunk_first = nullptr;
};
// where build_device_code puts the chunk table: behind the positions, in the same allocation
__host__ __device__ inline int code_chunk_table_offset(int n_punct) { return (n_punct + 7) & ~7; }

// A8..A11: FIC of n_frames frames.
hipError_t launch_fic_decode(const CodeTables &c, const int8_t *soft, size_t soft_stride, int n_frames,
                             uint8_t *fib, uint8_t *crc_ok, hipStream_t s);
// A9 on contiguous punctured codewords.
hipError_t launch_viterbi_plain(const CodeTables &c, const int8_t *punct, int n_codewords, uint8_t *out,
                                hipStream_t s);
// A12: one subchannel, time de-interleave fused into the fetch.
struct MscArgs {
    const int8_t *soft;
    size_t soft_stride;
    int n_streams;
    int frames_per_stream;
    int start_bit;             // start_address * 64
    int nbits;                 // length * 64 (== n_punct)
    const int8_t *hist_in;     // [n_streams][15][nbits] or nullptr
    int8_t *hist_out;          // [n_streams][15][nbits] or nullptr
    uint8_t *out;              // [n_streams][frames*4][(nsteps-6)/8]
};
hipError_t launch_msc_decode(const CodeTables &c, const MscArgs &a, hipStream_t s);
// Small batches of a whole multiplex: every sub-channel's codewords in one launch of the wave-per-codeword kernel and
// all history rings in a second one (instead of one pair of launches per sub-channel).  nsteps must pass
// wave_group_supported() (every DAB profile does).
struct WaveGroupItem {
    CodeTables code;
    MscArgs args;
};
// the FIC of the same frames, decoded by the same launch (nullptr = not)
struct WaveFicItem {
    CodeTables code;
    const int8_t *soft;
    size_t soft_stride;
    int n_frames;
    uint8_t *fib, *crc_ok;
};
bool wave_group_supported(int nsteps);
// true when `n_waves` codewords whose longest takes `max_nsteps` trellis steps are all resident at once (LDS per wave x
// waves <= the chip's LDS): a grouped launch then finishes with its longest codeword instead of queueing kinds up
bool wave_group_one_round(int max_nsteps, long n_waves);
hipError_t launch_msc_decode_group(const WaveGroupItem *items, int n, hipStream_t s, const WaveFicItem *fic = nullptr);

// ---- large-batch variant: one codeword per lane (viterbi_lane_kernels.hip) ----
struct LaneScratch {
    void *base;
    size_t bytes;
    bool unfused = false;      // DABGPU_FLAG_LANE_UNFUSED: depuncture in a pass of its own (lane_prep_kernel)
};
size_t lane_scratch_bytes(int nsteps, int n_codewords);
bool lane_supported(int nsteps);
// per-profile tables of the lane kernels: punct_idx [4*nsteps] (punctured index of each mother bit, -1 = erased);
// fused_desc [4*nsteps] and fused_tiles [2*ceil(nsteps/24)] from build_lane_fused_tables (may be null: prep path)
struct LaneTables {
    const int32_t *punct_idx;
    const int32_t *fused_desc;
    const int32_t *fused_tiles;
};
void build_lane_fused_tables(const uint8_t *mask, int nsteps, std::vector<int32_t> &desc, std::vector<int32_t> &tiles);
hipError_t launch_fic_decode_lane(const CodeTables &c, const LaneTables &lt, const int8_t *soft, size_t soft_stride,
                                  int n_frames, const LaneScratch &sc, uint8_t *fib, uint8_t *crc_ok, hipStream_t s);
hipError_t launch_viterbi_plain_lane(const CodeTables &c, const LaneTables &lt, const int8_t *punct,
                                     int n_codewords, const LaneScratch &sc, uint8_t *out, hipStream_t s);
hipError_t launch_msc_decode_lane(const CodeTables &c, const LaneTables &lt, const MscArgs &a, const LaneScratch &sc,
                                  hipStream_t s);
// Grouped launch: the FIC and/or several sub-channels of the same frames, each with its own profile, in ONE forward
// and ONE traceback launch.  Sub-channel items must pass lane_group_fusable(); all items lane_supported(nsteps).
// For the FIC item only args.soft / soft_stride / n_streams / frames_per_stream / out (= FIBs) and crc_ok are used.
// The history rings are updated by the caller (launch_msc_history per sub-channel).
struct LaneGroupItem {
    CodeTables code;
    LaneTables tables;
    MscArgs args;
    bool is_fic;
    uint8_t *crc_ok;
};
bool lane_group_fusable(const MscArgs &a);
size_t lane_group_scratch_bytes(const LaneGroupItem *items, int n);
// mid (optional, timing only): two events, recorded behind the forward pass and behind the traceback of the last pack
hipError_t launch_lane_group(const LaneGroupItem *items, int n, const LaneScratch &sc, hipStream_t s, hipEvent_t *mid = nullptr);
// Dynamic-LDS request (>= lds) that makes every CU hold the same number of workgroups of a `grid`-workgroup
// launch when at most `o_cap` fit per CU otherwise (the dispatcher fills CUs greedily).
size_t balanced_lds_bytes(unsigned grid, size_t lds, unsigned o_cap);
// history ring update alone (used by both MSC variants)
hipError_t launch_msc_history(const MscArgs &a, hipStream_t s);

// Let every kernel that takes dynamic LDS use the whole 160 KB of a CU: set once per context creation (on the
// context's device) instead of per launch.
hipError_t init_viterbi_kernel_attributes();
hipError_t init_lane_kernel_attributes();

// LDS bytes one codeword needs in the first (fallback) wave-per-codeword kernel
inline size_t viterbi_wave_lds_bytes(int nsteps) { return size_t(nsteps) * 12 + 64; }
// largest trellis the wave-per-codeword kernels take: it must fit one CU's 160 KB of LDS (8 B/step for DAB codeword
// lengths) and its mother-bit positions 16 bits (CodeTables::mother_pos) -- about 680 kbit/s.  Longer codewords go to
// the lane kernels, which keep survivors in HBM.
inline bool viterbi_fits(int nsteps) {
    const bool rot = nsteps >= 102 && (nsteps - 6) % 96 == 0;
    return 4 * size_t(nsteps) <= 65535 && (rot ? size_t(nsteps) * 8 + 4096 : viterbi_wave_lds_bytes(nsteps)) <= 160 * 1024;
}

}  // namespace dabk
