// sync_kernels.hip -- frame synchronisation on the phase reference symbol: what the
// reference's OFDM_Demod does in its RUNNING_COARSE_FREQ_SYNC and RUNNING_FINE_TIME_SYNC states
// (/root/reference/src/render_radio_block.cpp:195-196; knobs is_coarse_freq_correction,
// max_coarse_freq_correction_norm, impulse_peak_threshold_db at :213-231).
//
// One 256-thread workgroup per candidate frame:
//   X  = FFT2048(nco * x[504 .. 2552))                       (window assumed aligned to the useful part)
//   Q[b] = X[b+1] conj X[b]          differential between adjacent bins: immune to the unknown timing offset
//   D_k  = sum over adjacent carrier pairs b of Q[b+k] conj(S[b]),  S[b] = R[b+1] conj R[b] in {1,j,-1,-j}
//   coarse offset  k^ = argmax |D_k|^2, |k| <= max_coarse          (integer carriers)
//   h   = IFFT(X[b+k^] conj R[b])  channel impulse response;  time offset = argmax |h|^2 (signed)
// Both FFTs are the 256-thread LDS Stockham of fft_common.hpp; the correlation is adds only because R is a
// fourth root of unity.  Results are integers plus two peak-to-mean ratios for thresholding.
#include <algorithm>
#include <type_traits>

#include "kernels.hpp"
#include "dab_tables.hpp"
#include "fft_common.hpp"
#include "mem_stream.hpp"

namespace dabk {

using namespace dab;

namespace {

struct SyncLds {
    double red_d[2][WG];        // acquisition: cyclic-prefix correlation
    float2 tw[TW_TOTAL];        // natural table + the radix-8 passes' compact ones (fft_common.hpp)
    float2 t1[NB_FFT];
    float2 x[NB_FFT];
    float2 y[NB_FFT];           // coarse search: spectrum of Q, then the correlation over all 2048 shifts
    int8_t qt[NB_FFT];          // quarter turns of the PRS per bin, -1 = not a carrier
    float red_m[WG];
    int red_i[WG];
    float red_s[WG];
};
// The tracking pass of a batch call needs neither the whole-carrier search (no y) nor the cyclic-prefix estimate (no
// red_d), and reads its twiddles from the L1/L2-resident global table: 37 KB instead of 73 KB, four workgroups per CU
// instead of two -- the kernel is all barriers and latency, occupancy is what it runs on.
struct SyncLdsLite {
    float2 t1[NB_FFT];
    float2 x[NB_FFT];
    int8_t qt[NB_FFT];
    float red_m[WG];
    int red_i[WG];
    float red_s[WG];
};

// X * (-j)^q  (= X * conj(R) for R = j^q)
__device__ __forceinline__ float2 rot_mq(float2 v, int q) {
    switch (q & 3) {
    case 0: return v;
    case 1: return make_float2(v.y, -v.x);
    case 2: return make_float2(-v.x, -v.y);
    default: return make_float2(-v.y, v.x);
    }
}

// block-wide argmax with "first maximum" semantics (smaller index wins ties) and the sum of all values.  Inside a wave the
// reduction is a butterfly of lane exchanges (no barrier); the four waves' results meet in LDS: two barriers per call instead
// of the ten of a 256-wide LDS tree -- this kernel is all barriers and latency.
template <class Lds>
__device__ __forceinline__ void block_argmax_sum(Lds &sm, int tid, float m, int idx, float s, float &best_m,
                                                 int &best_i, float &total) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float m2 = __shfl_xor(m, off);
        const int i2 = __shfl_xor(idx, off);
        s += __shfl_xor(s, off);
        if (m2 > m || (m2 == m && i2 < idx)) { m = m2; idx = i2; }
    }
    if ((tid & 63) == 0) {
        sm.red_m[tid >> 6] = m;
        sm.red_i[tid >> 6] = idx;
        sm.red_s[tid >> 6] = s;
    }
    __syncthreads();
    best_m = sm.red_m[0];
    best_i = sm.red_i[0];
    total = sm.red_s[0];
#pragma unroll
    for (int w = 1; w < WG / 64; w++) {
        const float m2 = sm.red_m[w];
        const int i2 = sm.red_i[w];
        if (m2 > best_m || (m2 == best_m && i2 < best_i)) { best_m = m2; best_i = i2; }
        total += sm.red_s[w];
    }
    __syncthreads();
}

// Fractional frequency error from the cyclic prefix of the PRS: products first .. first+375 of the candidate's prefix
// against the samples 2048 later (inside the prefix for any candidate within +-64 samples of first - 64 early);
// -angle / (2 pi 2048) cycles per sample.  Block-wide; ends with a barrier.
__device__ __forceinline__ float cp_fine_offset(SyncLds &sm, const float2 *sym, int first, int tid) {
    double cr = 0.0, ci = 0.0;
    for (int i = first + tid; i < first + 376; i += WG) {
        const float2 a = sym[i], b = sym[i + NB_FFT];
        cr += double(__fadd_rn(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y)));      // conj(a) * b
        ci += double(__fsub_rn(__fmul_rn(a.x, b.y), __fmul_rn(a.y, b.x)));
    }
    sm.red_d[0][tid] = cr;
    sm.red_d[1][tid] = ci;
    __syncthreads();
    for (int off = WG / 2; off > 0; off >>= 1) {
        if (tid < off) {
            sm.red_d[0][tid] += sm.red_d[0][tid + off];
            sm.red_d[1][tid] += sm.red_d[1][tid + off];
        }
        __syncthreads();
    }
    const float fine = float(-atan2(sm.red_d[1][0], sm.red_d[0][0]) / (2.0 * 3.14159265358979323846 * double(NB_FFT)));
    __syncthreads();
    return fine;
}

constexpr int MODE_PLAIN = 0, MODE_ACQ = 1, MODE_TRACK = 2;
constexpr int FRAME_LEN = NB_FRAME_SYMBOLS * NB_SYM_PERIOD;

// where the stream's state says frame slot i starts: p_i = next + (j0 + i) * period, j0 = frames that begin before
// the capture does.  The same arithmetic (double) in track_sync, track_update and the oracle.
struct Predictor {
    double next, period;
    int j0;
    __device__ Predictor(const StreamState &st) {
        next = st.next_frame_start;
        period = double(NB_FRAME_SAMPLES) + double(st.drift);
        j0 = next < 0.0 ? int(ceil(-next / period)) : 0;
    }
    __device__ double at(int i) const { return next + double(j0 + i) * period; }
    __device__ static bool fits(int64_t cand, int64_t n_samples) { return cand >= 0 && cand + FRAME_LEN + 512 <= n_samples; }
};

// MODE_PLAIN: candidates at a fixed stride, correction given, SyncResult out.
// MODE_ACQ  : candidates from the null search; the fractional frequency error is first measured on the cyclic
//             prefix of the PRS (products 64..439 of the prefix against the samples 2048 later: inside the prefix
//             for any candidate within +-64 samples), then the same search; AcquiredFrame out.
// MODE_TRACK: candidates predicted from the stream states (TrackArgs); correction = the state's fine + coarse offset;
//             the whole-carrier search only when asked for (max_coarse > 0); AcquiredFrame out.
// geometry of an upload riding in a synchronisation launch (see the kernel's head): measured over eight streams of a process,
// 256 x 2: 81-92 us per frame call, 128 x 4: 70-80, 64 x 8 / 48 x 8 / 32 x 16: 68-70 on every stream
constexpr int RIDE_BLOCKS = 64, RIDE_UNROLL = 8;
template <int MODE, bool LITE = false>
__global__ __launch_bounds__(WG) void prs_sync_kernel(SyncTables tab, const float2 *iq, size_t frame_stride,
                                                      const float *freq_offset, int max_coarse_arg, SyncResult *out,
                                                      AcquireArgs acq, TrackArgs trk, int n_total) {
    using Lds = typename std::conditional<LITE, SyncLdsLite, SyncLds>::type;
    __shared__ Lds sm;
    const int tid = threadIdx.x;
    // workgroups behind the synchronising ones: the upload that rides in this launch (TrackArgs::copy_*)
    int n_sync_blocks = int(gridDim.x);
    if constexpr (MODE == MODE_TRACK) {
        n_sync_blocks -= trk.copy_blocks;
        if (int(blockIdx.x) >= n_sync_blocks) {
            // (few, fat workgroups: this kernel's workgroups carry the synchronisation's 50 kB of LDS, so only two or three
            // fit a CU, and how the dispatcher spreads several hundred of them over the XCDs turned out to depend on the
            // hardware queue the stream was given -- 37 / 42 / 52 us for the same upload on three streams of one process.
            // 64 workgroups are one per CU on any queue; each thread keeps eight 16-byte loads in flight)
            const unsigned stride = unsigned(trk.copy_blocks) * WG;
            for (unsigned i0 = unsigned(int(blockIdx.x) - n_sync_blocks) * WG + unsigned(tid); i0 < trk.copy_n16; i0 += RIDE_UNROLL * stride) {
                uint4 v[RIDE_UNROLL];
#pragma unroll
                for (int u = 0; u < RIDE_UNROLL; u++)
                    if (i0 + u * stride < trk.copy_n16) v[u] = trk.copy_src[i0 + u * stride];
#pragma unroll
                for (int u = 0; u < RIDE_UNROLL; u++)
                    if (i0 + u * stride < trk.copy_n16) trk.copy_dst[i0 + u * stride] = v[u];
            }
            return;
        }
    }
    // With an upload riding in this launch the synchronising workgroup asks for its 2048 samples BEFORE anything else: they
    // come over the same link as the upload's 1.5 MB, and a request issued after the copying workgroups have queued theirs
    // is served behind all of them (measured: the launch took upload + synchronisation, 55 us instead of 37).
    float2 pre[8];
    bool prefetched = false;
    if constexpr (MODE == MODE_TRACK && !LITE) {
        if (trk.copy_n16 != 0 && trk.sync_iq != nullptr) {
            prefetched = true;
#pragma unroll
            for (int r = 0; r < 8; r++) pre[r] = trk.sync_iq[NB_CP + tid + r * WG];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // Auto-acquisition beside tracked streams (a steady-state tracked call): when every candidate of this workgroup belongs to
    // a stream that is being tracked there is nothing to do -- not even the 20 kB of tables (24 us of an all-skipped launch).
    if constexpr (MODE == MODE_ACQ) {
        if (acq.skip_tracked) {
            bool any = false;
            for (int frame = blockIdx.x; frame < n_total && !any; frame += n_sync_blocks)
                any = acq.skip_tracked[frame / acq.max_out].tracking != 1;
            if (!any) return;                                  // (the same for every thread of the workgroup)
        }
    }
    // the tables go to LDS once per workgroup; a workgroup then takes every gridDim.x-th candidate
    const float2 *tw;
    float *pw;                                                 // |h|^2 of every tap (a buffer that is idle by then)
    if constexpr (LITE) {
        tw = tab.twiddle;
        pw = reinterpret_cast<float *>(sm.t1);
    } else {
        for (int i = tid; i < TW_TOTAL; i += WG) sm.tw[i] = tab.twiddle[i];
        tw = sm.tw;
        pw = reinterpret_cast<float *>(sm.y);
    }
    const float2 *const twc8 = tw + TWC8_OFF, *const twc64 = tw + TWC64_OFF;
    for (int i = tid; i < NB_FFT; i += WG) sm.qt[i] = tab.prs_qt[i];
    __syncthreads();
    for (int frame = blockIdx.x; frame < n_total; frame += n_sync_blocks) {
    int max_coarse = max_coarse_arg;
    const float2 *sym;
    uint32_t dphi;
    int64_t cand = 0;
    float fine = 0.0f, coarse = 0.0f;
    PeakRule rule;
    bool do_coarse = true;
    if constexpr (MODE == MODE_ACQ) {
        const int st = frame / acq.max_out, j = frame - st * acq.max_out;
        if (acq.skip_tracked && acq.skip_tracked[st].tracking == 1) continue;
        if (j >= acq.counts[st]) {
            if (tid == 0) acq.out[frame] = AcquiredFrame{-1, 0.f, 0, 0.f, 0.f, 0.f, 0};
            continue;
        }
        cand = acq.cands[frame];
        sym = acq.iq + size_t(st) * acq.stream_stride + cand;
        fine = cp_fine_offset(sm, sym, 64, tid);
        dphi = uint32_t(__double2ll_rn(double(fine) * 4294967296.0));
        max_coarse = acq.max_coarse;
        rule = acq.rule;
    } else if constexpr (MODE == MODE_TRACK) {
        const int st = frame / trk.max_out, i = frame - st * trk.max_out;
        const StreamState ss = trk.state[st];
        bool have = true;
        if (!trk.fixed_start) {
            const Predictor pr(ss);
            cand = __double2ll_rn(pr.at(i));
            have = ss.tracking == 1 && Predictor::fits(cand, trk.n_samples);
        }
        if (!have) {
            if (tid == 0) {
                trk.out[frame] = AcquiredFrame{-1, 0.f, 0, 0.f, 0.f, 0.f, 0};
                if (trk.sync_out) trk.sync_out[frame] = SyncResult{0, 0, 0.f, 0.f};
            }
            continue;
        }
        sym = (trk.sync_iq ? trk.sync_iq : trk.iq) + size_t(st) * trk.stream_stride + cand;
        fine = ss.fine_freq_offset;
        // first frame after a null-symbol detection: the fine offset starts from this PRS's own cyclic prefix, so that
        // the frame is already demodulated with it (the loop then refines it)
        if constexpr (!LITE)
            if (trk.fixed_start && trk.acquiring) fine = cp_fine_offset(sm, sym, trk.margin, tid);
        coarse = (trk.fixed_start && trk.acquiring && trk.max_coarse > 0) ? 0.0f : ss.coarse_freq_offset;
        dphi = uint32_t(__double2ll_rn(double(__fadd_rn(fine, coarse)) * 4294967296.0));
        max_coarse = trk.max_coarse;
        do_coarse = max_coarse > 0;
        rule = trk.rule;
    } else {
        sym = iq + size_t(frame) * frame_stride;
        dphi = dphi_of(freq_offset, frame);
    }
    // ---- X = FFT(nco * window) ----
    {
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int n = tid + r * WG;
            v[r] = prefetched ? pre[r] : sym[NB_CP + n];
            if (dphi != 0u) v[r] = cmul(v[r], nco(uint32_t(n), dphi));
        }
        block_fft2048(v, sm.t1, sm.x, tw, twc8, twc64, tid);
    }
    float best_m, total;
    int khat = 0;
    float coarse_ptm = 0.0f;
    if constexpr (!LITE)
    if (do_coarse) {                                           // (uniform over the workgroup)
        // ---- Q[b] = X[b+1] conj X[b] -> t1 ----
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int b = tid + r * WG;
            sm.t1[b] = cmulc(sm.x[(b + 1) & (NB_FFT - 1)], sm.x[b]);
        }
        __syncthreads();
        // ---- coarse frequency: D_k = sum_b Q[b+k] conj S[b] for ALL shifts at once as a circular correlation,
        //      D = IFFT(FFT(Q) conj FFT(S)); FFT(S) is a table, |IFFT(Z)| = |FFT(conj Z)| (the common 1/N does not
        //      change a peak-to-mean ratio).  Two more transforms instead of (2 max + 1) x 1535 complex adds. ----
        {
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = sm.t1[tid + r * WG];
            __syncthreads();                                   // t1 is the transform's scratch from here on
            block_fft2048(v, sm.t1, sm.y, sm.tw, twc8, twc64, tid);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int b = tid + r * WG;
                const float2 z = cmulc(sm.y[b], tab.pair_spectrum[b]);
                v[r] = make_float2(z.x, -z.y);
            }
            __syncthreads();
            block_fft2048(v, sm.t1, sm.y, sm.tw, twc8, twc64, tid);
        }
        float my_m = -1.0f, my_s = 0.0f;
        int my_k = 0x7fffffff;
        for (int idx = tid; idx <= 2 * max_coarse; idx += WG) {
            const float2 d = sm.y[(idx - max_coarse) & (NB_FFT - 1)];
            const float m = d.x * d.x + d.y * d.y;
            my_s += m;
            if (m > my_m) { my_m = m; my_k = idx; }
        }
        int best_idx;
        block_argmax_sum(sm, tid, my_m, my_k, my_s, best_m, best_idx, total);
        khat = best_idx - max_coarse;
        coarse_ptm = best_m / (total / float(2 * max_coarse + 1));
    }

    // ---- fine time: |IFFT(Z)| = |FFT(conj Z)|, Z[b] = X[b+k^] conj R[b] on carriers ----
    {
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int b = tid + r * WG;
            const int q = sm.qt[b];
            float2 z = make_float2(0.f, 0.f);
            if (q >= 0) {
                z = rot_mq(sm.x[(b + khat) & (NB_FFT - 1)], q);
                z.y = -z.y;
            }
            v[r] = z;
        }
        __syncthreads();          // everyone has read x[] before the FFT overwrites it
        block_fft2048(v, sm.t1, sm.x, tw, twc8, twc64, tid);
    }
    // power of every tap into pw[] (kept for the first-path scan), weighted score for the peak choice (PeakRule)
    float my_m = -1.0f, my_s = 0.0f;
    int my_n = 0x7fffffff;
    const float decay = __fmul_rn(__fsub_rn(1.0f, rule.distance_prob), 1.0f / float(NB_SYM_PERIOD));
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int n = tid + r * WG;
        const float2 h = sm.x[n];
        const float m = __fadd_rn(__fmul_rn(h.x, h.x), __fmul_rn(h.y, h.y));
        pw[n] = m;
        my_s += m;
        const int t = n < NB_FFT / 2 ? n : n - NB_FFT;
        const float w = __fsub_rn(1.0f, __fmul_rn(decay, float(abs(t - rule.expected))));
        const float sc = __fmul_rn(__fmul_rn(m, w), w);
        if (sc > my_m) { my_m = sc; my_n = n; }
    }
    int best_n;
    block_argmax_sum(sm, tid, my_m, my_n, my_s, best_m, best_n, total);   // (its barriers make y[] visible)
    const float peak = pw[best_n];
    const float mean = total / float(NB_FFT);
    if (rule.first_path_rel > 0.0f) {
        const float thr = fmaxf(__fmul_rn(rule.first_path_rel, peak), __fmul_rn(16.0f, mean));
        float my_d = 0.0f;
        for (int d = tid + 1; d <= NB_CP; d += WG)
            if (pw[(best_n - d) & (NB_FFT - 1)] >= thr) my_d = float(d);
        float dmax, dsum;
        int dummy;
        block_argmax_sum(sm, tid, my_d, tid, 0.0f, dmax, dummy, dsum);
        best_n = (best_n - int(dmax)) & (NB_FFT - 1);
    }
    if (tid == 0) {
        const int toff = best_n < NB_FFT / 2 ? best_n : best_n - NB_FFT;
        const float ptm = peak / mean;
        if constexpr (MODE == MODE_ACQ) {
            AcquiredFrame r;
            r.start = cand + toff - acq.margin;
            r.coarse_carriers = khat;
            r.fine_offset = fine;
            r.freq_offset = __fsub_rn(fine, float(khat) / float(NB_FFT));
            r.peak_to_mean = ptm;
            r.coarse_peak_to_mean = coarse_ptm;
            r.flags = (ptm >= acq.min_peak_to_mean ? 1 : 0) |
                      ((r.start >= 0 && r.start + int64_t(FRAME_LEN) <= acq.n_samples) ? 2 : 0);
            acq.out[frame] = r;
        } else if constexpr (MODE == MODE_TRACK) {
            AcquiredFrame r;
            const bool locked = ptm >= trk.min_peak_to_mean;
            if (trk.fixed_start) {
                // the host assembled the frame: it is demodulated where it lies; what moves is the coarse offset
                const int st = frame;                          // max_out == 1
                if (do_coarse && locked) {
                    if (trk.acquiring) coarse = -float(khat) / float(NB_FFT);
                    else if (khat != 0) coarse = __fsub_rn(coarse, __fmul_rn(trk.coarse_slow_beta, float(khat) / float(NB_FFT)));
                    trk.state[st].coarse_freq_offset = coarse;
                }
                if (trk.acquiring) trk.state[st].fine_freq_offset = fine;
                r.start = 0;
                r.flags = (locked ? 1 : 0) | ((toff >= 0 && toff <= NB_CP - 16) ? 2 : 0);
            } else {
                r.start = cand + toff - trk.margin;
                r.flags = (locked ? 1 : 0) | ((r.start >= 0 && r.start + int64_t(FRAME_LEN) <= trk.n_samples) ? 2 : 0);
            }
            r.coarse_carriers = khat;
            r.fine_offset = fine;
            r.freq_offset = __fadd_rn(fine, coarse);
            r.peak_to_mean = ptm;
            r.coarse_peak_to_mean = coarse_ptm;
            trk.out[frame] = r;
            if (trk.sync_out) trk.sync_out[frame] = SyncResult{khat, toff, ptm, coarse_ptm};
        } else {
            SyncResult r;
            r.coarse_carriers = khat;
            r.time_offset = toff;
            r.peak_to_mean = ptm;
            r.coarse_peak_to_mean = coarse_ptm;
            out[frame] = r;
        }
    }
    __syncthreads();                                           // the reduction buffers are reused by the next candidate
    }
}

// ---- timing tracking: state update after the demodulation of the tracked frames (TrackUpdateArgs) ----
constexpr int TU = 1024;          // threads per stream: the kernel is one workgroup per stream and lives on loads in flight
__global__ __launch_bounds__(TU) void track_update_kernel(TrackUpdateArgs a) {
    __shared__ double red[7][TU];
    __shared__ int red_last[TU];
    __shared__ StreamState st_in;
    const int s = blockIdx.x, tid = threadIdx.x;
    if (s >= a.n_streams) {                                    // the download that rides in this launch (TrackUpdateArgs::down)
        unsigned i = unsigned(s - a.n_streams) * TU + unsigned(tid);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const unsigned n16 = unsigned(a.down[k].bytes >> 4);
            if (i < n16) { static_cast<uint4 *>(a.down[k].dst)[i] = static_cast<const uint4 *>(a.down[k].src)[i]; return; }
            i -= n16;
        }
        return;
    }
    // one thread reads the state for everybody: thread 0 writes it back further down (and at once in the branch below),
    // and a wavefront scheduled late must not see that write
    if (tid == 0) st_in = a.state[s];
    __syncthreads();
    StreamState st = st_in;
    if (!a.fixed_start && st.tracking == 2) {                  // started in this very call (auto-acquisition): nothing to update
        if (tid == 0) a.state[s].tracking = 1;
        return;
    }
    if (a.settle_only) return;
    if (!a.fixed_start && !st.tracking) {
        if (tid == 0 && a.counts) a.counts[s] = 0;
        return;
    }
    const Predictor pr(st);
    // frame slots inside the capture (the same test track_sync made): the first slot that does not fit ends the list.  Slot i is
    // tested by thread i (every thread walking the list by itself was 15 us of double arithmetic at 256 slots)
    int count = a.max_out;
    if (!a.fixed_start) {
        if (tid == 0) red_last[0] = a.max_out;
        __syncthreads();
        for (int i = tid; i < a.max_out; i += TU)
            if (!Predictor::fits(__double2ll_rn(pr.at(i)), a.n_samples)) atomicMin(&red_last[0], i);
        __syncthreads();
        count = red_last[0];
        __syncthreads();                                       // (red_last is the reduction's scratch further down)
    }
    const AcquiredFrame *fr = a.frames + size_t(s) * a.max_out;
    const float2 *cyc = a.cyc + size_t(s) * a.max_out * NB_FRAME_SYMBOLS;
    // sums over the locked frames: n, i, i^2, r, i*r, and the cyclic-prefix angles
    double sn = 0, si = 0, sii = 0, sr = 0, sir = 0, sang = 0, sang2 = 0;     // (dd: sang / sang2 = real / imaginary part)
    int last = -1;
    for (int i = tid; i < count; i += TU) {
        const AcquiredFrame f = fr[i];
        if ((f.flags & 3) != 3) continue;
        const double r = double(f.start) - pr.at(i);
        sn += 1.0; si += double(i); sii += double(i) * double(i); sr += r; sir += double(i) * r;
        last = i;
    }
    {   // Entry k = tid, tid + TU, ...: its frame and symbol without a division per entry, and EIGHT entries' loads in flight at a
        // time -- one workgroup per stream has nothing to hide a load's latency behind, and one entry at a time (two dependent
        // loads each: the frame's flags, then the value) was ~19 x 2.5 us of this kernel's 65 at 256 frames per stream.
        constexpr int U = 8;
        const int n_entries = count * NB_FRAME_SYMBOLS;
        int i = tid / NB_FRAME_SYMBOLS, l = tid - i * NB_FRAME_SYMBOLS;
        for (int k0 = tid; k0 < n_entries; k0 += U * TU) {
            int fl[U], sym[U];
            float2 c[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int k = k0 + u * TU;
                const bool in = k < n_entries;
                fl[u] = in ? fr[in ? i : 0].flags : 0;
                c[u] = in ? cyc[k] : make_float2(1.f, 0.f);
                sym[u] = l;
                i += TU / NB_FRAME_SYMBOLS;
                l += TU % NB_FRAME_SYMBOLS;
                if (l >= NB_FRAME_SYMBOLS) { l -= NB_FRAME_SYMBOLS; i++; }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if ((fl[u] & 3) != 3) continue;
                if (a.dd) {
                    if (sym[u]) { sang += double(c[u].x); sang2 += double(c[u].y); }     // (entry 0: second pass below)
                } else {
                    sang += double(atan2f(c[u].y, c[u].x));
                }
            }
        }
    }
    red[0][tid] = sn; red[1][tid] = si; red[2][tid] = sii; red[3][tid] = sr; red[4][tid] = sir; red[5][tid] = sang;
    red[6][tid] = sang2;
    red_last[tid] = last;
    __syncthreads();
    for (int off = TU / 2; off > 0; off >>= 1) {
        if (tid < off) {
#pragma unroll
            for (int q = 0; q < 7; q++) red[q][tid] += red[q][tid + off];
            red_last[tid] = max(red_last[tid], red_last[tid + off]);
        }
        __syncthreads();
    }
    sn = red[0][0]; si = red[1][0]; sii = red[2][0]; sr = red[3][0]; sir = red[4][0]; sang = red[5][0]; sang2 = red[6][0];
    last = red_last[0];
    __syncthreads();
    // level of the last locked frame: its first 4096 samples
    float l1 = 0.f;
    if (last >= 0) {
        const float2 *x = a.iq + size_t(s) * a.stream_stride + fr[last].start;
        for (int i = tid; i < 4096; i += TU) l1 += fabsf(x[i].x) + fabsf(x[i].y);
    }
    // ... and, for the decision-directed loop, the angles of the locked frames' PRS cyclic-prefix correlations (entry 0)
    double ang0 = 0.0;
    if (a.dd)
        for (int i = tid; i < count; i += TU)
            if ((fr[i].flags & 3) == 3) ang0 += double(atan2f(cyc[size_t(i) * NB_FRAME_SYMBOLS].y, cyc[size_t(i) * NB_FRAME_SYMBOLS].x));
    red[0][tid] = double(l1);
    red[1][tid] = ang0;
    __syncthreads();
    for (int off = TU / 2; off > 0; off >>= 1) {
        if (tid < off) { red[0][tid] += red[0][tid + off]; red[1][tid] += red[1][tid + off]; }
        __syncthreads();
    }
    if (tid != 0) return;
    l1 = float(red[0][0] * (1.0 / 4096.0));
    ang0 = red[1][0];
    const int n_locked = int(sn);
    int desync = (a.fixed_start ? 0 : pr.j0) + (count - n_locked);
    bool level_lost = false;
    if (n_locked > 0) {
        // fine-frequency loop
        float err = float(sang / (sn * double(NB_FRAME_SYMBOLS))) * (1.0f / (6.283185307179586f * float(NB_FFT)));
        level_lost = st.signal_average > 0.f && l1 < a.thr_null_start * st.signal_average;
        // (ONE locked frame whose level is gone has nothing to steer the loop with)
        const bool steers = !(level_lost && n_locked == 1);
        if (a.dd) {
            StreamState scratch = st;          // an estimate that is not applied does not move the gate's memory either
            err = dd_loop_error(sang, sang2, float(ang0 / sn) * (1.0f / (6.283185307179586f * float(NB_FFT))),
                                sn * double(a.dd_terms_per_frame), a.dd_gate, st.total_frames_read == 0, steers ? st : scratch);
        }
        if (steers) {
            constexpr float HALF = 0.5f / float(NB_FFT);
            float f = st.fine_freq_offset - a.fine_beta * err;
            if (f > HALF) f -= 2.f * HALF;
            if (f < -HALF) f += 2.f * HALF;
            st.fine_freq_offset = f;
        }
        st.last_fine_error = err;
        if (!a.fixed_start) st.last_time_offset = int32_t(fr[last].start - __double2ll_rn(pr.at(last)));
        st.last_peak_to_mean = fr[last].peak_to_mean;
        if (level_lost) desync++;
        else st.signal_average = st.signal_average > 0.f ? a.signal_beta * st.signal_average + (1.0f - a.signal_beta) * l1 : l1;
    }
    if (!a.fixed_start) {
        double alpha = 0.0, slope = 0.0, slope_hat = 0.0, gain = 0.0;
        if (n_locked >= 2) {
            const double det = sn * sii - si * si;             // > 0: distinct slots
            slope = (sn * sir - si * sr) / det;
            alpha = (sr - slope * si) / sn;
            slope_hat = slope;
            gain = double(a.drift_beta) * fmin(1.0, sn * 0.25);
        } else if (n_locked == 1) {
            alpha = sr;
            // one frame per call: its residual IS the drift error of one period (the position was extrapolated one
            // period from the previous frame); a lone survivor among several slots says nothing about the slope
            slope_hat = count == 1 ? alpha : 0.0;
            gain = double(a.drift_beta) * 0.125;
        }
        if (count > 0 && n_locked == 0) st.tracking = 0;        // every frame of the call lost: acquire again
        st.next_frame_start = pr.at(count) + alpha + slope * double(count) - double(a.advance);
        st.drift = float(double(st.drift) + gain * slope_hat);
    }
    st.total_frames_read += n_locked - (level_lost ? 1 : 0);      // (a frame whose level is gone counts as lost, not as read)
    st.total_frames_desync += desync;
    a.state[s] = st;
    if (a.state_out) a.state_out[s] = st;
    if (a.counts) a.counts[s] = count;
}

// one wave per stream; see launch_track_start in kernels.hpp
__global__ __launch_bounds__(64) void track_start_kernel(StreamState *state, const AcquiredFrame *frames, const int32_t *counts,
                                                         int max_out, int64_t advance, int only_lost) {
    const int s = blockIdx.x, lane = threadIdx.x;
    if (only_lost && state[s].tracking == 1) return;
    const AcquiredFrame *fr = frames + size_t(s) * max_out;
    const int count = min(counts[s], max_out);
    // first and last locked frame
    int first = 0x7fffffff, last = -1;
    for (int i = lane; i < count; i += 64)
        if ((fr[i].flags & 3) == 3) { first = min(first, i); last = max(last, i); }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { first = min(first, __shfl_xor(first, off)); last = max(last, __shfl_xor(last, off)); }
    StreamState st = state[s];
    if (last < 0) {
        if (lane == 0) { st.tracking = 0; state[s] = st; }
        return;
    }
    const int64_t s0 = fr[first].start;
    double sn = 0, sj = 0, sjj = 0, sy = 0, sjy = 0, sf = 0;
    for (int i = lane; i < count; i += 64) {
        const AcquiredFrame f = fr[i];
        if ((f.flags & 3) != 3) continue;
        const double y = double(f.start - s0);
        const double j = double(__double2ll_rn(y / double(NB_FRAME_SAMPLES)));
        sn += 1.0; sj += j; sjj += j * j; sy += y; sjy += j * y; sf += double(f.fine_offset);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        sn += __shfl_xor(sn, off); sj += __shfl_xor(sj, off); sjj += __shfl_xor(sjj, off);
        sy += __shfl_xor(sy, off); sjy += __shfl_xor(sjy, off); sf += __shfl_xor(sf, off);
    }
    if (lane != 0) return;
    double drift = 0.0;
    if (sn >= 4.0) drift = (sn * sjy - sj * sy) / (sn * sjj - sj * sj) - double(NB_FRAME_SAMPLES);
    st.drift = float(drift);
    st.next_frame_start = double(fr[last].start) + double(NB_FRAME_SAMPLES) + double(st.drift) - double(advance);
    st.fine_freq_offset = float(sf / sn);
    st.coarse_freq_offset = -float(fr[last].coarse_carriers) / float(NB_FFT);
    st.tracking = only_lost ? 2 : 1;
    // (re)started from an acquisition: the decision-directed loop's gate is "pulling in" again -- its first estimate, on
    // whatever branch, is believed (a branch 0 remembered from before the loss would hold a correct +-1 back once)
    st.dd_branch = DD_NO_BRANCH;
    st.dd_pending = DD_NO_BRANCH;
    st.total_frames_read += int(sn);
    st.last_time_offset = 0;
    st.last_peak_to_mean = fr[last].peak_to_mean;
    state[s] = st;
}

// ---- null-symbol search (FINDING_NULL_POWER_DIP, /root/reference/src/render_radio_block.cpp:193) ----
// L1 norm of every 64-sample block.  A wave takes two blocks per trip: lane j of a half holds samples 2j, 2j+1,
// the 32 pair sums are combined by the XOR butterfly 1, 2, 4, 8, 16 (the fixed tree the oracle restates).  A wave
// owns 32 consecutive blocks (16 KB of samples, four trips' loads in flight at a time) and writes their norms as ONE
// 128-byte store: two dwords per wave, as a first version did, are partial cache lines that eight XCDs' L2s each
// hold a piece of.
__global__ __launch_bounds__(256) void null_l1_kernel(const float2 *iq, size_t stream_stride, int64_t nb, float *l1,
                                                      const StreamState *skip_tracked) {
    const int st = blockIdx.y;
    if (skip_tracked && skip_tracked[st].tracking == 1) return;
    const int lane = threadIdx.x & 63;
    const int64_t wave = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = int64_t(gridDim.x) * 4;
    const float2 *x = iq + size_t(st) * stream_stride;
    float *o = l1 + size_t(st) * nb;
    const int64_t n_chunks = (nb + 31) / 32;
    for (int64_t chunk = wave; chunk < n_chunks; chunk += n_waves) {
        const int64_t b0 = chunk * 32;
        float keep = 0.0f;
#pragma unroll
        for (int j0 = 0; j0 < 16; j0 += 4) {
            float2 s0[4], s1[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                // (a block past the end re-reads the last one: no branch between the loads, the store is predicated)
                const int64_t blk = min(b0 + 2 * (j0 + u) + (lane >> 5), nb - 1);
                const float2 *p = x + blk * 64 + 2 * (lane & 31);
                s0[u] = ld_stream(p);
                s1[u] = ld_stream(p + 1);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                float v = __fadd_rn(__fadd_rn(__fadd_rn(fabsf(s0[u].x), fabsf(s0[u].y)), fabsf(s1[u].x)), fabsf(s1[u].y));
                // XOR butterfly inside each half-wave, on the VALU (DPP / row swap) instead of the LDS crossbar
                v = __fadd_rn(v, lane_xor_f<1>(v, lane));
                v = __fadd_rn(v, lane_xor_f<2>(v, lane));
                v = __fadd_rn(v, lane_xor_f<4>(v, lane));
                v = __fadd_rn(v, lane_xor_f<8>(v, lane));
                v = __fadd_rn(v, lane_xor_f<16>(v, lane));
                if ((lane & 31) == j0 + u) keep = v;           // every lane of the half has the sum: lane j keeps trip j's
            }
        }
        const int64_t b = b0 + 2 * (lane & 31) + (lane >> 5);
        if ((lane & 31) < 16 && b < nb) o[b] = keep;
    }
}

// ---- dip search ----
// The search is a two-state machine over the block norms (below thr_start x mean: a dip begins; above thr_end x mean:
// it ends, and a dip of plausible length is a null symbol).  One wave walking a whole stream is pure latency, so the
// stream is cut into segments that are scanned in parallel; what a segment does depends on the state it is entered
// in only up to its first block above the upper threshold -- after that block both possible machines coincide.  A
// segment therefore reports (first high block h, first low block before it, the candidates and end state of a
// state-0 machine started at h + 1), and one wave per stream stitches the segments together in order: the result is
// exactly the sequential machine's (the oracle runs the plain loop).
constexpr int MEAN_THREADS = 1024;
constexpr int64_t SEG_BLOCKS = 16384;         // blocks per segment: 2^20 samples, about half a second

struct DipSegment {                            // 32 bytes
    int64_t h;                                 // first block above thr_end, or -1
    int64_t l0;                                // first block below thr_start before h (anywhere, if h = -1), or -1
    int64_t dip_begin_out;                     // tail machine: begin of the dip that is open at the segment's end
    int32_t n_tail;                            // candidates of the tail machine
    int32_t state_out;                         // tail machine's state at the segment's end
};

// Mean block norm of a stream: 1024 strided partial sums in double, XOR butterfly inside each wave, the sixteen wave
// sums added in order (the fixed tree the oracle restates).
__global__ __launch_bounds__(MEAN_THREADS) void null_mean_kernel(const float *l1_all, int64_t nb, float *avg,
                                                                 const StreamState *skip_tracked) {
    __shared__ double part[MEAN_THREADS / 64];
    const int st = blockIdx.x, tid = threadIdx.x;
    if (skip_tracked && skip_tracked[st].tracking == 1) return;
    const float *l1 = l1_all + size_t(st) * nb;
    double acc = 0.0;
    int64_t b = tid;
    for (; b + 7 * MEAN_THREADS < nb; b += 8 * MEAN_THREADS) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = l1[b + MEAN_THREADS * j];
#pragma unroll
        for (int j = 0; j < 8; j++) acc += double(v[j]);
    }
    for (; b < nb; b += MEAN_THREADS) acc += double(l1[b]);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
    if ((tid & 63) == 0) part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = part[0];
        for (int w = 1; w < MEAN_THREADS / 64; w++) t += part[w];
        avg[st] = float(t / double(nb));
    }
}

// Local level: the mean block norm of every chunk of `chunk` blocks -- one wave per chunk: 64 strided partial sums in
// double, XOR butterfly (the fixed tree the oracle restates).  The thresholds at a block use the mean of the chunk means
// c-2 .. c+2 that exist (local_level below): what the reference's running average (signal_l1.update_beta) does for a
// stream -- a slow fade must not look like a null symbol, nor hide one.
__global__ __launch_bounds__(64) void null_level_kernel(const float *l1_all, int64_t nb, int chunk, int64_t nc, double *cm_all,
                                                        const StreamState *skip_tracked) {
    const int st = blockIdx.y, lane = threadIdx.x;
    if (skip_tracked && skip_tracked[st].tracking == 1) return;
    const float *l1 = l1_all + size_t(st) * nb;
    for (int64_t c = blockIdx.x; c < nc; c += gridDim.x) {
        const int64_t b0 = c * chunk, b1 = min(nb, b0 + chunk);
        double acc = 0.0;
        for (int64_t b = b0 + lane; b < b1; b += 64) acc += double(l1[b]);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) cm_all[size_t(st) * nc + c] = acc / double(b1 - b0);
    }
}

__device__ __forceinline__ float local_level(const double *cm, int64_t nc, int64_t c) {
    double a = 0.0;
    int k = 0;
    for (int64_t j = c - 2; j <= c + 2; j++)
        if (j >= 0 && j < nc) { a += cm[j]; k++; }
    return float(a / double(k));
}

__device__ __forceinline__ bool dip_is_null_symbol(const AcquireArgs &a, int64_t dip_begin, int64_t end_block, int64_t *cand) {
    const int max_blocks = 2 * NB_NULL_PERIOD / 64;
    const int64_t len = end_block - dip_begin;
    const int64_t c = end_block * 64 - 48;
    *cand = c;
    return len >= a.min_blocks && len <= max_blocks && c >= 0 &&
           c + int64_t(NB_FRAME_SYMBOLS) * NB_SYM_PERIOD + 512 <= a.n_samples;
}

// one wave per (segment, stream); 64 blocks are classified per trip with two ballots, the scalar unit walks the events
__global__ __launch_bounds__(64) void null_segment_kernel(AcquireArgs a, int64_t nb, int n_seg, const float *avg,
                                                          DipSegment *segs, int64_t *seg_cands) {
    const int seg = blockIdx.x, st = blockIdx.y;
    const int lane = threadIdx.x;
    if (a.skip_tracked && a.skip_tracked[st].tracking == 1) return;
    const float *l1 = a.l1 + size_t(st) * nb;
    float ts = 0.f, te = 0.f;
    const int64_t nc = a.level_chunk ? (nb + a.level_chunk - 1) / a.level_chunk : 0;
    const double *cm = a.level_chunk ? a.chunk_mean + size_t(st) * nc : nullptr;
    if (!a.level_chunk) {
        const float mean = avg[st];
        ts = __fmul_rn(a.thr_start, mean);
        te = __fmul_rn(a.thr_end, mean);
    }
    const int64_t b0 = int64_t(seg) * SEG_BLOCKS, b1 = min(nb, b0 + SEG_BLOCKS);
    int64_t *cands = seg_cands + (size_t(st) * n_seg + seg) * a.max_out;
    int64_t h = -1, l0 = -1, dip_begin = 0;
    int count = 0, state = 0;
    constexpr int AHEAD = 8;                                   // trips fetched together
    float vbuf[AHEAD];
    for (int64_t base = b0; base < b1 && count < a.max_out; base += 64) {
        const int slot = int(((base - b0) >> 6) % AHEAD);
        if (slot == 0) {
#pragma unroll
            for (int j = 0; j < AHEAD; j++) {
                const int64_t bj = base + 64 * j + lane;
                vbuf[j] = bj < b1 ? l1[bj] : 0.0f;
            }
        }
        if (a.level_chunk && ((base - b0) & int64_t(a.level_chunk - 1)) == 0) {   // a trip of 64 blocks lies inside one chunk
            const float level = local_level(cm, nc, base / a.level_chunk);
            ts = __fmul_rn(a.thr_start, level);
            te = __fmul_rn(a.thr_end, level);
        }
        const int64_t b = base + lane;
        float v = vbuf[0];
#pragma unroll
        for (int j = 1; j < AHEAD; j++) v = (slot == j) ? vbuf[j] : v;
        const unsigned long long low = __ballot(b < b1 && v < ts);
        const unsigned long long high = __ballot(b < b1 && v > te);
        int pos = 0;
        if (h < 0) {                                           // head: up to the first high block
            if (l0 < 0 && low != 0ull && (high == 0ull || __builtin_ctzll(low) < __builtin_ctzll(high)))
                l0 = base + __builtin_ctzll(low);
            if (high == 0ull) continue;
            h = base + __builtin_ctzll(high);
            pos = __builtin_ctzll(high) + 1;                   // the tail machine starts behind it, in state 0
        }
        while (pos < 64 && count < a.max_out) {
            const unsigned long long m = (state == 0 ? low : high) >> pos;
            if (m == 0ull) break;
            const int bit = pos + __builtin_ctzll(m);
            const int64_t bb = base + bit;
            if (state == 0) {
                state = 1;
                dip_begin = bb;
            } else {
                int64_t c;
                if (dip_is_null_symbol(a, dip_begin, bb, &c)) {
                    if (lane == 0) cands[count] = c;
                    count++;
                }
                state = 0;
            }
            pos = bit + 1;
        }
    }
    if (lane == 0) segs[size_t(st) * n_seg + seg] = DipSegment{h, l0, dip_begin, count, state};
}

// one wave per stream: the segments in order
__global__ __launch_bounds__(64) void null_stitch_kernel(AcquireArgs a, int n_seg, const DipSegment *segs, const int64_t *seg_cands) {
    const int st = blockIdx.x, lane = threadIdx.x;
    if (a.skip_tracked && a.skip_tracked[st].tracking == 1) return;
    int64_t *cands = a.cands + size_t(st) * a.max_out;
    int count = 0, state = 0;
    int64_t dip_begin = 0;
    for (int s0 = 0; s0 < n_seg && count < a.max_out; s0 += 64) {
        DipSegment mine = DipSegment{-1, -1, 0, 0, 0};
        if (s0 + lane < n_seg) mine = segs[size_t(st) * n_seg + s0 + lane];
        const int n_here = min(64, n_seg - s0);
        for (int j = 0; j < n_here && count < a.max_out; j++) {
            const int64_t h = __shfl(mine.h, j), l0 = __shfl(mine.l0, j);
            if (h < 0) {                                       // no block above the upper threshold in this segment
                if (state == 0 && l0 >= 0) { state = 1; dip_begin = l0; }
                continue;
            }
            const bool open = state == 1 || l0 >= 0;
            const int64_t begin = state == 1 ? dip_begin : l0;
            int64_t c;
            if (open && dip_is_null_symbol(a, begin, h, &c)) {
                if (lane == 0) cands[count] = c;
                count++;
            }
            const int n_tail = __shfl(mine.n_tail, j);
            const int64_t *tc = seg_cands + (size_t(st) * n_seg + s0 + j) * a.max_out;
            const int take = min(n_tail, a.max_out - count);
            for (int i = lane; i < take; i += 64) cands[count + i] = tc[i];
            count += take;
            state = __shfl(mine.state_out, j);
            dip_begin = __shfl(mine.dip_begin_out, j);
        }
    }
    if (lane == 0) a.counts[st] = count;
}

}  // namespace

// two workgroups fit a CU (73 KB of LDS each); eight rounds' worth of workgroups, each walking the candidates with that
// stride, keeps the tail short and loads the tables once per eight candidates
static unsigned sync_grid(int n) { return unsigned(std::max(1, std::min(n, 4096))); }

hipError_t launch_prs_sync(const SyncTables &t, const float2 *iq, size_t frame_stride, int n_frames,
                           const float *freq_offset, int max_coarse, SyncResult *out, hipStream_t s) {
    if (n_frames <= 0) return hipSuccess;
    if (max_coarse < 0 || max_coarse > 1023) return hipErrorInvalidValue;
    hipLaunchKernelGGL(prs_sync_kernel<MODE_PLAIN>, dim3(sync_grid(n_frames)), dim3(WG), 0, s, t, iq, frame_stride,
                       freq_offset, max_coarse, out, AcquireArgs{}, TrackArgs{}, n_frames);
    return hipGetLastError();
}

hipError_t launch_track_sync(const SyncTables &t, const TrackArgs &a, hipStream_t s) {
    if (a.n_streams <= 0 || a.max_out <= 0) return hipSuccess;
    if (a.max_coarse < 0 || a.max_coarse > 1023 || (a.fixed_start && a.max_out != 1)) return hipErrorInvalidValue;
    TrackArgs b = a;
    b.rule.expected = a.margin;
    b.copy_blocks = 0;
    if (a.copy_n16) {
        // the riding upload's prefetch reads stream 0 at candidate 0 for the whole workgroup: one stream, one frame only
        if (!a.copy_dst || !a.copy_src || !a.fixed_start || a.n_streams * a.max_out != 1) return hipErrorInvalidValue;
        b.copy_blocks = int(std::min<unsigned>(unsigned(RIDE_BLOCKS), (a.copy_n16 + WG - 1) / WG));
    } else {
        b.copy_dst = nullptr;
        b.copy_src = nullptr;
    }
    if (a.copy_n16)
        hipLaunchKernelGGL(prs_sync_kernel<MODE_TRACK>, dim3(sync_grid(a.n_streams * a.max_out) + unsigned(b.copy_blocks)), dim3(WG), 0, s,
                           t, static_cast<const float2 *>(nullptr), size_t(0), static_cast<const float *>(nullptr), 0,
                           static_cast<SyncResult *>(nullptr), AcquireArgs{}, b, a.n_streams * a.max_out);
    else if (!a.fixed_start && a.max_coarse == 0)
        hipLaunchKernelGGL((prs_sync_kernel<MODE_TRACK, true>), dim3(sync_grid(a.n_streams * a.max_out)), dim3(WG), 0, s, t,
                           static_cast<const float2 *>(nullptr), size_t(0), static_cast<const float *>(nullptr), 0,
                           static_cast<SyncResult *>(nullptr), AcquireArgs{}, b, a.n_streams * a.max_out);
    else
        hipLaunchKernelGGL(prs_sync_kernel<MODE_TRACK>, dim3(sync_grid(a.n_streams * a.max_out)), dim3(WG), 0, s, t,
                           static_cast<const float2 *>(nullptr), size_t(0), static_cast<const float *>(nullptr), 0,
                           static_cast<SyncResult *>(nullptr), AcquireArgs{}, b, a.n_streams * a.max_out);
    return hipGetLastError();
}

hipError_t launch_track_update(const TrackUpdateArgs &a, hipStream_t s) {
    if (a.n_streams <= 0 || a.max_out <= 0) return hipSuccess;
    TrackUpdateArgs b = a;
    unsigned words = 0;
    for (const CopyPiece &c : b.down) {
        if ((c.bytes & 15) || ((reinterpret_cast<uintptr_t>(c.dst) | reinterpret_cast<uintptr_t>(c.src)) & 15)) return hipErrorInvalidValue;
        words += unsigned(c.bytes >> 4);
    }
    b.copy_blocks = int((words + TU - 1) / TU);
    hipLaunchKernelGGL(track_update_kernel, dim3(unsigned(a.n_streams + b.copy_blocks)), dim3(TU), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_track_start(StreamState *state, const AcquiredFrame *frames, const int32_t *counts, int n_streams,
                              int max_out, int64_t advance, int only_lost, hipStream_t s) {
    if (n_streams <= 0 || max_out <= 0) return hipSuccess;
    hipLaunchKernelGGL(track_start_kernel, dim3(unsigned(n_streams)), dim3(64), 0, s, state, frames, counts, max_out, advance, only_lost);
    return hipGetLastError();
}

static size_t dip_segments(int64_t nb) { return size_t((nb + SEG_BLOCKS - 1) / SEG_BLOCKS); }
static size_t al256(size_t v) { return (v + 255) & ~size_t(255); }

// scratch layout: block norms | candidates | stream means | segment records | segment candidates
size_t acquire_scratch_bytes(int n_streams, int64_t n_samples, int max_out) {
    const size_t nb = size_t(n_samples / 64), n_seg = dip_segments(int64_t(nb));
    return al256(size_t(n_streams) * nb * sizeof(float)) + al256(size_t(n_streams) * max_out * sizeof(int64_t)) +
           al256(size_t(n_streams) * sizeof(float)) + al256(size_t(n_streams) * n_seg * sizeof(DipSegment)) +
           al256(size_t(n_streams) * n_seg * max_out * sizeof(int64_t)) + al256(size_t(n_streams) * (nb / 64 + 1) * sizeof(double));
}

hipError_t launch_acquire(const SyncTables &t, const AcquireArgs &a, hipStream_t s) {
    if (a.n_streams <= 0 || a.max_out <= 0) return hipSuccess;
    const int64_t nb = a.n_samples / 64;
    if (nb <= 0 || a.max_coarse < 0 || a.max_coarse > 1023) return hipErrorInvalidValue;
    // enough workgroups for the whole chip over ALL streams (each walks its stream's chunks with the grid's stride): a
    // launch whose streams are all skipped (auto-acquisition beside tracked streams) then costs microseconds
    const int64_t per_stream = std::max<int64_t>(16, 8192 / std::max(1, a.n_streams));
    const unsigned gx = unsigned(std::min<int64_t>(((nb + 31) / 32 + 3) / 4, std::min<int64_t>(per_stream, 4096)));
    hipLaunchKernelGGL(null_l1_kernel, dim3(gx, unsigned(a.n_streams)), dim3(256), 0, s, a.iq, a.stream_stride, nb, a.l1, a.skip_tracked);
    // the rest of the scratch buffer follows the candidate lists (acquire_scratch_bytes)
    const int n_seg = int(dip_segments(nb));
    char *p = reinterpret_cast<char *>(a.cands) + al256(size_t(a.n_streams) * a.max_out * sizeof(int64_t));
    float *avg = reinterpret_cast<float *>(p);
    p += al256(size_t(a.n_streams) * sizeof(float));
    DipSegment *segs = reinterpret_cast<DipSegment *>(p);
    p += al256(size_t(a.n_streams) * n_seg * sizeof(DipSegment));
    int64_t *seg_cands = reinterpret_cast<int64_t *>(p);
    p += al256(size_t(a.n_streams) * n_seg * a.max_out * sizeof(int64_t));
    AcquireArgs b = a;
    if (a.level_chunk) {
        if (a.level_chunk < 64 || a.level_chunk > SEG_BLOCKS || (a.level_chunk & (a.level_chunk - 1))) return hipErrorInvalidValue;
        const int64_t nc = (nb + a.level_chunk - 1) / a.level_chunk;
        double *cm = reinterpret_cast<double *>(p);
        b.chunk_mean = cm;
        hipLaunchKernelGGL(null_level_kernel, dim3(unsigned(std::min<int64_t>(nc, 4 * per_stream)), unsigned(a.n_streams)), dim3(64), 0, s, a.l1, nb,
                           a.level_chunk, nc, cm, a.skip_tracked);
    } else {
        hipLaunchKernelGGL(null_mean_kernel, dim3(unsigned(a.n_streams)), dim3(MEAN_THREADS), 0, s, a.l1, nb, avg, a.skip_tracked);
    }
    hipLaunchKernelGGL(null_segment_kernel, dim3(unsigned(n_seg), unsigned(a.n_streams)), dim3(64), 0, s, b, nb, n_seg, avg, segs,
                       seg_cands);
    hipLaunchKernelGGL(null_stitch_kernel, dim3(unsigned(a.n_streams)), dim3(64), 0, s, b, n_seg, segs, seg_cands);
    hipLaunchKernelGGL(prs_sync_kernel<MODE_ACQ>, dim3(sync_grid(a.n_streams * a.max_out)), dim3(WG), 0, s, t,
                       static_cast<const float2 *>(nullptr), size_t(0), static_cast<const float *>(nullptr), 0,
                       static_cast<SyncResult *>(nullptr), b, TrackArgs{}, a.n_streams * a.max_out);
    return hipGetLastError();
}

}  // namespace dabk
