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

#include "kernels.hpp"
#include "dab_tables.hpp"
#include "fft_common.hpp"
#include "mem_stream.hpp"

namespace dabk {

using namespace dab;

namespace {

struct SyncLds {
    double red_d[2][WG];        // acquisition: cyclic-prefix correlation
    float2 tw[NB_FFT];
    float2 t1[NB_FFT];
    float2 x[NB_FFT];
    float2 y[NB_FFT];           // coarse search: spectrum of Q, then the correlation over all 2048 shifts
    int8_t qt[NB_FFT];          // quarter turns of the PRS per bin, -1 = not a carrier
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

// block-wide argmax with "first maximum" semantics (smaller index wins ties) and the sum of all values
__device__ __forceinline__ void block_argmax_sum(SyncLds &sm, int tid, float m, int idx, float s, float &best_m,
                                                 int &best_i, float &total) {
    sm.red_m[tid] = m;
    sm.red_i[tid] = idx;
    sm.red_s[tid] = s;
    __syncthreads();
    for (int off = WG / 2; off > 0; off >>= 1) {
        if (tid < off) {
            const float m2 = sm.red_m[tid + off];
            const int i2 = sm.red_i[tid + off];
            if (m2 > sm.red_m[tid] || (m2 == sm.red_m[tid] && i2 < sm.red_i[tid])) {
                sm.red_m[tid] = m2;
                sm.red_i[tid] = i2;
            }
            sm.red_s[tid] += sm.red_s[tid + off];
        }
        __syncthreads();
    }
    best_m = sm.red_m[0];
    best_i = sm.red_i[0];
    total = sm.red_s[0];
    __syncthreads();
}

// ACQ = false: candidates at a fixed stride, correction given, SyncResult out.
// ACQ = true : candidates from the null search; the fractional frequency error is first measured on the cyclic
//              prefix of the PRS (products 64..439 of the prefix against the samples 2048 later: inside the prefix
//              for any candidate within +-64 samples), then the same search; AcquiredFrame out.
template <bool ACQ>
__global__ __launch_bounds__(WG) void prs_sync_kernel(SyncTables tab, const float2 *iq, size_t frame_stride,
                                                      const float *freq_offset, int max_coarse, SyncResult *out,
                                                      AcquireArgs acq) {
    __shared__ SyncLds sm;
    const int tid = threadIdx.x;
    const int frame = blockIdx.x;
    const float2 *sym;
    uint32_t dphi;
    int64_t cand = 0;
    float fine = 0.0f;
    if constexpr (ACQ) {
        const int st = frame / acq.max_out, j = frame - st * acq.max_out;
        if (j >= acq.counts[st]) {
            if (tid == 0) acq.out[frame] = AcquiredFrame{-1, 0.f, 0, 0.f, 0.f, 0.f, 0};
            return;
        }
        cand = acq.cands[frame];
        sym = acq.iq + size_t(st) * acq.stream_stride + cand;
        double cr = 0.0, ci = 0.0;
        for (int i = 64 + tid; i < 440; i += WG) {
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
        fine = float(-atan2(sm.red_d[1][0], sm.red_d[0][0]) / (2.0 * 3.14159265358979323846 * double(NB_FFT)));
        dphi = uint32_t(__double2ll_rn(double(fine) * 4294967296.0));
        max_coarse = acq.max_coarse;
    } else {
        sym = iq + size_t(frame) * frame_stride;
        dphi = dphi_of(freq_offset, frame);
    }
    for (int i = tid; i < NB_FFT; i += WG) {
        sm.tw[i] = tab.twiddle[i];
        sm.qt[i] = tab.prs_qt[i];
    }
    __syncthreads();
    // ---- X = FFT(nco * window) ----
    {
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int n = tid + r * WG;
            v[r] = sym[NB_CP + n];
            if (dphi != 0u) v[r] = cmul(v[r], nco(uint32_t(n), dphi));
        }
        block_fft2048(v, sm.t1, sm.x, sm.tw, tid);
    }
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
        __syncthreads();                                       // t1 is the transform's scratch from here on
        block_fft2048(v, sm.t1, sm.y, sm.tw, tid);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int b = tid + r * WG;
            const float2 z = cmulc(sm.y[b], tab.pair_spectrum[b]);
            v[r] = make_float2(z.x, -z.y);
        }
        __syncthreads();
        block_fft2048(v, sm.t1, sm.y, sm.tw, tid);
    }
    float my_m = -1.0f, my_s = 0.0f;
    int my_k = 0x7fffffff;
    for (int idx = tid; idx <= 2 * max_coarse; idx += WG) {
        const float2 d = sm.y[(idx - max_coarse) & (NB_FFT - 1)];
        const float m = d.x * d.x + d.y * d.y;
        my_s += m;
        if (m > my_m) { my_m = m; my_k = idx; }
    }
    float best_m, total;
    int best_idx;
    block_argmax_sum(sm, tid, my_m, my_k, my_s, best_m, best_idx, total);
    const int khat = best_idx - max_coarse;
    const float coarse_ptm = best_m / (total / float(2 * max_coarse + 1));

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
        block_fft2048(v, sm.t1, sm.x, sm.tw, tid);
    }
    my_m = -1.0f;
    my_s = 0.0f;
    int my_n = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int n = tid + r * WG;
        const float2 h = sm.x[n];
        const float m = h.x * h.x + h.y * h.y;
        my_s += m;
        if (m > my_m) { my_m = m; my_n = n; }
    }
    int best_n;
    block_argmax_sum(sm, tid, my_m, my_n, my_s, best_m, best_n, total);
    if (tid == 0) {
        const int toff = best_n < NB_FFT / 2 ? best_n : best_n - NB_FFT;
        const float ptm = best_m / (total / float(NB_FFT));
        if constexpr (ACQ) {
            AcquiredFrame r;
            r.start = cand + toff - acq.margin;
            r.coarse_carriers = khat;
            r.fine_offset = fine;
            r.freq_offset = __fsub_rn(fine, float(khat) / float(NB_FFT));
            r.peak_to_mean = ptm;
            r.coarse_peak_to_mean = coarse_ptm;
            r.flags = (ptm >= acq.min_peak_to_mean ? 1 : 0) |
                      ((r.start >= 0 && r.start + int64_t(NB_FRAME_SYMBOLS) * NB_SYM_PERIOD <= acq.n_samples) ? 2 : 0);
            acq.out[frame] = r;
        } else {
            SyncResult r;
            r.coarse_carriers = khat;
            r.time_offset = toff;
            r.peak_to_mean = ptm;
            r.coarse_peak_to_mean = coarse_ptm;
            out[frame] = r;
        }
    }
}

// ---- null-symbol search (FINDING_NULL_POWER_DIP, /root/reference/src/render_radio_block.cpp:193) ----
// L1 norm of every 64-sample block.  A wave takes two blocks per trip: lane j of a half holds samples 2j, 2j+1,
// the 32 pair sums are combined by the XOR butterfly 1, 2, 4, 8, 16 (the fixed tree the oracle restates).  A wave
// owns 32 consecutive blocks (16 KB of samples, four trips' loads in flight at a time) and writes their norms as ONE
// 128-byte store: two dwords per wave, as a first version did, are partial cache lines that eight XCDs' L2s each
// hold a piece of.
__global__ __launch_bounds__(256) void null_l1_kernel(const float2 *iq, size_t stream_stride, int64_t nb, float *l1) {
    const int st = blockIdx.y;
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
__global__ __launch_bounds__(MEAN_THREADS) void null_mean_kernel(const float *l1_all, int64_t nb, float *avg) {
    __shared__ double part[MEAN_THREADS / 64];
    const int st = blockIdx.x, tid = threadIdx.x;
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
    const float *l1 = a.l1 + size_t(st) * nb;
    const float mean = avg[st];
    const float ts = __fmul_rn(a.thr_start, mean), te = __fmul_rn(a.thr_end, mean);
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

hipError_t launch_prs_sync(const SyncTables &t, const float2 *iq, size_t frame_stride, int n_frames,
                           const float *freq_offset, int max_coarse, SyncResult *out, hipStream_t s) {
    if (n_frames <= 0) return hipSuccess;
    if (max_coarse < 0 || max_coarse > 1023) return hipErrorInvalidValue;
    hipLaunchKernelGGL(prs_sync_kernel<false>, dim3(unsigned(n_frames)), dim3(WG), 0, s, t, iq, frame_stride,
                       freq_offset, max_coarse, out, AcquireArgs{});
    return hipGetLastError();
}

static size_t dip_segments(int64_t nb) { return size_t((nb + SEG_BLOCKS - 1) / SEG_BLOCKS); }
static size_t al256(size_t v) { return (v + 255) & ~size_t(255); }

// scratch layout: block norms | candidates | stream means | segment records | segment candidates
size_t acquire_scratch_bytes(int n_streams, int64_t n_samples, int max_out) {
    const size_t nb = size_t(n_samples / 64), n_seg = dip_segments(int64_t(nb));
    return al256(size_t(n_streams) * nb * sizeof(float)) + al256(size_t(n_streams) * max_out * sizeof(int64_t)) +
           al256(size_t(n_streams) * sizeof(float)) + al256(size_t(n_streams) * n_seg * sizeof(DipSegment)) +
           size_t(n_streams) * n_seg * max_out * sizeof(int64_t);
}

hipError_t launch_acquire(const SyncTables &t, const AcquireArgs &a, hipStream_t s) {
    if (a.n_streams <= 0 || a.max_out <= 0) return hipSuccess;
    const int64_t nb = a.n_samples / 64;
    if (nb <= 0 || a.max_coarse < 0 || a.max_coarse > 1023) return hipErrorInvalidValue;
    const unsigned gx = unsigned(std::min<int64_t>(((nb + 31) / 32 + 3) / 4, 4096));
    hipLaunchKernelGGL(null_l1_kernel, dim3(gx, unsigned(a.n_streams)), dim3(256), 0, s, a.iq, a.stream_stride, nb, a.l1);
    // the rest of the scratch buffer follows the candidate lists (acquire_scratch_bytes)
    const int n_seg = int(dip_segments(nb));
    char *p = reinterpret_cast<char *>(a.cands) + al256(size_t(a.n_streams) * a.max_out * sizeof(int64_t));
    float *avg = reinterpret_cast<float *>(p);
    p += al256(size_t(a.n_streams) * sizeof(float));
    DipSegment *segs = reinterpret_cast<DipSegment *>(p);
    p += al256(size_t(a.n_streams) * n_seg * sizeof(DipSegment));
    int64_t *seg_cands = reinterpret_cast<int64_t *>(p);
    hipLaunchKernelGGL(null_mean_kernel, dim3(unsigned(a.n_streams)), dim3(MEAN_THREADS), 0, s, a.l1, nb, avg);
    hipLaunchKernelGGL(null_segment_kernel, dim3(unsigned(n_seg), unsigned(a.n_streams)), dim3(64), 0, s, a, nb, n_seg, avg, segs,
                       seg_cands);
    hipLaunchKernelGGL(null_stitch_kernel, dim3(unsigned(a.n_streams)), dim3(64), 0, s, a, n_seg, segs, seg_cands);
    hipLaunchKernelGGL(prs_sync_kernel<true>, dim3(unsigned(a.n_streams) * unsigned(a.max_out)), dim3(WG), 0, s, t,
                       static_cast<const float2 *>(nullptr), size_t(0), static_cast<const float *>(nullptr), 0,
                       static_cast<SyncResult *>(nullptr), a);
    return hipGetLastError();
}

}  // namespace dabk
