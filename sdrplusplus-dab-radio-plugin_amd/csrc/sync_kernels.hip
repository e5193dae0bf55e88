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
#include "kernels.hpp"
#include "dab_tables.hpp"
#include "fft_common.hpp"

namespace dabk {

using namespace dab;

namespace {

struct SyncLds {
    float2 tw[NB_FFT];
    float2 t1[NB_FFT];
    float2 x[NB_FFT];
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

__global__ __launch_bounds__(WG) void prs_sync_kernel(SyncTables tab, const float2 *iq, size_t frame_stride,
                                                      const float *freq_offset, int max_coarse, SyncResult *out) {
    __shared__ SyncLds sm;
    const int tid = threadIdx.x;
    const int frame = blockIdx.x;
    const float2 *sym = iq + size_t(frame) * frame_stride;
    const uint32_t dphi = dphi_of(freq_offset, frame);
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
    // ---- coarse frequency: scan k = -max..+max, this thread takes every 256th candidate ----
    float my_m = -1.0f, my_s = 0.0f;
    int my_k = 0x7fffffff;
    for (int idx = tid; idx <= 2 * max_coarse; idx += WG) {
        const int k = idx - max_coarse;
        float dr = 0.f, di = 0.f;
        for (int e = 0; e < tab.n_pairs; e++) {
            const unsigned pr = tab.pairs[e];                      // bin | s << 11 (wave-uniform load)
            const float2 q = sm.t1[(int(pr & 2047u) + k) & (NB_FFT - 1)];
            switch (pr >> 11) {
            case 0: dr += q.x; di += q.y; break;
            case 1: dr += q.y; di -= q.x; break;
            case 2: dr -= q.x; di -= q.y; break;
            default: dr -= q.y; di += q.x; break;
            }
        }
        const float m = dr * dr + di * di;
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
        SyncResult r;
        r.coarse_carriers = khat;
        r.time_offset = best_n < NB_FFT / 2 ? best_n : best_n - NB_FFT;
        r.peak_to_mean = best_m / (total / float(NB_FFT));
        r.coarse_peak_to_mean = coarse_ptm;
        out[frame] = r;
    }
}

}  // namespace

hipError_t launch_prs_sync(const SyncTables &t, const float2 *iq, size_t frame_stride, int n_frames,
                           const float *freq_offset, int max_coarse, SyncResult *out, hipStream_t s) {
    if (n_frames <= 0) return hipSuccess;
    if (max_coarse < 0 || max_coarse > 1023) return hipErrorInvalidValue;
    hipLaunchKernelGGL(prs_sync_kernel, dim3(unsigned(n_frames)), dim3(WG), 0, s, t, iq, frame_stride, freq_offset,
                       max_coarse, out);
    return hipGetLastError();
}

}  // namespace dabk
