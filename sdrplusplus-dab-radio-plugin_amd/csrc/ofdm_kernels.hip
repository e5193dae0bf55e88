// ofdm_kernels.hip -- OFDM front end for DAB Mode I on gfx950 (CDNA4).
//
// Rows A2..A6 of SURVEY.md section 8a, i.e. what OFDM_Demod::Process does per frame in
// its READING_SYMBOLS state (/root/reference/src/dab_module.cpp:25) before the
// On_OFDM_Frame callback (/root/reference/src/radio_block.cpp:25):
//   A2  NCO frequency correction          y[n] = x[n] * exp(+j*2*pi*phase(n))
//   A3  2048-point forward FFT per symbol (the reference plans FFTW3f, CMakeLists.txt:55-64)
//   A4  differential demodulation         d[k] = X_l[k] * conj(X_{l-1}[k])
//   A5  frequency de-interleave           gather through the carrier mapper
//   A6  L-infinity normalise + int8 soft-bit quantise
// plus the cyclic-prefix correlation that drives the fine-frequency loop.
//
// One 256-thread workgroup walks `syms_per_group` consecutive data symbols of one
// frame (plus the symbol before them as differential reference), keeping the previous
// spectrum in LDS so every IQ sample is read from HBM once.  The FFT is a Stockham
// autosort radix-8/8/8/4 through LDS; the first radix-8 pass is fed straight from the
// coalesced global loads with the NCO rotation applied in registers.
#include "kernels.hpp"
#include "dab_tables.hpp"

namespace dabk {

using namespace dab;

namespace {

constexpr int WG = 256;
constexpr float SQRT1_2 = 0.70710678118654752440f;

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// multiply by -j
__device__ __forceinline__ float2 mul_mj(float2 a) { return make_float2(a.y, -a.x); }

__device__ __forceinline__ void fft4(float2 &a, float2 &b, float2 &c, float2 &d) {
    const float2 t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = mul_mj(csub(b, d));
    a = cadd(t0, t2);
    b = cadd(t1, t3);
    c = csub(t0, t2);
    d = csub(t1, t3);
}

// in-place 8-point forward DFT, natural order in and out
__device__ __forceinline__ void fft8(float2 *v) {
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    fft4(e0, e1, e2, e3);
    fft4(o0, o1, o2, o3);
    // W8^k * O[k]
    o1 = make_float2((o1.x + o1.y) * SQRT1_2, (o1.y - o1.x) * SQRT1_2);   // (1-j)/sqrt2
    o2 = mul_mj(o2);
    o3 = make_float2((o3.y - o3.x) * SQRT1_2, -(o3.x + o3.y) * SQRT1_2);  // (-1-j)/sqrt2
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

// NCO: unit phasor for sample index n (relative to the first PRS sample).
__device__ __forceinline__ float2 nco(uint32_t n, uint32_t dphi) {
    const int32_t ph = int32_t(n * dphi);
    const float rev2 = float(ph) * (1.0f / 2147483648.0f);   // 2 * revolutions in [-1,1)
    float s, c;
    sincospif(rev2, &s, &c);
    return make_float2(c, s);
}

__device__ __forceinline__ uint32_t dphi_of(const float *freq_offset, int frame) {
    if (!freq_offset) return 0u;
    const long long q = __double2ll_rn(double(freq_offset[frame]) * 4294967296.0);
    return uint32_t(q);
}

struct Smem {
    float2 tw[NB_FFT];
    float2 t1[NB_FFT];
    float2 x[2][NB_FFT];
    float2 red[4];
};

// One Stockham pass with radix 8 from LDS to LDS.
template <int NS>
__device__ __forceinline__ void pass8(const float2 *src, float2 *dst, const float2 *tw, int j) {
    float2 v[8];
    const int k = j & (NS - 1);
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = src[j + r * (NB_FFT / 8)];
#pragma unroll
    for (int r = 1; r < 8; r++) v[r] = cmul(v[r], tw[r * k * (NB_FFT / (NS * 8))]);
    fft8(v);
    const int j0 = (j - k) * 8 + k;
#pragma unroll
    for (int r = 0; r < 8; r++) dst[j0 + r * NS] = v[r];
}

template <bool FFT_ONLY>
__global__ __launch_bounds__(WG) void ofdm_kernel(OfdmTables tab, OfdmArgs a, int syms_per_group,
                                                  int groups_per_frame) {
    __shared__ Smem sm;
    const int tid = threadIdx.x;
    const int frame = blockIdx.x / groups_per_frame;
    const int group = blockIdx.x % groups_per_frame;
    const float2 *fiq = a.iq + size_t(frame) * a.frame_stride;
    const uint32_t dphi = dphi_of(a.freq_offset, frame);

    for (int i = tid; i < NB_FFT; i += WG) sm.tw[i] = tab.twiddle[i];

    // data indices this thread quantises: n0..n0+7 (threads 0..191)
    uint16_t bins[8];
    if (!FFT_ONLY && tid < NB_CARRIERS / 8) {
#pragma unroll
        for (int q = 0; q < 8; q++) bins[q] = tab.bin_of_n[tid * 8 + q];
    }

    // symbols [l_first, l_last]; in fused mode l_first is only the differential reference
    const int l_first = FFT_ONLY ? group * syms_per_group : group * syms_per_group;
    const int l_last = FFT_ONLY ? l_first + syms_per_group - 1 : l_first + syms_per_group;
    __syncthreads();

    for (int l = l_first; l <= l_last; l++) {
        const float2 *sym = fiq + size_t(l) * NB_SYM_PERIOD;
        float2 *xc = sm.x[l & 1];
        const float2 *xp = sm.x[(l & 1) ^ 1];
        const bool emit = FFT_ONLY || (l > l_first) || (l == 0);   // who owns the cyc of symbol l

        // ---- A2 + first radix-8 pass straight from global memory ----
        {
            float2 v[8];
            const uint32_t nbase = uint32_t(l * NB_SYM_PERIOD + NB_CP + tid);
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = sym[NB_CP + tid + r * WG];
            if (dphi != 0u) {
#pragma unroll
                for (int r = 0; r < 8; r++) v[r] = cmul(v[r], nco(nbase + uint32_t(r * WG), dphi));
            }
            fft8(v);
#pragma unroll
            for (int r = 0; r < 8; r++) sm.t1[tid * 8 + r] = v[r];
        }
        // ---- cyclic-prefix correlation (on raw samples, rotated once at the end) ----
        float2 acc = make_float2(0.f, 0.f);
        if (a.cyc && emit && tid < NB_CP / 2) {
            const float4 p = *reinterpret_cast<const float4 *>(sym + 2 * tid);
            const float4 q = *reinterpret_cast<const float4 *>(sym + NB_FFT + 2 * tid);
            // conj(p) * q
            acc.x = p.x * q.x + p.y * q.y + p.z * q.z + p.w * q.w;
            acc.y = p.x * q.y - p.y * q.x + p.z * q.w - p.w * q.z;
        }
        __syncthreads();
        pass8<8>(sm.t1, xc, sm.tw, tid);
        __syncthreads();
        pass8<64>(xc, sm.t1, sm.tw, tid);
        __syncthreads();
        // ---- last pass: radix 4, NS = 512, two work items per thread ----
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int j = tid + h * WG;   // k == j since j < 512
            float2 v0 = sm.t1[j], v1 = sm.t1[j + 512], v2 = sm.t1[j + 1024], v3 = sm.t1[j + 1536];
            v1 = cmul(v1, sm.tw[j]);
            v2 = cmul(v2, sm.tw[2 * j]);
            v3 = cmul(v3, sm.tw[3 * j]);
            fft4(v0, v1, v2, v3);
            if (FFT_ONLY) {
                float2 *o = a.spectra + (size_t(frame) * NB_FRAME_SYMBOLS + l) * NB_FFT;
                o[j] = v0; o[j + 512] = v1; o[j + 1024] = v2; o[j + 1536] = v3;
            } else {
                xc[j] = v0; xc[j + 512] = v1; xc[j + 1024] = v2; xc[j + 1536] = v3;
            }
        }
        // ---- reduce the cyclic-prefix correlation ----
        if (a.cyc && emit) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                acc.x += __shfl_down(acc.x, off);
                acc.y += __shfl_down(acc.y, off);
            }
            if ((tid & 63) == 0) sm.red[tid >> 6] = acc;
        }
        __syncthreads();
        if (a.cyc && emit && tid == 0) {
            float2 c = cadd(cadd(sm.red[0], sm.red[1]), cadd(sm.red[2], sm.red[3]));
            c = cmul(c, nco(uint32_t(NB_FFT), dphi));   // conj(w_i) * w_{i+2048}
            a.cyc[size_t(frame) * NB_FRAME_SYMBOLS + l] = c;
        }
        if (FFT_ONLY || l == l_first) continue;

        // ---- A4..A6: differential demod, frequency de-interleave, quantise ----
        if (tid < NB_CARRIERS / 8) {
            uint32_t re_lo = 0, re_hi = 0, im_lo = 0, im_hi = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const float2 d = cmulc(xc[bins[q]], xp[bins[q]]);
                const float A = fmaxf(fabsf(d.x), fabsf(d.y));
                int br = 0, bi = 0;
                if (A != 0.0f) {
                    br = __float2int_rz(-127.0f * (d.x / A));
                    bi = __float2int_rz(-127.0f * (d.y / A));
                }
                const uint32_t ur = uint32_t(br) & 0xFFu, ui = uint32_t(bi) & 0xFFu;
                if (q < 4) { re_lo |= ur << (8 * q); im_lo |= ui << (8 * q); }
                else       { re_hi |= ur << (8 * (q - 4)); im_hi |= ui << (8 * (q - 4)); }
            }
            int8_t *o = a.soft + size_t(frame) * NB_FRAME_BITS + size_t(l - 1) * NB_SYM_BITS + tid * 8;
            *reinterpret_cast<uint2 *>(o) = make_uint2(re_lo, re_hi);
            *reinterpret_cast<uint2 *>(o + NB_CARRIERS) = make_uint2(im_lo, im_hi);
        }
        if (a.dqpsk) {
            float2 *o = a.dqpsk + (size_t(frame) * NB_DATA_SYMBOLS + (l - 1)) * NB_CARRIERS;
            for (int i = tid; i < NB_CARRIERS; i += WG) {
                const int bin = (i < NB_CARRIERS / 2) ? (NB_FFT - NB_CARRIERS / 2 + i) : (i - NB_CARRIERS / 2 + 1);
                o[i] = cmulc(xc[bin], xp[bin]);
            }
        }
        // the barrier after the next symbol's first pass orders these LDS reads before
        // the next overwrite of x[]
    }
}

}  // namespace

hipError_t launch_ofdm_demod(const OfdmTables &t, const OfdmArgs &a, int syms_per_group, hipStream_t s) {
    if (a.n_frames <= 0) return hipSuccess;
    if (syms_per_group <= 0 || NB_DATA_SYMBOLS % syms_per_group) return hipErrorInvalidValue;
    const int groups = NB_DATA_SYMBOLS / syms_per_group;
    hipLaunchKernelGGL(ofdm_kernel<false>, dim3(unsigned(a.n_frames) * groups), dim3(WG), 0, s, t, a,
                       syms_per_group, groups);
    return hipGetLastError();
}

hipError_t launch_fft_symbols(const OfdmTables &t, const OfdmArgs &a, int syms_per_group, hipStream_t s) {
    if (a.n_frames <= 0) return hipSuccess;
    if (syms_per_group <= 0 || NB_FRAME_SYMBOLS % syms_per_group) return hipErrorInvalidValue;
    const int groups = NB_FRAME_SYMBOLS / syms_per_group;
    hipLaunchKernelGGL(ofdm_kernel<true>, dim3(unsigned(a.n_frames) * groups), dim3(WG), 0, s, t, a,
                       syms_per_group, groups);
    return hipGetLastError();
}

}  // namespace dabk
