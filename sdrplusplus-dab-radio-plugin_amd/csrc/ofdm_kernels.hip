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
#include <cstdlib>

#include "kernels.hpp"
#include "dab_tables.hpp"
#include "fft_common.hpp"

namespace dabk {

using namespace dab;

namespace {

struct Smem {
    float2 tw[NB_FFT];
    float2 t1[NB_FFT];
    float2 x[2][NB_FFT];
    float2 red[4];
};

template <bool FFT_ONLY>
__global__ __launch_bounds__(WG) void ofdm_kernel(OfdmTables tab, OfdmArgs a, int parts) {
    __shared__ Smem sm;
    const int tid = threadIdx.x;
    const int frame = blockIdx.x / parts;
    const int part = blockIdx.x % parts;
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
    // fused: data symbols (l_first, l_last], l_first is only the differential reference
    const int l_first = FFT_ONLY ? (NB_FRAME_SYMBOLS * part) / parts : (NB_DATA_SYMBOLS * part) / parts;
    const int l_last = FFT_ONLY ? (NB_FRAME_SYMBOLS * (part + 1)) / parts - 1 : (NB_DATA_SYMBOLS * (part + 1)) / parts;
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


// ============================================================================
// Kernel 2 ("wave" variant): one 64-lane wavefront owns a run of consecutive symbols of
// one frame and keeps a whole 2048-point symbol in registers (32 complex per lane).
//
//   n = 128*n1 + 8*n2 + n3   (n1<16, n2<16, n3<8)      k = k1 + 16*k2 + 256*k3
//   step 1  lane (n2, n3/2): two 16-point DFTs over n1   (inputs are the lane's own 16-byte loads)
//           twiddle W256^(n2*k1), exchange through LDS
//   step 2  lane (k1, n3/2): two 16-point DFTs over n2,  twiddle W2048^(n3*(k1+16*k2)), exchange
//   step 3  lane v = (k1 | k2%4 << 4): four 8-point DFTs over n3 -> bins v + 64*m, m = 0..31
//
// Because every lane ends each symbol with the SAME bins, the previous spectrum stays in
// registers and the differential demodulation needs no memory at all.  The two LDS
// exchanges use XOR-swizzled layouts that are bank-conflict free for both the
// ds_write_b64 and the ds_read_b64 side (tools: see DESIGN.md).  No workgroup barrier is
// needed after the twiddle table is loaded: a wave only talks to itself.
// Soft bits are scattered as bytes into the (then idle) exchange buffer and leave as
// three coalesced 16-byte stores per lane.
// ============================================================================
#ifndef DAB_OFDM_WAVES
#define DAB_OFDM_WAVES 4
#endif
constexpr int WAVES = DAB_OFDM_WAVES;

struct WaveLds {
    float2 tw[NB_FFT];               // exp(-2*pi*i*m/2048)                                   16 KB
    uint32_t nidx[12 * 64];          // frequency de-interleave table, two data indices per dword  3 KB
    float2 ex[WAVES][NB_FFT / 2];    // per-wave exchange buffer (half a symbol per pass), reused
                                     // as soft-bit staging                                    8 KB each
};   // 51 KB per 4-wave workgroup -> 3 workgroups = 12 waves per CU

__device__ __forceinline__ float2 cmul_k(float2 a, float c, float s) {   // a * (c + j*s)
    return make_float2(a.x * c - a.y * s, a.x * s + a.y * c);
}

// in-place 16-point forward DFT, natural order in and out (radix 4 x 4)
__device__ __forceinline__ void fft16(float2 *x) {
    constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;
#pragma unroll
    for (int b = 0; b < 4; b++) fft4(x[b], x[4 + b], x[8 + b], x[12 + b]);   // x[4c+b] = Y[b][c]
    // Y[b][c] *= W16^(b*c)
    x[5] = cmul_k(x[5], C1, -S1);                 // W16^1
    x[6] = cmul_k(x[6], SQRT1_2, -SQRT1_2);       // W16^2
    x[7] = cmul_k(x[7], S1, -C1);                 // W16^3
    x[9] = cmul_k(x[9], SQRT1_2, -SQRT1_2);       // W16^2
    x[10] = mul_mj(x[10]);                        // W16^4
    x[11] = cmul_k(x[11], -SQRT1_2, -SQRT1_2);    // W16^6
    x[13] = cmul_k(x[13], S1, -C1);               // W16^3
    x[14] = cmul_k(x[14], -SQRT1_2, -SQRT1_2);    // W16^6
    x[15] = cmul_k(x[15], -C1, S1);               // W16^9
    float2 o[16];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float2 a = x[4 * c], b = x[4 * c + 1], d = x[4 * c + 2], e = x[4 * c + 3];
        fft4(a, b, d, e);
        o[c] = a; o[c + 4] = b; o[c + 8] = d; o[c + 12] = e;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = o[i];
}

// wave-wide sum by XOR butterflies on the VALU (DPP / permlane swaps), result in every lane
template <int XORMASK>
__device__ __forceinline__ float lane_xor_f(float v, int lane) {
    const int m = __float_as_int(v);
    int r;
    if constexpr (XORMASK == 1) {
        r = __builtin_amdgcn_update_dpp(0, m, 0xB1, 0xF, 0xF, false);
    } else if constexpr (XORMASK == 2) {
        r = __builtin_amdgcn_update_dpp(0, m, 0x4E, 0xF, 0xF, false);
    } else if constexpr (XORMASK == 4) {
        r = __builtin_amdgcn_update_dpp(0, __builtin_amdgcn_update_dpp(0, m, 0x141, 0xF, 0xF, false), 0x1B, 0xF, 0xF, false);
    } else if constexpr (XORMASK == 8) {
        r = __builtin_amdgcn_update_dpp(0, m, 0x128, 0xF, 0xF, false);
    } else if constexpr (XORMASK == 16) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const u2 t = __builtin_amdgcn_permlane16_swap(unsigned(m), unsigned(m), false, false);
        r = (lane & 16) ? int(t.x) : int(t.y);
    } else {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const u2 t = __builtin_amdgcn_permlane32_swap(unsigned(m), unsigned(m), false, false);
        r = (lane & 32) ? int(t.x) : int(t.y);
    }
    return __int_as_float(r);
}
__device__ __forceinline__ float wave_sum(float v, int lane) {
    v += lane_xor_f<1>(v, lane);
    v += lane_xor_f<2>(v, lane);
    v += lane_xor_f<4>(v, lane);
    v += lane_xor_f<8>(v, lane);
    v += lane_xor_f<16>(v, lane);
    v += lane_xor_f<32>(v, lane);
    return v;
}

// SELECT: soft-bit selection table in use (a separate instantiation so that the plain kernel keeps its registers)
// NCO: a frequency correction is applied (the launch has a freq_offset array or acquired frames).  A compile-time
// switch, not a per-frame branch: the branch cost 31 register moves per symbol where its two paths re-joined.
template <bool FFT_ONLY, bool WITH_DQPSK, bool SELECT = false, bool NCO = true>
__global__ __launch_bounds__(64 * WAVES, 3) void ofdm_wave_kernel(OfdmTables tab, OfdmArgs a, int parts, int n_items) {
    __shared__ WaveLds sm;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    for (int i = tid; i < NB_FFT; i += 64 * WAVES) sm.tw[i] = tab.twiddle[i];
    for (int i = tid; i < 12 * 64; i += 64 * WAVES) sm.nidx[i] = reinterpret_cast<const uint32_t *>(tab.n_of_vj)[i];
    __syncthreads();
    // The item is the same for all lanes of a wave; saying so keeps everything derived from it (frame pointers, the
    // symbol counter, the NCO increment) in SGPRs: -5 % VALU instructions, -8 % time.
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + wave);
    if (item >= n_items) return;
    const int frame = item / parts;
    const int part = item - frame * parts;
    const float2 *fiq = a.iq + size_t(frame) * a.frame_stride;
    // fused: data symbols (l_first, l_last]; l_first is only the differential reference
    const int l_first = FFT_ONLY ? (NB_FRAME_SYMBOLS * part) / parts : (NB_DATA_SYMBOLS * part) / parts;
    const int l_last = FFT_ONLY ? (NB_FRAME_SYMBOLS * (part + 1)) / parts - 1 : (NB_DATA_SYMBOLS * (part + 1)) / parts;
    uint32_t dphi_acq = 0u;
    if (a.acq) {
        // frames found by the acquisition kernels: arbitrary (8-byte aligned) start inside their stream
        const AcquiredFrame m = a.acq[frame];
        if ((m.flags & 3) != 3) {
            if constexpr (!FFT_ONLY) {                        // not a demodulable frame: erased soft bits
                uint4 *o = reinterpret_cast<uint4 *>(a.soft + size_t(frame) * NB_FRAME_BITS + size_t(l_first) * NB_SYM_BITS);
                const int n16 = (l_last - l_first) * NB_SYM_BITS / 16;
                for (int i = lane; i < n16; i += 64) o[i] = make_uint4(0u, 0u, 0u, 0u);
            }
            return;
        }
        fiq = a.iq + size_t(frame / a.acq_per_stream) * a.frame_stride + m.start;
        dphi_acq = uint32_t(__double2ll_rn(double(m.freq_offset) * 4294967296.0));
    }
    const bool aligned16 = (reinterpret_cast<uintptr_t>(fiq) & 15u) == 0;
#ifdef DAB_OFDM_STAGGER
    // de-phase the co-resident waves so that their load / FFT / LDS phases do not line up
    for (int i = 0; i < (item % 12) * DAB_OFDM_STAGGER; i++) __builtin_amdgcn_s_sleep(32);
#endif
#ifdef DAB_EXP_NOPLL
    const uint32_t dphi = 0u;
#else
    const uint32_t dphi = a.acq ? dphi_acq : dphi_of(a.freq_offset, frame);
#endif
#ifdef DAB_EXP_NOCYC
    a.cyc = nullptr;
#endif
    float2 *ex = sm.ex[wave];
    const float2 *tw = sm.tw;

    // lane roles
    const int n2 = lane >> 2, p = lane & 3;            // step 1 (and k1 = n2, p' = p in step 2)
    const int k1v = lane & 15, kk = lane >> 4;         // step 3
    // exchange-1 slots (one pass per e = n3 & 1): (n3/2)*256 + k1*16 + (n2 ^ (n3/2)<<2 ^ (k1/2)&3)
    const int w1_base = p * 256, w1_r = n2 ^ (p << 2);
    // exchange-2 slots (one pass per half of k2): n3*128 + (k2 % 8)*16 + (k1 ^ (n3/2)<<2)
    const int w2_base = p * 256 + (n2 ^ (p << 2));
    const int r2_base = kk * 16;

    // step-1 twiddles W256^(n2*k1) are the same for every symbol: keep them in registers? (30 VGPRs) -- no,
    // they are re-read from LDS per symbol; the table index is 8*n2*k1.
    // frequency de-interleave: data index of each carrier register (24 per lane)
    float2 prev[24];
#pragma unroll
    for (int j = 0; j < 24; j++) prev[j] = make_float2(0.f, 0.f);

    const float2 r1 = nco(1u, dphi), r128 = nco(128u, dphi), rot2048 = nco(uint32_t(NB_FFT), dphi);

    for (int l = l_first; l <= l_last; l++) {
        const float2 *sym = fiq + size_t(l) * NB_SYM_PERIOD;
        const bool emit = FFT_ONLY || (l > l_first) || (l == 0);
        // Opaque per-iteration copies of the lane roles: every LDS address below is one or two VALU ops
        // from these, which is cheaper than letting LICM park ~60 loop-invariant addresses in VGPRs (spills).
        int n2i = n2, pi = p;
        asm volatile("" : "+v"(n2i), "+v"(pi));
        // phasor of this lane's first sample, computed before the loads so sincospi's temporaries are dead
        // by the time 64 data registers are live
        if constexpr (SELECT) {
            // A symbol is transformed only if some of its own soft bits are wanted or it is the differential
            // reference of a wanted one.  The others give at most their cyclic-prefix correlation, which needs the
            // prefix and the last 512 samples only (8 of the 20 loads).
            const int ls = __builtin_amdgcn_readfirstlane(l);
            bool need = false;
            if (ls > l_first) { const unsigned long long *kw = a.keep + 3 * (ls - 1); need |= (kw[0] | kw[1] | kw[2]) != 0ull; }
            if (ls < l_last) { const unsigned long long *kw = a.keep + 3 * ls; need |= (kw[0] | kw[1] | kw[2]) != 0ull; }
            if (!need) {
                if (a.cyc && emit) {
                    float2 acc = make_float2(0.f, 0.f);
                    const float2 *cp = sym + 2 * (lane - 4);
                    const float2 *tail = sym + NB_CP + 128 * 12 + 2 * lane;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (i > 0 || lane >= 4) {
                            const float2 c0 = cp[128 * i], c1 = cp[128 * i + 1];
                            const float2 u0 = tail[128 * i], u1 = tail[128 * i + 1];
                            acc.x += c0.x * u0.x + c0.y * u0.y + c1.x * u1.x + c1.y * u1.y;      // conj(c) * u
                            acc.y += c0.x * u0.y - c0.y * u0.x + c1.x * u1.y - c1.y * u1.x;
                        }
                    }
                    acc.x = wave_sum(acc.x, lane);
                    acc.y = wave_sum(acc.y, lane);
                    if (lane == 0) a.cyc[size_t(frame) * NB_FRAME_SYMBOLS + l] = cmul(acc, rot2048);
                }
                continue;
            }
        }
        float2 w = make_float2(1.f, 0.f);
#ifdef DAB_EXP_NCO_BRANCH
        if (dphi != 0u) w = nco(uint32_t(l * NB_SYM_PERIOD + NB_CP + 2 * lane), dphi);
#else
        if constexpr (NCO) w = nco(uint32_t(l * NB_SYM_PERIOD + NB_CP + 2 * lane), dphi);
#endif
        __builtin_amdgcn_sched_barrier(0);
        // ---- loads: 16 x 16 B per lane, row n1 = samples 128*n1 + 2*lane, +1 ----
        float2 x0[16], x1[16];
        if (aligned16) {
            const float4 *rows = reinterpret_cast<const float4 *>(sym + NB_CP) + lane;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
#ifdef DAB_EXP_NT_LD
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(rows + 64 * n1));
                const float4 v = make_float4(t.x, t.y, t.z, t.w);
#else
                const float4 v = rows[64 * n1];
#endif
                x0[n1] = make_float2(v.x, v.y);
                x1[n1] = make_float2(v.z, v.w);
            }
        } else {                                              // odd sample offset: 8-byte loads
            const float2 *rows = sym + NB_CP + 2 * lane;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                x0[n1] = rows[128 * n1];
                x1[n1] = rows[128 * n1 + 1];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- cyclic-prefix correlation on raw samples: CP pair c = lane-4+64*i <-> row 12+i of this lane ----
        if (a.cyc && emit) {
            float2 acc = make_float2(0.f, 0.f);
            const float2 *cp = sym + 2 * (lane - 4);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (i > 0 || lane >= 4) {
                    float4 c;
                    if (aligned16) {
                        c = *reinterpret_cast<const float4 *>(cp + 128 * i);
                    } else {
                        const float2 c0 = cp[128 * i], c1 = cp[128 * i + 1];
                        c = make_float4(c0.x, c0.y, c1.x, c1.y);
                    }
                    const float2 u0 = x0[12 + i], u1 = x1[12 + i];
                    acc.x += c.x * u0.x + c.y * u0.y + c.z * u1.x + c.w * u1.y;      // conj(c) * u
                    acc.y += c.x * u0.y - c.y * u0.x + c.z * u1.y - c.w * u1.x;
                }
            }
            acc.x = wave_sum(acc.x, lane);
            acc.y = wave_sum(acc.y, lane);
            if (lane == 0) a.cyc[size_t(frame) * NB_FRAME_SYMBOLS + l] = cmul(acc, rot2048);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- A2: NCO ----
#ifdef DAB_EXP_NCO_BRANCH
        if (dphi != 0u) {
#else
        if constexpr (NCO) {
#endif
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                x0[n1] = cmul(x0[n1], w);
                x1[n1] = cmul(x1[n1], cmul(w, r1));
                // pin the order row by row: without this the whole phasor chain is computed up front and
                // ~100 extra VGPRs are live next to the 64 data registers (spills)
                asm volatile("" : "+v"(x0[n1].x), "+v"(x0[n1].y), "+v"(x1[n1].x), "+v"(x1[n1].y));
                w = cmul(w, r128);
                asm volatile("" : "+v"(w.x), "+v"(w.y));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- step 1 ----
        fft16(x0);
        __builtin_amdgcn_sched_barrier(0);
        fft16(x1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k1 = 1; k1 < 16; k1++) {
            const float2 t = tw[(8 * n2i) * k1];          // index <= 8*15*15 = 1800
            x0[k1] = cmul(x0[k1], t);
            x1[k1] = cmul(x1[k1], t);
        }
        // ---- exchange 1 + step 2, one pass per column parity e (the buffer holds half a symbol) ----
        const int r1b = pi * 256 + n2i * 16, r1x = (pi << 2) ^ ((n2i >> 1) & 3);
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++) ex[w1_base + k1 * 16 + (w1_r ^ ((k1 >> 1) & 3))] = x0[k1];
#pragma unroll
        for (int m = 0; m < 16; m++) x0[m] = ex[r1b + (m ^ r1x)];
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++) ex[w1_base + k1 * 16 + (w1_r ^ ((k1 >> 1) & 3))] = x1[k1];
#pragma unroll
        for (int m = 0; m < 16; m++) x1[m] = ex[r1b + (m ^ r1x)];
        __builtin_amdgcn_sched_barrier(0);
        fft16(x0);
        __builtin_amdgcn_sched_barrier(0);
        fft16(x1);
        __builtin_amdgcn_sched_barrier(0);
        {
            // W2048^(n3*(k1+16*k2)), k1 = n2 of this lane, n3 = 2p (+1)
            const int i0 = 2 * pi * n2i, st0 = 32 * pi, i1 = i0 + n2i, st1 = st0 + 16;
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) {
                x0[k2] = cmul(x0[k2], tw[i0 + st0 * k2]);
                x1[k2] = cmul(x1[k2], tw[i1 + st1 * k2]);
            }
        }
        // ---- exchange 2 + step 3 in two passes over k2 (k2 < 8, then k2 >= 8) ----
        float2 X[4][8];
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) {
                ex[w2_base + k2 * 16] = x0[8 * h + k2];
                ex[w2_base + k2 * 16 + 128] = x1[8 * h + k2];
            }
            // lane v reads B[n3][k1v][k2 = 8h + 4c' + kk], c' = 0,1
#pragma unroll
            for (int cc = 0; cc < 2; cc++) {
#pragma unroll
                for (int n3 = 0; n3 < 8; n3++)
                    X[2 * h + cc][n3] = ex[n3 * 128 + cc * 64 + r2_base + (k1v ^ ((n3 >> 1) << 2))];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; c++) fft8(X[c]);
        __builtin_amdgcn_sched_barrier(0);
        // bin of X[c][k3] is lane + 64*(c + 4*k3)
        if (FFT_ONLY) {
            float2 *o = a.spectra + (size_t(frame) * NB_FRAME_SYMBOLS + l) * NB_FFT + lane;
#pragma unroll
            for (int k3 = 0; k3 < 8; k3++)
#pragma unroll
                for (int c = 0; c < 4; c++) o[64 * (c + 4 * k3)] = X[c][k3];
            continue;
        }
        // ---- carrier registers: j<12 -> m=j ; j>=12 -> m=j+8 ; lane 0 holds bin 768 (m=12) instead of DC ----
        float2 cur[24];
#pragma unroll
        for (int j = 0; j < 24; j++) {
            const int m = j < 12 ? j : j + 8;
            cur[j] = X[m & 3][m >> 2];
        }
        if (lane == 0) cur[0] = X[0][3];
        // soft-bit selection: bit k of the symbol's 192-bit word = its 16-byte chunk k is wanted.  A symbol nobody
        // wants skips the whole epilogue (its spectrum is still the next symbol's reference).
        bool wanted = true;
        if constexpr (SELECT) {
            if (l > l_first) {
                const unsigned long long *kw = a.keep + 3 * __builtin_amdgcn_readfirstlane(l - 1);
                wanted = (kw[0] | kw[1] | kw[2]) != 0ull;
            }
        }
#ifdef DAB_EXP_NOEPI
        if (l > l_first && cur[3].x == 12345.f) {
#else
        if (l > l_first && wanted) {
#endif
            uint8_t *stg = reinterpret_cast<uint8_t *>(ex);
#pragma unroll
            for (int j = 0; j < 24; j++) {
                const float2 d = cmulc(cur[j], prev[j]);
                // A6: trunc(-127 * c / max(|re|,|im|)).  One v_rcp (1 ulp) instead of two IEEE divisions; the
                // 2^-22 head-room makes the larger component land on exactly +-127 after the clamp, as an exact
                // division gives.  A == 0 (erased carrier) yields 0 because d == 0 and the floor keeps sc finite.
                const float Amax = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), 1.0e-30f);
                const float sc = -127.00003f * __builtin_amdgcn_rcpf(Amax);
                const int br = int(__builtin_amdgcn_fmed3f(d.x * sc, -127.0f, 127.0f));
                const int bi = int(__builtin_amdgcn_fmed3f(d.y * sc, -127.0f, 127.0f));
                const uint32_t nd = sm.nidx[(j >> 1) * 64 + lane];
                const uint32_t ni = (j & 1) ? (nd >> 16) : (nd & 0xFFFFu);
                stg[ni] = uint8_t(br);
                stg[NB_CARRIERS + ni] = uint8_t(bi);
                if constexpr (WITH_DQPSK) {
                    // carrier-order index: bins 1..768 -> 767+bin ; bins 1280..2047 -> bin-1280
                    float2 *dq = a.dqpsk + (size_t(frame) * NB_DATA_SYMBOLS + (l - 1)) * NB_CARRIERS;
                    const int m = j < 12 ? j : j + 8;
                    int bin = lane + 64 * m;
                    if (j == 0 && lane == 0) bin = 768;
                    dq[bin >= 1280 ? bin - 1280 : bin + 767] = d;
                }
            }
            const uint4 *sv = reinterpret_cast<const uint4 *>(stg) + lane;
            uint4 *o = reinterpret_cast<uint4 *>(a.soft + size_t(frame) * NB_FRAME_BITS + size_t(l - 1) * NB_SYM_BITS) + lane;
            const uint4 s0 = sv[0], s1 = sv[64], s2 = sv[128];
#ifdef DAB_EXP_NT_ST
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(v4u{s0.x, s0.y, s0.z, s0.w}, reinterpret_cast<v4u *>(o));
            __builtin_nontemporal_store(v4u{s1.x, s1.y, s1.z, s1.w}, reinterpret_cast<v4u *>(o + 64));
            __builtin_nontemporal_store(v4u{s2.x, s2.y, s2.z, s2.w}, reinterpret_cast<v4u *>(o + 128));
#else
            if constexpr (SELECT) {
                // wave-uniform selection words; one predicated 16-byte store per chunk
                const unsigned long long *kw = a.keep + 3 * __builtin_amdgcn_readfirstlane(l - 1);
                const unsigned long long me = 1ull << lane;
                if (kw[0] & me) o[0] = s0;
                if (kw[1] & me) o[64] = s1;
                if (kw[2] & me) o[128] = s2;
            } else {
                o[0] = s0; o[64] = s1; o[128] = s2;
            }
#endif
        }
#pragma unroll
        for (int j = 0; j < 24; j++) prev[j] = cur[j];
    }
}

}  // namespace

static bool use_v0() {
    static const bool v = std::getenv("DABGPU_OFDM_V0") != nullptr;
    return v;
}

hipError_t launch_ofdm_demod(const OfdmTables &t, const OfdmArgs &a, int parts, hipStream_t s) {
    if (a.n_frames <= 0) return hipSuccess;
    if (parts <= 0 || parts > NB_DATA_SYMBOLS) return hipErrorInvalidValue;
    const int items = a.n_frames * parts;
    if (use_v0() && (a.acq || a.keep)) return hipErrorInvalidValue;   // the first-generation kernel: aligned frames, all bits
    if (use_v0()) {
        hipLaunchKernelGGL(ofdm_kernel<false>, dim3(unsigned(items)), dim3(WG), 0, s, t, a, parts);
    } else {
        const bool nco = a.freq_offset != nullptr || a.acq != nullptr;
        const dim3 grid(unsigned((items + WAVES - 1) / WAVES)), block(64 * WAVES);
        OfdmArgs b = a;
        if (a.dqpsk) b.keep = nullptr;      // the constellation output covers every symbol: a selection is ignored there
        if (b.dqpsk) {
            if (nco) hipLaunchKernelGGL((ofdm_wave_kernel<false, true, false, true>), grid, block, 0, s, t, b, parts, items);
            else hipLaunchKernelGGL((ofdm_wave_kernel<false, true, false, false>), grid, block, 0, s, t, b, parts, items);
        } else if (b.keep) {
            if (nco) hipLaunchKernelGGL((ofdm_wave_kernel<false, false, true, true>), grid, block, 0, s, t, b, parts, items);
            else hipLaunchKernelGGL((ofdm_wave_kernel<false, false, true, false>), grid, block, 0, s, t, b, parts, items);
        } else {
            if (nco) hipLaunchKernelGGL((ofdm_wave_kernel<false, false, false, true>), grid, block, 0, s, t, b, parts, items);
            else hipLaunchKernelGGL((ofdm_wave_kernel<false, false, false, false>), grid, block, 0, s, t, b, parts, items);
        }
    }
    return hipGetLastError();
}

hipError_t launch_fft_symbols(const OfdmTables &t, const OfdmArgs &a, int parts, hipStream_t s) {
    if (a.n_frames <= 0) return hipSuccess;
    if (parts <= 0 || parts > NB_FRAME_SYMBOLS) return hipErrorInvalidValue;
    const int items = a.n_frames * parts;
    if (use_v0()) {
        hipLaunchKernelGGL(ofdm_kernel<true>, dim3(unsigned(items)), dim3(WG), 0, s, t, a, parts);
    } else {
        const dim3 grid(unsigned((items + WAVES - 1) / WAVES)), block(64 * WAVES);
        if (a.freq_offset) hipLaunchKernelGGL((ofdm_wave_kernel<true, false, false, true>), grid, block, 0, s, t, a, parts, items);
        else hipLaunchKernelGGL((ofdm_wave_kernel<true, false, false, false>), grid, block, 0, s, t, a, parts, items);
    }
    return hipGetLastError();
}

}  // namespace dabk
