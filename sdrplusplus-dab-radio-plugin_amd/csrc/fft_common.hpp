// fft_common.hpp -- device helpers shared by the OFDM and synchronisation kernels: complex arithmetic on
// float2, radix-4/8 butterflies, the NCO phasor, and a 256-thread 2048-point Stockham FFT through LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "dab_tables.hpp"

namespace dabk {
namespace {

using namespace dab;

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

// Where entry L of an intermediate buffer lives.  A float2 takes two of the 64 LDS banks, so the 32 lanes of one pass of a
// ds_write_b64 must hit 32 different slots modulo 32.  The first stage writes L = 8 tid + r (lanes 64 bytes apart: sixteen
// lanes per slot pair as it stood), the second L = 64 (j >> 3) + (j & 7) + 8 r (four groups on the same slots): an XOR with
// bits that are CONSTANT over the 32 / 64 consecutive entries a read takes keeps the reads conflict-free and spreads the writes.
__device__ __forceinline__ int swz1(int L) { return L ^ ((L >> 5) & 7); }
__device__ __forceinline__ int swz2(int L) { return L ^ (((L >> 6) & 3) << 3); }

// The twiddles of the two radix-8 passes, behind the natural table exp(-2 pi i m / 2048), m < 2048, in the order the lanes read
// them: pass NS = 8 takes W^(32 r k), k = j & 7 (eight lanes a bank pair apart as it stood: 256-byte stride), pass NS = 64
// W^(4 r k), k = j & 63 (32-byte stride) -- here [r][k], consecutive lanes on consecutive entries.
constexpr int TWC8_OFF = NB_FFT, TWC64_OFF = NB_FFT + 64, TW_TOTAL = NB_FFT + 64 + 512;

// One Stockham pass with radix 8 from LDS to LDS (SRC / DST: 0 natural, 1 swz1, 2 swz2); twc: this pass's compact twiddles [8][NS].
template <int NS, int SRC, int DST>
__device__ __forceinline__ void pass8(const float2 *src, float2 *dst, const float2 *twc, int j) {
    float2 v[8];
    const int k = j & (NS - 1);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int L = j + r * (NB_FFT / 8);
        v[r] = src[SRC == 1 ? swz1(L) : SRC == 2 ? swz2(L) : L];
    }
#pragma unroll
    for (int r = 1; r < 8; r++) v[r] = cmul(v[r], twc[r * NS + k]);
    fft8(v);
    const int j0 = (j - k) * 8 + k;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int L = j0 + r * NS;
        dst[DST == 1 ? swz1(L) : DST == 2 ? swz2(L) : L] = v[r];
    }
}


// 2048-point forward FFT by one 256-thread workgroup.  Each thread passes its eight inputs x[tid + 256*r]
// in v[]; the result lands in `out` (natural order).  t1 and out are 2048-entry LDS buffers (distinct), tw the
// exp(-2*pi*i*m/2048) table (LDS or global), twc8 / twc64 the compact tables of the two radix-8 passes.  Ends with a barrier.
__device__ __forceinline__ void block_fft2048(float2 (&v)[8], float2 *t1, float2 *out, const float2 *tw, const float2 *twc8,
                                              const float2 *twc64, int tid) {
    fft8(v);
#pragma unroll
    for (int r = 0; r < 8; r++) t1[swz1(tid * 8 + r)] = v[r];
    __syncthreads();
    pass8<8, 1, 2>(t1, out, twc8, tid);
    __syncthreads();
    pass8<64, 2, 0>(out, t1, twc64, tid);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int j = tid + h * WG;
        float2 v0 = t1[j], v1 = t1[j + 512], v2 = t1[j + 1024], v3 = t1[j + 1536];
        v1 = cmul(v1, tw[j]);
        v2 = cmul(v2, tw[2 * j]);
        v3 = cmul(v3, tw[3 * j]);
        fft4(v0, v1, v2, v3);
        out[j] = v0; out[j + 512] = v1; out[j + 1024] = v2; out[j + 1536] = v3;
    }
    __syncthreads();
}

}  // namespace
}  // namespace dabk
