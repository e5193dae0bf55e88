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
// One 64-lane wavefront owns a run of consecutive symbols of one frame and keeps a whole 2048-point symbol in
// registers (32 complex per lane).
//
//   n = 128*n1 + 8*n2 + n3   (n1<16, n2<16, n3<8)      k = k1 + 16*k2 + 256*k3
//   step 1  lane (n2, n3/2): two 16-point DFTs over n1   (inputs are the lane's own 16-byte loads)
//           twiddle W256^(n2*k1), exchange through LDS
//   step 2  lane (k1, n3/2): two 16-point DFTs over n2,  twiddle W2048^(n3*(k1+16*k2)), exchange
//   step 3  lane v = (k1 | k2%4 << 4): four 8-point DFTs over n3 -> bins v + 64*m, m = 0..31
//
// Because every lane ends each symbol with the SAME bins, the previous spectrum stays in registers and the
// differential demodulation needs no memory at all.  The two LDS exchanges use XOR-swizzled layouts and the
// twiddle tables a lane-permuted layout; all of them are bank-conflict free on the hardware (measured:
// tools/ubench/lds_conflict.hip, profiles/r02_ofdm_bound.md).  No workgroup barrier is needed after the tables are
// loaded: a wave only talks to itself.  Soft bits are scattered as bytes into the (then idle) exchange buffer and
// leave as three coalesced 16-byte stores per lane.
#include <algorithm>

#include "kernels.hpp"
#include "dab_tables.hpp"
#include "fft_common.hpp"
#include "mem_stream.hpp"

namespace dabk {

using namespace dab;

namespace {

constexpr int WAVES = 4;

// Twiddle table layout (float2 entries), 16 KB:
//   [0, TW_E1)       even samples (n3 = 2p, p = 1..3): block k2 has 49 entries -- rows k1>>3 of 24 entries at
//                    (k1&3) + 4(p-1) + 12((k1>>2)&1), then one entry 1.0 that the p = 0 lanes (n3 = 0) read
//   [TW_E1, TW_T1)   odd samples (n3 = 2p+1): block k2 has 64 entries -- rows k1>>3 of 32 entries at
//                    (k1&3) + 4p + 16((k1>>2)&1)
//   [TW_T1, 2048)    step-1 twiddles W256^(n2*k1) as [k1-1][n2]
// With lane = 4*k1 + p each 16-lane group of a ds_read2_b64 touches 16 distinct bank pairs and each 32-lane group
// of a ds_read_b64 32 distinct ones; k2 (step 2) and k1 (step 1) are immediate offsets, so a symbol needs three
// address registers instead of 47 computed addresses.
constexpr int TW_E1 = 16 * 49;             // 784
constexpr int TW_T1 = TW_E1 + 16 * 64;     // 1808
static_assert(TW_T1 + 15 * 16 == NB_FFT, "twiddle table fills 16 KB exactly");

struct WaveLds {
    float2 tw[NB_FFT];               // twiddles, layout above                                  16 KB
    uint32_t nidx[12 * 64];          // frequency de-interleave table, two data indices per dword  3 KB
    float2 ex[WAVES][NB_FFT / 2];    // per-wave exchange buffer (half a symbol per pass), reused
                                     // as soft-bit staging                                    8 KB each
    float2 cyc[WAVES][32];           // cyclic-prefix correlations waiting to leave as one wide store  1 KB
};   // 52 KB per 4-wave workgroup -> 3 workgroups = 12 waves per CU

__device__ __forceinline__ float2 cmul_k(float2 a, float c, float s) {   // a * (c + j*s)
    return make_float2(a.x * c - a.y * s, a.x * s + a.y * c);
}

// (x, y) / |(x, y)| for a vector whose larger component has magnitude 127 (the quantiser's scaling); (0, 0) stays (0, 0)
__device__ __forceinline__ float2 unit_of(float x, float y) {
    const float g = __builtin_amdgcn_rsqf(fmaxf(x * x + y * y, 1.0f));
    return make_float2(x * g, y * g);
}

// a wave-uniform value, kept in a scalar register
__device__ __forceinline__ float uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ float2 uniform(float2 v) { return make_float2(uniform(v.x), uniform(v.y)); }

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

// 8-point forward DFT whose odd-indexed inputs still lack a common factor r: c[k] = r * W8^k (wave-uniform).
// The frequency correction of the odd samples (one more sample period than their even neighbours) rides on the
// last butterfly stage this way instead of costing a phasor product per row of the load.
__device__ __forceinline__ void fft8_odd_scaled(float2 *v, const float2 (&c)[4]) {
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    fft4(e0, e1, e2, e3);
    fft4(o0, o1, o2, o3);
    o0 = cmul(o0, c[0]); o1 = cmul(o1, c[1]); o2 = cmul(o2, c[2]); o3 = cmul(o3, c[3]);
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

// The wave's private LDS exchanges are ordered by the hardware (one wave's DS operations execute in order); these
// keep the compiler from moving the loads of a phase above the stores of the one before.  No instruction is emitted.
__device__ __forceinline__ void lds_stores_done() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); }
__device__ __forceinline__ void lds_loads_may_start() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// frequency correction of frame `frame`, as the NCO's 32-bit phase increment per sample
__device__ __forceinline__ uint32_t frame_dphi(const OfdmArgs &a, int frame) {
    if (a.state) {
        const StreamState st = a.state[frame / a.frames_per_stream];
        return uint32_t(__double2ll_rn(double(st.fine_freq_offset + st.coarse_freq_offset) * 4294967296.0));
    }
    return dphi_of(a.freq_offset, frame);
}

// SELECT: soft-bit selection table in use (a separate instantiation so that the plain kernel keeps its registers)
// NCO: a frequency correction is applied (the launch has a freq_offset array, stream states or acquired frames).  A
// compile-time switch, not a per-frame branch: the branch cost 31 register moves per symbol where its two paths
// re-joined.
template <bool FFT_ONLY, bool WITH_DQPSK, bool SELECT = false, bool NCO = true>
__global__ __launch_bounds__(64 * WAVES, 3) void ofdm_wave_kernel(OfdmTables tab, OfdmArgs a, int parts, int n_items) {
    __shared__ WaveLds sm;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    for (int i = tid; i < NB_FFT; i += 64 * WAVES) {
        int m;                                                  // exponent of exp(-2*pi*i/2048)
        if (i < TW_E1) {
            const int k2 = i / 49, r = i - 49 * k2;
            const int hi = r / 24, s = r - 24 * hi, kq = s / 12, s2 = s - 12 * kq;
            m = r == 48 ? 0 : 2 * ((s2 >> 2) + 1) * (hi * 8 + kq * 4 + (s2 & 3) + 16 * k2);
        } else if (i < TW_T1) {
            const int j = i - TW_E1, k2 = j >> 6, r = j & 63;
            m = (2 * ((r >> 2) & 3) + 1) * ((r >> 5) * 8 + ((r >> 4) & 1) * 4 + (r & 3) + 16 * k2);
        } else {
            const int j = i - TW_T1;
            m = 8 * (j & 15) * ((j >> 4) + 1);
        }
        sm.tw[i] = tab.twiddle[m];                              // m <= 1800
    }
    for (int i = tid; i < 12 * 64; i += 64 * WAVES) sm.nidx[i] = reinterpret_cast<const uint32_t *>(tab.n_of_vj)[i];
    __syncthreads();
    // The item is the same for all lanes of a wave; saying so keeps everything derived from it (frame pointers, the
    // symbol counter, the NCO increment) in SGPRs: -5 % VALU instructions, -8 % time.
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + wave);
    if (item >= n_items) return;
    // whole frames first (one run each), then the cut ones: the launch ends on short items
    int frame = item, part = 0;
    if (item >= a.uncut_frames) {
        const int j = item - a.uncut_frames;
        frame = a.uncut_frames + j / parts;
        part = j - (frame - a.uncut_frames) * parts;
    } else {
        parts = 1;
    }
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
                for (int i = lane; i < n16; i += 64) st_stream(o + i, make_uint4(0u, 0u, 0u, 0u));
            }
            return;
        }
        fiq = a.iq + size_t(frame / a.acq_per_stream) * a.frame_stride + m.start;
        dphi_acq = uint32_t(__double2ll_rn(double(m.freq_offset) * 4294967296.0));
    }
    const bool aligned16 = (reinterpret_cast<uintptr_t>(fiq) & 15u) == 0;
    const uint32_t dphi = a.acq ? dphi_acq : frame_dphi(a, frame);
    float2 *ex = sm.ex[wave];
    const float2 *tw = sm.tw;

    // the previous symbol's carriers (unrolling the symbol loop by two to alternate between two register sets and
    // save the 48 moves per symbol was tried: the allocator then spills ~40 registers)
    float2 prev[24];
#pragma unroll
    for (int j = 0; j < 24; j++) prev[j] = make_float2(0.f, 0.f);

    // NCO phasors.  Wave-uniform ones live in SGPRs: rotation per row of 128 samples and per 2048 samples (cyclic-prefix
    // correlation); the odd samples' extra sample period times W8^k for the last stage.
    const float2 r128 = uniform(nco(128u, dphi)), rot2048 = uniform(nco(uint32_t(NB_FFT), dphi));
    float2 codd[4];
    {
        const float2 r1 = nco(1u, dphi);
        codd[0] = uniform(r1);
        codd[1] = uniform(cmul_k(r1, SQRT1_2, -SQRT1_2));
        codd[2] = uniform(mul_mj(r1));
        codd[3] = uniform(cmul_k(r1, -SQRT1_2, -SQRT1_2));
    }
    // The correlation of symbol l is one 8-byte value: written on its own it is a partial cache line that the
    // memory side has to read-modify-write.  They are collected in LDS and leave 32 at a time (or at the end of the
    // run; the last symbol of a run always has a value: only a run's reference symbol l_first > 0 has none).
    float2 *const qout = a.cyc;
    const bool dd = !FFT_ONLY && a.cyc == nullptr && a.dd4 != nullptr;
    float2 ddacc = make_float2(0.f, 0.f);                       // decision-directed frequency-error sum of the run (wave-uniform: scalar registers)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);    // (scalar: the row's address costs no VGPR across the loop)
    auto put_cyc = [&](const int l, const float2 c) {
        const int slot = (l - l_first) & 31;                    // wave-uniform
        if (lane == 0) sm.cyc[wave_u][slot] = c;
        if (slot == 31 || l == l_last) {
            lds_stores_done();
            lds_loads_may_start();
            int lq = lane;                                      // opaque copy: the row address is formed here, every 32nd
            asm volatile("" : "+v"(lq));                        // symbol, instead of living in a VGPR across the loop
            const int li = l - slot + lq;                       // symbol whose value lane `lane` carries out
            if (lq <= slot && (FFT_ONLY || li > l_first || li == 0))
                st_stream(qout + size_t(frame) * NB_FRAME_SYMBOLS + li, sm.cyc[wave_u][lq]);
        }
    };

    for (int l = l_first; l <= l_last; l++) {
        const float2 *sym = fiq + size_t(l) * NB_SYM_PERIOD;
        const bool emit = FFT_ONLY || (l > l_first) || (l == 0);
        // Lane roles, derived per iteration from an opaque copy of the lane number: every LDS address below is one or
        // two VALU ops from these, which is cheaper than letting LICM park ~60 loop-invariant addresses in VGPRs (spills).
        int li = lane;
        asm volatile("" : "+v"(li));
        const int n2i = li >> 2, pi = li & 3;              // step 1 (and k1 = n2, p' = p in step 2)
        const int k1v = li & 15, kk = li >> 4;             // step 3
        // exchange-1 slots (one pass per e = n3 & 1): (n3/2)*256 + k1*16 + (n2 ^ (n3/2)<<2 ^ (k1/2)&3)
        const int w1_base = pi * 256, w1_r = n2i ^ (pi << 2);
        // exchange-2 slots (one pass per half of k2): n3*128 + (k2 % 8)*16 + (k1 ^ (n3/2)<<2)
        const int w2_base = pi * 256 + (n2i ^ (pi << 2));
        const int r2_base = kk * 16;
        if constexpr (SELECT) {
            // A symbol is transformed only if some of its own soft bits are wanted or it is the differential
            // reference of a wanted one.  The others give at most their cyclic-prefix correlation, which needs the
            // prefix and the last 512 samples only (8 of the 20 loads).
            const int ls = __builtin_amdgcn_readfirstlane(l);
            bool need = false;
            if (ls > l_first) { const unsigned long long *kw = a.keep + 3 * (ls - 1); need |= (kw[0] | kw[1] | kw[2]) != 0ull; }
            if (ls < l_last) { const unsigned long long *kw = a.keep + 3 * ls; need |= (kw[0] | kw[1] | kw[2]) != 0ull; }
            if (!need) {
                if ((a.cyc && emit) || (dd && l == 0)) {
                    float2 acc = make_float2(0.f, 0.f);
                    const float2 *cp = sym + 2 * (lane - 4);
                    const float2 *tail = sym + NB_CP + 128 * 12 + 2 * lane;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (i > 0 || lane >= 4) {
                            const float2 c0 = ld_stream(cp + 128 * i), c1 = ld_stream(cp + 128 * i + 1);
                            const float2 u0 = ld_stream(tail + 128 * i), u1 = ld_stream(tail + 128 * i + 1);
                            acc.x += c0.x * u0.x + c0.y * u0.y + c1.x * u1.x + c1.y * u1.y;      // conj(c) * u
                            acc.y += c0.x * u0.y - c0.y * u0.x + c1.x * u1.y - c1.y * u1.x;
                        }
                    }
                    acc.x = wave_sum(acc.x, lane);
                    acc.y = wave_sum(acc.y, lane);
                    if (dd) { if (lane == 0) st_stream(a.dd4 + size_t(frame) * NB_FRAME_SYMBOLS, cmul(acc, rot2048)); }
                    else put_cyc(l, cmul(acc, rot2048));
                }
                continue;
            }
        }
        // phasor of this lane's first sample, from the exact 32-bit phase (a recurrence from the previous symbol would
        // save ~45 instructions but make a frame's soft bits depend on how the batch was cut into runs); computed
        // before the loads so sincospi's temporaries are dead by the time 64 data registers are live
        float2 w = make_float2(1.f, 0.f);
        if constexpr (NCO) w = nco(uint32_t(l * NB_SYM_PERIOD + NB_CP + 2 * lane), dphi);
        __builtin_amdgcn_sched_barrier(0);
        // ---- loads: 16 x 16 B per lane, row n1 = samples 128*n1 + 2*lane, +1 ----
        float2 x0[16], x1[16];
        if (aligned16) {
            const float4 *rows = reinterpret_cast<const float4 *>(sym + NB_CP) + lane;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const float4 v = ld_stream(rows + 64 * n1);
                x0[n1] = make_float2(v.x, v.y);
                x1[n1] = make_float2(v.z, v.w);
            }
        } else {                                              // odd sample offset: 8-byte loads
            const float2 *rows = sym + NB_CP + 2 * lane;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                x0[n1] = ld_stream(rows + 128 * n1);
                x1[n1] = ld_stream(rows + 128 * n1 + 1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- cyclic-prefix correlation on raw samples: CP pair c = lane-4+64*i <-> row 12+i of this lane ----
        // (decision-directed mode: of the PRS only -- its angle picks the branch of the fourth-power estimate, 4.1)
        if ((a.cyc && emit) || (dd && l == 0)) {
            float2 acc = make_float2(0.f, 0.f);
            const float2 *cp = sym + 2 * (lane - 4);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (i > 0 || lane >= 4) {
                    float4 c;
                    if (aligned16) {
                        c = ld_stream(reinterpret_cast<const float4 *>(cp + 128 * i));
                    } else {
                        const float2 c0 = ld_stream(cp + 128 * i), c1 = ld_stream(cp + 128 * i + 1);
                        c = make_float4(c0.x, c0.y, c1.x, c1.y);
                    }
                    const float2 u0 = x0[12 + i], u1 = x1[12 + i];
                    acc.x += c.x * u0.x + c.y * u0.y + c.z * u1.x + c.w * u1.y;      // conj(c) * u
                    acc.y += c.x * u0.y - c.y * u0.x + c.z * u1.y - c.w * u1.x;
                }
            }
            acc.x = wave_sum(acc.x, lane);
            acc.y = wave_sum(acc.y, lane);
            if (dd) { if (lane == 0) st_stream(a.dd4 + size_t(frame) * NB_FRAME_SYMBOLS, cmul(acc, rot2048)); }
            else put_cyc(l, cmul(acc, rot2048));
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- A2: NCO.  Even and odd sample of a load get the same phasor here; the odd one's missing sample
        // period is applied in step 3 (fft8_odd_scaled) ----
        if constexpr (NCO) {
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                x0[n1] = cmul(x0[n1], w);
                x1[n1] = cmul(x1[n1], w);
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
        {
            const float2 *t1 = tw + TW_T1 + n2i;
#pragma unroll
            for (int k1 = 1; k1 < 16; k1++) {
                const float2 t = t1[(k1 - 1) * 16];
                x0[k1] = cmul(x0[k1], t);
                x1[k1] = cmul(x1[k1], t);
            }
        }
        // ---- exchange 1 + step 2, one pass per column parity e (the buffer holds half a symbol) ----
        const int r1b = pi * 256 + n2i * 16, r1x = (pi << 2) ^ ((n2i >> 1) & 3);
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++) ex[w1_base + k1 * 16 + (w1_r ^ ((k1 >> 1) & 3))] = x0[k1];
        lds_stores_done();
        lds_loads_may_start();
#pragma unroll
        for (int m = 0; m < 16; m++) x0[m] = ex[r1b + (m ^ r1x)];
        lds_stores_done();
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++) ex[w1_base + k1 * 16 + (w1_r ^ ((k1 >> 1) & 3))] = x1[k1];
        lds_stores_done();
        lds_loads_may_start();
#pragma unroll
        for (int m = 0; m < 16; m++) x1[m] = ex[r1b + (m ^ r1x)];
        __builtin_amdgcn_sched_barrier(0);
        fft16(x0);
        __builtin_amdgcn_sched_barrier(0);
        fft16(x1);
        __builtin_amdgcn_sched_barrier(0);
        {
            // W2048^(n3*(k1+16*k2)), k1 = n2 of this lane, n3 = 2p (+1): see the table layout at WaveLds
            const int hi = n2i >> 3, kq = (n2i >> 2) & 1, klo = n2i & 3;
            const float2 *t0 = tw + (pi ? hi * 24 + kq * 12 + (pi - 1) * 4 + klo : 48);
            const float2 *t1 = tw + TW_E1 + hi * 32 + kq * 16 + pi * 4 + klo;
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) {
                x0[k2] = cmul(x0[k2], t0[49 * k2]);
                x1[k2] = cmul(x1[k2], t1[64 * k2]);
            }
        }
        // ---- exchange 2 + step 3 in two passes over k2 (k2 < 8, then k2 >= 8) ----
        float2 X[4][8];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            lds_stores_done();                                 // (the reads of the pass before are program-ordered loads)
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) {
                ex[w2_base + k2 * 16] = x0[8 * h + k2];
                ex[w2_base + k2 * 16 + 128] = x1[8 * h + k2];
            }
            lds_stores_done();
            lds_loads_may_start();
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
        for (int c = 0; c < 4; c++) {
            if constexpr (NCO) fft8_odd_scaled(X[c], codd);
            else fft8(X[c]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // bin of X[c][k3] is lane + 64*(c + 4*k3)
        if (FFT_ONLY) {
            float2 *o = a.spectra + (size_t(frame) * NB_FRAME_SYMBOLS + l) * NB_FFT + lane;
#pragma unroll
            for (int k3 = 0; k3 < 8; k3++)
#pragma unroll
                for (int c = 0; c < 4; c++) st_stream(o + 64 * (c + 4 * k3), X[c][k3]);
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
        // ---- decision-directed frequency error: the fourth power of a differential symbol is -|d|^4 exp(j 4 theta)
        // whatever its two bits are, theta = 2 pi (residual offset) 2552.  Each symbol enters with unit magnitude (its
        // direction, taken from the soft-bit quantiser's scaling): every carrier weighs the same, the estimator's gain
        // is 1, and no input level can overflow the sum.  With a soft-bit selection only the symbols whose bits are wanted
        // contribute (their differential reference is then a transformed symbol by construction).  Four of the lane's
        // 24 carriers are used: the 256 nearest the centre (bins lane, lane + 64, lane + 1920, lane + 1984 = carriers -128..127), because a sample
        // clock that is off by e rotates carrier k by 2 pi k e 2552/2048 per symbol on top of theta -- times four, 150 ppm
        // would turn the outer carriers' terms around (cos(4 x 0.9) < 0) while these lose 4 % and, being symmetric
        // about the centre, stay unbiased.  Reduced over the wave per symbol and summed over the run in scalar registers
        // (two more live VGPRs would spill): what the loop needs, without a single cyclic-prefix sample being read.
        // soft-bit selection: bit k of the symbol's 192-bit word = its 16-byte chunk k is wanted.  A symbol nobody
        // wants skips the whole epilogue (its spectrum is still the next symbol's reference).
        bool wanted = true;
        if constexpr (SELECT) {
            if (l > l_first) {
                const unsigned long long *kw = a.keep + 3 * __builtin_amdgcn_readfirstlane(l - 1);
                wanted = (kw[0] | kw[1] | kw[2]) != 0ull;
            }
        }
        if (l > l_first && wanted) {
            uint8_t *stg = reinterpret_cast<uint8_t *>(ex);
            lds_stores_done();
            float2 t = make_float2(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 24; j++) {
                const float2 d = cmulc(cur[j], prev[j]);
                // A6: trunc(-127 * c / max(|re|,|im|)).  One v_rcp (1 ulp) instead of two IEEE divisions; the 2^-22
                // head-room keeps the larger component's product in [127, 127.0001], which the truncating conversion
                // turns into exactly +-127 as an exact division gives (no clamp needed: nothing exceeds 127.0001).
                // A == 0 (erased carrier) yields 0 because d == 0 and the floor keeps sc finite.
                const float Amax = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), 1.0e-30f);
                const float sc = -127.00003f * __builtin_amdgcn_rcpf(Amax);
                const float fx = d.x * sc, fy = d.y * sc;
                const int br = int(fx), bi = int(fy);
                if ((j < 2 || j >= 22) && dd) {
                    // the decision-directed sum takes the differential symbol's direction only (from the quantiser's
                    // scaling, larger component = 127: nothing can overflow or vanish whatever the level of the input)
                    const float2 u = unit_of(fx, fy);
                    const float2 z = make_float2(u.x * u.x - u.y * u.y, 2.0f * u.x * u.y);
                    t.x += z.x * z.x - z.y * z.y;
                    t.y += 2.0f * z.x * z.y;
                }
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
                    st_stream(dq + (bin >= 1280 ? bin - 1280 : bin + 767), d);
                }
            }
            if (dd) ddacc = uniform(make_float2(ddacc.x + wave_sum(t.x, lane), ddacc.y + wave_sum(t.y, lane)));
            lds_stores_done();
            lds_loads_may_start();
            const uint4 *sv = reinterpret_cast<const uint4 *>(stg) + lane;
            uint4 *o = reinterpret_cast<uint4 *>(a.soft + size_t(frame) * NB_FRAME_BITS + size_t(l - 1) * NB_SYM_BITS) + lane;
            const uint4 s0 = sv[0], s1 = sv[64], s2 = sv[128];
            if constexpr (SELECT) {
                // wave-uniform selection words; one predicated 16-byte store per chunk
                const unsigned long long *kw = a.keep + 3 * __builtin_amdgcn_readfirstlane(l - 1);
                const unsigned long long me = 1ull << lane;
                if (kw[0] & me) st_stream(o, s0);
                if (kw[1] & me) st_stream(o + 64, s1);
                if (kw[2] & me) st_stream(o + 128, s2);
            } else {
                st_stream(o, s0); st_stream(o + 64, s1); st_stream(o + 128, s2);
            }
        }
#pragma unroll
        for (int j = 0; j < 24; j++) prev[j] = cur[j];
    }
    if (dd) {
        // the run's sum goes to the entry of its last symbol, zeros to its other entries (one narrow store per lane)
        const float sx = ddacc.x, sy = ddacc.y;
        const int nrun = l_last - l_first;
        for (int i = lane; i < nrun; i += 64)
            st_stream(a.dd4 + size_t(frame) * NB_FRAME_SYMBOLS + l_first + 1 + i,
                      i == nrun - 1 ? make_float2(sx, sy) : make_float2(0.f, 0.f));
    }
}

// One 1024-thread workgroup per stream.  Restated by oracle.py stream_update() for the parity test.
constexpr int SU_THREADS = 1024;
__global__ __launch_bounds__(SU_THREADS) void stream_update_kernel(StreamState *state, const float2 *cyc, const float2 *iq,
                                                                   size_t frame_stride, int frames_per_stream, float beta,
                                                                   float thr_null_start, float signal_beta, int dd, float dd_gate,
                                                                   int dd_terms_per_frame) {
    __shared__ float red[2][SU_THREADS / 64];
    __shared__ double red_dd[2][SU_THREADS / 64];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float2 *c = cyc + size_t(s) * frames_per_stream * NB_FRAME_SYMBOLS;
    const int n = frames_per_stream * NB_FRAME_SYMBOLS;
    float acc = 0.f;
    double sx = 0.0, sy = 0.0;
    // level of the stream's most recent frame: first 4096 samples (PRS and the start of the first data symbol); read first
    const float4 *x = reinterpret_cast<const float4 *>(iq + (size_t(s) * frames_per_stream + (frames_per_stream - 1)) * frame_stride);
    float4 lv[2048 / SU_THREADS];
#pragma unroll
    for (int k = 0; k < 2048 / SU_THREADS; k++) lv[k] = x[tid + k * SU_THREADS];
    // (eight entries in flight per thread; each thread still adds its entries in ascending order)
    constexpr int SU_BATCH = 8;
    if (dd) {
        for (int i0 = tid; i0 < n; i0 += SU_THREADS * SU_BATCH) {
            float2 v[SU_BATCH];
#pragma unroll
            for (int u = 0; u < SU_BATCH; u++) {
                const int i = i0 + u * SU_THREADS;
                v[u] = i < n ? c[i] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < SU_BATCH; u++) {
                const int i = i0 + u * SU_THREADS;
                if (i >= n) break;
                if (i % NB_FRAME_SYMBOLS) { sx += double(v[u].x); sy += double(v[u].y); }
                else acc += atan2f(v[u].y, v[u].x);               // entry 0 of a frame: the PRS's cyclic-prefix correlation
            }
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { sx += __shfl_xor(sx, off); sy += __shfl_xor(sy, off); }
        if (lane == 0) { red_dd[0][wave] = sx; red_dd[1][wave] = sy; }
    } else {
        for (int i0 = tid; i0 < n; i0 += SU_THREADS * SU_BATCH) {
            float2 v[SU_BATCH];
#pragma unroll
            for (int u = 0; u < SU_BATCH; u++) {
                const int i = i0 + u * SU_THREADS;
                v[u] = i < n ? c[i] : make_float2(1.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < SU_BATCH; u++)
                if (i0 + u * SU_THREADS < n) acc += atan2f(v[u].y, v[u].x);
        }
    }
    float l1 = 0.f;
#pragma unroll
    for (int k = 0; k < 2048 / SU_THREADS; k++) l1 += fabsf(lv[k].x) + fabsf(lv[k].y) + fabsf(lv[k].z) + fabsf(lv[k].w);
    acc = wave_sum(acc, lane);
    l1 = wave_sum(l1, lane);
    if (lane == 0) { red[0][wave] = acc; red[1][wave] = l1; }
    __syncthreads();
    if (tid == 0) {
        acc = 0.f; l1 = 0.f;
        for (int w = 0; w < SU_THREADS / 64; w++) { acc += red[0][w]; l1 += red[1][w]; }
        l1 *= 1.0f / 4096.0f;
        StreamState st = state[s];
        float err = acc / float(n) * (1.0f / (6.283185307179586f * float(NB_FFT)));
        const bool level_lost = st.signal_average > 0.f && l1 < thr_null_start * st.signal_average;
        // (a call of ONE frame whose level is gone -- a null symbol, a dropout -- has nothing to steer the loop with)
        const bool steers = !(level_lost && frames_per_stream == 1);
        if (dd) {
            sx = 0.0; sy = 0.0;
            for (int w = 0; w < SU_THREADS / 64; w++) { sx += red_dd[0][w]; sy += red_dd[1][w]; }
            // sum = -A exp(j 4 theta), theta = 2 pi r 2552: r modulo 1 / (4 2552); the branch from the PRS prefixes; gated.
            // An estimate that is not applied must not move the gate's memory either (branch, pending, gated count)
            StreamState scratch = st;
            err = dd_loop_error(sx, sy, acc / float(frames_per_stream) * (1.0f / (6.283185307179586f * float(NB_FFT))),
                                double(frames_per_stream) * double(dd_terms_per_frame), dd_gate, st.total_frames_read == 0,
                                steers ? st : scratch);
        }
        if (steers) {
            constexpr float HALF = 0.5f / float(NB_FFT);
            float f = st.fine_freq_offset - beta * err;
            if (f > HALF) f -= 2.f * HALF;
            if (f < -HALF) f += 2.f * HALF;
            st.fine_freq_offset = f;
        }
        st.last_fine_error = err;
        if (level_lost) {
            st.total_frames_desync += 1;
            st.total_frames_read += frames_per_stream - 1;
        } else {
            st.total_frames_read += frames_per_stream;
            st.signal_average = st.signal_average > 0.f ? signal_beta * st.signal_average + (1.0f - signal_beta) * l1 : l1;
        }
        state[s] = st;
    }
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void fill_noise_kernel(uint32_t *p, size_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        uint32_t x = uint32_t(i) * 2654435761u + uint32_t(i >> 32) * 40503u + 12345u;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (x & 0x807fffffu) | 0x3f000000u;
    }
}
}  // namespace

namespace {
// the front end's access shape: a wave owns consecutive chunks, reads 20 KB (twenty 16-byte loads per lane) and writes
// 3 KB per chunk, 12 waves per CU
__global__ __launch_bounds__(256) void placement_probe_kernel(const char *in, char *out, int n_chunks, int chunks_per_wave) {
    __shared__ char occupancy[52 * 1024];                       // three workgroups per CU, as the front end
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (n_chunks < 0) occupancy[threadIdx.x] = 1;               // (never: keeps the array)
    uint4 acc = make_uint4(1u, 2u, 3u, 4u);
    for (int c = 0; c < chunks_per_wave; c++) {
        const int chunk = wave * chunks_per_wave + c;
        if (chunk >= n_chunks) break;
        const uint4 *p = reinterpret_cast<const uint4 *>(in + size_t(chunk) * 20480) + lane;
        uint4 v[20];
#pragma unroll
        for (int i = 0; i < 20; i++) v[i] = ld_stream(p + 64 * i);
#pragma unroll
        for (int i = 0; i < 20; i++) { acc.x += v[i].x; acc.y ^= v[i].y; acc.z += v[i].z; acc.w ^= v[i].w; }
        uint4 *o = reinterpret_cast<uint4 *>(out + size_t(chunk) * 3072) + lane;
#pragma unroll
        for (int i = 0; i < 3; i++) st_stream(o + 64 * i, acc);
    }
}
}  // namespace

namespace {
// The fused front end's memory geometry without its arithmetic (dabgpu_mover_frames_dev): the same items (whole frames
// first, then frames cut into `parts` runs), one wavefront per item, per symbol of the run sixteen 16-byte loads per lane
// from the useful part (PREFIXES: four more covering the cyclic prefix; else the prefix of symbol 0 only), per data symbol
// three 16-byte stores per lane; streaming accesses, the same 52 KB of LDS per 4-wave workgroup.
template <bool PREFIXES>
__global__ __launch_bounds__(64 * WAVES, 3) void geometry_mover_kernel(const float2 *iq, size_t frame_stride, int8_t *soft,
                                                                        int uncut_frames, int parts, int n_items) {
    __shared__ char occupancy[sizeof(WaveLds)];
    const int lane = threadIdx.x & 63;
    if (n_items < 0) occupancy[threadIdx.x] = 1;               // (never: keeps the array)
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + (threadIdx.x >> 6));
    if (item >= n_items) return;
    int frame = item, part = 0;
    if (item >= uncut_frames) {
        const int j = item - uncut_frames;
        frame = uncut_frames + j / parts;
        part = j - (frame - uncut_frames) * parts;
    } else {
        parts = 1;
    }
    const float2 *fiq = iq + size_t(frame) * frame_stride;
    const int l_first = (NB_DATA_SYMBOLS * part) / parts, l_last = (NB_DATA_SYMBOLS * (part + 1)) / parts;
    uint4 acc = make_uint4(1u, 2u, 3u, 4u);
    for (int l = l_first; l <= l_last; l++) {
        const float2 *sym = fiq + size_t(l) * NB_SYM_PERIOD;
        const uint4 *rows = reinterpret_cast<const uint4 *>(sym + NB_CP) + lane;
        uint4 v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = ld_stream(rows + 64 * i);
        if (PREFIXES ? (l > l_first || l == 0) : l == 0) {
            const uint4 *cp = reinterpret_cast<const uint4 *>(sym + 2 * (lane - 4));
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (i > 0 || lane >= 4) { const uint4 c = ld_stream(cp + 64 * i); acc.x += c.x; acc.y ^= c.y; acc.z += c.z; acc.w ^= c.w; }
        }
#pragma unroll
        for (int i = 0; i < 16; i++) { acc.x += v[i].x; acc.y ^= v[i].y; acc.z += v[i].z; acc.w ^= v[i].w; }
        if (l > l_first) {
            uint4 *o = reinterpret_cast<uint4 *>(soft + size_t(frame) * NB_FRAME_BITS + size_t(l - 1) * NB_SYM_BITS) + lane;
            st_stream(o, acc); st_stream(o + 64, acc); st_stream(o + 128, acc);
        }
    }
}
}  // namespace

hipError_t launch_geometry_mover(const float2 *iq, size_t frame_stride, int n_frames, int8_t *soft, int uncut_frames, int parts,
                                 bool prefixes, hipStream_t s) {
    if (n_frames <= 0) return hipSuccess;
    if (parts <= 0 || parts > NB_DATA_SYMBOLS || uncut_frames < 0 || uncut_frames > n_frames) return hipErrorInvalidValue;
    const int items = uncut_frames + (n_frames - uncut_frames) * parts;
    const dim3 grid(unsigned((items + WAVES - 1) / WAVES)), block(64 * WAVES);
    if (prefixes) hipLaunchKernelGGL(geometry_mover_kernel<true>, grid, block, 0, s, iq, frame_stride, soft, uncut_frames, parts, items);
    else hipLaunchKernelGGL(geometry_mover_kernel<false>, grid, block, 0, s, iq, frame_stride, soft, uncut_frames, parts, items);
    return hipGetLastError();
}

hipError_t launch_placement_probe(const void *in, size_t in_bytes, void *out, size_t out_bytes, hipStream_t s) {
    const size_t n_chunks = std::min<size_t>(std::min(in_bytes / 20480, out_bytes / 3072), size_t(1) << 20);
    if (n_chunks == 0) return hipSuccess;
    const int cpw = 8;
    const unsigned grid = unsigned(((n_chunks + cpw - 1) / cpw + 3) / 4);
    hipLaunchKernelGGL(placement_probe_kernel, dim3(grid), dim3(256), 0, s, static_cast<const char *>(in), static_cast<char *>(out),
                       int(n_chunks), cpw);
    return hipGetLastError();
}

namespace {
constexpr int COPY_PIECES_MAX = 24;
struct CopyPieces16 {
    uint4 *dst[COPY_PIECES_MAX];
    const uint4 *src[COPY_PIECES_MAX];
    unsigned n[COPY_PIECES_MAX];                               // in 16-byte words
    int count;
};
__global__ __launch_bounds__(256) void copy_pieces_kernel(CopyPieces16 p) {
    unsigned i = blockIdx.x * 256u + threadIdx.x;
    for (int k = 0; k < p.count; k++) {
        if (i < p.n[k]) { p.dst[k][i] = p.src[k][i]; return; }
        i -= p.n[k];
    }
}
}  // namespace

int copy_pieces_max() { return COPY_PIECES_MAX; }

namespace {
__global__ void signal_kernel(unsigned long long *flag, unsigned long long value) {
    __threadfence_system();
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
hipError_t launch_signal(unsigned long long *flag, unsigned long long value, hipStream_t s) {
    hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, s, flag, value);
    return hipGetLastError();
}

hipError_t launch_copy_pieces(const CopyPiece *pieces, int n, hipStream_t s) {
    if (n < 0 || n > COPY_PIECES_MAX) return hipErrorInvalidValue;
    CopyPieces16 p{};
    unsigned total = 0;
    for (int k = 0; k < n; k++) {
        const CopyPiece &c = pieces[k];
        if (c.bytes == 0) continue;
        if ((c.bytes & 15) || ((reinterpret_cast<uintptr_t>(c.dst) | reinterpret_cast<uintptr_t>(c.src)) & 15) || (c.bytes >> 4) > 0x7fffffffu)
            return hipErrorInvalidValue;
        p.dst[p.count] = static_cast<uint4 *>(c.dst);
        p.src[p.count] = static_cast<const uint4 *>(c.src);
        p.n[p.count] = unsigned(c.bytes >> 4);
        total += p.n[p.count];
        p.count++;
    }
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(copy_pieces_kernel, dim3((total + 255) / 256), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_fill_noise(void *p, size_t bytes, hipStream_t s) {
    if (bytes < 4) return hipSuccess;
    hipLaunchKernelGGL(fill_noise_kernel, dim3(4096), dim3(256), 0, s, static_cast<uint32_t *>(p), bytes / 4);
    return hipGetLastError();
}

hipError_t launch_stream_update(StreamState *state, const float2 *cyc, const float2 *iq, size_t frame_stride,
                                int n_streams, int frames_per_stream, float beta, float thr_null_start, float signal_beta,
                                int dd, float dd_gate, int dd_terms_per_frame, hipStream_t s) {
    if (n_streams <= 0 || frames_per_stream <= 0) return hipSuccess;
    hipLaunchKernelGGL(stream_update_kernel, dim3(unsigned(n_streams)), dim3(SU_THREADS), 0, s, state, cyc, iq, frame_stride,
                       frames_per_stream, beta, thr_null_start, signal_beta, dd, dd_gate, dd_terms_per_frame);
    return hipGetLastError();
}

hipError_t launch_ofdm_demod(const OfdmTables &t, const OfdmArgs &a, int parts, hipStream_t s) {
    if (a.n_frames <= 0) return hipSuccess;
    if (parts <= 0 || parts > NB_DATA_SYMBOLS) return hipErrorInvalidValue;
    if (a.state && a.frames_per_stream <= 0) return hipErrorInvalidValue;
    if (a.uncut_frames < 0 || a.uncut_frames > a.n_frames) return hipErrorInvalidValue;
    const int items = a.uncut_frames + (a.n_frames - a.uncut_frames) * parts;
    const bool nco = a.freq_offset != nullptr || a.acq != nullptr || a.state != nullptr;
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
    return hipGetLastError();
}

hipError_t launch_fft_symbols(const OfdmTables &t, const OfdmArgs &a, int parts, hipStream_t s) {
    if (a.n_frames <= 0) return hipSuccess;
    if (parts <= 0 || parts > NB_FRAME_SYMBOLS) return hipErrorInvalidValue;
    if (a.state && a.frames_per_stream <= 0) return hipErrorInvalidValue;
    if (a.uncut_frames < 0 || a.uncut_frames > a.n_frames) return hipErrorInvalidValue;
    const int items = a.uncut_frames + (a.n_frames - a.uncut_frames) * parts;
    const dim3 grid(unsigned((items + WAVES - 1) / WAVES)), block(64 * WAVES);
    if (a.freq_offset || a.state) hipLaunchKernelGGL((ofdm_wave_kernel<true, false, false, true>), grid, block, 0, s, t, a, parts, items);
    else hipLaunchKernelGGL((ofdm_wave_kernel<true, false, false, false>), grid, block, 0, s, t, a, parts, items);
    return hipGetLastError();
}

}  // namespace dabk
