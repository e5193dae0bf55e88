// viterbi_kernels.hip -- DAB channel decoder on gfx950 (CDNA4): rows A7..A12 of
// SURVEY.md section 8a, i.e. what BasicRadio::Process does with one frame of soft bits
// (/root/reference/src/radio_block.cpp:42): depuncture, K=7 rate-1/4 soft Viterbi
// (the `viterbi` package, /root/reference/CMakeLists.txt:53-54), energy dispersal,
// FIB CRC16, and for MSC the 16-CIF time de-interleaver fused into the fetch.
//
// Kernel 1 (this file, "wave" variant): one 64-lane wavefront per codeword, lane =
// trellis state.  Path metrics live in a VGPR per lane, predecessor metrics arrive
// by cross-lane permute, the 64 survivor decisions of a step are one __ballot word
// kept in LDS together with the depunctured codeword; traceback runs from LDS.
// Decisions follow the exact integer correlation metric,
// strict-greater tie-break, start state 0, end state 0).
#include <algorithm>

#include "kernels.hpp"
#include "dab_tables.hpp"

namespace dabk {

using namespace dab;

namespace {

constexpr int WAVES_PER_WG = 4;
constexpr int WGV = 64 * WAVES_PER_WG;

__device__ __forceinline__ int parity32(unsigned x) { return __popc(x) & 1; }

// ---- where a codeword's punctured soft bits come from -----------------------
struct FetchFic {
    const int8_t *soft;
    size_t stride;
    __device__ __forceinline__ int8_t operator()(int cw, int i) const {
        return soft[size_t(cw >> 2) * stride + size_t(cw & 3) * NB_FIC_GROUP_BITS + i];
    }
};
struct FetchPlain {
    const int8_t *punct;
    int n_punct;
    __device__ __forceinline__ int8_t operator()(int cw, int i) const { return punct[size_t(cw) * n_punct + i]; }
};
// A12 time de-interleave: logical frame completed by CIF t takes bit i from CIF
// t - 15 + d(i % 16); CIFs before the call come from the history ring.
struct FetchMsc {
    const int8_t *soft;
    size_t stride;
    const int8_t *hist;
    int frames_per_stream;
    int start_bit;
    int nbits;
    __device__ __forceinline__ int8_t operator()(int cw, int i) const {
        const int cifs = frames_per_stream * NB_CIFS;
        const int s = cw / cifs, t = cw - s * cifs;
        const int src = t - 15 + int(__brev(unsigned(i) & 15u) >> 28);   // d(i%16) = bitrev4
        if (src >= 0) {
            const size_t f = size_t(s) * frames_per_stream + (src >> 2);
            return soft[f * stride + NB_FIC_BITS + size_t(src & 3) * NB_CIF_BITS + start_bit + i];
        }
        if (!hist) return 0;
        return hist[(size_t(s) * 15 + (15 + src)) * nbits + i];
    }
};

enum class Tail { kBytes, kFic };

// One wavefront decodes one codeword.  Dynamic LDS per wave:
//   [0, 4*nsteps)            depunctured mother codeword (int8), later the decoded bits
//   [align8, +8*nsteps)      survivor words, later the packed output bytes
template <class Fetch, Tail TAIL>
__global__ __launch_bounds__(WGV) void viterbi_wave_kernel(Fetch fetch, CodeTables code, int n_codewords,
                                                           uint8_t *out, uint8_t *crc_ok, int lds_per_wave) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int cw_raw = blockIdx.x * WAVES_PER_WG + wave;
    const bool active = cw_raw < n_codewords;
    const int cw = active ? cw_raw : n_codewords - 1;
    const int nsteps = code.nsteps;

    int8_t *mother = reinterpret_cast<int8_t *>(smem + size_t(wave) * lds_per_wave);
    const int dec_off = (4 * nsteps + 15) & ~15;
    unsigned long long *dec = reinterpret_cast<unsigned long long *>(mother + dec_off);

    // ---- A8: depuncture into LDS (erasures = 0) ----
    for (int i = lane; i < nsteps; i += 64) reinterpret_cast<int *>(mother)[i] = 0;
    __syncthreads();
    for (int i = lane; i < code.n_punct; i += 64) mother[code.mother_pos[i]] = fetch(cw, i);
    __syncthreads();

    // ---- A9: add-compare-select, lane = new state n ----
    // branch (pred n>>1, input n&1) emits parity(n & POLY[p]); the branch from pred (n>>1)+32
    // emits the complement because every generator taps a[i-6].
    const int sg0 = parity32(lane & 109) ? 1 : -1;   // POLY {109, 79, 83, 109}
    const int sg1 = parity32(lane & 79) ? 1 : -1;
    const int sg2 = parity32(lane & 83) ? 1 : -1;
    const int p0 = lane >> 1, p1 = (lane >> 1) + 32;
    int metric = (lane == 0) ? 0 : -VITERBI_INIT_PENALTY;
    const int *m4 = reinterpret_cast<const int *>(mother);
    for (int t = 0; t < nsteps; t++) {
        const int w = m4[t];
        const int s0 = int8_t(w), s1 = int8_t(w >> 8), s2 = int8_t(w >> 16), s3 = w >> 24;
        const int c = sg0 * (s0 + s3) + sg1 * s1 + sg2 * s2;
        const int cand0 = __shfl(metric, p0) + c;
        const int cand1 = __shfl(metric, p1) - c;
        const bool d = cand1 > cand0;
        metric = d ? cand1 : cand0;
        const unsigned long long word = __ballot(d);
        if (lane == 0) dec[t] = word;
    }
    __syncthreads();

    // ---- traceback from state 0; decoded bit t = newest bit of the state after step t ----
    uint8_t *bits = reinterpret_cast<uint8_t *>(mother);
    if (lane == 0) {
        unsigned s = 0;
        for (int t = nsteps - 1; t >= 0; t--) {
            bits[t] = uint8_t(s & 1u);
            const unsigned h = unsigned(dec[t] >> s) & 1u;
            s = (s >> 1) | (h << 5);
        }
    }
    __syncthreads();

    // ---- A10: pack MSB-first + energy dispersal ----
    const int nbytes = (nsteps - 6) >> 3;
    uint8_t *bytes = reinterpret_cast<uint8_t *>(dec);
    uint8_t *o = out + size_t(cw) * nbytes;
    for (int k = lane; k < nbytes; k += 64) {
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) v = (v << 1) | bits[8 * k + b];
        if (code.prbs_bytes) v ^= code.prbs_bytes[k];
        bytes[k] = uint8_t(v);
        if (active) o[k] = uint8_t(v);
    }
    if (TAIL == Tail::kFic) {
        __syncthreads();
        // ---- A11: CRC16 (x^16+x^12+x^5+1, init FFFF, complemented) of the three FIBs ----
        if (lane < 3) {
            const uint8_t *p = bytes + 32 * lane;
            unsigned crc = 0xFFFFu;
            for (int i = 0; i < 30; i++) {
                crc ^= unsigned(p[i]) << 8;
#pragma unroll
                for (int b = 0; b < 8; b++) crc = (crc & 0x8000u) ? ((crc << 1) ^ 0x1021u) : (crc << 1);
                crc &= 0xFFFFu;
            }
            crc ^= 0xFFFFu;
            if (active) crc_ok[size_t(cw) * 3 + lane] = uint8_t(crc == ((unsigned(p[30]) << 8) | p[31]));
        }
    }
}

// history ring update: hist_out[s][h] = CIF (4F - 15 + h), h = 0..14
__global__ void msc_history_kernel(MscArgs a) {
    const int cifs = a.frames_per_stream * NB_CIFS;
    const size_t total = size_t(a.n_streams) * 15 * a.nbits;
    for (size_t idx = size_t(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += size_t(gridDim.x) * blockDim.x) {
        const int i = int(idx % a.nbits);
        const int h = int((idx / a.nbits) % 15);
        const int s = int(idx / (size_t(a.nbits) * 15));
        const int src = cifs - 15 + h;
        int8_t v = 0;
        if (src >= 0) {
            const size_t f = size_t(s) * a.frames_per_stream + (src >> 2);
            v = a.soft[f * a.soft_stride + NB_FIC_BITS + size_t(src & 3) * NB_CIF_BITS + a.start_bit + i];
        } else if (a.hist_in) {
            v = a.hist_in[(size_t(s) * 15 + (15 + src)) * a.nbits + i];
        }
        a.hist_out[idx] = v;
    }
}

template <class Fetch, Tail TAIL>
hipError_t launch_wave(Fetch f, const CodeTables &c, int n_codewords, uint8_t *out, uint8_t *crc_ok,
                       hipStream_t s) {
    if (n_codewords <= 0) return hipSuccess;
    const int lds_per_wave = int((viterbi_wave_lds_bytes(c.nsteps) + 15) & ~size_t(15));
    const size_t lds = size_t(lds_per_wave) * WAVES_PER_WG;
    auto kern = viterbi_wave_kernel<Fetch, TAIL>;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    const unsigned grid = unsigned((n_codewords + WAVES_PER_WG - 1) / WAVES_PER_WG);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WGV), lds, s, f, c, n_codewords, out, crc_ok, lds_per_wave);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_fic_decode(const CodeTables &c, const int8_t *soft, size_t soft_stride, int n_frames,
                             uint8_t *fib, uint8_t *crc_ok, hipStream_t s) {
    return launch_wave<FetchFic, Tail::kFic>(FetchFic{soft, soft_stride}, c, n_frames * NB_FIC_GROUPS, fib,
                                             crc_ok, s);
}

hipError_t launch_viterbi_plain(const CodeTables &c, const int8_t *punct, int n_codewords, uint8_t *out,
                                hipStream_t s) {
    return launch_wave<FetchPlain, Tail::kBytes>(FetchPlain{punct, c.n_punct}, c, n_codewords, out, nullptr, s);
}

hipError_t launch_msc_decode(const CodeTables &c, const MscArgs &a, hipStream_t s) {
    FetchMsc f{a.soft, a.soft_stride, a.hist_in, a.frames_per_stream, a.start_bit, a.nbits};
    hipError_t e = launch_wave<FetchMsc, Tail::kBytes>(f, c, a.n_streams * a.frames_per_stream * NB_CIFS, a.out,
                                                       nullptr, s);
    if (e != hipSuccess) return e;
    if (a.hist_out) {
        const size_t total = size_t(a.n_streams) * 15 * a.nbits;
        const unsigned grid = unsigned(std::min<size_t>((total + 255) / 256, 2048));
        hipLaunchKernelGGL(msc_history_kernel, dim3(grid), dim3(256), 0, s, a);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace dabk
