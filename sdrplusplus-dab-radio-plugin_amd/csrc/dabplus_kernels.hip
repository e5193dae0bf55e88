// dabplus_kernels.hip -- DAB+ audio super-frame checks on the GPU: the step after the
// MSC Viterbi for DAB+ subchannels.  The reference shows the results as the "Firecode / RS / AU" error flags and
// GetSuperFrameHeader() (/root/reference/src/render_radio_block.cpp:414-437); the code is in the absent
// DAB-Radio submodule, so this restates ETSI TS 102 563 (see oracle/dabplus_oracle.c for the conventions).
//
// One 64-thread workgroup per super-frame (120*s bytes, s = bitrate/8 <= 64):
//   thread j < s     RS(120,110) decode of byte-interleaved column j: syndromes by Horner, and only if one is
//                    non-zero Berlekamp-Massey + Chien + Forney (<= 5 byte errors), GF(2^8)/0x11D via log tables
//   thread 0         Fire code over header bytes 2..10, AU table (2/3/4/6 access units, 12-bit starts)
//   all 64 threads   CRC16-CCITT of each access unit in turn: every lane takes a contiguous piece from register state 0 (the first
//                    piece from 0xFFFF), multiplies it by x^(8 * bytes behind the piece) mod the generator, and the wave XORs
//                    the sixty-four results -- the CRC is linear over GF(2)
// Byte-serial integer work at kB/s rates: one LDS-resident super-frame, no attempt at a roofline.
#include "kernels.hpp"

namespace dabk {

namespace {

struct alignas(16) GfLds {
    uint8_t exp[512];
    uint8_t log[256];
};
// GF(2^8) tables, x^8 + x^4 + x^3 + x^2 + 1, alpha = 2: made by the compiler, copied into LDS by all lanes (one thread
// building them with a 512-step loop at the head of every launch was ~3 us of a ~40 us kernel)
constexpr GfLds make_gf_tables() {
    GfLds g{};
    unsigned x = 1;
    for (int i = 0; i < 255; i++) {
        g.exp[i] = uint8_t(x);
        g.log[x] = uint8_t(i);
        x <<= 1;
        if (x & 0x100u) x ^= 0x11Du;
    }
    for (int i = 255; i < 512; i++) g.exp[i] = g.exp[i - 255];
    g.log[0] = 0;
    return g;
}
__device__ const GfLds GF_TABLES = make_gf_tables();
static_assert(sizeof(GfLds) == 768, "copied as 192 words");

// x^(8 k) mod x^16 + x^12 + x^5 + 1 for k = 0 .. 110 * 64 + 65: what a CRC register that has taken a piece of an access unit still
// has to be multiplied by when k more bytes follow the piece
constexpr int CRC_Z_MAX = 110 * 64 + 66;
struct alignas(16) CrcShift {
    uint16_t z[CRC_Z_MAX];                                     // (copied into LDS as 32-bit words)
};
constexpr CrcShift make_crc_shift() {
    CrcShift t{};
    unsigned v = 1;
    for (int k = 0; k < CRC_Z_MAX; k++) {
        t.z[k] = uint16_t(v);
        for (int b = 0; b < 8; b++) v = (v & 0x8000u) ? (((v << 1) ^ 0x1021u) & 0xFFFFu) : (v << 1);
    }
    return t;
}
__device__ const CrcShift CRC_SHIFT = make_crc_shift();

// a(x) b(x) mod x^16 + x^12 + x^5 + 1 (16-bit operands)
__device__ __forceinline__ unsigned crc_mulmod(unsigned a, unsigned b) {
    unsigned r = 0;
#pragma unroll
    for (int bit = 15; bit >= 0; bit--) {
        r = (r & 0x8000u) ? (((r << 1) ^ 0x1021u) & 0xFFFFu) : (r << 1);
        if ((a >> bit) & 1u) r ^= b;
    }
    return r;
}

__device__ __forceinline__ unsigned gmul(const GfLds &g, unsigned a, unsigned b) {
    return (a && b) ? g.exp[g.log[a] + g.log[b]] : 0u;
}
__device__ __forceinline__ unsigned gdiv(const GfLds &g, unsigned a, unsigned b) {   // b != 0
    return a ? g.exp[g.log[a] + 255 - g.log[b]] : 0u;
}
// coefficient c (non-zero handled by caller) times X^-j, with xinv = log(X^-1)
__device__ __forceinline__ unsigned gpow_term(const GfLds &g, unsigned c, int xinv, int j) {
    return c ? g.exp[(g.log[c] + xinv * j) % 255] : 0u;
}

__device__ __forceinline__ unsigned crc_ccitt_byte(unsigned crc, unsigned byte) {
    crc = ((crc >> 8) | (crc << 8)) & 0xFFFFu;
    crc ^= byte;
    crc ^= (crc & 0xFFu) >> 4;
    crc ^= (crc << 12) & 0xFFFFu;
    crc ^= ((crc & 0xFFu) << 5) & 0xFFFFu;
    return crc;
}

// The ten syndromes of all s columns, on all 64 lanes: S_k = sum_i r_i alpha^(k (119 - i)).  Lane (column c, part q) takes
// the bytes i = q len .. (q + 1) len - 1 of its column term by term -- one log look-up per byte, then ten INDEPENDENT exp
// look-ups (the index advances by 119 - i per k) -- and XORs its partial sums into syn[c][k].  (Horner on one lane per
// column, as before, was 1 200 dependent three-look-up multiplications: ~50 us of a super-frame's ~100; this is ~2.)
__device__ __forceinline__ void rs_syndromes(const GfLds &g, const uint8_t *sf, int s, int tid, unsigned (*syn)[10]) {
    constexpr int N = 120, T2 = 10;
    const int parts = max(1, 64 / s);
    const int len = (N + parts - 1) / parts;
    const int c = tid % s, q = tid / s;
    if (q >= parts) return;
    unsigned S[T2];
#pragma unroll
    for (int k = 0; k < T2; k++) S[k] = 0;
    const int i1 = min(N, (q + 1) * len);
    for (int i = q * len; i < i1; i++) {
        const unsigned r = sf[c + s * i];
        if (!r) continue;
        const int p = N - 1 - i;                                // < 255
        int idx = g.log[r];                                     // k = 0: alpha^0
#pragma unroll
        for (int k = 0; k < T2; k++) {
            S[k] ^= g.exp[idx];
            idx += p;
            if (idx >= 255) idx -= 255;
        }
    }
#pragma unroll
    for (int k = 0; k < T2; k++)
        if (S[k]) atomicXor(&syn[c][k], S[k]);
}

// in-place RS(120,110) decode of the column {sf[j + s*i]} from its syndromes; returns corrected bytes or -1
__device__ int rs_decode_column(const GfLds &g, uint8_t *sf, int j, int s, const unsigned *syn) {
    constexpr int N = 120, T2 = 10;
    unsigned S[T2];
    unsigned any = 0;
#pragma unroll
    for (int k = 0; k < T2; k++) { S[k] = syn[k]; any |= S[k]; }
    if (!any) return 0;
    // Berlekamp-Massey (fixed-bound loops so everything stays in registers)
    unsigned L[T2 + 1], B[T2 + 1], Tm[T2 + 1];
#pragma unroll
    for (int i = 0; i <= T2; i++) { L[i] = 0; B[i] = 0; }
    L[0] = 1; B[0] = 1;
    int ll = 0, m = 1;
    unsigned bb = 1;
#pragma unroll
    for (int n = 0; n < T2; n++) {
        unsigned d = S[n];
#pragma unroll
        for (int i = 1; i <= T2; i++)
            if (i <= ll && i <= n) d ^= gmul(g, L[i], S[n - i]);
        if (d == 0) { m++; continue; }
#pragma unroll
        for (int i = 0; i <= T2; i++) Tm[i] = L[i];
        const unsigned coef = gdiv(g, d, bb);
#pragma unroll
        for (int i = 0; i <= T2; i++)
            if (i >= m) L[i] ^= gmul(g, coef, B[(i - m) < 0 ? 0 : (i - m)]);
        if (2 * ll <= n) {
            ll = n + 1 - ll;
#pragma unroll
            for (int i = 0; i <= T2; i++) B[i] = Tm[i];
            bb = d;
            m = 1;
        } else {
            m++;
        }
    }
    if (ll > 5) return -1;
    unsigned Om[T2];
#pragma unroll
    for (int i = 0; i < T2; i++) {
        unsigned v = 0;
#pragma unroll
        for (int jj = 0; jj <= 5; jj++)
            if (jj <= i && jj <= ll) v ^= gmul(g, L[jj], S[i - jj]);
        Om[i] = v;
    }
    int nerr = 0;
    int pos[5];
    unsigned val[5];
    for (int i = 0; i < N; i++) {
        const int p = N - 1 - i;
        const int xinv = (255 - p) % 255;
        unsigned ev = 0;
#pragma unroll
        for (int jj = 0; jj <= 5; jj++) ev ^= gpow_term(g, L[jj], xinv, jj);
        if (ev) continue;
        if (nerr == 5) return -1;
        unsigned om = 0, dl = 0;
#pragma unroll
        for (int jj = 0; jj < T2; jj++) om ^= gpow_term(g, Om[jj], xinv, jj);
        dl = gpow_term(g, L[1], xinv, 0) ^ gpow_term(g, L[3], xinv, 2) ^ gpow_term(g, L[5], xinv, 4);
        if (dl == 0) return -1;
        const unsigned e = gmul(g, g.exp[p % 255], gdiv(g, om, dl));
#pragma unroll
        for (int q = 0; q < 5; q++)
            if (q == nerr) { pos[q] = i; val[q] = e; }
        nerr++;
    }
    if (nerr != ll) return -1;
#pragma unroll
    for (int q = 0; q < 5; q++)
        if (q < nerr) sf[j + s * pos[q]] ^= uint8_t(val[q]);
    return nerr;
}

__global__ __launch_bounds__(64) void dabplus_superframe_kernel(const uint8_t *in, size_t in_stride, int s,
                                                                uint8_t *out, SuperframeStatus *status,
                                                                unsigned long long *done_flag, unsigned long long done_seq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GfLds &g = *reinterpret_cast<GfLds *>(smem);
    uint8_t *sf = smem + sizeof(GfLds);
    const int n_z = 110 * s + 66;                                  // shift table entries this super-frame can ask for
    uint16_t *zt = reinterpret_cast<uint16_t *>(smem + sizeof(GfLds) + ((size_t(120) * s + 15) & ~size_t(15)));
    __shared__ int sh_corrected, sh_bad, sh_mask, sh_naus;
    __shared__ int sh_start[8];
    __shared__ unsigned sh_syn[64][10];
    const int tid = threadIdx.x;
    const int nbytes = 120 * s;
    const uint8_t *src = in + size_t(blockIdx.x) * in_stride;
#ifdef DABPLUS_PHASE_TIMING
    const long long t0 = wall_clock64();
    long long t1 = 0, t2 = 0, t3 = 0;
#define STAMP(v) v = wall_clock64()
#else
#define STAMP(v)
#endif
    for (int i = tid; i < 192; i += 64) reinterpret_cast<uint32_t *>(&g)[i] = reinterpret_cast<const uint32_t *>(&GF_TABLES)[i];
    for (int i = tid; i < n_z / 2; i += 64) reinterpret_cast<uint32_t *>(zt)[i] = reinterpret_cast<const uint32_t *>(&CRC_SHIFT)[i];
    if (tid == 0) { sh_corrected = 0; sh_bad = 0; sh_mask = 0; sh_naus = 0; }
    // the super-frame into LDS: 16 bytes per lane and trip when the source allows (one or two round trips -- the source
    // may be the caller's page-locked host buffer), bytes otherwise
    if (((reinterpret_cast<uintptr_t>(src) | size_t(nbytes)) & 15) == 0) {
        for (int i = tid; i < nbytes / 16; i += 64) reinterpret_cast<uint4 *>(sf)[i] = reinterpret_cast<const uint4 *>(src)[i];
    } else {
        for (int i = tid; i < nbytes; i += 64) sf[i] = src[i];
    }
    if (tid < 8) sh_start[tid] = 0;
    for (int i = tid; i < 64 * 10; i += 64) (&sh_syn[0][0])[i] = 0;
    __syncthreads();
    STAMP(t1);
    rs_syndromes(g, sf, s, tid, sh_syn);
    __syncthreads();
    if (tid < s) {
        const int r = rs_decode_column(g, sf, tid, s, sh_syn[tid]);
        if (r < 0) atomicAdd(&sh_bad, 1);
        else if (r > 0) atomicAdd(&sh_corrected, r);
    }
    __syncthreads();
    STAMP(t2);
    int fire_ok = 0;
    if (tid == 0) {
        unsigned crc = 0, any = unsigned(sf[0]) | sf[1];
        for (int i = 2; i < 11; i++) {
            any |= sf[i];
            crc ^= unsigned(sf[i]) << 8;
            for (int b = 0; b < 8; b++) crc = (crc & 0x8000u) ? ((crc << 1) ^ 0x782Fu) : (crc << 1);
            crc &= 0xFFFFu;
        }
        // an all-zero header is its own (zero) check word: that is silence or erasures, not a super-frame
        fire_ok = any != 0 && crc == ((unsigned(sf[0]) << 8) | sf[1]);
        if (fire_ok) {
            const int dac_rate = (sf[2] >> 6) & 1, sbr = (sf[2] >> 5) & 1;
            const int naus = dac_rate ? (sbr ? 3 : 6) : (sbr ? 2 : 4);
            sh_start[0] = naus == 2 ? 5 : naus == 3 ? 6 : naus == 4 ? 8 : 11;
            // the 12-bit start addresses follow the header byte back to back: bytes 3 .. 10 as one 64-bit word
            unsigned long long w = 0;
            for (int i = 3; i < 11; i++) w = (w << 8) | sf[i];
            for (int a = 1; a < naus; a++) sh_start[a] = int((w >> (64 - 12 * a)) & 0xFFFu);
            sh_start[naus] = 110 * s;
            sh_naus = naus;
        }
    }
    __syncthreads();
    {
        const int naus = sh_naus;
        for (int a = 0; a < naus; a++) {
            const int b0 = sh_start[a], b1 = sh_start[a + 1];                 // (the same for every lane)
            if (!(b0 >= 3 && b1 <= 110 * s && b1 - b0 >= 3)) continue;
            // len payload bytes, cut into 64 pieces of m, padded AT THE FRONT: lane l takes the padded positions
            // [l m, (l + 1) m), and (63 - l) m bytes follow its piece
            const int len = b1 - 2 - b0, m = (len + 63) >> 6, pad = 64 * m - len;
            int lo = tid * m - pad;
            const int hi = lo + m;
            unsigned crc = 0;
            if (hi > 0) {
                if (lo <= 0) { lo = 0; crc = 0xFFFFu; }                        // the piece the message starts in
                for (int i = lo; i < hi; i++) crc = crc_ccitt_byte(crc, sf[b0 + i]);
                crc = crc_mulmod(crc, zt[(63 - tid) * m]);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) crc ^= unsigned(__shfl_xor(int(crc), off));
            crc ^= 0xFFFFu;
            if (tid == 0 && crc == ((unsigned(sf[b1 - 2]) << 8) | sf[b1 - 1])) sh_mask |= 1 << a;
        }
    }
    __syncthreads();
    STAMP(t3);
    uint8_t *dst = out + size_t(blockIdx.x) * size_t(110 * s);
    for (int i = tid; i < 110 * s; i += 64) dst[i] = sf[i];
    if (tid == 0) {
        SuperframeStatus st;
        st.firecode_ok = fire_ok;
        st.rs_corrected = sh_corrected;
        st.rs_uncorrectable = sh_bad;
        st.num_aus = sh_naus;
        st.au_crc_mask = sh_mask;
        for (int a = 0; a < 8; a++) st.au_start[a] = sh_start[a];
        st.reserved[0] = st.reserved[1] = st.reserved[2] = 0;
#ifdef DABPLUS_PHASE_TIMING
        st.reserved[0] = int(t1 - t0); st.reserved[1] = int(t2 - t1); st.reserved[2] = int(t3 - t2);   // 100 MHz ticks: staging | RS | header + CRCs
#endif
        status[blockIdx.x] = st;
    }
    if (done_flag) {
        // (a launch of ONE workgroup: launch_dabplus_superframes.)  Every lane's stores -- the corrected bytes, lane 0's status
        // record -- are out at system scope before the barrier; only then is the word stored that the host is watching.
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace

hipError_t launch_dabplus_superframes(const uint8_t *in, size_t in_stride, int n_superframes, int s, uint8_t *out,
                                      SuperframeStatus *status, hipStream_t stream, unsigned long long *done_flag,
                                      unsigned long long done_seq) {
    if (n_superframes <= 0) return hipSuccess;
    if (s < 1 || s > 64 || (done_flag && n_superframes != 1)) return hipErrorInvalidValue;
    const size_t lds = sizeof(GfLds) + ((size_t(120) * s + 15) & ~size_t(15)) + (((size_t(110) * s + 66) * 2 + 3) & ~size_t(3));
    hipLaunchKernelGGL(dabplus_superframe_kernel, dim3(unsigned(n_superframes)), dim3(64), lds, stream, in, in_stride,
                       s, out, status, done_flag, done_seq);
    return hipGetLastError();
}

}  // namespace dabk
