// viterbi_lane_kernels.hip -- the K=7 rate-1/4 soft Viterbi for LARGE batches: one codeword per LANE.
//
// viterbi_kernels.hip spends a whole wavefront on one codeword (lane = state); per trellis step that costs
// ~10 VALU + ~6 SALU instructions for 64 add-compare-selects because branch metrics, the lane exchange and the
// survivor bookkeeping are paid per state.  Here a lane owns a codeword and keeps all 64 path metrics in 32
// VGPRs as packed int16 pairs, so per step a wave does 64 codewords x 64 ACS in ~250 instructions (~4 per
// codeword-step) with no cross-lane traffic at all.  It needs >= ~25k codewords per launch to fill the chip
// (one wave = 64 codewords), so the launcher picks it by batch size (rows A8..A12 of SURVEY.md section 8a,
// same results bit for bit).
//
// Exactness.  Metrics are stored doubled (2m, LSB clear).  The candidate through the older-bit-1 predecessor
// is even, the one through the older-bit-0 predecessor is made odd by adding 1, so one packed max yields both
// the survivor metric (after clearing the LSB, exactly 2m again) and, in its LSB, the survivor tag with the
// reference tie rule (older-bit-1 wins only when strictly larger: 2*m0 + 1 > 2*m1  <=>  m0 >= m1).  Every 12 steps the metric of one slot is subtracted
// from all (a common offset never changes a decision); with start penalty 6144 (> 6*2*508, the largest lead a
// wrong start state can gain before its path merges, so equivalent to the reference's "known start state") the
// doubled values stay within +-24480 < 2^15.
//
// Layout.  Slot p (0..63) = register p/2, half p%2 holds state rotl6(p, t mod 6) at step t.  Going from t to t+1
// only bit q = (5 - t) mod 6 of the slot index changes meaning (oldest bit out, newest bit in), so the two
// predecessors of slot p are the slots p with bit q cleared / set: for q >= 1 that is a PAIR OF REGISTERS updated
// in place by packed add/sub/max on both halves at once; for q = 0 it is the two halves of one register (one
// half-swap).  Nothing ever moves.
#include <algorithm>
#include <type_traits>
#include <vector>

#include "mem_stream.hpp"
#include "kernels.hpp"
#include "dab_tables.hpp"

namespace dabk {

using namespace dab;

namespace {

#ifndef LANE_FWD_ATTR
#define LANE_FWD_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif

constexpr int LANE_INIT2 = 2 * 6144;      // doubled start penalty of states != 0
// staging window of the fused forward pass (K2')
constexpr int FT = 24;                          // steps per tile (four phase cycles)
constexpr int FCOLS = 8;                        // 16-byte chunks per staged row (span <= 15 + 4*FT = 111 bytes)
constexpr int FPITCH = FCOLS * 16 + 4;          // 132 B = 33 dwords (odd); bytes 128..131 stay zero (erased bits)
constexpr int FERASED = FCOLS * 16;             // column of the always-zero byte

typedef short s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned as_u(s2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ s2 as_s2(unsigned v) { return __builtin_bit_cast(s2, v); }
__device__ __forceinline__ unsigned pk_add(unsigned a, unsigned b) { return as_u(as_s2(a) + as_s2(b)); }
__device__ __forceinline__ unsigned pk_sub(unsigned a, unsigned b) { return as_u(as_s2(a) - as_s2(b)); }
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b) {
    return as_u(__builtin_elementwise_max(as_s2(a), as_s2(b)));
}
// (lo16(lo), lo16(hi))
__device__ __forceinline__ unsigned pack16(int lo, int hi) { return __builtin_amdgcn_perm(unsigned(hi), unsigned(lo), 0x05040100u); }
// a packed add / subtract whose FIRST operand has its halves swapped: (hi(a) +- lo(k), lo(a) +- hi(k)).  The swap rides on the
// instruction's op_sel bits (the compiler does not fold a shuffle or a byte permute into them: it emits a second instruction)
__device__ __forceinline__ unsigned pk_add_swapped(unsigned a, unsigned k) {
    unsigned d;
    asm("v_pk_add_u16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(k));
    return d;
}
__device__ __forceinline__ unsigned pk_sub_swapped(unsigned a, unsigned k) {
    unsigned d;
    asm("v_pk_sub_i16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(k));
    return d;
}
// (m & mask) | 1 in one instruction (a VOP3 cannot carry a 32-bit literal on this target: the mask travels in a scalar register)
__device__ __forceinline__ unsigned and_or1(unsigned m, unsigned mask) {
    unsigned d;
    asm("v_and_or_b32 %0, %1, %2, 1" : "=v"(d) : "v"(m), "s"(mask));
    return d;
}

__host__ __device__ constexpr int par7(int x) { x ^= x >> 4; x ^= x >> 2; x ^= x >> 1; return x & 1; }
__host__ __device__ constexpr int rotl6c(int v, int r) { return r == 0 ? v : (((v << r) | (v >> (6 - r))) & 63); }
// sign pattern of the branch (state n reached from its older-bit-0 predecessor): bit0 <-> (s0+s3), bit1 <-> s1, bit2 <-> s2
__host__ __device__ constexpr int sig_of(int n) { return par7(n & 109) | (par7(n & 79) << 1) | (par7(n & 83) << 2); }

// ---- where a codeword's punctured soft bits come from ----
// row(g): first punctured byte of codeword g.  PRE = how many earlier codewords a codeword draws from (the time
// de-interleaver reads CIFs t-15..t of its own stream; consecutive codeword indices are consecutive CIFs).
struct LSrcFic {
    static constexpr int PRE = 0;
    const int8_t *soft;
    size_t stride;
    __device__ __forceinline__ const int8_t *row(int g) const {
        return soft + size_t(g >> 2) * stride + size_t(g & 3) * NB_FIC_GROUP_BITS;
    }
};
struct LSrcPlain {
    static constexpr int PRE = 0;
    const int8_t *punct;
    int n_punct;
    __device__ __forceinline__ const int8_t *row(int g) const { return punct + size_t(g) * n_punct; }
};
struct LSrcMsc {
    static constexpr int PRE = 15;
    const int8_t *soft;
    size_t stride;
    const int8_t *hist;
    int cifs_per_stream;
    int base_off;             // first byte of the codeword's part in CIF 0 of a frame (FIC: 0; MSC: 9216 + 64*start CU)
    int per_cif;              // distance between the four codewords of a frame (FIC group: 2304; CIF: 55296)
    int nbits;
    int d_force;              // -1: time de-interleaver delays from the descriptor table; 15: no interleaving (FIC)
    __device__ __forceinline__ const int8_t *row(int g) const {
        return soft + size_t(g >> 2) * stride + base_off + size_t(g & 3) * per_cif;
    }
};
__host__ __device__ inline LSrcMsc make_msc_src(const MscArgs &a) {
    return LSrcMsc{a.soft, a.soft_stride, a.hist_in, a.frames_per_stream * NB_CIFS, NB_FIC_BITS + a.start_bit, NB_CIF_BITS,
                   a.nbits, -1};
}
// A descriptor is (delay * FPITCH + column) | column << 16: the low half is the window offset of a time-interleaved
// bit, the high half the column alone for sources without interleaving (delay 0: FIC / plain codewords) or with a
// forced delay (the FIC riding in a grouped launch: d_force = 15, its own row).  desc_shift / desc_extra pick the
// half and the constant to add -- wave-uniform, so the per-byte work is one bit-field extract and one add.
template <class Src>
__device__ __forceinline__ int desc_shift(const Src &) { return 16; }
template <class Src>
__device__ __forceinline__ int desc_extra(const Src &) { return 0; }
__device__ __forceinline__ int desc_shift(const LSrcMsc &s) { return s.d_force >= 0 ? 16 : 0; }
__device__ __forceinline__ int desc_extra(const LSrcMsc &s) { return s.d_force >= 0 ? s.d_force * FPITCH : 0; }

// ---------------------------------------------------------------------------------------------------------
// K1: depuncture (+ time de-interleave) + transpose.  Block = (group of 64 codewords, tile of 64 steps).
// The punctured bytes the tile needs -- one contiguous span of <= 256 bytes per source codeword -- are staged
// in LDS with 16-byte loads along the codewords; then every thread assembles the 4 soft bytes of (step, lane)
// from LDS and the stores run along the lanes: M[group][step][lane].
// ---------------------------------------------------------------------------------------------------------
constexpr int PREP_STEPS = 64;
// bytes per staged row: the span (<= 4 bytes per step) rounded up to 16-byte chunks + one chunk of alignment slack,
// padded to an odd number of dwords (64 steps: 17 chunks + 4 = 276 B = 69 dwords)
constexpr int PREP_PITCH = ((4 * PREP_STEPS + 15) / 16 + 1) * 16 + 4;

template <class Src>
__global__ __launch_bounds__(256) void lane_prep_kernel(Src src, const int32_t *punct_idx, int nsteps, int n_codewords,
                                                        int vec16, uint32_t *M) {
    constexpr int ROWS = 64 + Src::PRE;
    __shared__ __attribute__((aligned(16))) uint8_t win[ROWS * PREP_PITCH];
    __shared__ int s_idx[4 * PREP_STEPS];
    __shared__ int s_lo, s_hi;
    const int tid = threadIdx.x;
    const int group = blockIdx.y;
    const int t0 = blockIdx.x * PREP_STEPS;
    const int cw0 = group * 64;
    if (tid == 0) { s_lo = 0x7fffffff; s_hi = 0; }
    __syncthreads();
    for (int i = tid; i < 4 * PREP_STEPS; i += 256) {
        const int p = 4 * t0 + i;
        const int idx = (p < 4 * nsteps) ? punct_idx[p] : -1;
        s_idx[i] = idx;
        if (idx >= 0) { atomicMin(&s_lo, idx); atomicMax(&s_hi, idx + 1); }
    }
    __syncthreads();
    const int lo = s_lo, hi = s_hi;
    const int lo_al = lo & ~15;
    if (hi > lo) {
        // stage rows cw0-PRE .. cw0+63 (those that exist)
        if (vec16) {
            const int nchunks = (hi - lo_al + 15) >> 4;       // <= 17
            for (int i = tid; i < ROWS * nchunks; i += 256) {
                const int row = i / nchunks, c = i - row * nchunks;
                const int g = cw0 - Src::PRE + row;
                if (g < 0 || g >= n_codewords) continue;
                const uint4 v = *reinterpret_cast<const uint4 *>(src.row(g) + lo_al + 16 * c);
                uint32_t *d = reinterpret_cast<uint32_t *>(win + row * PREP_PITCH + 16 * c);
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            const int span = hi - lo_al;
            for (int i = tid; i < ROWS * span; i += 256) {
                const int row = i / span, c = i - row * span;
                const int g = cw0 - Src::PRE + row;
                if (g < 0 || g >= n_codewords) continue;
                win[row * PREP_PITCH + c] = uint8_t(src.row(g)[lo_al + c]);
            }
        }
    }
    __syncthreads();
    const int lane = tid & 63;
    const int r_eff = min(lane, n_codewords - 1 - cw0);       // lanes past the end repeat the last codeword
    int t_in_stream = 0, stream = 0;
    if constexpr (Src::PRE > 0) {
        const int cw = cw0 + r_eff;
        stream = cw / src.cifs_per_stream;
        t_in_stream = cw - stream * src.cifs_per_stream;
    }
    for (int sstep = tid >> 6; sstep < PREP_STEPS; sstep += 4) {
        const int t = t0 + sstep;
        if (t >= nsteps) break;
        uint32_t w = 0;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int idx = s_idx[4 * sstep + m];
            if (idx < 0) continue;
            unsigned v;
            if constexpr (Src::PRE > 0) {
                const int d = int(__brev(unsigned(idx) & 15u) >> 28);
                if (t_in_stream + d >= 15) {
                    v = win[(r_eff + d) * PREP_PITCH + (idx - lo_al)];
                } else {                                      // before the stream's first CIF: carried history
                    v = src.hist ? uint8_t(src.hist[(size_t(stream) * 15 + t_in_stream + d) * src.nbits + idx]) : 0u;
                }
            } else {
                v = win[r_eff * PREP_PITCH + (idx - lo_al)];
            }
            w |= v << (8 * m);
        }
        M[(size_t(group) * nsteps + t) * 64 + lane] = w;
    }
}

// ---------------------------------------------------------------------------------------------------------
// K2: forward pass.  One wave = 64 codewords.
// ---------------------------------------------------------------------------------------------------------
// The four soft bits of a trellis step (sign-extended) and the packed branch constants a phase needs.
// A state's doubled branch correlation is 2c[S] = +-a +-b +-c with a = 2(s0 + s3), b = 2 s1, c = 2 s2 and the signs given
// by the bits of its sign pattern S = sig_of(state) (bit set = plus), so c[S ^ 7] = -c[S].  The two states of a register
// (its two 16-bit halves) differ in ONE state bit, i.e. their patterns differ by a constant D of the phase: the packed
// constant of a register is (c[S], c[S ^ D]), and only the four odd S are formed -- with packed arithmetic, from a, b, c
// packed with their second half negated where D says so; an even S uses the negative of its odd partner by swapping
// the roles of add and subtract (lane_pair / lane_self1).
struct Soft4 {
    int s0, s1, s2, s3;
};
__device__ __forceinline__ Soft4 unpack_soft(uint32_t w) {
    return Soft4{int(int8_t(w)), int(int8_t(w >> 8)), int(int8_t(w >> 16)), int(w) >> 24};
}
struct KC {
    unsigned k[4];             // index S >> 1 for S = 1, 3, 5, 7
};
template <int D>
__device__ __forceinline__ KC branch_consts(const Soft4 &q) {
    const int a = 2 * (q.s0 + q.s3), b = 2 * q.s1, c = 2 * q.s2;
    const unsigned A2 = pack16(a, (D & 1) ? -a : a), B2 = pack16(b, (D & 2) ? -b : b), C2 = pack16(c, (D & 4) ? -c : c);
    const unsigned e1 = pk_add(A2, B2), e2 = pk_sub(A2, B2);
    KC k;
    k.k[3] = pk_add(e1, C2);   // S = 7 (+,+,+)
    k.k[1] = pk_sub(e1, C2);   // S = 3 (+,+,-)
    k.k[2] = pk_add(e2, C2);   // S = 5 (+,-,+)
    k.k[0] = pk_sub(e2, C2);   // S = 1 (+,-,-)
    return k;
}

// Record the survivor tags (bits 0 and 16 of the max results) of a pair of registers: one byte gather, one mask,
// one shift-or.  Pair I puts its 4 tags into bits I%8 + {0, 8, 16, 24} of word I/8, in the order (first register
// low half, high half, second register low half, high half).  The tags also stay in the registers: the next step
// forces them to 1 or 0 anyway when it makes the operand odd or even.
template <int I>
__device__ __forceinline__ void take_tags(unsigned first, unsigned second, unsigned &acc0, unsigned &acc1) {
    const unsigned g = __builtin_amdgcn_perm(second, first, 0x06040200u) & 0x01010101u;
    if constexpr (I < 8) asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(acc0) : "v"(g), "n"(I & 7));
    else asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(acc1) : "v"(g), "n"(I & 7));
}

template <int PH, int I>
__device__ __forceinline__ void lane_pair(unsigned (&M)[32], const KC &kc, unsigned &acc0, unsigned &acc1) {
    constexpr int Q = 5 - PH;                 // slot bit replaced in this step, 1..5 here
    constexpr int RB = Q - 1;                 // the same bit in register-index space
    constexpr int RA = ((I >> RB) << (RB + 1)) | (I & ((1 << RB) - 1));
    constexpr int RBI = RA | (1 << RB);
    constexpr int ROT = (PH + 1) % 6;         // layout after the step
    constexpr int SLO = sig_of(rotl6c(2 * RA, ROT));           // pattern of the register's low half; the high half's is SLO ^ D
    static_assert(sig_of(rotl6c(2 * RA + 1, ROT)) == (SLO ^ sig_of(rotl6c(1, ROT))), "the halves differ by the phase's constant");
    constexpr bool NEG = (SLO & 1) == 0;      // even pattern: minus the constant of its odd partner SLO ^ 7
    const unsigned k2 = kc.k[(NEG ? (SLO ^ 7) : SLO) >> 1];
    const unsigned a1 = M[RA] | 0x00010001u;                   // older-bit-0 candidates are odd (metric + 1)
    const unsigned b = M[RBI] & 0xFFFEFFFEu;                   // older-bit-1 candidates even (metric)
    // new states with newest bit 0 (register RA): a1 + 2c, b - 2c; newest bit 1 (register RB): c flips sign
    const unsigned x = NEG ? pk_sub(a1, k2) : pk_add(a1, k2), y = NEG ? pk_add(b, k2) : pk_sub(b, k2);
    const unsigned u = NEG ? pk_add(a1, k2) : pk_sub(a1, k2), v = NEG ? pk_sub(b, k2) : pk_add(b, k2);
    M[RA] = pk_max(x, y);
    M[RBI] = pk_max(u, v);
    take_tags<I>(M[RA], M[RBI], acc0, acc1);
}

template <int R>
__device__ __forceinline__ unsigned lane_self1(unsigned m, const KC &kc) {
    // phase 5: the replaced bit is the half bit; both predecessors live in this register.  After the step the
    // layout rotation is 0, so the new state of slot 2R is 2R itself, and both halves share one pattern (D = 0).
    constexpr int S = sig_of(2 * R);
    constexpr bool NEG = (S & 1) == 0;
    const unsigned k = kc.k[(NEG ? (S ^ 7) : S) >> 1];
    const unsigned a = and_or1(m, 0xFFFEFFFFu);          // (lo + 1, hi): odd older-bit-0, even older-bit-1 operand
    const unsigned p = NEG ? pk_sub(a, k) : pk_add(a, k);                 // (lo + 1 + 2c, hi + 2c)
    const unsigned q = NEG ? pk_add_swapped(a, k) : pk_sub_swapped(a, k); // (hi - 2c, lo + 1 - 2c)
    return pk_max(p, q);
}
template <int I>
__device__ __forceinline__ void lane_self(unsigned (&M)[32], const KC &kc, unsigned &acc0, unsigned &acc1) {
    M[2 * I] = lane_self1<2 * I>(M[2 * I], kc);
    M[2 * I + 1] = lane_self1<2 * I + 1>(M[2 * I + 1], kc);
    take_tags<I>(M[2 * I], M[2 * I + 1], acc0, acc1);
}

template <int PH>
__device__ __forceinline__ void lane_step(unsigned (&M)[32], const Soft4 &q, uint2 *dec_out) {
    unsigned acc0 = 0, acc1 = 0;
    if constexpr (PH < 5) {
        const KC kc = branch_consts<sig_of(rotl6c(1, (PH + 1) % 6))>(q);
        lane_pair<PH, 0>(M, kc, acc0, acc1);  lane_pair<PH, 1>(M, kc, acc0, acc1);
        lane_pair<PH, 2>(M, kc, acc0, acc1);  lane_pair<PH, 3>(M, kc, acc0, acc1);
        lane_pair<PH, 4>(M, kc, acc0, acc1);  lane_pair<PH, 5>(M, kc, acc0, acc1);
        lane_pair<PH, 6>(M, kc, acc0, acc1);  lane_pair<PH, 7>(M, kc, acc0, acc1);
        lane_pair<PH, 8>(M, kc, acc0, acc1);  lane_pair<PH, 9>(M, kc, acc0, acc1);
        lane_pair<PH, 10>(M, kc, acc0, acc1); lane_pair<PH, 11>(M, kc, acc0, acc1);
        lane_pair<PH, 12>(M, kc, acc0, acc1); lane_pair<PH, 13>(M, kc, acc0, acc1);
        lane_pair<PH, 14>(M, kc, acc0, acc1); lane_pair<PH, 15>(M, kc, acc0, acc1);
    } else {
        const KC kc = branch_consts<0>(q);
        lane_self<0>(M, kc, acc0, acc1);          lane_self<1>(M, kc, acc0, acc1);
        lane_self<2>(M, kc, acc0, acc1);          lane_self<3>(M, kc, acc0, acc1);
        lane_self<4>(M, kc, acc0, acc1);          lane_self<5>(M, kc, acc0, acc1);
        lane_self<6>(M, kc, acc0, acc1);          lane_self<7>(M, kc, acc0, acc1);
        lane_self<8>(M, kc, acc0, acc1);          lane_self<9>(M, kc, acc0, acc1);
        lane_self<10>(M, kc, acc0, acc1);          lane_self<11>(M, kc, acc0, acc1);
        lane_self<12>(M, kc, acc0, acc1);          lane_self<13>(M, kc, acc0, acc1);
        lane_self<14>(M, kc, acc0, acc1);          lane_self<15>(M, kc, acc0, acc1);
    }
    st_stream(dec_out, make_uint2(acc0, acc1));
}
template <int PH>
__device__ __forceinline__ void lane_step(unsigned (&M)[32], uint32_t w, uint2 *dec_out) {
    lane_step<PH>(M, unpack_soft(w), dec_out);
}

// Four groups per workgroup (one per SIMD); the launcher pads the LDS request so that every CU gets the same
// number of workgroups -- the dispatcher otherwise packs some SIMDs with 3 waves while others hold 1.
__global__ __launch_bounds__(256) void lane_forward_kernel(const uint32_t *Msoft, int nsteps, int groups, uint2 *dec) {
    const int lane = threadIdx.x & 63;
    const int group = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (group >= groups) return;
    const size_t base = size_t(group) * nsteps * 64 + lane;
    const uint32_t *src = Msoft + base;
    uint2 *dst = dec + base;
    unsigned M[32];
#pragma unroll
    for (int r = 0; r < 32; r++) M[r] = pack16(-LANE_INIT2, -LANE_INIT2);
    M[0] = pack16(0, -LANE_INIT2);                         // state 0 sits in slot 0 at t = 0
    uint32_t w[6], wn[6];
#pragma unroll
    for (int i = 0; i < 6; i++) w[i] = src[size_t(i) * 64];
    for (int t = 0; t < nsteps; t += 6) {
        const int tn = (t + 6 < nsteps) ? t + 6 : t;       // prefetch the next phase cycle (last one re-reads)
#pragma unroll
        for (int i = 0; i < 6; i++) wn[i] = src[size_t(tn + i) * 64];
        lane_step<0>(M, w[0], dst + size_t(t + 0) * 64);
        lane_step<1>(M, w[1], dst + size_t(t + 1) * 64);
        lane_step<2>(M, w[2], dst + size_t(t + 2) * 64);
        lane_step<3>(M, w[3], dst + size_t(t + 3) * 64);
        lane_step<4>(M, w[4], dst + size_t(t + 4) * 64);
        lane_step<5>(M, w[5], dst + size_t(t + 5) * 64);
        if (((t / 6) & 1) == 1) {                          // every 12 steps (two phase cycles)
            const unsigned ref = pack16(int(M[0]), int(M[0])) & 0xFFFEFFFEu;
#pragma unroll
            for (int r = 0; r < 32; r++) M[r] = pk_sub(M[r], ref);
        }
#pragma unroll
        for (int i = 0; i < 6; i++) w[i] = wn[i];
    }
}

// ---------------------------------------------------------------------------------------------------------
// K2': forward pass that depunctures for itself ("fused" variant: no lane_prep_kernel, no M[] round trip).
// Each wave stages the punctured bytes its 64 codewords need for the next FT steps -- rows = the codewords
// themselves plus, for the time de-interleaver, the 15 CIFs before the first one (history rows at a stream start)
// -- in a wave-private LDS window with 16-byte loads issued one tile ahead, and every lane picks its 4 soft bytes
// per step from the window: row = lane + delay, column = punctured index - tile start, both from a per-profile
// descriptor table that the scalar unit walks.  Requires that a group of 64 codewords never straddles two streams
// (CIFs per stream a multiple of 64); other shapes take the prep kernel.
// ---------------------------------------------------------------------------------------------------------

// row r (0 .. 63+PRE) of the window of the group whose first codeword is cw0 -> first punctured byte, or nullptr
// for a row that does not exist (reads as erasures)
__device__ __forceinline__ const int8_t *fused_row(const LSrcFic &s, int cw0, int r, int n_codewords) {
    const int g = cw0 + r;
    return g < n_codewords ? s.row(g) : nullptr;
}
__device__ __forceinline__ const int8_t *fused_row(const LSrcPlain &s, int cw0, int r, int n_codewords) {
    const int g = cw0 + r;
    return g < n_codewords ? s.row(g) : nullptr;
}
__device__ __forceinline__ const int8_t *fused_row(const LSrcMsc &s, int cw0, int r, int n_codewords) {
    const int stream = cw0 / s.cifs_per_stream;
    const int v = cw0 - stream * s.cifs_per_stream - 15 + r;          // CIF of this stream, < 0 = carried history
    if (v < 0) return s.hist ? s.hist + (size_t(stream) * 15 + (15 + v)) * s.nbits : nullptr;
    const int g = stream * s.cifs_per_stream + v;
    return g < n_codewords ? s.row(g) : nullptr;
}

// one wave = one group of 64 codewords; `win` is the wave's private window
template <class Src>
__device__ __forceinline__ void lane_forward_fused_body(const Src &src, const int32_t *desc, const int32_t *tiles,
                                                        int nsteps, int group, int n_codewords, uint2 *dec, uint8_t *win,
                                                        int lane) {
    constexpr int ROWS = 64 + Src::PRE;
    constexpr int NK = (ROWS * FCOLS + 63) / 64;
    const int cw0 = group * 64;
    uint2 *dst = dec + size_t(group) * nsteps * 64 + lane;
    // this lane's staging duties: chunk (row, col) = (i >> 3, i & 7), i = lane + 64 k
    const int8_t *rowp[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const int row = (lane + 64 * k) >> 3;
        rowp[k] = row < ROWS ? fused_row(src, cw0, row, n_codewords) : nullptr;
    }
    // rows that do not exist read as erasures for the whole codeword: zero them once, never stage them
    for (int i = lane; i < ROWS * (FPITCH / 4); i += 64) reinterpret_cast<uint32_t *>(win)[i] = 0u;
    const int r_eff = min(lane, n_codewords - 1 - cw0);       // lanes past the end repeat the last codeword
    const uint8_t *my = win + r_eff * FPITCH;

    uint4 R[NK];
    auto fetch = [&](int tile) {
        const int lo_al = tiles[2 * tile], nch = tiles[2 * tile + 1];
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const int c = (lane + 64 * k) & 7;
            if (rowp[k] && c < nch) R[k] = *reinterpret_cast<const uint4 *>(rowp[k] + lo_al + 16 * c);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const int i = lane + 64 * k;
            if (rowp[k]) {                                     // (columns past the tile's span hold stale bytes nobody reads)
                uint32_t *d = reinterpret_cast<uint32_t *>(win + (i >> 3) * FPITCH + 16 * (i & 7));
                d[0] = R[k].x; d[1] = R[k].y; d[2] = R[k].z; d[3] = R[k].w;
            }
        }
    };
    unsigned M[32];
#pragma unroll
    for (int r = 0; r < 32; r++) M[r] = pack16(-LANE_INIT2, -LANE_INIT2);
    M[0] = pack16(0, -LANE_INIT2);
    // The window is wave-private and one wave's DS operations execute in order; these fences (no instruction) keep the
    // compiler from moving window reads across the stores that refill it.
    auto window_fence = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    fetch(0);
    stash();
    window_fence();
    // Descriptors of the next six steps are fetched while the current six are worked on: with one wave per SIMD
    // nothing else would hide their latency.  They are deliberately NOT restrict-qualified: as scalar loads they
    // share the LDS counter (lgkmcnt) and every wait for them also drains the window reads (measured slower).
    const int dsh = desc_shift(src), dex = desc_extra(src);
    // where this lane's 24 soft bytes of the coming six steps sit in the window: descriptor -> offset -> address once, in
    // vector registers (two VALU per byte; as scalars they cost a v_readfirstlane, two SALU and a v_add per byte, and a
    // lone wave pays an issue turn for every one of them)
    const int8_t *mys = reinterpret_cast<const int8_t *>(my);
    const int8_t *pcur[24];
    int32_t dnxt[24];
    auto soft_address = [&](int32_t d) {
        asm volatile("" : "+v"(d));
        return mys + (int((unsigned(d) >> dsh) & 0xFFFFu) + dex);
    };
#pragma unroll
    for (int j = 0; j < 24; j++) pcur[j] = soft_address(desc[j]);
    int tile = 0;
    for (int t0 = 0; t0 < nsteps; t0 += FT, tile++) {
        const bool more = t0 + FT < nsteps;
        if (more) fetch(tile + 1);
        const int t_end = min(t0 + FT, nsteps);
        for (int t = t0; t < t_end; t += 6) {
            // The descriptors of the NEXT six steps: six 16-byte loads that stay here, in flight under the six steps below, and
            // stay in vector registers until the bottom of the iteration, where they become the next six steps' addresses.  (Left to itself the compiler turns each load into
            // load + wait + v_readfirstlane through ONE register quad, six serial round trips per iteration: that cost more
            // than this round's instruction diet saved, profiles/r04_pmc_ofdm_summary.md.)
            const int32_t *dp = desc + 4 * min(t + 6, nsteps - 6);
#pragma unroll
            for (int j = 0; j < 24; j++) dnxt[j] = dp[j];
            __builtin_amdgcn_sched_barrier(0);
            // (signed byte reads straight into the four values a step works with: nothing is packed and unpacked again)
            Soft4 w[6];
#pragma unroll
            for (int i = 0; i < 6; i++) w[i] = Soft4{int(*pcur[4 * i]), int(*pcur[4 * i + 1]), int(*pcur[4 * i + 2]), int(*pcur[4 * i + 3])};
            lane_step<0>(M, w[0], dst + size_t(t + 0) * 64);
            lane_step<1>(M, w[1], dst + size_t(t + 1) * 64);
            lane_step<2>(M, w[2], dst + size_t(t + 2) * 64);
            lane_step<3>(M, w[3], dst + size_t(t + 3) * 64);
            lane_step<4>(M, w[4], dst + size_t(t + 4) * 64);
            lane_step<5>(M, w[5], dst + size_t(t + 5) * 64);
            if (((t / 6) & 1) == 1) {                          // every 12 steps (two phase cycles)
                const unsigned ref = pack16(int(M[0]), int(M[0])) & 0xFFFEFFFEu;
#pragma unroll
                for (int r = 0; r < 32; r++) M[r] = pk_sub(M[r], ref);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 24; j++) pcur[j] = soft_address(dnxt[j]);   // now they are wanted
        }
        if (more) {                                            // the wave's own LDS reads above are already issued
            window_fence();
            stash();
            window_fence();
        }
    }
}

template <class Src>
__global__ LANE_FWD_ATTR __launch_bounds__(256) void lane_forward_fused_kernel(Src src, const int32_t *desc, const int32_t *tiles,
                                                                 int nsteps, int groups, int n_codewords, uint2 *dec) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fused_lds[];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int group = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);     // wave-uniform: pointers stay in SGPRs
    if (group >= groups) return;
    lane_forward_fused_body(src, desc, tiles, nsteps, group, n_codewords, dec, fused_lds + wv * ((64 + Src::PRE) * FPITCH),
                            lane);
}

// Grouped launch (every sub-channel of a multiplex in one launch): a wave looks up which
// sub-channel its group belongs to and runs the same body with that entry's parameters.  Codeword lengths differ
// between entries; nothing else does.  The entry table travels BY VALUE in the kernel arguments: the lookup then
// compiles to scalar loads, every field stays in SGPRs and the pointers are known to be global memory (as a table
// in device memory they were generic pointers in VGPRs: flat loads/stores that also tick the LDS counter, and a
// quarter-rate v_mul_lo_u32 per soft byte for the window address).
constexpr int LANE_GROUP_MAX = 16;        // entries per launch (16 x 112 B of kernel arguments); longer lists are chunked
struct LaneEntry {
    LSrcMsc src;
    const int32_t *desc, *tiles;
    const uint8_t *prbs;
    uint8_t *out;
    uint8_t *crc_ok;          // FIC entry: CRC flag per FIB; nullptr for sub-channels
    uint2 *dec;
    int nsteps, n_codewords, first_group, groups;
};
struct LaneEntryPack {
    int n;
    int total_groups;
    int prio_nsteps;          // waves of entries at least this long get the issue slots first (0: nobody); see the forward kernel
    LaneEntry e[LANE_GROUP_MAX];
};

__device__ __forceinline__ int find_entry(const LaneEntryPack &pack, int group) {
    int e = 0;
    while (e + 1 < pack.n && group >= pack.e[e + 1].first_group) e++;
    return e;
}

__global__ LANE_FWD_ATTR __launch_bounds__(256) void lane_forward_grouped_kernel(const LaneEntryPack pack) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fused_lds[];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int group = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
    if (group >= pack.total_groups) return;
    const LaneEntry &en = pack.e[find_entry(pack, group)];
    const LSrcMsc src = en.src;
    // Codeword lengths differ between entries (FIC 774 steps, a 64 kbit/s sub-channel 1542).  When the whole launch is
    // resident at once -- two waves per SIMD, one of each -- the longer wave would run alone, at a single wave's issue rate,
    // for the last third of the launch.  The longest entry's waves therefore get the issue slots first; the shorter ones
    // fill in and are still around when the long ones end: 1.17 -> 1.10 ms for the bench's pair of launches
    // (profiles/r04_lane_forward.txt).  Launches with more waves than the chip holds balance themselves by dispatch order, and
    // there a priority only hurt (whole multiplex +2.7 %): the launcher sets prio_nsteps to 0 for them.
    if (pack.prio_nsteps > 0 && en.nsteps >= pack.prio_nsteps) __builtin_amdgcn_s_setprio(1);
    lane_forward_fused_body(src, en.desc, en.tiles, en.nsteps, group - en.first_group, en.n_codewords, en.dec,
                            fused_lds + wv * ((64 + LSrcMsc::PRE) * FPITCH), lane);
}

// ---------------------------------------------------------------------------------------------------------
// K3: traceback + output.  One wave per group, one lane per codeword, from slot 0 (end state 0 sits in slot 0
// in every layout).  The survivor tag of slot p sits where take_tags() put it, set when the older-bit-0
// predecessor survived; the bit the step replaces in the slot index is the decoded input bit.
// Decoded words go to an LDS tile [codeword][word]; the group's output bytes are one contiguous block, written
// with dword stores after energy dispersal; FIB CRCs are checked from the tile.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lane_tb_step(const uint2 d, int t, int nsteps, unsigned &p, int &q, unsigned &word,
                                             uint32_t *row) {
    // where the forward pass put the tag of slot p: registers were paired across register-index bit rb
    const unsigned rb = unsigned(q > 0 ? q - 1 : 0);
    const unsigned R = p >> 1;
    const unsigned I = ((R >> (rb + 1)) << rb) | (R & ((1u << rb) - 1u));
    const unsigned sel = (I & 8u) ? d.y : d.x;
    const unsigned h = (~sel >> (((((R >> rb) & 1u) << 1 | (p & 1u)) << 3) + (I & 7u))) & 1u;
    const unsigned b = (p >> q) & 1u;
    if (t < nsteps - 6) {
        word |= b << (31 - (t & 31));
        if ((t & 31) == 0) {
            row[t >> 5] = word;
            word = 0;
        }
    }
    p = (p & ~(1u << q)) | (h << q);
    q = (q == 5) ? 0 : q + 1;                                // q(t-1) = (5 - (t-1)) mod 6
}

// The same step with everything that does not depend on the data known at compile time.  In the main loop the
// replaced slot bit q and the output bit position follow a fixed pattern: blocks of 32 steps start where t & 31 == 31
// (nsteps - 6 is a multiple of 32), so step i of a block writes bit i of the block's word, and q advances by one per
// step from 0, 2, 4 in the three blocks of a 96-step round (nsteps - 6 is a multiple of 96 as well).  No scalar
// compares, no branches, shifts by constants: about half the instructions of the generic step on the serial chain.
template <int Q, int BIT>
__device__ __forceinline__ void lane_tb_step_static(const uint2 d, unsigned &p, unsigned &word) {
    constexpr unsigned rb = unsigned(Q > 0 ? Q - 1 : 0);
    const unsigned R = p >> 1;
    const unsigned I = ((R >> (rb + 1)) << rb) | (R & ((1u << rb) - 1u));
    const unsigned sel = (I & 8u) ? d.y : d.x;
    const unsigned h = (~sel >> (((((R >> rb) & 1u) << 1 | (p & 1u)) << 3) + (I & 7u))) & 1u;
    word |= ((p >> Q) & 1u) << BIT;
    p = (p & ~(1u << Q)) | (h << Q);
}

template <int Q0, int I>
__device__ __forceinline__ void lane_tb_block_steps(const uint2 (&d)[32], unsigned &p, unsigned &word) {
    if constexpr (I < 32) {
        lane_tb_step_static<(Q0 + I) % 6, I>(d[I], p, word);
        lane_tb_block_steps<Q0, I + 1>(d, p, word);
    }
}

// does the traceback's output tile ([64][words | 1] dwords) fit into LDS?
__host__ __device__ __forceinline__ bool lane_tile_fits(int nsteps) {
    return size_t(64) * size_t(((nsteps - 6) >> 5) | 1) * 4 <= size_t(150) * 1024;
}

__device__ __forceinline__ unsigned crc16_byte_l(unsigned crc, unsigned byte) {
    crc = ((crc >> 8) | (crc << 8)) & 0xFFFFu;
    crc ^= byte;
    crc ^= (crc & 0xFFu) >> 4;
    crc ^= (crc << 12) & 0xFFFFu;
    crc ^= ((crc & 0xFFu) << 5) & 0xFFFFu;
    return crc;
}

// tile == nullptr: codewords too long for a [64][words] tile in LDS (above ~800 kbit/s): every decoded word goes
// straight to its place in the output (one 4-byte store per lane per 32 steps); never the FIC.
__device__ __forceinline__ void lane_traceback_body(const uint2 *dec, int nsteps, int n_codewords, int group,
                                                    const uint8_t *prbs_bytes, uint8_t *out, uint8_t *crc_ok, uint32_t *tile,
                                                    int lane) {
    const uint2 *src = dec + size_t(group) * nsteps * 64 + lane;
    const int nwords = (nsteps - 6) >> 5;
    const int pitch = nwords | 1;                             // odd -> rows start in different banks
    const bool direct = tile == nullptr;
    const int nvalid_d = min(64, n_codewords - group * 64);
    const uint32_t *prbs32_d = reinterpret_cast<const uint32_t *>(prbs_bytes);
    uint32_t *out_d = reinterpret_cast<uint32_t *>(out + (size_t(group) * 64 + lane) * nwords * 4);
    uint32_t *row = direct ? nullptr : tile + lane * pitch;
    unsigned p = 0, word = 0;
    int q = 5 - ((nsteps - 1) % 6);
    // The survivor words do not depend on the path: they are fetched a block of 32 steps ahead, only the slot
    // update is serial.
    uint2 d[32], dn[32];
#pragma unroll
    for (int i = 0; i < 6; i++) d[i] = ld_stream(src + size_t(nsteps - 1 - i) * 64);
#pragma unroll
    for (int i = 0; i < 32; i++) dn[i] = ld_stream(src + size_t(nsteps - 7 - i) * 64);
#pragma unroll
    for (int i = 0; i < 6; i++) lane_tb_step(d[i], nsteps - 1 - i, nsteps, p, q, word, row);
    // (after the six tail steps q is back at 0 and word is empty)
    auto block = [&](auto q0, int t1) {                         // steps t1 .. t1 - 31 -> word t1 >> 5
#pragma unroll
        for (int i = 0; i < 32; i++) d[i] = dn[i];
        if (t1 >= 32) {
#pragma unroll
            for (int i = 0; i < 32; i++) dn[i] = ld_stream(src + size_t(t1 - 32 - i) * 64);
        }
        unsigned w = 0;
        lane_tb_block_steps<decltype(q0)::value, 0>(d, p, w);
        if (!direct) {
            row[t1 >> 5] = w;
        } else if (lane < nvalid_d) {
            uint32_t v = __builtin_bswap32(w);
            if (prbs32_d) v ^= prbs32_d[t1 >> 5];
            out_d[t1 >> 5] = v;
        }
    };
    for (int t1 = nsteps - 7; t1 >= 0; t1 -= 96) {            // nsteps - 6 is a multiple of 96
        block(std::integral_constant<int, 0>{}, t1);
        block(std::integral_constant<int, 2>{}, t1 - 32);
        block(std::integral_constant<int, 4>{}, t1 - 64);
    }
    if (direct) return;                                       // (wave-uniform: every lane of the workgroup leaves)
    __syncthreads();
    // output: codewords of a group are adjacent in the output, so the tile is one contiguous run of dwords
    const int nvalid = min(64, n_codewords - group * 64);
    const uint32_t *prbs32 = reinterpret_cast<const uint32_t *>(prbs_bytes);
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out + size_t(group) * 64 * nwords * 4);
    for (int i = lane; i < nvalid * nwords; i += 64) {
        const int r = i / nwords, w = i - r * nwords;
        uint32_t v = __builtin_bswap32(tile[r * pitch + w]);  // MSB-first bytes in memory order
        if (prbs32) v ^= prbs32[w];
        out32[i] = v;
        if (crc_ok) tile[r * pitch + w] = v;                  // descrambled, memory byte order, for the CRC
    }
    if (crc_ok) {                                             // FIC: three 32-byte FIBs per codeword
        __syncthreads();
        const int nfib = nwords >> 3;
        for (int j = lane; j < nvalid * nfib; j += 64) {
            const int r = j / nfib, fi = j - r * nfib;
            const uint32_t *wp = tile + r * pitch + 8 * fi;
            unsigned crc = 0xFFFFu;
#pragma unroll
            for (int k = 0; k < 30; k++) crc = crc16_byte_l(crc, (wp[k >> 2] >> (8 * (k & 3))) & 0xFFu);
            const unsigned rx = ((wp[7] >> 16) & 0xFFu) << 8 | (wp[7] >> 24);
            crc_ok[(size_t(group) * 64) * nfib + j] = uint8_t((crc ^ 0xFFFFu) == rx);
        }
    }
}

__global__ __launch_bounds__(64) void lane_traceback_kernel(const uint2 *dec, int nsteps, int n_codewords,
                                                            const uint8_t *prbs_bytes, uint8_t *out, uint8_t *crc_ok) {
    extern __shared__ uint32_t tile[];                        // [64][nwords + 1], or nothing for over-long codewords
    lane_traceback_body(dec, nsteps, n_codewords, blockIdx.x, prbs_bytes, out, crc_ok, lane_tile_fits(nsteps) ? tile : nullptr,
                        threadIdx.x);
}

__global__ __launch_bounds__(64) void lane_traceback_grouped_kernel(const LaneEntryPack pack) {
    extern __shared__ uint32_t tile[];                        // sized for the longest entry
    const int group = blockIdx.x;
    const LaneEntry &en = pack.e[find_entry(pack, group)];
    lane_traceback_body(en.dec, en.nsteps, en.n_codewords, group - en.first_group, en.prbs, en.out, en.crc_ok,
                        lane_tile_fits(en.nsteps) ? tile : nullptr, threadIdx.x);
}

template <class Src>
hipError_t run_lane(Src f, bool vec16, bool fusable, const CodeTables &c, const LaneTables &lt, int n_codewords,
                    const LaneScratch &sc, uint8_t *out, uint8_t *crc_ok, hipStream_t s) {
    const int groups = (n_codewords + 63) / 64;
    const int nwords = (c.nsteps - 6) >> 5;
    const size_t need = lane_scratch_bytes(c.nsteps, n_codewords);
    if (!sc.base || sc.bytes < need || (reinterpret_cast<uintptr_t>(out) & 3)) return hipErrorInvalidValue;
    uint32_t *M = reinterpret_cast<uint32_t *>(sc.base);
    uint2 *dec = reinterpret_cast<uint2 *>(M + size_t(groups) * c.nsteps * 64);
    const unsigned fgrid = unsigned((groups + 3) / 4);
    if (fusable && vec16 && lt.fused_desc && lt.fused_tiles && !sc.unfused) {
        const size_t lds = balanced_lds_bytes(fgrid, size_t(4) * (64 + Src::PRE) * FPITCH, 3);
        hipLaunchKernelGGL((lane_forward_fused_kernel<Src>), dim3(fgrid), dim3(256), lds, s, f, lt.fused_desc, lt.fused_tiles,
                           c.nsteps, groups, n_codewords, dec);
    } else {
        hipLaunchKernelGGL((lane_prep_kernel<Src>), dim3(unsigned((c.nsteps + PREP_STEPS - 1) / PREP_STEPS), unsigned(groups)),
                           dim3(256), 0, s, f, lt.punct_idx, c.nsteps, n_codewords, int(vec16), M);
        const size_t fwd_lds = balanced_lds_bytes(fgrid, 0, 8);
        hipLaunchKernelGGL(lane_forward_kernel, dim3(fgrid), dim3(256), fwd_lds, s, M, c.nsteps, groups, dec);
    }
    const size_t tb_lds = lane_tile_fits(c.nsteps) ? size_t(64) * (nwords | 1) * 4 : 0;
    hipLaunchKernelGGL(lane_traceback_kernel, dim3(unsigned(groups)), dim3(64), tb_lds, s, dec, c.nsteps, n_codewords,
                       c.prbs_bytes, out, crc_ok);
    return hipGetLastError();
}

inline bool aligned16(const void *p, size_t stride) { return ((reinterpret_cast<uintptr_t>(p) | stride) & 15) == 0; }

}  // namespace

hipError_t init_lane_kernel_attributes() {
    for (const void *k : {reinterpret_cast<const void *>(lane_forward_kernel),
                          reinterpret_cast<const void *>(lane_forward_fused_kernel<LSrcFic>),
                          reinterpret_cast<const void *>(lane_forward_fused_kernel<LSrcPlain>),
                          reinterpret_cast<const void *>(lane_forward_fused_kernel<LSrcMsc>),
                          reinterpret_cast<const void *>(lane_forward_grouped_kernel),
                          reinterpret_cast<const void *>(lane_traceback_kernel),
                          reinterpret_cast<const void *>(lane_traceback_grouped_kernel)}) {
        const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

size_t lane_scratch_bytes(int nsteps, int n_codewords) {
    const size_t groups = size_t((n_codewords + 63) / 64);
    const size_t nwords = size_t((nsteps - 6) >> 5);
    (void)nwords;
    return groups * 64 * size_t(nsteps) * 12 + 256;           // soft dwords + survivor words
}

bool lane_supported(int nsteps) {
    // whole phase cycles and whole 32-bit output words (any length: a codeword whose output tile does not fit into LDS
    // writes its words directly, lane_traceback_body)
    return nsteps >= 38 && nsteps % 6 == 0 && ((nsteps - 6) & 31) == 0;
}

hipError_t launch_fic_decode_lane(const CodeTables &c, const LaneTables &lt, const int8_t *soft, size_t soft_stride,
                                  int n_frames, const LaneScratch &sc, uint8_t *fib, uint8_t *crc_ok, hipStream_t s) {
    return run_lane(LSrcFic{soft, soft_stride}, aligned16(soft, soft_stride), true, c, lt, n_frames * NB_FIC_GROUPS, sc,
                    fib, crc_ok, s);
}

hipError_t launch_viterbi_plain_lane(const CodeTables &c, const LaneTables &lt, const int8_t *punct,
                                     int n_codewords, const LaneScratch &sc, uint8_t *out, hipStream_t s) {
    return run_lane(LSrcPlain{punct, c.n_punct}, aligned16(punct, size_t(c.n_punct)), true, c, lt, n_codewords, sc, out,
                    nullptr, s);
}

hipError_t launch_msc_decode_lane(const CodeTables &c, const LaneTables &lt, const MscArgs &a, const LaneScratch &sc,
                                  hipStream_t s) {
    const LSrcMsc f = make_msc_src(a);
    // the fused forward pass wants whole groups inside one stream and 16-byte aligned history rows
    const bool fusable = (a.frames_per_stream * NB_CIFS) % 64 == 0 &&
                         (!a.hist_in || ((reinterpret_cast<uintptr_t>(a.hist_in) | size_t(a.nbits)) & 15) == 0);
    return run_lane(f, aligned16(a.soft, a.soft_stride) && (a.start_bit & 15) == 0, fusable, c, lt,
                    a.n_streams * a.frames_per_stream * NB_CIFS, sc, a.out, nullptr, s);
}

bool lane_group_fusable(const MscArgs &a) {
    return (a.frames_per_stream * NB_CIFS) % 64 == 0 && aligned16(a.soft, a.soft_stride) && (a.start_bit & 15) == 0 &&
           (!a.hist_in || ((reinterpret_cast<uintptr_t>(a.hist_in) | size_t(a.nbits)) & 15) == 0) &&
           (reinterpret_cast<uintptr_t>(a.out) & 3) == 0;
}

static size_t item_codewords(const LaneGroupItem &it) {
    return size_t(it.args.n_streams) * it.args.frames_per_stream * NB_CIFS;      // FIC: 4 groups per frame as well
}

size_t lane_group_scratch_bytes(const LaneGroupItem *items, int n) {
    size_t total = 0;
    for (int i = 0; i < n; i++) total += ((item_codewords(items[i]) + 63) / 64) * 64 * size_t(items[i].code.nsteps) * sizeof(uint2);
    return total + 512;
}

static int resident_cus() {
    static const int n = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
    }();
    return n;
}

hipError_t launch_lane_group(const LaneGroupItem *items, int n, const LaneScratch &sc, hipStream_t s, hipEvent_t *mid) {
    if (n <= 0) return hipSuccess;
    if (!sc.base || sc.bytes < lane_group_scratch_bytes(items, n)) return hipErrorInvalidValue;
    char *p = static_cast<char *>(sc.base);
    for (int i0 = 0; i0 < n; i0 += LANE_GROUP_MAX) {
        LaneEntryPack pack{};
        pack.n = std::min(LANE_GROUP_MAX, n - i0);
        int max_nwords = 0;
        for (int i = 0; i < pack.n; i++) {
            const LaneGroupItem &it = items[i0 + i];
            const MscArgs &a = it.args;
            if (!it.tables.fused_desc || !it.tables.fused_tiles || !lane_supported(it.code.nsteps)) return hipErrorInvalidValue;
            LaneEntry &e = pack.e[i];
            e.n_codewords = int(item_codewords(it));
            if (it.is_fic) {
                // the FIC as one more entry: four 2304-bit groups per frame, no interleaving (every bit "delay 15" =
                // the codeword's own row), never a history row
                if (!aligned16(a.soft, a.soft_stride) || (reinterpret_cast<uintptr_t>(a.out) & 3)) return hipErrorInvalidValue;
                e.src = LSrcMsc{a.soft, a.soft_stride, nullptr, e.n_codewords + 128, 0, NB_FIC_GROUP_BITS, NB_FIC_GROUP_BITS, 15};
            } else {
                if (!lane_group_fusable(a)) return hipErrorInvalidValue;
                e.src = make_msc_src(a);
            }
            e.desc = it.tables.fused_desc;
            e.tiles = it.tables.fused_tiles;
            e.prbs = it.code.prbs_bytes;
            e.out = a.out;
            e.crc_ok = it.is_fic ? it.crc_ok : nullptr;
            e.dec = reinterpret_cast<uint2 *>(p);
            e.nsteps = it.code.nsteps;
            e.first_group = pack.total_groups;
            e.groups = (e.n_codewords + 63) / 64;
            pack.total_groups += e.groups;
            p += size_t(e.groups) * 64 * size_t(e.nsteps) * sizeof(uint2);
            if (lane_tile_fits(e.nsteps)) max_nwords = std::max(max_nwords, (e.nsteps - 6) >> 5);
        }
        const unsigned fgrid = unsigned((pack.total_groups + 3) / 4);
        const size_t lds = balanced_lds_bytes(fgrid, size_t(4) * (64 + LSrcMsc::PRE) * FPITCH, 3);
        {   // every wave resident at once (<= 2 per SIMD) and entries of different lengths: the longest go first
            int longest = 0, shortest = 0x7fffffff;
            for (int i = 0; i < pack.n; i++) { longest = std::max(longest, pack.e[i].nsteps); shortest = std::min(shortest, pack.e[i].nsteps); }
            pack.prio_nsteps = (pack.total_groups <= 2 * 4 * resident_cus() && longest > shortest) ? longest : 0;
        }
        const bool timed = mid && i0 + LANE_GROUP_MAX >= n;
        hipLaunchKernelGGL(lane_forward_grouped_kernel, dim3(fgrid), dim3(256), lds, s, pack);
        if (timed) (void)hipEventRecord(mid[0], s);
        const size_t tb_lds = size_t(64) * (max_nwords | 1) * 4;
        hipLaunchKernelGGL(lane_traceback_grouped_kernel, dim3(unsigned(pack.total_groups)), dim3(64), tb_lds, s, pack);
        if (timed) (void)hipEventRecord(mid[1], s);
    }
    return hipGetLastError();
}

void build_lane_fused_tables(const uint8_t *mask, int nsteps, std::vector<int32_t> &desc, std::vector<int32_t> &tiles) {
    std::vector<int> idx(size_t(4) * nsteps, -1);
    for (int p = 0, j = 0; p < 4 * nsteps; p++)
        if (mask[p]) idx[p] = j++;
    const int ntiles = (nsteps + FT - 1) / FT;
    desc.assign(size_t(4) * nsteps, FERASED | (FERASED << 16));
    tiles.assign(size_t(2) * ntiles, 0);
    for (int tile = 0; tile < ntiles; tile++) {
        const int p0 = 4 * FT * tile, p1 = std::min(4 * nsteps, p0 + 4 * FT);
        int lo = -1, hi = 0;
        for (int p = p0; p < p1; p++)
            if (idx[p] >= 0) { if (lo < 0) lo = idx[p]; hi = idx[p] + 1; }
        if (lo < 0) continue;                                   // a tile of erasures only
        const int lo_al = lo & ~15;
        tiles[2 * tile] = lo_al;
        tiles[2 * tile + 1] = (hi - lo_al + 15) / 16;           // <= FCOLS by construction
        for (int p = p0; p < p1; p++) {
            if (idx[p] < 0) continue;
            unsigned i = unsigned(idx[p]) & 15u, d = ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3);
            const int col = idx[p] - lo_al;
            desc[p] = (int(d) * FPITCH + col) | (col << 16);
        }
    }
}

}  // namespace dabk
