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
#include <cstdlib>

#include "kernels.hpp"
#include "dab_tables.hpp"

namespace dabk {

using namespace dab;

namespace {

constexpr int WAVES_PER_WG = 4;
constexpr int WGV = 64 * WAVES_PER_WG;

__device__ __forceinline__ int parity32(unsigned x) { return __popc(x) & 1; }

// CRC-16/CCITT (x^16+x^12+x^5+1), one byte per call, table-free byte-wise form
__device__ __forceinline__ unsigned crc16_byte(unsigned crc, unsigned byte) {
    crc = ((crc >> 8) | (crc << 8)) & 0xFFFFu;
    crc ^= byte;
    crc ^= (crc & 0xFFu) >> 4;
    crc ^= (crc << 12) & 0xFFFFu;
    crc ^= ((crc & 0xFFu) << 5) & 0xFFFFu;
    return crc;
}

// ---- where a codeword's punctured soft bits come from -----------------------
struct FetchFic {
    static constexpr int kBatch = 1;     // loads the rot kernel keeps in flight per lane while depuncturing
    static constexpr bool kContiguous = true;   // a codeword's punctured bits lie side by side: 16 of them per load
    const int8_t *soft;
    size_t stride;
    __device__ __forceinline__ bool aligned16() const { return ((reinterpret_cast<uintptr_t>(soft) | stride) & 15) == 0; }
    __device__ __forceinline__ const int8_t *row(int cw) const { return soft + size_t(cw >> 2) * stride + size_t(cw & 3) * NB_FIC_GROUP_BITS; }
    __device__ __forceinline__ int8_t operator()(int cw, int i) const {
        return soft[size_t(cw >> 2) * stride + size_t(cw & 3) * NB_FIC_GROUP_BITS + i];
    }
};
struct FetchPlain {
    static constexpr int kBatch = 1;
    static constexpr bool kContiguous = false;
    const int8_t *punct;
    int n_punct;
    __device__ __forceinline__ int8_t operator()(int cw, int i) const { return punct[size_t(cw) * n_punct + i]; }
};
// A12 time de-interleave: logical frame completed by CIF t takes bit i from CIF
// t - 15 + d(i % 16); CIFs before the call come from the history ring.
struct FetchMsc {
    static constexpr int kBatch = 8;
    static constexpr bool kContiguous = false;
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
    // the same as an address, branch-free (for loads that are issued long before their values are used): `erased` = there
    // is no such byte (a CIF before the call and no history), the address is then just a valid one
    __device__ __forceinline__ const int8_t *addr(int cw, int i, bool &erased) const {
        const int cifs = frames_per_stream * NB_CIFS;
        const int s = cw / cifs, t = cw - s * cifs;
        const int src = t - 15 + int(__brev(unsigned(i) & 15u) >> 28);
        const size_t f = size_t(s) * frames_per_stream + size_t(max(src, 0) >> 2);
        const int8_t *from_soft = soft + f * stride + NB_FIC_BITS + size_t(max(src, 0) & 3) * NB_CIF_BITS + start_bit + i;
        const int8_t *from_hist = hist ? hist + (size_t(s) * 15 + size_t(15 + min(src, -1))) * nbits + i : from_soft;
        erased = src < 0 && !hist;
        return src >= 0 ? from_soft : from_hist;
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
    const int cw_raw = __builtin_amdgcn_readfirstlane(blockIdx.x * int(blockDim.x >> 6) + wave);   // wave-uniform
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
            for (int i = 0; i < 30; i++) crc = crc16_byte(crc, p[i]);
            crc ^= 0xFFFFu;
            if (active) crc_ok[size_t(cw) * 3 + lane] = uint8_t(crc == ((unsigned(p[30]) << 8) | p[31]));
        }
    }
}


// ============================================================================
// Kernel 2 ("rot" variant): same decoder, restructured around the dependency chain.
//
// Lane l holds trellis state rotl6(l, t mod 6) at step t.  With that layout the two
// predecessors of the state a lane will hold next are the lane itself and the lane
// differing in ONE bit position q = (5 - t) mod 6, so the ACS butterfly is a fixed
// XOR exchange: DPP quad_perm / row_half_mirror / row_ror for 1,2,4,8 and
// v_permlane16/32_swap for 16,32 -- all VALU latency, no LDS crossbar.  Branch
// metrics are one v_dot4_i32_i8 of a per-lane sign vector with the step's soft word
// (an LDS broadcast read, one phase cycle ahead), computed off the chain
// metric -> exchange -> subtract -> max.  Each lane shifts its own survivor bit
// into a register straight from VCC; every 32 steps the 64 words are transposed
// across the wave into one 8-byte LDS slot per step; the traceback runs on all lanes
// at once, every lane its own segment of steps in the lane domain (the state update
// is "flip bit q of the lane index"), and checks itself: see rot_decode.
// Requires nsteps = 96k + 6, which every DAB codeword satisfies (FIC 768+6,
// EEP 192n+6); other lengths use kernel 1.
// ============================================================================
template <int XORMASK>
__device__ __forceinline__ int lane_xchg(int m, int lane) {
    if constexpr (XORMASK == 1) {
        return __builtin_amdgcn_update_dpp(0, m, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    } else if constexpr (XORMASK == 2) {
        return __builtin_amdgcn_update_dpp(0, m, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    } else if constexpr (XORMASK == 4) {
        const int h = __builtin_amdgcn_update_dpp(0, m, 0x141, 0xF, 0xF, false);   // row_half_mirror: i ^ 7
        return __builtin_amdgcn_update_dpp(0, h, 0x1B, 0xF, 0xF, false);           // quad_perm [3,2,1,0]: i ^ 3
    } else if constexpr (XORMASK == 8) {
        return __builtin_amdgcn_update_dpp(0, m, 0x128, 0xF, 0xF, false);  // row_ror:8
    } else if constexpr (XORMASK == 16) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const u2 r = __builtin_amdgcn_permlane16_swap(unsigned(m), unsigned(m), false, false);
        return (lane & 16) ? int(r.x) : int(r.y);
    } else {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const u2 r = __builtin_amdgcn_permlane32_swap(unsigned(m), unsigned(m), false, false);
        return (lane & 32) ? int(r.x) : int(r.y);
    }
}

struct RotTables {
    int tab_cs[6];    // int8x4 signs giving  sigma * c  (sigma = -1 where the lane itself is the older-bit-1 predecessor)
    int thr[6];       // 0 where bit q of the lane is 0, -1 where it is 1 (strict-compare threshold, see rot_step)
};

__device__ __forceinline__ int pack_i8x4(int a, int b, int c, int d) {
    return (a & 0xFF) | ((b & 0xFF) << 8) | ((c & 0xFF) << 16) | ((d & 0xFF) << 24);
}

// one trellis step at phase PH (= t mod 6); c = sigma * (the step's branch metric), ct = c + the lane's threshold -- both
// depend on the soft bits alone and are computed ahead of the chain metric -> exchange -> subtract -> max
//   x = M_self + c, y = M_other - c are the two candidates of the state this lane holds next.
//   Lanes with bit q clear are the older-bit-0 predecessor themselves: survivor bit = (y > x).
//   Lanes with bit q set are the older-bit-1 predecessor:               survivor bit = (x > y) = !(y > x - 1).
//   So ONE compare against x + threshold (0 / -1) gives r = survivor bit XOR (bit q of the lane) for both: r says whether
//   the traceback FLIPS bit q of the lane index when it goes back over this step.  r is shifted into the lane's own
//   32-step word straight from VCC (v_cmp + v_addc_co: 2.5 cycles on top of the compare for a wave that runs alone;
//   bringing the step's 64 bits together at once -- v_cmp to an SGPR pair, s_xor, two v_writelane -- cost 13 to 27:
//   tools/ubench/wave_step_cycles.hip, profiles/r04_wave_step_cycles.txt); the words are transposed 32 steps at a time.
template <int PH>
__device__ __forceinline__ void rot_step(int lane, int c, int ct, int &metric, unsigned &dec) {
    constexpr int Q = 5 - PH;
    const int other = lane_xchg<(1 << Q)>(metric, lane);
    const int x = metric + c, xt = metric + ct;
    const int y = other - c;
    metric = max(x, y);
    // (one statement, e32 forms with VCC implicit: the pair the compiler itself emits back to back)
    asm("v_cmp_gt_i32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(dec) : "v"(y), "v"(xt) : "vcc");
}

// 32 x 32 bit transpose across each half of the wave: lane a of a half holds row a, bit c = element (a, c); afterwards
// lane b holds the column b, bit a = element (a, b).  Five rounds of the block-swap transpose (block sizes 16 .. 1): a
// lane exchanges with the lane 16 / 8 / .. / 1 away (the ACS's own XOR exchanges), rotates what it got by the block
// size and merges under the round's column mask: three to six instructions a round, once per 32 trellis steps.
template <int J>
__device__ __forceinline__ unsigned transpose_round(unsigned x, int lane) {
    constexpr unsigned M0 = J == 16 ? 0x0000FFFFu : J == 8 ? 0x00FF00FFu : J == 4 ? 0x0F0F0F0Fu : J == 2 ? 0x33333333u : 0x55555555u;
    const bool upper = (lane & J) != 0;                           // the row index has bit J set: it gives its low blocks away
    const unsigned p = unsigned(lane_xchg<J>(int(x), lane));
    const unsigned rot = __builtin_amdgcn_alignbit(p, p, upper ? J : 32 - J);   // upper: p >> J, lower: p << J (the rest is masked)
    const unsigned keep = upper ? ~M0 : M0;
    return (x & keep) | (rot & ~keep);
}
__device__ __forceinline__ unsigned transpose32(unsigned x, int lane) {
    x = transpose_round<16>(x, lane);
    x = transpose_round<8>(x, lane);
    x = transpose_round<4>(x, lane);
    x = transpose_round<2>(x, lane);
    return transpose_round<1>(x, lane);
}

// The per-wave LDS slab of the rot kernel (host and device agree through this one function).
//   survivors: one 8-byte slot per trellis step t (bit l = lane l's r of the step: flip bit q or not), at slot t + (t - 6) / seg_steps
//              for t >= 6 -- one slot of skew per traceback segment, so that the lanes of the traceback, each reading
//              its own segment, spread over the LDS banks (segments are 12..48 slots long: unskewed, 8- to 16-way conflicts)
//   mother:    the depunctured codeword, consumed from a region that starts half-way up the survivors' final extent: a
//              survivor row is only written after the codeword bytes it overlaps have been read (4 B/step consumed vs
//              8 B/step produced); the skew slots are added to the offset
//   out6:      the decoded bits, six per byte;  zero: six all-zero slots (what the traceback reads above the last step:
//              nothing flips)
struct RotLayout {
    int seg_cycles;       // six-step phase cycles per traceback lane
    int seg_steps;        // = 6 * seg_cycles
    int n_lanes;          // lanes that own a segment
    int mother_off, mother_bytes, out6_off, zero_off, total;
};
__host__ __device__ inline RotLayout rot_layout(int nsteps) {
    RotLayout L;
    const int groups = (nsteps - 6) / 6;
    L.seg_cycles = (groups + 63) / 64;
    L.seg_steps = 6 * L.seg_cycles;
    L.n_lanes = (groups + L.seg_cycles - 1) / L.seg_cycles;
    const int skew_slots = (nsteps - 6) / L.seg_steps + 2;
    L.mother_off = 128 * ((nsteps - 6) / 32 + 1) + ((8 * skew_slots + 255) & ~255);
    L.mother_bytes = (4 * nsteps + 255) & ~255;
    L.out6_off = L.mother_off + L.mother_bytes;
    L.zero_off = L.out6_off + ((groups + 255) & ~255);
    L.total = L.zero_off + 256;                                   // (the survivors end below out6_off)
    return L;
}
constexpr int ROT_WARM_CYCLES = 16;                                // 96 steps run in before a lane's own segment

// One wavefront decodes codeword `cw` in its LDS slab (every wave of the workgroup must call it: it contains
// workgroup barriers, none of them inside a loop whose trip count depends on the codeword).
template <class Fetch, Tail TAIL>
__device__ __forceinline__ void rot_decode(const Fetch &fetch, const CodeTables &code, int cw, bool active, uint8_t *out,
                                           uint8_t *crc_ok, unsigned char *slab, int lane) {
    const int nsteps = code.nsteps;
    const int nchunks = (nsteps - 6) / 96;

    const RotLayout L = rot_layout(nsteps);
    const int mother_bytes = L.mother_bytes;
    int8_t *mother = reinterpret_cast<int8_t *>(slab + L.mother_off);
    int *m4 = reinterpret_cast<int *>(mother);
    uint8_t *out6 = slab + L.out6_off;                                       // [16*nchunks] six-bit groups

    // ---- A8: depuncture into LDS ----
    for (int i = lane; i < mother_bytes / 4; i += 64) m4[i] = 0;
    if (lane < 16) reinterpret_cast<int *>(slab + L.zero_off)[lane] = 0;
    __syncthreads();
    bool wide = false;
    if constexpr (Fetch::kContiguous) {
        wide = fetch.aligned16() && (code.n_punct & 15) == 0;
        if (wide) {
            // sixteen bits and their sixteen positions per lane and round trip (the FIC: 144 such pieces)
            const uint4 *src = reinterpret_cast<const uint4 *>(fetch.row(cw));
            const uint4 *ptab = reinterpret_cast<const uint4 *>(code.mother_pos);
            const int pieces = code.n_punct >> 4;
            for (int p0 = lane; p0 < pieces; p0 += 64) {
                const uint4 v = src[p0], pa = ptab[2 * p0], pb = ptab[2 * p0 + 1];
                const unsigned vb[4] = {v.x, v.y, v.z, v.w};
                const unsigned pp[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
                for (int u = 0; u < 16; u++)
                    mother[(pp[u >> 1] >> (16 * (u & 1))) & 0xFFFFu] = int8_t(vb[u >> 2] >> (8 * (u & 3)));
            }
        }
    }
    // The sub-channel's gather -- every byte from one of sixteen CIFs, latency-bound: 13-20 us of an 80 us call when done
    // up front -- runs BESIDE the forward pass instead.  Lane l always takes the punctured bits l, l + 64, l + 128, ...:
    // their CIF (bit i comes from CIF t - 15 + bitrev4(i mod 16)) is the same for all of them, so a lane's source is ONE
    // base pointer for the whole codeword and a load costs an add.  Only the rows (of 64 bits) that cover the first two
    // 96-step chunks are fetched before the first step (CodeTables::chunk_first); at the top of every later chunk the
    // next RB rows leave (a chunk consumes at most six), at its bottom -- 2 us of add-compare-select later -- they are
    // scattered into the codeword in LDS, at least one chunk before the forward pass reads them (it fetches its soft
    // words one phase cycle ahead: chunk c's last cycle already touches chunk c + 1).
    constexpr bool kStream = Fetch::kBatch > 1;
    constexpr int RB = 7;
    const bool stream = kStream && !wide && code.chunk_first != nullptr;
    const int total_rows = (code.n_punct + 63) >> 6;
    const int first_rows = stream ? min(total_rows, (int(code.chunk_first[min(2, nchunks + 1)]) + 63) >> 6) : total_rows;
    const int gather_end = min(code.n_punct, first_rows << 6);
    if (!wide) {
        if constexpr (Fetch::kBatch > 1) {
            // the de-interleaver's bytes come from sixteen CIFs at once: eight of them (and their positions) in flight per lane
            for (int i0 = lane; i0 < gather_end; i0 += 64 * Fetch::kBatch) {
                int8_t v[Fetch::kBatch];
                uint16_t pos[Fetch::kBatch];
#pragma unroll
                for (int u = 0; u < Fetch::kBatch; u++) {
                    const int i = i0 + 64 * u;
                    if (i < gather_end) { v[u] = fetch(cw, i); pos[u] = code.mother_pos[i]; }
                }
#pragma unroll
                for (int u = 0; u < Fetch::kBatch; u++)
                    if (i0 + 64 * u < gather_end) mother[pos[u]] = v[u];
            }
        } else {
            for (int i = lane; i < code.n_punct; i += 64) mother[code.mother_pos[i]] = fetch(cw, i);
        }
    }
    __syncthreads();

    // ---- per-lane sign tables for the six layout phases ----
    RotTables T;
#pragma unroll
    for (int ph = 0; ph < 6; ph++) {
        const int rot = (ph + 1) % 6;                                       // state held AFTER the step
        const int n = ((lane << rot) | (lane >> (6 - rot))) & 63;
        const int s0 = parity32(n & 109) ? 1 : -1, s1 = parity32(n & 79) ? 1 : -1, s2 = parity32(n & 83) ? 1 : -1;
        const int qb = (lane >> (5 - ph)) & 1;
        const int sg = qb ? -1 : 1;
        T.thr[ph] = -qb;
        T.tab_cs[ph] = pack_i8x4(sg * s0, sg * s1, sg * s2, sg * s0);
    }

    // ---- A9 forward pass ----
    // Every lane collects its own r of 32 steps (bit 31 - k = step k of the row); a row is then transposed across the
    // wave -- lane b of either half gets the bits of step 31 - b, bit a = lane a of that half -- and goes to the
    // survivor slots: low half of a step's slot from lanes 0..31, high half from lanes 32..63.
    const unsigned seg_magic = unsigned((0x100000000ull + unsigned(L.seg_steps) - 1) / unsigned(L.seg_steps));
    auto flush = [&](int row_t0, unsigned row_bits) {
        const unsigned col = transpose32(row_bits, lane);
        const int t = row_t0 + 31 - (lane & 31);
        const int skew = t >= 6 ? int(__umulhi(unsigned(t - 6), seg_magic)) : 0;    // (t - 6) / seg_steps, exact below 2^16
        if (t < nsteps) *reinterpret_cast<unsigned *>(slab + 8 * (t + skew) + 4 * (lane >> 5)) = col;
    };
    int metric = (lane == 0) ? 0 : -VITERBI_INIT_PENALTY;
    unsigned dec = 0;
    // the step's four soft bits: one LDS word, the same address in every lane (a broadcast read), fetched one phase
    // cycle ahead of its use
    int wc[6];
#pragma unroll
    for (int ph = 0; ph < 6; ph++) wc[ph] = m4[ph];
    int8_t gv[RB];
    uint16_t gp[RB];
    int k_next = first_rows;                                   // first row of 64 punctured bits not fetched yet
    const int8_t *g_base = nullptr;                            // byte i of this lane's rows is g_base[i]
    bool g_erased = false;                                     // ... or nothing at all (a CIF before the call, no history)
    if constexpr (kStream) {
        if (stream) g_base = fetch.addr(cw, lane, g_erased) - lane;
    }
    for (int c = 0; c < nchunks; c++) {
        const int *mw = m4 + 96 * c;
        const bool more = stream && k_next < total_rows;       // (wave-uniform)
        if constexpr (kStream) {
            if (more) {
                // (straight-line: every lane loads from a valid address and decides at the bottom of the chunk what to
                // keep; a branch around a load would bring its s_waitcnt up here)
#pragma unroll
                for (int u = 0; u < RB; u++) {
                    const int ic = min(lane + ((k_next + u) << 6), code.n_punct - 1);
                    gv[u] = g_base[ic];
                    gp[u] = code.mother_pos[ic];
                }
                __builtin_amdgcn_sched_barrier(0);             // the loads leave before the chunk's first step, not after its last
            }
        }
#define DAB_ROT_METRIC(I, PH)                                                                    \
        constexpr int j##PH = 6 * (I) + (PH);                                                    \
        const int c##PH = __builtin_amdgcn_sdot4(T.tab_cs[PH], wc[PH], 0, false);                \
        const int ct##PH = c##PH + T.thr[PH];
#define DAB_ROT_STEP(PH)                                                                         \
    {                                                                                            \
        rot_step<PH>(lane, c##PH, ct##PH, metric, dec);                                          \
        if constexpr ((j##PH & 31) == 31) flush(c * 96 + j##PH - 31, dec);                       \
    }
#define DAB_ROT_CYCLE(I)                                                                         \
    {                                                                                            \
        int wn[6];                                                                               \
        _Pragma("unroll") for (int ph = 0; ph < 6; ph++) wn[ph] = mw[6 * ((I) + 1) + ph];        \
        DAB_ROT_METRIC(I, 0) DAB_ROT_METRIC(I, 1) DAB_ROT_METRIC(I, 2) DAB_ROT_METRIC(I, 3) DAB_ROT_METRIC(I, 4) DAB_ROT_METRIC(I, 5) \
        DAB_ROT_STEP(0) DAB_ROT_STEP(1) DAB_ROT_STEP(2) DAB_ROT_STEP(3) DAB_ROT_STEP(4) DAB_ROT_STEP(5) \
        _Pragma("unroll") for (int ph = 0; ph < 6; ph++) wc[ph] = wn[ph];                        \
    }
        DAB_ROT_CYCLE(0) DAB_ROT_CYCLE(1) DAB_ROT_CYCLE(2) DAB_ROT_CYCLE(3) DAB_ROT_CYCLE(4) DAB_ROT_CYCLE(5) DAB_ROT_CYCLE(6) DAB_ROT_CYCLE(7)
        DAB_ROT_CYCLE(8) DAB_ROT_CYCLE(9) DAB_ROT_CYCLE(10) DAB_ROT_CYCLE(11) DAB_ROT_CYCLE(12) DAB_ROT_CYCLE(13) DAB_ROT_CYCLE(14) DAB_ROT_CYCLE(15)
#undef DAB_ROT_CYCLE
#undef DAB_ROT_STEP
#undef DAB_ROT_METRIC
        if constexpr (kStream) {
            if (more) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < RB; u++)
                    if (!g_erased && lane + ((k_next + u) << 6) < code.n_punct) mother[gp[u]] = gv[u];
                k_next += RB;
                // (wave-private LDS: the wave's own DS operations execute in order; the fences pin the compiler's)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    {   // the six tail steps (zero tail bits): one more phase cycle (its words came with the last cycle above)
        dec = 0;
#define DAB_ROT_TAIL(PH)                                                                         \
    {                                                                                            \
        const int cb = __builtin_amdgcn_sdot4(T.tab_cs[PH], wc[PH], 0, false);                   \
        rot_step<PH>(lane, cb, cb + T.thr[PH], metric, dec);                                     \
    }
        DAB_ROT_TAIL(0) DAB_ROT_TAIL(1) DAB_ROT_TAIL(2) DAB_ROT_TAIL(3) DAB_ROT_TAIL(4) DAB_ROT_TAIL(5)
#undef DAB_ROT_TAIL
        flush(nchunks * 96, dec << 26);                        // (a row of six steps: only those below nsteps are written)
    }
    __syncthreads();

    // ---- traceback in the lane domain, all segments at once ----
    // End state 0 sits in lane 0 in every layout.  Going back over step t flips bit q_t of the lane index l exactly when
    // bit l of the step's slot says so (r, see rot_step): the bit then holds the survivor bit h_t, and h_t is the input
    // bit of step t-6 (it becomes the oldest bit of the earlier state).  After
    // the six steps of one phase cycle (q = 0,1,..,5 going backwards) the lane index therefore IS six consecutive
    // decoded bits, earliest in bit 5: out6[g] = the index after the cycle of steps [6g+6, 6g+12).
    // Lane c owns the cycles of steps [6 + S c, 6 + S (c+1)) (S = seg_steps).  It starts ROT_WARM_CYCLES cycles above
    // them from index 0 -- above the last step the slots read as zero, which keeps index 0 down to the true end state --
    // and has, with all but negligible probability, merged with the survivor path when it reaches its own segment.
    // That is then CHECKED, not assumed: the index a lane entered its segment with must be the one the lane above left
    // its segment with; while any pair disagrees, the lanes re-enter with their neighbour's exit index and trace their
    // segments again.  The top lane's entry is exact, so is, by induction, everything below once all pairs agree:
    // the bits are those of the serial traceback, on any input (at most n_lanes passes, one on ordinary ones).
    {
        const int S = L.seg_steps;
        const int t_above = 6 + S * (lane + 1);                   // first step above the lane's segment
        const int slot_above = t_above + (lane + 1);              // ... and its slot (skew = lane + 1 segments)
        int l = 0;
        // one phase cycle of steps [t_above + wi, +6), wi a multiple of 6 in [-S, 6 * ROT_WARM_CYCLES)
        auto cycle = [&](int wi, bool emit) {
            const int fl = wi >= 0 ? wi / S : -1;                  // floor(wi / S): wave-uniform
            const int tb = t_above + wi;
            const bool real = tb < nsteps;
            const unsigned long long *p = reinterpret_cast<const unsigned long long *>(slab + (real ? 8 * (slot_above + wi + fl) : L.zero_off));
            unsigned long long M[6];
#pragma unroll
            for (int ph = 0; ph < 6; ph++) M[ph] = p[ph];
#pragma unroll
            for (int ph = 5; ph >= 0; ph--) {
                const int q = 5 - ph;
                l ^= int((unsigned(M[ph] >> l) & 1u) << q);
            }
            if (emit && real) out6[(tb - 6) / 6] = uint8_t(l);
        };
        for (int i = ROT_WARM_CYCLES - 1; i >= 0; i--) cycle(6 * i, false);
        int entry = l;
        for (int i = 1; i <= L.seg_cycles; i++) cycle(-6 * i, true);
        const bool chained = lane < L.n_lanes - 1;
        int above = __shfl_down(l, 1);
        while (__ballot(chained && entry != above)) {
            if (chained) entry = above;
            l = entry;
            for (int i = 1; i <= L.seg_cycles; i++) cycle(-6 * i, true);
            above = __shfl_down(l, 1);
        }
    }
    __syncthreads();

    // ---- A10: bytes out (big-endian within each word) + energy dispersal ----
    const int nbytes = (nsteps - 6) >> 3;
    uint8_t *bytes = slab;                                                 // (the survivors are done with)
    uint8_t *o = out + size_t(cw) * nbytes;
    for (int k = lane; k < nbytes; k += 64) {
        // byte k = bits 8k..8k+7 = tail of six-bit group g and head of group g+1
        const int g = (8 * k) / 6, off = 8 * k - 6 * g;                      // off in {0, 2, 4}
        const unsigned two = (unsigned(out6[g]) << 6) | unsigned(out6[g + 1]);   // 12 bits (g+1 < 16*nchunks always)
        unsigned v = (two >> (4 - off)) & 0xFFu;
        if (code.prbs_bytes) v ^= code.prbs_bytes[k];
        bytes[k] = uint8_t(v);
        if (active) o[k] = uint8_t(v);
    }
    if (TAIL == Tail::kFic) {
        __syncthreads();
        if (lane < 3) {
            const uint8_t *p = bytes + 32 * lane;
            unsigned crc = 0xFFFFu;
            for (int i = 0; i < 30; i++) crc = crc16_byte(crc, p[i]);
            crc ^= 0xFFFFu;
            if (active) crc_ok[size_t(cw) * 3 + lane] = uint8_t(crc == ((unsigned(p[30]) << 8) | p[31]));
        }
    }
}

// Workgroup b of `total` runs on XCD b % 8; this gives it the index whose neighbours (index +- 1) run on the same XCD.
__device__ __forceinline__ int same_xcd(int b, int total) {
    const int per = (total + 7) >> 3, x = b & 7, r = b >> 3;
    const int full = total - (per - 1) * 8;               // XCDs that take `per` indices (the others per - 1)
    return x < full ? x * per + r : full * per + (x - full) * (per - 1) + r;
}

template <class Fetch, Tail TAIL>
__global__ __launch_bounds__(WGV) void viterbi_rot_kernel(Fetch fetch, CodeTables code, int n_codewords,
                                                          uint8_t *out, uint8_t *crc_ok, int lds_per_wave) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // consecutive workgroups' codewords to the same XCD (see viterbi_rot_grouped_kernel)
    const int wg = same_xcd(int(blockIdx.x), int(gridDim.x));
    const int cw_raw = __builtin_amdgcn_readfirstlane(wg * int(blockDim.x >> 6) + wave);           // wave-uniform
    const bool active = cw_raw < n_codewords;
    rot_decode<Fetch, TAIL>(fetch, code, active ? cw_raw : n_codewords - 1, active, out, crc_ok,
                            smem + size_t(wave) * lds_per_wave, lane);
}

// history ring update: hist_out[s][h] = CIF (4F - 15 + h), h = 0..14
__device__ __forceinline__ void msc_history_body(const MscArgs &a, size_t first, size_t step) {
    const int cifs = a.frames_per_stream * NB_CIFS;
    const size_t total = size_t(a.n_streams) * 15 * a.nbits;
    for (size_t idx = first; idx < total; idx += step) {
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
// Grouped launch for small batches (the plugin's one frame at a time): the codewords of several sub-channels, each
// with its own profile and length, in ONE launch -- one wavefront (= one workgroup) per codeword, the entry table by
// value in the kernel arguments.  A whole multiplex is then three launches (FIC, sub-channels, history rings)
// instead of one pair per sub-channel queueing up behind each other on the stream.
constexpr int WAVE_GROUP_MAX = 24;
struct WaveEntry {
    FetchMsc fetch;
    CodeTables code;
    uint8_t *out;
    int first_cw;
    int n_streams;                    // (with fetch: everything the entry's history ring update needs)
    int8_t *hist_out;
};
constexpr int HIST_BLOCKS = 32;       // workgroups per entry that write its history ring, behind the decoding ones
struct WaveEntryPack {
    int n;
    int n_dec;                        // workgroups that decode (FIC + sub-channel codewords); the rest update history rings
    int n_fic;                        // FIC codewords (4 per frame) decoded by the first n_fic workgroups, or 0
    FetchFic fic_fetch;
    CodeTables fic_code;
    uint8_t *fib, *crc_ok;
    WaveEntry e[WAVE_GROUP_MAX];
};
__global__ __launch_bounds__(64) void viterbi_rot_grouped_kernel(const WaveEntryPack pack) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // consecutive codewords to the same XCD (workgroup b runs on XCD b % 8): the sixteen CIFs a sub-channel codeword is
    // gathered from are the next codeword's too, and each XCD has an L2 of its own.  The FIC's (shorter) codewords and
    // the sub-channels' are spread separately, so that every XCD gets its share of both.
    int cw = blockIdx.x;
    if (cw >= pack.n_dec) {
        // The sub-channels' history rings (the time de-interleaver's state for the next call: soft bits and old rings in,
        // new rings out -- nothing the decoding workgroups write) ride in this launch instead of one of their own behind it
        // (4 us of the plugin's 70 us decode call): HIST_BLOCKS workgroups per entry.
        const int hb = cw - pack.n_dec, k = hb / HIST_BLOCKS;
        const WaveEntry &en = pack.e[k];
        if (en.hist_out) {
            const MscArgs a{en.fetch.soft, en.fetch.stride, en.n_streams, en.fetch.frames_per_stream, en.fetch.start_bit, en.fetch.nbits,
                            en.fetch.hist, en.hist_out, en.out};
            msc_history_body(a, size_t(hb - k * HIST_BLOCKS) * 64 + threadIdx.x, size_t(HIST_BLOCKS) * 64);
        }
        return;
    }
    if (cw < pack.n_fic) {            // one frame at a time the FIC's four codewords ride along with the sub-channels'
        rot_decode<FetchFic, Tail::kFic>(pack.fic_fetch, pack.fic_code, same_xcd(cw, pack.n_fic), true, pack.fib, pack.crc_ok, smem,
                                         int(threadIdx.x));
        return;
    }
    cw = same_xcd(cw - pack.n_fic, pack.n_dec - pack.n_fic);
    int k = 0;
    while (k + 1 < pack.n && cw >= pack.e[k + 1].first_cw) k++;
    const WaveEntry &en = pack.e[k];
    rot_decode<FetchMsc, Tail::kBytes>(en.fetch, en.code, cw - en.first_cw, true, en.out, nullptr, smem, int(threadIdx.x));
}

struct HistoryPack {
    int n;
    MscArgs a[WAVE_GROUP_MAX];
};

inline size_t viterbi_rot_lds_bytes(int nsteps) { return size_t(rot_layout(nsteps).total); }

__global__ void msc_history_kernel(MscArgs a) {
    msc_history_body(a, size_t(blockIdx.x) * blockDim.x + threadIdx.x, size_t(gridDim.x) * blockDim.x);
}
__global__ void msc_history_grouped_kernel(const HistoryPack pack) {
    const MscArgs &a = pack.a[blockIdx.y];
    if (a.hist_out) msc_history_body(a, size_t(blockIdx.x) * blockDim.x + threadIdx.x, size_t(gridDim.x) * blockDim.x);
}

template <class Fetch, Tail TAIL>
hipError_t launch_wave(Fetch f, const CodeTables &c, int n_codewords, uint8_t *out, uint8_t *crc_ok,
                       hipStream_t s) {
    if (n_codewords <= 0) return hipSuccess;
    if (!viterbi_fits(c.nsteps) || !c.mother_pos) return hipErrorInvalidValue;     // longer codewords: lane kernels only
    const bool rot = c.nsteps >= 102 && (c.nsteps - 6) % 96 == 0 && viterbi_rot_lds_bytes(c.nsteps) <= 160 * 1024;
    const size_t per_wave = rot ? viterbi_rot_lds_bytes(c.nsteps) : viterbi_wave_lds_bytes(c.nsteps);
    int lds_per_wave = int((per_wave + 255) & ~size_t(255));
    // 4 waves per workgroup normally; long codewords (high bit rates) need more LDS per wave -> fewer waves
    int waves = WAVES_PER_WG;
    while (waves > 1 && size_t(lds_per_wave) * waves > 160 * 1024) waves >>= 1;
    size_t lds = size_t(lds_per_wave) * waves;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const unsigned grid = unsigned((n_codewords + waves - 1) / waves);
    lds = balanced_lds_bytes(grid, lds, 8);
    if (rot)
        hipLaunchKernelGGL((viterbi_rot_kernel<Fetch, TAIL>), dim3(grid), dim3(64 * waves), lds, s, f, c, n_codewords, out,
                           crc_ok, lds_per_wave);
    else
        hipLaunchKernelGGL((viterbi_wave_kernel<Fetch, TAIL>), dim3(grid), dim3(64 * waves), lds, s, f, c, n_codewords, out,
                           crc_ok, lds_per_wave);
    return hipGetLastError();
}

template <class Fetch, Tail TAIL>
hipError_t allow_full_lds() {
    for (const void *k : {reinterpret_cast<const void *>(viterbi_rot_kernel<Fetch, TAIL>),
                          reinterpret_cast<const void *>(viterbi_wave_kernel<Fetch, TAIL>)}) {
        const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

bool wave_group_supported(int nsteps) {
    return nsteps >= 102 && (nsteps - 6) % 96 == 0 && viterbi_fits(nsteps) && viterbi_rot_lds_bytes(nsteps) <= 160 * 1024;
}

static unsigned cu_count() {
    static const unsigned n_cu = [] {
        int dev = 0;
        unsigned n = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n = unsigned(prop.multiProcessorCount);
        return n;
    }();
    return n_cu;
}

bool wave_group_one_round(int max_nsteps, long n_waves) {
    const size_t lds = (viterbi_rot_lds_bytes(max_nsteps) + 255) & ~size_t(255);
    return n_waves >= 0 && size_t(n_waves) <= size_t(cu_count()) * (size_t(160 * 1024) / lds);
}

hipError_t launch_msc_decode_group(const WaveGroupItem *items, int n, hipStream_t s, const WaveFicItem *fic) {
    for (int i0 = 0; i0 < n; i0 += WAVE_GROUP_MAX) {
        const int m = std::min(WAVE_GROUP_MAX, n - i0);
        WaveEntryPack pack{};
        HistoryPack hp{};
        pack.n = hp.n = m;
        int total = 0;
        size_t lds = 0, hist_items = 0;
        if (fic && i0 == 0 && fic->n_frames > 0) {             // the FIC goes with the first launch
            if (!wave_group_supported(fic->code.nsteps)) return hipErrorInvalidValue;
            pack.n_fic = fic->n_frames * NB_FIC_GROUPS;
            pack.fic_fetch = FetchFic{fic->soft, fic->soft_stride};
            pack.fic_code = fic->code;
            pack.fib = fic->fib;
            pack.crc_ok = fic->crc_ok;
            lds = (viterbi_rot_lds_bytes(fic->code.nsteps) + 255) & ~size_t(255);
        }
        for (int i = 0; i < m; i++) {
            const WaveGroupItem &it = items[i0 + i];
            const MscArgs &a = it.args;
            if (!wave_group_supported(it.code.nsteps)) return hipErrorInvalidValue;
            pack.e[i].fetch = FetchMsc{a.soft, a.soft_stride, a.hist_in, a.frames_per_stream, a.start_bit, a.nbits};
            pack.e[i].code = it.code;
            pack.e[i].out = a.out;
            pack.e[i].first_cw = total;
            pack.e[i].n_streams = a.n_streams;
            pack.e[i].hist_out = a.hist_out;
            total += a.n_streams * a.frames_per_stream * NB_CIFS;
            lds = std::max(lds, (viterbi_rot_lds_bytes(it.code.nsteps) + 255) & ~size_t(255));
            hp.a[i] = a;
            if (a.hist_out) hist_items = std::max(hist_items, size_t(a.n_streams) * 15 * a.nbits);
        }
        if (total + pack.n_fic <= 0) continue;
        pack.n_dec = total + pack.n_fic;
        // small rings (the plugin's one stream) ride in the decoding launch; large ones keep a launch of their own, whose
        // 256-thread workgroups move bytes faster than 64-thread ones sharing CUs with the decoder
        const bool ride = hist_items > 0 && hist_items <= size_t(1) << 20;
        hipLaunchKernelGGL(viterbi_rot_grouped_kernel, dim3(unsigned(pack.n_dec + (ride ? m * HIST_BLOCKS : 0))), dim3(64), lds, s, pack);
        if (hist_items && !ride)
            hipLaunchKernelGGL(msc_history_grouped_kernel, dim3(unsigned(std::min<size_t>((hist_items + 255) / 256, 256)), unsigned(m)),
                               dim3(256), 0, s, hp);
    }
    return hipGetLastError();
}

hipError_t init_viterbi_kernel_attributes() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_rot_grouped_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = allow_full_lds<FetchFic, Tail::kFic>();
    if (e == hipSuccess) e = allow_full_lds<FetchPlain, Tail::kBytes>();
    if (e == hipSuccess) e = allow_full_lds<FetchMsc, Tail::kBytes>();
    return e;
}

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
    return launch_msc_history(a, s);
}

// Balance a grid over the CUs: with o_max workgroups resident per CU the dispatcher fills CUs greedily, so a grid
// that is not a multiple of o_max * CUs leaves some CUs with more co-resident (slower) waves than others.  Ask for
// just enough LDS that every CU holds the same number of workgroups per round.
size_t balanced_lds_bytes(unsigned grid, size_t lds, unsigned o_cap) {
    const unsigned n_cu = cu_count();
    const size_t cu_lds = 160 * 1024;
    const unsigned o_max = unsigned(std::min<size_t>(lds ? cu_lds / lds : o_cap, o_cap));
    if (o_max == 0 || grid == 0) return lds;
    const unsigned rounds = (grid + o_max * n_cu - 1) / (o_max * n_cu);
    const unsigned o_need = std::max(1u, (grid + rounds * n_cu - 1) / (rounds * n_cu));
    if (o_need < o_max) {
        const size_t padded = (cu_lds / o_need) & ~size_t(1023);
        if (padded >= lds) lds = padded;
    }
    return lds;
}

hipError_t launch_msc_history(const MscArgs &a, hipStream_t s) {
    if (!a.hist_out) return hipSuccess;
    const size_t total = size_t(a.n_streams) * 15 * a.nbits;
    const unsigned grid = unsigned(std::min<size_t>((total + 255) / 256, 2048));
    hipLaunchKernelGGL(msc_history_kernel, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace dabk
