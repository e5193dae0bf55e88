// dab_tables.hpp -- Mode-I constants and host-side table builders for libdabgpu.
//
// Product code: restated from ETSI EN 300 401 independently of the CPU test oracle
// so that tests can cross-check the two.  Stands behind
// get_DAB_OFDM_params / get_dab_parameters / get_DAB_PRS_reference /
// get_DAB_mapper_ref (/root/reference/src/radio_block.cpp:12-21).
#pragma once
#include <cstdint>
#include <vector>

namespace dab {

constexpr int NB_FFT = 2048;
constexpr int NB_CP = 504;
constexpr int NB_SYM_PERIOD = NB_FFT + NB_CP;      // 2552
constexpr int NB_NULL_PERIOD = 2656;
constexpr int NB_FRAME_SYMBOLS = 76;               // PRS + 75
constexpr int NB_DATA_SYMBOLS = 75;
constexpr int NB_CARRIERS = 1536;
constexpr int NB_SYM_BITS = 2 * NB_CARRIERS;       // 3072
constexpr int NB_FRAME_BITS = NB_DATA_SYMBOLS * NB_SYM_BITS;  // 230400
constexpr int NB_FRAME_SAMPLES = NB_NULL_PERIOD + NB_FRAME_SYMBOLS * NB_SYM_PERIOD;  // 196608
constexpr int NB_FIC_SYMBOLS = 3;
constexpr int NB_FIC_BITS = NB_FIC_SYMBOLS * NB_SYM_BITS;     // 9216
constexpr int NB_FIC_GROUPS = 4;
constexpr int NB_FIC_GROUP_BITS = NB_FIC_BITS / NB_FIC_GROUPS;  // 2304
constexpr int NB_FIC_INFO_BITS = 768;
constexpr int NB_FIC_STEPS = NB_FIC_INFO_BITS + 6;           // 774
constexpr int NB_FIBS = 12;
constexpr int NB_CIFS = 4;
constexpr int NB_CIF_BITS = 864 * 64;                         // 55296
constexpr int CU_BITS = 64;
constexpr int TDI_DEPTH = 16;
constexpr int VITERBI_INIT_PENALTY = 8192;   // start metric of states != 0 is -8192

static_assert(NB_FRAME_SAMPLES == 196608, "Mode-I frame is 96 ms at 2.048 MSPS");
static_assert(NB_FIC_BITS + NB_CIFS * NB_CIF_BITS == NB_FRAME_BITS, "FIC + 4 CIF fill the frame");

// ETSI clause 14.6: frequency interleaver.  out[n] = index of data carrier n in the
// carrier vector ordered k = -768..-1, 1..768.
inline std::vector<int32_t> make_mapper() {
    std::vector<int32_t> out;
    out.reserve(NB_CARRIERS);
    int pi = 0;
    for (int i = 0; i < NB_FFT; i++) {
        if (i) pi = (13 * pi + NB_FFT / 4 - 1) % NB_FFT;
        const int k = pi - NB_FFT / 2;
        if (k < -NB_CARRIERS / 2 || k > NB_CARRIERS / 2 || k == 0) continue;
        out.push_back(k < 0 ? k + NB_CARRIERS / 2 : k + NB_CARRIERS / 2 - 1);
    }
    return out;
}

// FFT bin of carrier-vector index i.
inline int carrier_bin(int i) {
    const int k = (i < NB_CARRIERS / 2) ? (i - NB_CARRIERS / 2) : (i - NB_CARRIERS / 2 + 1);
    return (k + NB_FFT) % NB_FFT;
}

// ETSI clause 14.3.2: phase reference symbol; returns quarter-turn counts per bin
// (-1 = unused bin) so callers can build cf32 or phases.
// (noinline + a volatile-free plain loop: hipcc's host pass was observed to mis-vectorise the inlined copy of
// this function inside dabgpu_create -- rows indexed as H[i][j] instead of H[i][j % 16] -- while the copy in
// dabgpu_get_prs_reference was correct; tests/test_sync.py now pins the device table against the oracle.)
__attribute__((noinline)) inline std::vector<int8_t> make_prs_quarter_turns() {
    static const uint8_t H[4][16] = {
        {0, 2, 0, 0, 0, 0, 1, 1, 2, 0, 0, 0, 2, 2, 1, 1},
        {0, 3, 2, 3, 0, 1, 3, 0, 2, 1, 2, 3, 2, 3, 3, 0},
        {0, 0, 0, 2, 0, 2, 1, 3, 2, 2, 0, 2, 2, 0, 1, 3},
        {0, 1, 2, 1, 0, 3, 3, 2, 2, 3, 2, 1, 2, 1, 3, 2},
    };
    // table 39 (Mode I): i and n per block of 32 carriers, k' = -768 + 32b (b<24), 1 + 32(b-24)
    static const uint8_t I_[48] = {0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2, 3,
                                   0, 3, 2, 1, 0, 3, 2, 1, 0, 3, 2, 1, 0, 3, 2, 1, 0, 3, 2, 1, 0, 3, 2, 1};
    static const uint8_t N_[48] = {1, 2, 0, 1, 3, 2, 2, 3, 2, 1, 2, 3, 1, 2, 3, 3, 2, 2, 2, 1, 1, 3, 1, 2,
                                   3, 1, 1, 1, 2, 2, 1, 0, 2, 2, 3, 3, 0, 2, 1, 3, 3, 3, 3, 0, 3, 0, 1, 1};
    std::vector<int8_t> q(NB_FFT, -1);
    for (int b = 0; b < 48; b++) {
        const int k0 = (b < 24) ? (-768 + 32 * b) : (1 + 32 * (b - 24));
        const uint8_t *row = H[I_[b]];
        for (int half = 0; half < 2; half++)          // the 32-entry row of the standard is the 16-entry row twice
            for (int j = 0; j < 16; j++) {
                const int k = k0 + 16 * half + j;
                q[size_t((k + NB_FFT) % NB_FFT)] = int8_t((row[j] + N_[b]) & 3);
            }
    }
    return q;
}

// ETSI table 29: puncturing vector V_PI as 32 flags.
inline void puncture_vector(int pi, uint8_t out[32]) {
    // group g is promoted (gains one more leading 1) at steps where bitrev3(step % 8) == g
    static const int BITREV3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    int ones[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int s = 0; s < pi; s++) ones[BITREV3[s & 7]]++;
    for (int g = 0; g < 8; g++)
        for (int b = 0; b < 4; b++) out[4 * g + b] = b < ones[g];
}

struct PunctureProfile {
    int nsteps = 0;                 // trellis steps = info bits + 6
    int n_punct = 0;                // transmitted bits
    std::vector<uint8_t> mask;      // 4*nsteps flags
};

inline void append_blocks(std::vector<uint8_t> &m, int nblocks, int pi) {
    uint8_t v[32];
    puncture_vector(pi, v);
    for (int b = 0; b < 4 * nblocks; b++) m.insert(m.end(), v, v + 32);
}
inline void append_tail(std::vector<uint8_t> &m) {
    for (int i = 0; i < 6; i++) { m.push_back(1); m.push_back(1); m.push_back(0); m.push_back(0); }
}
inline void finish_profile(PunctureProfile &p) {
    p.nsteps = int(p.mask.size() / 4);
    p.n_punct = 0;
    for (uint8_t f : p.mask) p.n_punct += f;
}

// clause 11.2.1, Mode I
inline PunctureProfile make_fic_profile() {
    PunctureProfile p;
    append_blocks(p.mask, 21, 16);
    append_blocks(p.mask, 3, 15);
    append_tail(p.mask);
    finish_profile(p);
    return p;
}

// clause 11.3.2 (EEP); returns false if the profile is invalid. size_cu receives the size.
inline bool make_eep_profile(int type, int level, int bitrate, PunctureProfile &p, int &size_cu) {
    if (level < 1 || level > 4 || bitrate <= 0) return false;
    int L1, L2, P1, P2;
    if (type == 0) {
        if (bitrate % 8) return false;
        const int n = bitrate / 8;
        if (level == 1) { L1 = 6 * n - 3; L2 = 3; P1 = 24; P2 = 23; size_cu = 12 * n; }
        else if (level == 2) {
            if (n == 1) { L1 = 5; L2 = 1; P1 = 13; P2 = 12; } else { L1 = 2 * n - 3; L2 = 4 * n + 3; P1 = 14; P2 = 13; }
            size_cu = 8 * n;
        }
        else if (level == 3) { L1 = 6 * n - 3; L2 = 3; P1 = 8; P2 = 7; size_cu = 6 * n; }
        else { L1 = 4 * n - 3; L2 = 2 * n + 3; P1 = 3; P2 = 2; size_cu = 4 * n; }
    } else if (type == 1) {
        if (bitrate % 32) return false;
        const int n = bitrate / 32;
        static const int PB[4][2] = {{10, 9}, {6, 5}, {4, 3}, {2, 1}};
        static const int CB[4] = {27, 21, 18, 15};
        L1 = 24 * n - 3; L2 = 3; P1 = PB[level - 1][0]; P2 = PB[level - 1][1];
        size_cu = CB[level - 1] * n;
    } else {
        return false;
    }
    p = PunctureProfile();
    append_blocks(p.mask, L1, P1);
    append_blocks(p.mask, L2, P2);
    append_tail(p.mask);
    finish_profile(p);
    return p.nsteps == bitrate * 24 + 6 && p.n_punct == size_cu * CU_BITS && size_cu <= 864;
}

// clause 12: energy dispersal PRBS (x^9 + x^5 + 1, all ones), packed MSB first.
inline std::vector<uint8_t> make_prbs_bytes(int nbytes) {
    std::vector<uint8_t> out(nbytes);
    unsigned reg = 0x1FF;
    for (int i = 0; i < nbytes; i++) {
        unsigned v = 0;
        for (int b = 0; b < 8; b++) {
            const unsigned x = ((reg >> 8) ^ (reg >> 4)) & 1u;
            reg = ((reg << 1) | x) & 0x1FF;
            v = (v << 1) | x;
        }
        out[i] = uint8_t(v);
    }
    return out;
}

// clause 12: time interleaving delay (in CIFs) of bit i is TDI_DELAY[i % 16] = bitrev4(i % 16)
constexpr int TDI_DELAY[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};

}  // namespace dab
