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

// clause 11.3.1 (UEP), Table 8: the 64 protection profiles of the sub-channel short form, in table-index order
// (FIG 0/1 "table index"): {bit rate kbit/s, protection level 1..5, size in CUs, L1..L4 (blocks of 128 mother bits),
// PI1..PI4, padding bits}.  Restated from memory; every row satisfies both L1+L2+L3+L4 = 24*bitrate/32 and
// sum Li*(32+4*PIi) + 12 + padding = 64*size (checked at build time below and in tests/test_tables.py).
struct UepProfileRow {
    int bitrate, level, size, L[4], PI[4], padding;
};
constexpr UepProfileRow UEP_TABLE[64] = {
    {32, 5, 16, {3, 4, 17, 0}, {5, 3, 2, 0}, 0},
    {32, 4, 21, {3, 3, 18, 0}, {11, 6, 5, 0}, 0},
    {32, 3, 24, {3, 4, 14, 3}, {15, 9, 6, 8}, 0},
    {32, 2, 29, {3, 4, 14, 3}, {22, 13, 8, 13}, 0},
    {32, 1, 35, {3, 5, 13, 3}, {24, 17, 12, 17}, 4},
    {48, 5, 24, {4, 3, 26, 3}, {5, 4, 2, 3}, 0},
    {48, 4, 29, {3, 4, 26, 3}, {9, 6, 4, 6}, 0},
    {48, 3, 35, {3, 4, 26, 3}, {15, 10, 6, 9}, 4},
    {48, 2, 42, {3, 4, 26, 3}, {24, 14, 8, 15}, 0},
    {48, 1, 52, {3, 5, 25, 3}, {24, 18, 13, 18}, 0},
    {56, 5, 29, {6, 10, 23, 3}, {5, 4, 2, 3}, 0},
    {56, 4, 35, {6, 10, 23, 3}, {9, 6, 4, 5}, 0},
    {56, 3, 42, {6, 12, 21, 3}, {16, 7, 6, 9}, 0},
    {56, 2, 52, {6, 10, 23, 3}, {23, 13, 8, 13}, 8},
    {64, 5, 32, {6, 9, 31, 2}, {5, 3, 2, 3}, 0},
    {64, 4, 42, {6, 9, 33, 0}, {11, 6, 5, 0}, 0},
    {64, 3, 48, {6, 12, 27, 3}, {16, 8, 6, 9}, 0},
    {64, 2, 58, {6, 10, 29, 3}, {23, 13, 8, 13}, 8},
    {64, 1, 70, {6, 11, 28, 3}, {24, 18, 12, 18}, 4},
    {80, 5, 40, {6, 10, 41, 3}, {6, 3, 2, 3}, 0},
    {80, 4, 52, {6, 10, 41, 3}, {11, 6, 5, 6}, 0},
    {80, 3, 58, {6, 11, 40, 3}, {16, 8, 6, 7}, 0},
    {80, 2, 70, {6, 10, 41, 3}, {23, 13, 8, 13}, 8},
    {80, 1, 84, {6, 10, 41, 3}, {24, 17, 12, 18}, 4},
    {96, 5, 48, {7, 9, 53, 3}, {5, 4, 2, 4}, 0},
    {96, 4, 58, {7, 10, 52, 3}, {9, 6, 4, 6}, 0},
    {96, 3, 70, {6, 12, 51, 3}, {16, 9, 6, 10}, 4},
    {96, 2, 84, {6, 10, 53, 3}, {22, 12, 9, 12}, 0},
    {96, 1, 104, {6, 13, 50, 3}, {24, 18, 13, 19}, 0},
    {112, 5, 58, {14, 17, 50, 3}, {5, 4, 2, 5}, 0},
    {112, 4, 70, {11, 21, 49, 3}, {9, 6, 4, 8}, 0},
    {112, 3, 84, {11, 23, 47, 3}, {16, 8, 6, 9}, 0},
    {112, 2, 104, {11, 21, 49, 3}, {23, 12, 9, 14}, 4},
    {128, 5, 64, {12, 19, 62, 3}, {5, 3, 2, 4}, 0},
    {128, 4, 84, {11, 21, 61, 3}, {11, 6, 5, 7}, 0},
    {128, 3, 96, {11, 22, 60, 3}, {16, 9, 6, 10}, 4},
    {128, 2, 116, {11, 21, 61, 3}, {22, 12, 9, 14}, 0},
    {128, 1, 140, {11, 20, 62, 3}, {24, 17, 13, 19}, 8},
    {160, 5, 80, {11, 19, 87, 3}, {5, 4, 2, 4}, 0},
    {160, 4, 104, {11, 23, 83, 3}, {11, 6, 5, 9}, 0},
    {160, 3, 116, {11, 24, 82, 3}, {16, 8, 6, 11}, 0},
    {160, 2, 140, {11, 21, 85, 3}, {22, 11, 9, 13}, 0},
    {160, 1, 168, {11, 22, 84, 3}, {24, 18, 12, 19}, 0},
    {192, 5, 96, {11, 20, 110, 3}, {6, 4, 2, 5}, 0},
    {192, 4, 116, {11, 22, 108, 3}, {10, 6, 4, 9}, 0},
    {192, 3, 140, {11, 24, 106, 3}, {16, 10, 6, 11}, 0},
    {192, 2, 168, {11, 20, 110, 3}, {22, 13, 9, 13}, 8},
    {192, 1, 208, {11, 21, 109, 3}, {24, 20, 13, 24}, 0},
    {224, 5, 116, {12, 22, 131, 3}, {8, 6, 2, 6}, 4},
    {224, 4, 140, {12, 26, 127, 3}, {12, 8, 4, 11}, 0},
    {224, 3, 168, {11, 20, 134, 3}, {16, 10, 7, 9}, 0},
    {224, 2, 208, {11, 22, 132, 3}, {24, 16, 10, 15}, 0},
    {224, 1, 232, {11, 24, 130, 3}, {24, 20, 12, 20}, 4},
    {256, 5, 128, {11, 24, 154, 3}, {6, 5, 2, 5}, 0},
    {256, 4, 168, {11, 24, 154, 3}, {12, 9, 5, 10}, 4},
    {256, 3, 192, {11, 27, 151, 3}, {16, 10, 7, 10}, 0},
    {256, 2, 232, {11, 22, 156, 3}, {24, 14, 10, 13}, 8},
    {256, 1, 280, {11, 26, 152, 3}, {24, 19, 14, 18}, 4},
    {320, 5, 160, {11, 26, 200, 3}, {8, 5, 2, 6}, 4},
    {320, 4, 208, {11, 25, 201, 3}, {13, 9, 5, 10}, 8},
    {320, 2, 280, {11, 26, 200, 3}, {24, 17, 9, 17}, 0},
    {384, 5, 192, {11, 27, 247, 3}, {8, 6, 2, 7}, 0},
    {384, 3, 280, {11, 24, 250, 3}, {16, 9, 7, 10}, 4},
    {384, 1, 416, {12, 28, 245, 3}, {24, 20, 14, 23}, 8},
};
constexpr bool uep_table_consistent() {
    for (const UepProfileRow &r : UEP_TABLE) {
        int blocks = 0, bits = 12 + r.padding;
        for (int i = 0; i < 4; i++) { blocks += r.L[i]; bits += r.L[i] * (32 + 4 * r.PI[i]); }
        if (blocks * 32 != r.bitrate * 24 || bits != r.size * 64) return false;
    }
    return true;
}
static_assert(uep_table_consistent(), "UEP protection profile table");

// index into UEP_TABLE for (bit rate, protection level), -1 if there is no such profile
inline int uep_table_index(int bitrate, int level) {
    for (int i = 0; i < 64; i++)
        if (UEP_TABLE[i].bitrate == bitrate && UEP_TABLE[i].level == level) return i;
    return -1;
}

// UEP puncturing: four runs of blocks with their own puncturing index, the tail, then `padding` zero bits that the
// decoder never looks at (n_punct excludes them; the sub-channel is size*64 bits long including them).
inline bool make_uep_profile(int table_index, PunctureProfile &p, int &size_cu) {
    if (table_index < 0 || table_index >= 64) return false;
    const UepProfileRow &r = UEP_TABLE[table_index];
    p = PunctureProfile();
    for (int i = 0; i < 4; i++)
        if (r.L[i] > 0) append_blocks(p.mask, r.L[i], r.PI[i]);
    append_tail(p.mask);
    finish_profile(p);
    size_cu = r.size;
    return p.nsteps == r.bitrate * 24 + 6 && p.n_punct + r.padding == r.size * CU_BITS;
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
