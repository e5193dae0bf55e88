/*
 * dab_oracle.c -- CPU restatement of the DAB Mode-I receive chain (see dab_oracle.h).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (reference DSP source absent, no reference
 * tests).  Each function cites the reference call site it stands behind and the
 * ETSI EN 300 401 clause it restates.  Plain scalar C99, no -ffast-math.
 */
#include "dab_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------- */
/* A0': carrier mapper.  Call site: get_DAB_mapper_ref(span<int>[1536], 2048)   */
/* /root/reference/src/radio_block.cpp:20-21.  ETSI clause 14.6 (frequency      */
/* interleaving): PI(0)=0, PI(i)=(13*PI(i-1)+511) mod 2048, keep 256..1792\1024 */
/* ------------------------------------------------------------------------- */
void oracle_get_mapper(int32_t *out)
{
    int pi = 0, n = 0;
    for (int i = 0; i < DAB_NB_FFT; i++) {
        if (i > 0) pi = (13 * pi + 511) % DAB_NB_FFT;
        if (pi < 256 || pi > 1792 || pi == 1024) continue;
        /* carrier k = pi-1024 in [-768,768]\{0}; index in the -768..768 sans-DC vector */
        out[n++] = (pi < 1024) ? (pi - 256) : (pi - 257);
    }
}

int oracle_carrier_bin(int i)
{
    /* i<768 <-> k=-768+i -> bin 2048+k ; i>=768 <-> k=i-767 -> bin k */
    return (i < 768) ? (1280 + i) : (i - 767);
}

/* ------------------------------------------------------------------------- */
/* A0': phase reference symbol.  Call site: get_DAB_PRS_reference(1, span)      */
/* /root/reference/src/radio_block.cpp:18-19.  ETSI clause 14.3.2, tables 39/43 */
/* ------------------------------------------------------------------------- */
static const uint8_t PRS_H[4][32] = {
    {0,2,0,0,0,0,1,1,2,0,0,0,2,2,1,1, 0,2,0,0,0,0,1,1,2,0,0,0,2,2,1,1},
    {0,3,2,3,0,1,3,0,2,1,2,3,2,3,3,0, 0,3,2,3,0,1,3,0,2,1,2,3,2,3,3,0},
    {0,0,0,2,0,2,1,3,2,2,0,2,2,0,1,3, 0,0,0,2,0,2,1,3,2,2,0,2,2,0,1,3},
    {0,1,2,1,0,3,3,2,2,3,2,1,2,1,3,2, 0,1,2,1,0,3,3,2,2,3,2,1,2,1,3,2},
};
/* (i, n) for the 48 blocks of 32 carriers: k' = -768,-736,...,-32, 1,33,...,737 */
static const uint8_t PRS_IN[48][2] = {
    {0,1},{1,2},{2,0},{3,1},{0,3},{1,2},{2,2},{3,3},{0,2},{1,1},{2,2},{3,3},
    {0,1},{1,2},{2,3},{3,3},{0,2},{1,2},{2,2},{3,1},{0,1},{1,3},{2,1},{3,2},
    {0,3},{3,1},{2,1},{1,1},{0,2},{3,2},{2,1},{1,0},{0,2},{3,2},{2,3},{1,3},
    {0,0},{3,2},{2,1},{1,3},{0,3},{3,3},{2,3},{1,0},{0,3},{3,0},{2,1},{1,1},
};

void oracle_get_prs(float *out)
{
    static const float RE[4] = {1.f, 0.f, -1.f, 0.f};
    static const float IM[4] = {0.f, 1.f, 0.f, -1.f};
    memset(out, 0, sizeof(float) * 2 * DAB_NB_FFT);
    for (int blk = 0; blk < 48; blk++) {
        const int kstart = (blk < 24) ? (-768 + 32 * blk) : (1 + 32 * (blk - 24));
        for (int j = 0; j < 32; j++) {
            const int k = kstart + j;
            const int q = (PRS_H[PRS_IN[blk][0]][j] + PRS_IN[blk][1]) & 3;
            const int bin = (k + DAB_NB_FFT) % DAB_NB_FFT;
            out[2 * bin] = RE[q];
            out[2 * bin + 1] = IM[q];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Puncturing (ETSI clause 11.1.2, table 29): V_PI promotes the 8 groups of 4   */
/* mother bits in the order g0,g4,g2,g6,g1,g5,g3,g7 from 1000 -> 1100 -> 1110   */
/* -> 1111.  Tail vector V_T = 1100 x 6.                                        */
/* ------------------------------------------------------------------------- */
void oracle_puncture_vector(int pi, uint8_t out[32])
{
    static const int ORDER[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    int ones[8];
    for (int g = 0; g < 8; g++) ones[g] = 1;
    for (int step = 0; step < pi; step++) ones[ORDER[step % 8]]++;
    for (int g = 0; g < 8; g++)
        for (int b = 0; b < 4; b++) out[4 * g + b] = (uint8_t)(b < ones[g]);
}

static int append_blocks(uint8_t *mask, int pos, int nblocks, int pi)
{
    uint8_t v[32];
    oracle_puncture_vector(pi, v);
    for (int b = 0; b < nblocks; b++)
        for (int r = 0; r < 4; r++) { /* one block = 128 mother bits = 4 x V_PI */
            memcpy(mask + pos, v, 32);
            pos += 32;
        }
    return pos;
}

static int append_tail(uint8_t *mask, int pos)
{
    static const uint8_t VT[4] = {1, 1, 0, 0};
    for (int i = 0; i < 6; i++) { memcpy(mask + pos, VT, 4); pos += 4; }
    return pos;
}

static int count_ones(const uint8_t *m, int n)
{
    int c = 0;
    for (int i = 0; i < n; i++) c += m[i];
    return c;
}

/* FIC Mode I (clause 11.2.1): 21 blocks PI=16, 3 blocks PI=15, tail. */
int oracle_fic_puncture_mask(uint8_t *mask)
{
    int pos = 0;
    pos = append_blocks(mask, pos, 21, 16);
    pos = append_blocks(mask, pos, 3, 15);
    pos = append_tail(mask, pos);
    return (pos == DAB_NB_FIC_MOTHER) ? count_ones(mask, pos) : -1;
}

/* EEP (clause 11.3.2, tables 32/33). */
int oracle_eep_puncture_mask(int option, int level, int bitrate, uint8_t *mask, int *out_nsteps, int *out_cu)
{
    int L1, L2, P1, P2, cu;
    if (level < 1 || level > 4 || bitrate <= 0) return -1;
    if (option == 0) {
        if (bitrate % 8) return -1;
        const int n = bitrate / 8;
        switch (level) {
        case 1: L1 = 6 * n - 3; L2 = 3; P1 = 24; P2 = 23; cu = 12 * n; break;
        case 2:
            if (n == 1) { L1 = 5; L2 = 1; P1 = 13; P2 = 12; }
            else { L1 = 2 * n - 3; L2 = 4 * n + 3; P1 = 14; P2 = 13; }
            cu = 8 * n; break;
        case 3: L1 = 6 * n - 3; L2 = 3; P1 = 8; P2 = 7; cu = 6 * n; break;
        default: L1 = 4 * n - 3; L2 = 2 * n + 3; P1 = 3; P2 = 2; cu = 4 * n; break;
        }
    } else if (option == 1) {
        if (bitrate % 32) return -1;
        const int n = bitrate / 32;
        static const int PB[4][2] = {{10, 9}, {6, 5}, {4, 3}, {2, 1}};
        static const int CB[4] = {27, 21, 18, 15};
        L1 = 24 * n - 3; L2 = 3; P1 = PB[level - 1][0]; P2 = PB[level - 1][1];
        cu = CB[level - 1] * n;
    } else {
        return -1;
    }
    const int nsteps = bitrate * 24 + 6;
    int pos = 0;
    pos = append_blocks(mask, pos, L1, P1);
    pos = append_blocks(mask, pos, L2, P2);
    pos = append_tail(mask, pos);
    if (pos != 4 * nsteps) return -1;
    if (out_nsteps) *out_nsteps = nsteps;
    if (out_cu) *out_cu = cu;
    return count_ones(mask, pos);
}

/* ------------------------------------------------------------------------- */
/* A10: energy dispersal PRBS x^9+x^5+1, register all ones (clause 12).        */
/* ------------------------------------------------------------------------- */
void oracle_prbs(uint8_t *bits, int n)
{
    unsigned reg = 0x1FF;
    for (int i = 0; i < n; i++) {
        const unsigned b = ((reg >> 8) ^ (reg >> 4)) & 1u;
        reg = ((reg << 1) | b) & 0x1FF;
        bits[i] = (uint8_t)b;
    }
}

/* A11: FIB CRC (clause 5.2.1): x^16+x^12+x^5+1, init all ones, complemented. */
uint16_t oracle_crc16(const uint8_t *bytes, int n)
{
    unsigned crc = 0xFFFF;
    for (int i = 0; i < n; i++) {
        crc ^= (unsigned)bytes[i] << 8;
        for (int b = 0; b < 8; b++)
            crc = (crc & 0x8000) ? ((crc << 1) ^ 0x1021) : (crc << 1);
        crc &= 0xFFFF;
    }
    return (uint16_t)(crc ^ 0xFFFF);
}

/* ------------------------------------------------------------------------- */
/* Convolutional code (clause 11.1.1): K=7, rate 1/4, generators 133,171,145,  */
/* 133 octal (MSB = current bit).  With reg bit k = a[i-k] the tap masks are    */
/* the bit reversals 109, 79, 83, 109.                                          */
/* ------------------------------------------------------------------------- */
static const unsigned POLY[4] = {109, 79, 83, 109};

static inline unsigned parity7(unsigned x)
{
    x ^= x >> 4; x ^= x >> 2; x ^= x >> 1;
    return x & 1u;
}

void oracle_conv_encode(const uint8_t *bits, int nbits, uint8_t *out)
{
    unsigned reg = 0;
    for (int i = 0; i < nbits + 6; i++) {
        const unsigned b = (i < nbits) ? (bits[i] & 1u) : 0u;
        reg = ((reg << 1) | b) & 0x7F;
        for (int p = 0; p < 4; p++) out[4 * i + p] = (uint8_t)parity7(reg & POLY[p]);
    }
}

/* ------------------------------------------------------------------------- */
/* A9: Viterbi.  Stands behind the `viterbi` package used by the reference      */
/* (/root/reference/CMakeLists.txt:53-54).  Exact int32 correlation metric.     */
/* ------------------------------------------------------------------------- */
void oracle_viterbi(const int8_t *soft, int nsteps, uint8_t *out_bits)
{
    enum { NS = 64 };
    int32_t ma[NS], mb[NS];
    int32_t *old = ma, *cur = mb;
    uint64_t *dec = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nsteps);
    /* sign[j][p]: expected bit (+1/-1) of output p on the branch (state j, h=0 i.e. j<32, input 0) */
    int sgn[32][4];
    for (int j = 0; j < 32; j++)
        for (int p = 0; p < 4; p++) sgn[j][p] = parity7(((unsigned)j << 1) & POLY[p]) ? 1 : -1;

    old[0] = 0;
    for (int s = 1; s < NS; s++) old[s] = -8192;

    for (int t = 0; t < nsteps; t++) {
        const int8_t *s4 = soft + 4 * t;
        uint64_t d = 0;
        for (int j = 0; j < 32; j++) {
            /* correlation of branch (pred j, input 0); the other three branches of the
               butterfly are +/- this value because every generator taps a[i] and a[i-6]. */
            const int32_t c = sgn[j][0] * s4[0] + sgn[j][1] * s4[1] + sgn[j][2] * s4[2] + sgn[j][3] * s4[3];
            const int32_t a0 = old[j] + c,      a1 = old[j + 32] - c; /* -> state 2j   */
            const int32_t b0 = old[j] - c,      b1 = old[j + 32] + c; /* -> state 2j+1 */
            const int d0 = a1 > a0, d1 = b1 > b0;
            cur[2 * j]     = d0 ? a1 : a0;
            cur[2 * j + 1] = d1 ? b1 : b0;
            d |= (uint64_t)d0 << (2 * j);
            d |= (uint64_t)d1 << (2 * j + 1);
        }
        dec[t] = d;
        int32_t *tmp = old; old = cur; cur = tmp;
    }
    /* traceback from the zero state (six zero tail bits) */
    unsigned s = 0;
    for (int t = nsteps - 1; t >= 0; t--) {
        const unsigned b = s & 1u;
        if (t < nsteps - 6) out_bits[t] = (uint8_t)b;
        const unsigned h = (unsigned)((dec[t] >> s) & 1u);
        s = (s >> 1) | (h << 5);
    }
    free(dec);
}

void oracle_depuncture(const int8_t *punct, const uint8_t *mask, int n_mother, int8_t *mother)
{
    int j = 0;
    for (int i = 0; i < n_mother; i++) mother[i] = mask[i] ? punct[j++] : 0;
}

static void pack_bits(const uint8_t *bits, int nbits, uint8_t *bytes)
{
    for (int i = 0; i < nbits / 8; i++) {
        unsigned v = 0;
        for (int b = 0; b < 8; b++) v = (v << 1) | (bits[8 * i + b] & 1u);
        bytes[i] = (uint8_t)v;
    }
}

/* A7/A8..A11: FIC.  Stands behind BasicRadio::Process (radio_block.cpp:42), FIC branch. */
void oracle_fic_decode(const int8_t *soft, uint8_t *fib, uint8_t *crc_ok)
{
    uint8_t mask[DAB_NB_FIC_MOTHER], prbs[768], bits[768];
    int8_t mother[DAB_NB_FIC_MOTHER];
    oracle_fic_puncture_mask(mask);
    oracle_prbs(prbs, 768);
    for (int g = 0; g < DAB_NB_FIC_GROUPS; g++) {
        oracle_depuncture(soft + g * DAB_NB_FIC_GROUP_BITS, mask, DAB_NB_FIC_MOTHER, mother);
        oracle_viterbi(mother, DAB_NB_FIC_STEPS, bits);
        for (int i = 0; i < 768; i++) bits[i] ^= prbs[i];
        pack_bits(bits, 768, fib + 96 * g);
        for (int f = 0; f < 3; f++) {
            const uint8_t *p = fib + 96 * g + 32 * f;
            const uint16_t crc = oracle_crc16(p, 30);
            crc_ok[3 * g + f] = (uint8_t)(crc == (((unsigned)p[30] << 8) | p[31]));
        }
    }
}

/* A12: time de-interleaving (clause 12): bit i is delayed by d(i mod 16) CIFs. */
static const int TDI_DELAY[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};

void oracle_time_deinterleave(const int8_t *const cifs[16], int nbits, int8_t *out)
{
    for (int i = 0; i < nbits; i++) out[i] = cifs[TDI_DELAY[i & 15]][i];
}

void oracle_msc_decode_lf(const int8_t *deint, const uint8_t *mask, int nsteps, uint8_t *out_bytes)
{
    const int nbits = nsteps - 6;
    int8_t *mother = (int8_t *)malloc((size_t)4 * nsteps);
    uint8_t *bits = (uint8_t *)malloc((size_t)nbits);
    uint8_t *prbs = (uint8_t *)malloc((size_t)nbits);
    oracle_depuncture(deint, mask, 4 * nsteps, mother);
    oracle_viterbi(mother, nsteps, bits);
    oracle_prbs(prbs, nbits);
    for (int i = 0; i < nbits; i++) bits[i] ^= prbs[i];
    pack_bits(bits, nbits, out_bytes);
    free(mother); free(bits); free(prbs);
}

/* ------------------------------------------------------------------------- */
/* A3: 2048-point forward FFT, fp32 (the reference calls FFTW3f,               */
/* /root/reference/CMakeLists.txt:55-64).  Iterative radix-2 DIT; twiddles are  */
/* generated in double and rounded once.                                        */
/* ------------------------------------------------------------------------- */
static float g_tw_re[DAB_NB_FFT / 2], g_tw_im[DAB_NB_FFT / 2];
static uint16_t g_brev[DAB_NB_FFT];
static int g_fft_init = 0;

static void fft_init(void)
{
    for (int i = 0; i < DAB_NB_FFT / 2; i++) {
        const double a = -2.0 * M_PI * (double)i / (double)DAB_NB_FFT;
        g_tw_re[i] = (float)cos(a);
        g_tw_im[i] = (float)sin(a);
    }
    for (int i = 0; i < DAB_NB_FFT; i++) {
        unsigned r = 0;
        for (int b = 0; b < 11; b++) r |= ((i >> b) & 1u) << (10 - b);
        g_brev[i] = (uint16_t)r;
    }
    g_fft_init = 1;
}

void oracle_fft2048(const float *in, float *out)
{
    if (!g_fft_init) fft_init();
    for (int i = 0; i < DAB_NB_FFT; i++) {
        out[2 * g_brev[i]] = in[2 * i];
        out[2 * g_brev[i] + 1] = in[2 * i + 1];
    }
    for (int len = 2; len <= DAB_NB_FFT; len <<= 1) {
        const int half = len >> 1, step = DAB_NB_FFT / len;
        for (int base = 0; base < DAB_NB_FFT; base += len)
            for (int k = 0; k < half; k++) {
                const float wr = g_tw_re[k * step], wi = g_tw_im[k * step];
                float *a = out + 2 * (base + k), *b = out + 2 * (base + k + half);
                const float tr = b[0] * wr - b[1] * wi;
                const float ti = b[0] * wi + b[1] * wr;
                b[0] = a[0] - tr; b[1] = a[1] - ti;
                a[0] = a[0] + tr; a[1] = a[1] + ti;
            }
    }
}

/* ------------------------------------------------------------------------- */
/* A2..A6: OFDM demodulation of one time-aligned frame.  Stands behind the      */
/* READING_SYMBOLS branch of OFDM_Demod::Process (/root/reference/src/          */
/* dab_module.cpp:25) up to the On_OFDM_Frame callback (radio_block.cpp:25).    */
/* ------------------------------------------------------------------------- */
void oracle_ofdm_demod_frame(const float *iq, float freq_offset, int8_t *soft,
                             float *spectra, float *cyc, float *dqpsk)
{
    oracle_ofdm_demod_frame_dd(iq, freq_offset, soft, spectra, cyc, dqpsk, NULL);
}

void oracle_ofdm_demod_frame_dd(const float *iq, float freq_offset, int8_t *soft,
                                float *spectra, float *cyc, float *dqpsk, float *dd4)
{
    const int32_t dphi = (int32_t)lrint((double)freq_offset * 4294967296.0);
    float *y = (float *)malloc(sizeof(float) * 2 * DAB_NB_SYM_PERIOD);
    float *X[2];
    X[0] = (float *)malloc(sizeof(float) * 2 * DAB_NB_FFT);
    X[1] = (float *)malloc(sizeof(float) * 2 * DAB_NB_FFT);
    float *d = (float *)malloc(sizeof(float) * 2 * DAB_NB_CARRIERS);
    int32_t mapper[DAB_NB_CARRIERS];
    oracle_get_mapper(mapper);

    for (int l = 0; l < DAB_NB_FRAME_SYMBOLS; l++) {
        /* A2: NCO frequency correction over the whole symbol (CP included) */
        for (int i = 0; i < DAB_NB_SYM_PERIOD; i++) {
            const uint32_t n = (uint32_t)(l * DAB_NB_SYM_PERIOD + i);
            const uint32_t ph = n * (uint32_t)dphi;
            const double ang = 2.0 * M_PI * ((double)ph / 4294967296.0);
            const float wr = (float)cos(ang), wi = (float)sin(ang);
            const float xr = iq[2 * (size_t)n], xi = iq[2 * (size_t)n + 1];
            y[2 * i] = xr * wr - xi * wi;
            y[2 * i + 1] = xr * wi + xi * wr;
        }
        /* cyclic-prefix correlation (fine frequency error estimator) */
        if (cyc || (dd4 && l == 0)) {
            double cr = 0.0, ci = 0.0;
            for (int i = 0; i < DAB_NB_CP; i++) {
                const float ar = y[2 * i], ai = y[2 * i + 1];
                const float br = y[2 * (i + DAB_NB_FFT)], bi = y[2 * (i + DAB_NB_FFT) + 1];
                cr += (double)(ar * br + ai * bi);   /* conj(a)*b */
                ci += (double)(ar * bi - ai * br);
            }
            if (cyc) {
                cyc[2 * l] = (float)cr;
                cyc[2 * l + 1] = (float)ci;
            }
            if (dd4 && l == 0) {               /* entry 0: the PRS's cyclic-prefix correlation (resolves the sums' ambiguity) */
                dd4[0] = (float)cr;
                dd4[1] = (float)ci;
            }
        }
        /* A3: FFT of the useful part */
        float *Xc = X[l & 1], *Xp = X[(l & 1) ^ 1];
        oracle_fft2048(y + 2 * DAB_NB_CP, Xc);
        if (spectra) memcpy(spectra + (size_t)l * 2 * DAB_NB_FFT, Xc, sizeof(float) * 2 * DAB_NB_FFT);
        if (l == 0) continue;
        /* A4: differential demodulation on the 1536 carriers */
        for (int i = 0; i < DAB_NB_CARRIERS; i++) {
            const int bin = oracle_carrier_bin(i);
            const float ar = Xc[2 * bin], ai = Xc[2 * bin + 1];
            const float br = Xp[2 * bin], bi = Xp[2 * bin + 1];
            d[2 * i] = ar * br + ai * bi;       /* a * conj(b) */
            d[2 * i + 1] = ai * br - ar * bi;
        }
        if (dqpsk) memcpy(dqpsk + (size_t)(l - 1) * 2 * DAB_NB_CARRIERS, d, sizeof(float) * 2 * DAB_NB_CARRIERS);
        /* decision-directed frequency error: sum of (X_l conj X_{l-1})^4 over the 256 bins v + 64 m, v = 0..63,
           m = 0, 1, 30, 31 (the carriers nearest the centre, which a sample-clock offset rotates least), bin 0 (DC,
           not a carrier) replaced by bin 768 */
        if (dd4) {
            static const int ms[4] = {0, 1, 30, 31};
            double sr = 0.0, si = 0.0;
            for (int k = 0; k < 4; k++)
                for (int v = 0; v < 64; v++) {
                    int bin = v + 64 * ms[k];
                    if (bin == 0) bin = 768;
                    const float ar = Xc[2 * bin], ai = Xc[2 * bin + 1];
                    const float br = Xp[2 * bin], bi = Xp[2 * bin + 1];
                    const float dr = ar * br + ai * bi, di = ai * br - ar * bi;
                    /* its direction only: scaled as the quantiser scales it (A6: larger component = 127, not
                       truncated), then to unit magnitude -- no input level overflows or vanishes */
                    const float A = fmaxf(fabsf(dr), fabsf(di));
                    if (A == 0.0f) continue;
                    const float fr = -127.0f * (dr / A), fi = -127.0f * (di / A);
                    const float g = 1.0f / sqrtf(fr * fr + fi * fi);
                    const float ur = fr * g, ui = fi * g;
                    const float zr = ur * ur - ui * ui, zi = 2.0f * ur * ui;
                    sr += (double)(zr * zr - zi * zi);
                    si += (double)(2.0f * zr * zi);
                }
            dd4[2 * l] = (float)sr;
            dd4[2 * l + 1] = (float)si;
        }
        /* A5+A6: frequency de-interleave + L-infinity normalise + quantise */
        int8_t *o = soft + (size_t)(l - 1) * DAB_NB_SYM_BITS;
        for (int n = 0; n < DAB_NB_CARRIERS; n++) {
            const float re = d[2 * mapper[n]], im = d[2 * mapper[n] + 1];
            const float A = fmaxf(fabsf(re), fabsf(im));
            if (A == 0.0f) { o[n] = 0; o[n + DAB_NB_CARRIERS] = 0; continue; }
            o[n] = (int8_t)(int)(-127.0f * (re / A));
            o[n + DAB_NB_CARRIERS] = (int8_t)(int)(-127.0f * (im / A));
        }
    }
    free(y); free(X[0]); free(X[1]); free(d);
}

/* ------------------------------------------------------------------------- */
/* UEP protection profiles (EN 300 401 clause 11.3.1, Table 8), table-index      */
/* order: bitrate, level, size (CU), L1..L4, PI1..PI4, padding.  Restated from   */
/* memory; oracle_uep_puncture_mask refuses a row that fails the two identities   */
/* sum L = 24*bitrate/32 and sum L*(32+4*PI) + 12 + padding = 64*size.            */
/* ------------------------------------------------------------------------- */
static const int16_t UEP_ROWS[64][12] = {
    { 32,   5,  16,   3,   4,  17,   0,   5,   3,   2,   0,   0},
    { 32,   4,  21,   3,   3,  18,   0,  11,   6,   5,   0,   0},
    { 32,   3,  24,   3,   4,  14,   3,  15,   9,   6,   8,   0},
    { 32,   2,  29,   3,   4,  14,   3,  22,  13,   8,  13,   0},
    { 32,   1,  35,   3,   5,  13,   3,  24,  17,  12,  17,   4},
    { 48,   5,  24,   4,   3,  26,   3,   5,   4,   2,   3,   0},
    { 48,   4,  29,   3,   4,  26,   3,   9,   6,   4,   6,   0},
    { 48,   3,  35,   3,   4,  26,   3,  15,  10,   6,   9,   4},
    { 48,   2,  42,   3,   4,  26,   3,  24,  14,   8,  15,   0},
    { 48,   1,  52,   3,   5,  25,   3,  24,  18,  13,  18,   0},
    { 56,   5,  29,   6,  10,  23,   3,   5,   4,   2,   3,   0},
    { 56,   4,  35,   6,  10,  23,   3,   9,   6,   4,   5,   0},
    { 56,   3,  42,   6,  12,  21,   3,  16,   7,   6,   9,   0},
    { 56,   2,  52,   6,  10,  23,   3,  23,  13,   8,  13,   8},
    { 64,   5,  32,   6,   9,  31,   2,   5,   3,   2,   3,   0},
    { 64,   4,  42,   6,   9,  33,   0,  11,   6,   5,   0,   0},
    { 64,   3,  48,   6,  12,  27,   3,  16,   8,   6,   9,   0},
    { 64,   2,  58,   6,  10,  29,   3,  23,  13,   8,  13,   8},
    { 64,   1,  70,   6,  11,  28,   3,  24,  18,  12,  18,   4},
    { 80,   5,  40,   6,  10,  41,   3,   6,   3,   2,   3,   0},
    { 80,   4,  52,   6,  10,  41,   3,  11,   6,   5,   6,   0},
    { 80,   3,  58,   6,  11,  40,   3,  16,   8,   6,   7,   0},
    { 80,   2,  70,   6,  10,  41,   3,  23,  13,   8,  13,   8},
    { 80,   1,  84,   6,  10,  41,   3,  24,  17,  12,  18,   4},
    { 96,   5,  48,   7,   9,  53,   3,   5,   4,   2,   4,   0},
    { 96,   4,  58,   7,  10,  52,   3,   9,   6,   4,   6,   0},
    { 96,   3,  70,   6,  12,  51,   3,  16,   9,   6,  10,   4},
    { 96,   2,  84,   6,  10,  53,   3,  22,  12,   9,  12,   0},
    { 96,   1, 104,   6,  13,  50,   3,  24,  18,  13,  19,   0},
    {112,   5,  58,  14,  17,  50,   3,   5,   4,   2,   5,   0},
    {112,   4,  70,  11,  21,  49,   3,   9,   6,   4,   8,   0},
    {112,   3,  84,  11,  23,  47,   3,  16,   8,   6,   9,   0},
    {112,   2, 104,  11,  21,  49,   3,  23,  12,   9,  14,   4},
    {128,   5,  64,  12,  19,  62,   3,   5,   3,   2,   4,   0},
    {128,   4,  84,  11,  21,  61,   3,  11,   6,   5,   7,   0},
    {128,   3,  96,  11,  22,  60,   3,  16,   9,   6,  10,   4},
    {128,   2, 116,  11,  21,  61,   3,  22,  12,   9,  14,   0},
    {128,   1, 140,  11,  20,  62,   3,  24,  17,  13,  19,   8},
    {160,   5,  80,  11,  19,  87,   3,   5,   4,   2,   4,   0},
    {160,   4, 104,  11,  23,  83,   3,  11,   6,   5,   9,   0},
    {160,   3, 116,  11,  24,  82,   3,  16,   8,   6,  11,   0},
    {160,   2, 140,  11,  21,  85,   3,  22,  11,   9,  13,   0},
    {160,   1, 168,  11,  22,  84,   3,  24,  18,  12,  19,   0},
    {192,   5,  96,  11,  20, 110,   3,   6,   4,   2,   5,   0},
    {192,   4, 116,  11,  22, 108,   3,  10,   6,   4,   9,   0},
    {192,   3, 140,  11,  24, 106,   3,  16,  10,   6,  11,   0},
    {192,   2, 168,  11,  20, 110,   3,  22,  13,   9,  13,   8},
    {192,   1, 208,  11,  21, 109,   3,  24,  20,  13,  24,   0},
    {224,   5, 116,  12,  22, 131,   3,   8,   6,   2,   6,   4},
    {224,   4, 140,  12,  26, 127,   3,  12,   8,   4,  11,   0},
    {224,   3, 168,  11,  20, 134,   3,  16,  10,   7,   9,   0},
    {224,   2, 208,  11,  22, 132,   3,  24,  16,  10,  15,   0},
    {224,   1, 232,  11,  24, 130,   3,  24,  20,  12,  20,   4},
    {256,   5, 128,  11,  24, 154,   3,   6,   5,   2,   5,   0},
    {256,   4, 168,  11,  24, 154,   3,  12,   9,   5,  10,   4},
    {256,   3, 192,  11,  27, 151,   3,  16,  10,   7,  10,   0},
    {256,   2, 232,  11,  22, 156,   3,  24,  14,  10,  13,   8},
    {256,   1, 280,  11,  26, 152,   3,  24,  19,  14,  18,   4},
    {320,   5, 160,  11,  26, 200,   3,   8,   5,   2,   6,   4},
    {320,   4, 208,  11,  25, 201,   3,  13,   9,   5,  10,   8},
    {320,   2, 280,  11,  26, 200,   3,  24,  17,   9,  17,   0},
    {384,   5, 192,  11,  27, 247,   3,   8,   6,   2,   7,   0},
    {384,   3, 280,  11,  24, 250,   3,  16,   9,   7,  10,   4},
    {384,   1, 416,  12,  28, 245,   3,  24,  20,  14,  23,   8},
};

int oracle_uep_profile(int index, int *bitrate, int *level, int *size_cu)
{
    if (index < 0 || index >= 64) return -1;
    *bitrate = UEP_ROWS[index][0];
    *level = UEP_ROWS[index][1];
    *size_cu = UEP_ROWS[index][2];
    return 0;
}

int oracle_uep_puncture_mask(int index, uint8_t *mask, int *nsteps, int *n_kept, int *size_cu)
{
    if (index < 0 || index >= 64) return -1;
    const int16_t *r = UEP_ROWS[index];
    int blocks = 0, bits = 12 + r[11];
    for (int i = 0; i < 4; i++) { blocks += r[3 + i]; bits += r[3 + i] * (32 + 4 * r[7 + i]); }
    if (blocks * 32 != r[0] * 24 || bits != r[2] * 64) return -1;
    int pos = 0, kept = 0;
    uint8_t v[32];
    for (int i = 0; i < 4; i++) {
        if (r[3 + i] == 0) continue;
        oracle_puncture_vector(r[7 + i], v);
        for (int b = 0; b < 4 * r[3 + i]; b++)
            for (int k = 0; k < 32; k++) { mask[pos++] = v[k]; kept += v[k]; }
    }
    for (int i = 0; i < 6; i++) { mask[pos++] = 1; mask[pos++] = 1; mask[pos++] = 0; mask[pos++] = 0; kept += 2; }
    *nsteps = pos / 4;
    *n_kept = kept;
    *size_cu = r[2];
    return (kept + r[11] == r[2] * 64 && pos / 4 == r[0] * 24 + 6) ? 0 : -1;
}

/* ------------------------------------------------------------------------- */
/* f-1: null-symbol search on a whole capture (see header).                       */
/* Summation trees are fixed so the result does not depend on who computes it:    */
/* a block is 32 pairs of samples; pair j contributes ((|x0|+|y0|)+|x1|)+|y1|,    */
/* the 32 pair sums are combined by the butterfly p[j] += p[j ^ off],             */
/* off = 1,2,4,8,16.                                                              */
/* ------------------------------------------------------------------------- */
void oracle_null_block_l1(const float *iq, int64_t n_samples, float *l1)
{
    const int64_t nb = n_samples / 64;
    for (int64_t b = 0; b < nb; b++) {
        const float *x = iq + (size_t)b * 128;
        float p[32], q[32];
        for (int j = 0; j < 32; j++)
            p[j] = ((fabsf(x[4 * j]) + fabsf(x[4 * j + 1])) + fabsf(x[4 * j + 2])) + fabsf(x[4 * j + 3]);
        for (int off = 1; off < 32; off <<= 1) {
            for (int j = 0; j < 32; j++) q[j] = p[j] + p[j ^ off];
            memcpy(p, q, sizeof(p));
        }
        l1[b] = p[0];
    }
}

int oracle_null_search(const float *iq, int64_t n_samples, float thr_start, float thr_end, int min_blocks,
                       int max_out, int64_t *cands)
{
    return oracle_null_search_ex(iq, n_samples, thr_start, thr_end, min_blocks, 0, max_out, cands);
}

int oracle_null_search_ex(const float *iq, int64_t n_samples, float thr_start, float thr_end, int min_blocks,
                          int level_chunk, int max_out, int64_t *cands)
{
    const int64_t nb = n_samples / 64;
    if (nb <= 0 || max_out <= 0) return 0;
    float *l1 = (float *)malloc(sizeof(float) * (size_t)nb);
    oracle_null_block_l1(iq, n_samples, l1);
    /* mean: 1024 strided partial sums in double, the butterfly over each group of 64, the 16 group sums in order */
    double p[1024], q[64];
    for (int j = 0; j < 1024; j++) {
        double a = 0.0;
        for (int64_t b = j; b < nb; b += 1024) a += (double)l1[b];
        p[j] = a;
    }
    for (int w = 0; w < 16; w++) {
        double *g = p + 64 * w;
        for (int off = 1; off < 64; off <<= 1) {
            for (int j = 0; j < 64; j++) q[j] = g[j] + g[j ^ off];
            memcpy(g, q, sizeof(q));
        }
    }
    for (int w = 1; w < 16; w++) p[0] += p[64 * w];
    const float avg = (float)(p[0] / (double)nb);
    /* local level: mean of every chunk of level_chunk blocks (64 strided partial sums in double, the butterfly over
       them), the level at a block = mean of the chunk means c-2 .. c+2 that exist */
    double *cm = NULL;
    int64_t nc = 0;
    if (level_chunk > 0) {
        nc = (nb + level_chunk - 1) / level_chunk;
        cm = (double *)malloc(sizeof(double) * (size_t)nc);
        for (int64_t c = 0; c < nc; c++) {
            const int64_t b0 = c * level_chunk, b1 = (b0 + level_chunk < nb) ? b0 + level_chunk : nb;
            double g[64], h[64];
            for (int j = 0; j < 64; j++) {
                double a = 0.0;
                for (int64_t b = b0 + j; b < b1; b += 64) a += (double)l1[b];
                g[j] = a;
            }
            for (int off = 1; off < 64; off <<= 1) {
                for (int j = 0; j < 64; j++) h[j] = g[j] + g[j ^ off];
                memcpy(g, h, sizeof(h));
            }
            cm[c] = g[0] / (double)(b1 - b0);
        }
    }
    float ts = thr_start * avg, te = thr_end * avg;
    const int max_blocks = 2 * DAB_NB_NULL_PERIOD / 64;                       /* 83 */
    int count = 0, state = 0;
    int64_t dip_begin = 0;
    for (int64_t b = 0; b < nb && count < max_out; b++) {
        if (cm && (b % level_chunk) == 0) {
            const int64_t c = b / level_chunk;
            double a = 0.0;
            int k = 0;
            for (int64_t j = c - 2; j <= c + 2; j++)
                if (j >= 0 && j < nc) { a += cm[j]; k++; }
            const float level = (float)(a / (double)k);
            ts = thr_start * level;
            te = thr_end * level;
        }
        if (state == 0) {
            if (l1[b] < ts) { state = 1; dip_begin = b; }
        } else if (l1[b] > te) {
            const int64_t len = b - dip_begin;
            const int64_t c = b * 64 - 48;
            if (len >= min_blocks && len <= max_blocks && c >= 0 &&
                c + (int64_t)DAB_NB_FRAME_SYMBOLS * DAB_NB_SYM_PERIOD + 512 <= n_samples)
                cands[count++] = c;
            state = 0;
        }
    }
    free(l1);
    free(cm);
    return count;
}

void oracle_acquire_candidate(const float *iq, int64_t n_samples, int64_t cand, int max_coarse,
                              float min_peak_to_mean, int margin, oracle_acquired_frame *out)
{
    oracle_acquire_candidate_ex(iq, n_samples, cand, max_coarse, min_peak_to_mean, margin, 1.0f, 0.0f, out);
}

void oracle_acquire_candidate_ex(const float *iq, int64_t n_samples, int64_t cand, int max_coarse,
                                 float min_peak_to_mean, int margin, float distance_prob, float first_path_rel,
                                 oracle_acquired_frame *out)
{
    const float *x = iq + 2 * (size_t)cand;
    double cr = 0.0, ci = 0.0;
    for (int i = 64; i < 440; i++) {
        const float ar = x[2 * i], ai = x[2 * i + 1];
        const float br = x[2 * (i + DAB_NB_FFT)], bi = x[2 * (i + DAB_NB_FFT) + 1];
        cr += (double)(ar * br + ai * bi);                              /* conj(a) * b */
        ci += (double)(ar * bi - ai * br);
    }
    const float fine = (float)(-atan2(ci, cr) / (2.0 * M_PI * (double)DAB_NB_FFT));
    int32_t k, toff;
    float ptm, cptm;
    oracle_sync_prs_ex(x, fine, max_coarse, 0, distance_prob, first_path_rel, &k, &toff, &ptm, &cptm);
    out->start = cand + toff - margin;
    out->coarse_carriers = k;
    out->fine_offset = fine;
    out->freq_offset = fine - (float)k / (float)DAB_NB_FFT;
    out->peak_to_mean = ptm;
    out->coarse_peak_to_mean = cptm;
    out->flags = (ptm >= min_peak_to_mean ? 1 : 0) |
                 ((out->start >= 0 && out->start + (int64_t)DAB_NB_FRAME_SYMBOLS * DAB_NB_SYM_PERIOD <= n_samples) ? 2 : 0);
}

/* ------------------------------------------------------------------------- */
/* f-1: coarse frequency + fine time synchronisation on the PRS (see header).   */
/* ------------------------------------------------------------------------- */
void oracle_sync_prs(const float *sym, float freq_offset, int max_coarse, int32_t *k_out, int32_t *toff,
                     float *peak_to_mean, float *coarse_peak_to_mean)
{
    oracle_sync_prs_ex(sym, freq_offset, max_coarse, 0, 1.0f, 0.0f, k_out, toff, peak_to_mean, coarse_peak_to_mean);
}

void oracle_sync_prs_ex(const float *sym, float freq_offset, int max_coarse, int expected, float distance_prob,
                        float first_path_rel, int32_t *k_out, int32_t *toff, float *peak_to_mean,
                        float *coarse_peak_to_mean)
{
    const int32_t dphi = (int32_t)lrint((double)freq_offset * 4294967296.0);
    float *y = (float *)malloc(sizeof(float) * 2 * DAB_NB_FFT);
    float *X = (float *)malloc(sizeof(float) * 2 * DAB_NB_FFT);
    float *Q = (float *)malloc(sizeof(float) * 2 * DAB_NB_FFT);
    float *Z = (float *)malloc(sizeof(float) * 2 * DAB_NB_FFT);
    float *H = (float *)malloc(sizeof(float) * 2 * DAB_NB_FFT);
    float R[2 * DAB_NB_FFT];
    int8_t qt[DAB_NB_FFT];              /* quarter turns of R per bin, -1 = not a carrier */
    oracle_get_prs(R);
    for (int b = 0; b < DAB_NB_FFT; b++) {
        const float re = R[2 * b], im = R[2 * b + 1];
        qt[b] = (re == 0.f && im == 0.f) ? -1 : (re > 0.5f ? 0 : (im > 0.5f ? 1 : (re < -0.5f ? 2 : 3)));
    }
    for (int n = 0; n < DAB_NB_FFT; n++) {
        const uint32_t ph = (uint32_t)n * (uint32_t)dphi;
        const double ang = 2.0 * M_PI * ((double)ph / 4294967296.0);
        const float wr = (float)cos(ang), wi = (float)sin(ang);
        const float xr = sym[2 * (DAB_NB_CP + n)], xi = sym[2 * (DAB_NB_CP + n) + 1];
        y[2 * n] = xr * wr - xi * wi;
        y[2 * n + 1] = xr * wi + xi * wr;
    }
    oracle_fft2048(y, X);
    for (int b = 0; b < DAB_NB_FFT; b++) {
        const int b1 = (b + 1) & (DAB_NB_FFT - 1);
        const float ar = X[2 * b1], ai = X[2 * b1 + 1], br = X[2 * b], bi = X[2 * b + 1];
        Q[2 * b] = ar * br + ai * bi;
        Q[2 * b + 1] = ai * br - ar * bi;
    }
    /* coarse: scan k = -max..+max */
    float best = -1.0f, sum = 0.0f;
    int best_k = 0;
    for (int k = -max_coarse; k <= max_coarse; k++) {
        float dr = 0.f, di = 0.f;
        for (int b = 0; b < DAB_NB_FFT - 1; b++) {
            if (qt[b] < 0 || qt[b + 1] < 0) continue;
            const int s = (qt[b + 1] - qt[b]) & 3;                 /* S[b] = j^s ; multiply Q by (-j)^s */
            const int i = (b + k) & (DAB_NB_FFT - 1);
            const float qr = Q[2 * i], qi = Q[2 * i + 1];
            switch (s) {
            case 0: dr += qr; di += qi; break;
            case 1: dr += qi; di -= qr; break;
            case 2: dr -= qr; di -= qi; break;
            default: dr -= qi; di += qr; break;
            }
        }
        const float m = dr * dr + di * di;
        sum += m;
        if (m > best) { best = m; best_k = k; }
    }
    *k_out = best_k;
    *coarse_peak_to_mean = best / (sum / (float)(2 * max_coarse + 1));
    /* fine time: |IFFT(Z)| == |FFT(conj Z)| index for index */
    for (int b = 0; b < DAB_NB_FFT; b++) {
        if (qt[b] < 0) { Z[2 * b] = 0.f; Z[2 * b + 1] = 0.f; continue; }
        const int i = (b + best_k) & (DAB_NB_FFT - 1);
        const float xr = X[2 * i], xi = X[2 * i + 1];
        float zr, zi;                                              /* X * conj(R) = X * (-j)^qt */
        switch (qt[b]) {
        case 0: zr = xr; zi = xi; break;
        case 1: zr = xi; zi = -xr; break;
        case 2: zr = -xr; zi = -xi; break;
        default: zr = -xi; zi = xr; break;
        }
        Z[2 * b] = zr; Z[2 * b + 1] = -zi;                          /* conj */
    }
    oracle_fft2048(Z, H);
    /* which tap: score = |h|^2 w^2, w = 1 - (1 - p) |t - expected| / 2552 (first maximum); then, if asked for, the
       earliest tap within a cyclic prefix before it that reaches max(rel * peak power, 16 * mean) */
    float *M = (float *)malloc(sizeof(float) * DAB_NB_FFT);
    float pk = -1.0f, tot = 0.0f;
    int pi = 0;
    const float decay = (1.0f - distance_prob) * (1.0f / (float)DAB_NB_SYM_PERIOD);
    for (int n = 0; n < DAB_NB_FFT; n++) {
        const float m = H[2 * n] * H[2 * n] + H[2 * n + 1] * H[2 * n + 1];
        M[n] = m;
        tot += m;
        const int t = (n < DAB_NB_FFT / 2) ? n : n - DAB_NB_FFT;
        const float w = 1.0f - decay * (float)abs(t - expected);
        const float sc = (m * w) * w;
        if (sc > pk) { pk = sc; pi = n; }
    }
    const float peak = M[pi], mean = tot / (float)DAB_NB_FFT;
    if (first_path_rel > 0.0f) {
        const float a = first_path_rel * peak, b = 16.0f * mean;
        const float thr = a > b ? a : b;
        for (int d = DAB_NB_CP; d >= 1; d--)
            if (M[(pi - d) & (DAB_NB_FFT - 1)] >= thr) { pi = (pi - d) & (DAB_NB_FFT - 1); break; }
    }
    free(M);
    *toff = (pi < DAB_NB_FFT / 2) ? pi : pi - DAB_NB_FFT;
    *peak_to_mean = peak / mean;
    free(y); free(X); free(Q); free(Z); free(H);
}
