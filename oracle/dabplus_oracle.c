/*
 * dabplus_oracle.c -- CPU restatement of the DAB+ audio super-frame checks: Fire code on the
 * header, RS(120,110) over GF(2^8) across the byte-interleaved columns, access-unit table and AU CRC16.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (restated from ETSI TS 102 563 clauses 5.2-5.4, 6.1 from memory; the
 * reference shows these checks only as GUI flags "Firecode / RS / AU", /root/reference/src/render_radio_block.cpp:431-437,
 * and GetSuperFrameHeader() :414-423 -- the implementation is in the absent DAB-Radio submodule).
 *
 *   super-frame  = 5 logical frames = 120*s bytes, s = bitrate/8; bytes [0,110*s) data, [110*s,120*s) RS parity
 *   RS codeword j (0..s-1) = bytes j, j+s, j+2s, ... (120 of them); GF(2^8) with x^8+x^4+x^3+x^2+1, alpha = 2,
 *                  generator prod_{k=0..9} (x - alpha^k), shortened from (255,245), corrects 5 byte errors
 *   Fire code    = x^16+x^14+x^13+x^12+x^11+x^5+x^3+x^2+x+1 (0x782F), zero start, over bytes 2..10, stored in 0..1
 *   header byte 2: rfa dac_rate sbr_flag aac_channel_mode ps_flag mpeg_surround(3)
 *   num_aus      = dac_rate/sbr: 0/1 -> 2, 1/1 -> 3, 0/0 -> 4, 1/0 -> 6; au_start[0] = 5, 6, 8, 11; further
 *                  starts are 12-bit fields from byte 3; au_start[num_aus] = 110*s
 *   AU CRC       = CRC16-CCITT (init FFFF, inverted) over the AU body, last two bytes of the AU
 */
#include <stdint.h>
#include <string.h>

#include "dab_oracle.h"

static uint8_t gf_exp[512], gf_log[256];
static int gf_ready = 0;

static void gf_init(void)
{
    unsigned x = 1;
    for (int i = 0; i < 255; i++) {
        gf_exp[i] = (uint8_t)x;
        gf_log[x] = (uint8_t)i;
        x <<= 1;
        if (x & 0x100) x ^= 0x11D;
    }
    for (int i = 255; i < 512; i++) gf_exp[i] = gf_exp[i - 255];
    gf_log[0] = 0;
    gf_ready = 1;
}

static inline uint8_t gf_mul(uint8_t a, uint8_t b)
{
    return (a && b) ? gf_exp[gf_log[a] + gf_log[b]] : 0;
}
static inline uint8_t gf_div(uint8_t a, uint8_t b) /* b != 0 */
{
    return a ? gf_exp[gf_log[a] + 255 - gf_log[b]] : 0;
}

uint16_t oracle_firecode(const uint8_t *bytes, int n)
{
    unsigned crc = 0;
    for (int i = 0; i < n; i++) {
        crc ^= (unsigned)bytes[i] << 8;
        for (int b = 0; b < 8; b++) crc = (crc & 0x8000) ? ((crc << 1) ^ 0x782F) : (crc << 1);
        crc &= 0xFFFF;
    }
    return (uint16_t)crc;
}

/* systematic RS(120,110) parity of 110 data bytes */
void oracle_rs_encode(const uint8_t *data110, uint8_t *parity10)
{
    if (!gf_ready) gf_init();
    uint8_t gp[11];
    memset(gp, 0, sizeof gp);
    gp[0] = 1;                                  /* gp[i] = coefficient of x^i of prod (x + alpha^k) */
    int deg = 0;
    for (int k = 0; k < 10; k++) {
        uint8_t nx[11];
        memset(nx, 0, sizeof nx);
        for (int i = 0; i <= deg; i++) {
            nx[i + 1] ^= gp[i];                       /* * x */
            nx[i] ^= gf_mul(gp[i], gf_exp[k]);        /* * alpha^k */
        }
        deg++;
        memcpy(gp, nx, sizeof nx);
    }
    uint8_t rem[10];
    memset(rem, 0, sizeof rem);                 /* rem[0] = highest-degree coefficient of the remainder */
    for (int i = 0; i < 110; i++) {
        const uint8_t fb = (uint8_t)(data110[i] ^ rem[0]);
        for (int j = 0; j < 9; j++) rem[j] = (uint8_t)(rem[j + 1] ^ gf_mul(fb, gp[9 - j]));
        rem[9] = gf_mul(fb, gp[0]);
    }
    memcpy(parity10, rem, 10);
}

/* decode one 120-byte codeword in place; returns the number of corrected bytes (0..5) or -1 if uncorrectable */
int oracle_rs_decode(uint8_t *cw120)
{
    if (!gf_ready) gf_init();
    enum { N = 120, T2 = 10 };
    uint8_t S[T2];
    int any = 0;
    for (int k = 0; k < T2; k++) {
        uint8_t s = 0;
        for (int i = 0; i < N; i++) s = (uint8_t)(gf_mul(s, gf_exp[k]) ^ cw120[i]);
        S[k] = s;
        any |= s;
    }
    if (!any) return 0;
    /* Berlekamp-Massey */
    uint8_t L[T2 + 1], B[T2 + 1], Tm[T2 + 1];
    memset(L, 0, sizeof L); memset(B, 0, sizeof B);
    L[0] = 1; B[0] = 1;
    int ll = 0, m = 1;
    uint8_t bb = 1;
    for (int n = 0; n < T2; n++) {
        uint8_t d = S[n];
        for (int i = 1; i <= ll; i++) d ^= gf_mul(L[i], S[n - i]);
        if (d == 0) { m++; continue; }
        memcpy(Tm, L, sizeof L);
        const uint8_t coef = gf_div(d, bb);
        for (int i = 0; i + m <= T2; i++) L[i + m] ^= gf_mul(coef, B[i]);
        if (2 * ll <= n) { ll = n + 1 - ll; memcpy(B, Tm, sizeof B); bb = d; m = 1; }
        else m++;
    }
    if (ll > 5) return -1;
    /* Omega = S(x) * Lambda(x) mod x^10 */
    uint8_t Om[T2];
    for (int i = 0; i < T2; i++) {
        uint8_t v = 0;
        for (int j = 0; j <= i && j <= ll; j++) v ^= gf_mul(L[j], S[i - j]);
        Om[i] = v;
    }
    /* Chien search over the 120 positions: byte i has power p = N-1-i, locator X = alpha^p */
    int nerr = 0, pos[5];
    uint8_t val[5];
    for (int i = 0; i < N; i++) {
        const int p = N - 1 - i;
        const int xinv = (255 - p) % 255;              /* log of X^-1 */
        uint8_t ev = 0;
        for (int j = 0; j <= ll; j++) if (L[j]) ev ^= gf_exp[(gf_log[L[j]] + xinv * j) % 255];
        if (ev) continue;
        if (nerr == 5) return -1;
        /* Forney with first root alpha^0: e = X * Omega(X^-1) / Lambda'(X^-1) */
        uint8_t om = 0, dl = 0;
        for (int j = 0; j < T2; j++) if (Om[j]) om ^= gf_exp[(gf_log[Om[j]] + xinv * j) % 255];
        for (int j = 1; j <= ll; j += 2) if (L[j]) dl ^= gf_exp[(gf_log[L[j]] + xinv * (j - 1)) % 255];
        if (dl == 0) return -1;
        pos[nerr] = i;
        val[nerr] = gf_mul(gf_exp[p % 255], gf_div(om, dl));
        nerr++;
    }
    if (nerr != ll) return -1;
    for (int e = 0; e < nerr; e++) cw120[pos[e]] ^= val[e];
    return nerr;
}

/* One super-frame: sf = 120*s bytes (modified in place by the RS correction).
 * status[0] = firecode ok (after RS), status[1] = total corrected bytes, status[2] = uncorrectable codewords,
 * status[3] = num_aus (0 if the header is not valid), status[4] = bit mask of AUs whose CRC passes,
 * au_start[0..num_aus] = AU boundaries. */
void oracle_dabplus_superframe(uint8_t *sf, int s, int32_t status[5], int32_t au_start[8])
{
    uint8_t cw[120];
    int corrected = 0, bad = 0;
    for (int j = 0; j < s; j++) {
        for (int i = 0; i < 120; i++) cw[i] = sf[j + s * i];
        const int r = oracle_rs_decode(cw);
        if (r < 0) { bad++; continue; }
        corrected += r;
        for (int i = 0; i < 120; i++) sf[j + s * i] = cw[i];
    }
    status[1] = corrected;
    status[2] = bad;
    {   /* an all-zero header matches its own (zero) check word: not a super-frame */
        unsigned any = 0;
        for (int i = 0; i < 11; i++) any |= sf[i];
        status[0] = any != 0 && oracle_firecode(sf + 2, 9) == (((unsigned)sf[0] << 8) | sf[1]);
    }
    status[3] = 0;
    status[4] = 0;
    memset(au_start, 0, sizeof(int32_t) * 8);
    if (!status[0]) return;
    const int dac_rate = (sf[2] >> 6) & 1, sbr = (sf[2] >> 5) & 1;
    const int num_aus = dac_rate ? (sbr ? 3 : 6) : (sbr ? 2 : 4);
    static const int FIRST[7] = {0, 0, 5, 6, 8, 0, 11};
    au_start[0] = FIRST[num_aus];
    int bitpos = 24;                                  /* 12-bit fields start at byte 3 */
    for (int a = 1; a < num_aus; a++) {
        int v = 0;
        for (int b = 0; b < 12; b++, bitpos++) v = (v << 1) | ((sf[bitpos >> 3] >> (7 - (bitpos & 7))) & 1);
        au_start[a] = v;
    }
    au_start[num_aus] = 110 * s;
    status[3] = num_aus;
    for (int a = 0; a < num_aus; a++) {
        const int b0 = au_start[a], b1 = au_start[a + 1];
        if (b0 < 3 || b1 > 110 * s || b1 - b0 < 3) continue;          /* malformed table: CRC flag stays 0 */
        const uint16_t crc = oracle_crc16(sf + b0, b1 - b0 - 2);
        if (crc == (((unsigned)sf[b1 - 2] << 8) | sf[b1 - 1])) status[4] |= 1 << a;
    }
}
