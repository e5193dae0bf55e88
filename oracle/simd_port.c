/*
 * simd_port.c -- a CPU implementation of the hot path OF THE REFERENCE'S CLASS, for bench.py's cpu_baseline leg only.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE.  Not the oracle (dab_oracle.c stays the checker: exact integer Viterbi, libm NCO),
 * not part of the product.  The reference links FFTW3f and a SIMD Viterbi decoder and builds with
 * `-ffast-math -march=native` (/root/reference/CMakeLists.txt:53-64, /root/reference/CMakePresets.json:40-47,
 * /root/reference/README.md:37); neither library is in the image or on the pool boxes, so the scalar oracle
 * (325 frames/s/core: libm sin/cos per sample, radix-2 FFT, int32 Viterbi) was the only CPU row.  This file is the
 * same path written the way such a CPU implementation is written:
 *   NCO        one sincos per symbol, then a 16-sample phasor table times a block recurrence (no libm per sample)
 *   FFT        2048 = 32 x 64 four-step: 64 column transforms of 32 points and 32 of 64, every butterfly a loop over a
 *              contiguous row of 64 / 32 floats in split re / im arrays (vectorises to AVX2 / AVX-512 as compiled)
 *   DQPSK      on the whole spectrum, then gather through the frequency de-interleaver + L-infinity quantiser
 *   Viterbi    K = 7 add-compare-select on 64 x 16-bit saturating path metrics in four AVX2 registers, error metrics
 *              |soft - expected| per branch, renormalised every 8 steps, byte-mask decisions, traceback from state 0
 *              (the scheme of Karn / Spiral decoders and of the `viterbi` package the reference links)
 * Soft bits follow the oracle's convention (+127 = 1, trunc(-127 c / max)); decoded bytes are NOT bit-identical to the
 * oracle's on noise (different metric, ties) -- tests hold this file to the TRANSMITTED data instead.
 * Built `-O3 -march=native -ffast-math` on the box that runs it (oracle/Makefile target `simd`).
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "dab_oracle.h"

#if defined(__AVX2__)
#include <immintrin.h>
#endif

#define N 2048
#define N1 32   /* n = 64 n1 + n2,  k = k1 + 32 k2 */
#define N2 64

static pthread_once_t once = PTHREAD_ONCE_INIT;
static float tw32_re[N1 / 2], tw32_im[N1 / 2];      /* exp(-2 pi i m / 32) */
static float tw64_re[N2 / 2], tw64_im[N2 / 2];
static float twm_re[N1 * N2], twm_im[N1 * N2];      /* W2048^(n2 k1) at [k1][n2] */
static int bitrev32[N1], bitrev64[N2];
static int bin_of_n[DAB_NB_CARRIERS];               /* FFT bin of data index n */
static uint8_t fic_mask[DAB_NB_FIC_MOTHER];
static uint8_t prbs_bits[8192];
static uint8_t branch_exp[4][32];                   /* expected output bit r of the transition (state j, input 0) */

static int rev_bits(int v, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

static void init_tables(void)
{
    for (int m = 0; m < N1 / 2; m++) { tw32_re[m] = (float)cos(2 * M_PI * m / N1); tw32_im[m] = (float)-sin(2 * M_PI * m / N1); }
    for (int m = 0; m < N2 / 2; m++) { tw64_re[m] = (float)cos(2 * M_PI * m / N2); tw64_im[m] = (float)-sin(2 * M_PI * m / N2); }
    for (int k1 = 0; k1 < N1; k1++)
        for (int n2 = 0; n2 < N2; n2++) {
            twm_re[k1 * N2 + n2] = (float)cos(2 * M_PI * (double)(n2 * k1) / N);
            twm_im[k1 * N2 + n2] = (float)-sin(2 * M_PI * (double)(n2 * k1) / N);
        }
    for (int i = 0; i < N1; i++) bitrev32[i] = rev_bits(i, 5);
    for (int i = 0; i < N2; i++) bitrev64[i] = rev_bits(i, 6);
    int32_t map[DAB_NB_CARRIERS];
    oracle_get_mapper(map);
    for (int n = 0; n < DAB_NB_CARRIERS; n++) bin_of_n[n] = oracle_carrier_bin(map[n]);
    oracle_fic_puncture_mask(fic_mask);
    oracle_prbs(prbs_bits, (int)sizeof(prbs_bits));
    static const int polys[4] = {109, 79, 83, 109};
    for (int r = 0; r < 4; r++)
        for (int j = 0; j < 32; j++)
            branch_exp[r][j] = (uint8_t)__builtin_parity((unsigned)(polys[r] & (j << 1)));
}

/* in-place decimation-in-time transform of `n` rows of width `w` (row = one "sample", the `w` columns are independent
   transforms); rows arrive bit-reversed.  Radix-4 passes (two radix-2 levels fused: half the trips through the data),
   one radix-2 pass first when log2(n) is odd.  The column loop is the vector loop.  tw = exp(-2 pi i m / n), m < n/2. */
static void row_fft(float *restrict re, float *restrict im, int n, int w, const float *twr, const float *twi)
{
    int half = 1;
    int lg = 0;
    while ((1 << lg) < n) lg++;
    if (lg & 1) {                                            /* radix-2 level with trivial twiddles */
        for (int base = 0; base < n; base += 2) {
            float *restrict ar = re + (size_t)base * w, *restrict ai = im + (size_t)base * w;
            float *restrict br = ar + w, *restrict bi = ai + w;
#pragma GCC ivdep
            for (int x = 0; x < w; x++) {
                const float tr = br[x], ti = bi[x];
                br[x] = ar[x] - tr; bi[x] = ai[x] - ti;
                ar[x] += tr; ai[x] += ti;
            }
        }
        half = 2;
    }
    for (; half < n; half <<= 2) {
        /* levels `half` and `2 half` together: rows j, j+half, j+2half, j+3half of every block of 4 half */
        const int step1 = n / (2 * half), step2 = n / (4 * half);
        for (int base = 0; base < n; base += 4 * half)
            for (int j = 0; j < half; j++) {
                const float c1 = twr[j * step1], s1 = twi[j * step1];               /* W_{2half}^j   */
                const float c2 = twr[j * step2], s2 = twi[j * step2];               /* W_{4half}^j   */
                const float c3 = twr[(j + half) * step2], s3 = twi[(j + half) * step2];   /* W_{4half}^{j+half} */
                float *restrict r0 = re + (size_t)(base + j) * w, *restrict i0 = im + (size_t)(base + j) * w;
                float *restrict r1 = r0 + (size_t)half * w, *restrict i1 = i0 + (size_t)half * w;
                float *restrict r2 = r1 + (size_t)half * w, *restrict i2 = i1 + (size_t)half * w;
                float *restrict r3 = r2 + (size_t)half * w, *restrict i3 = i2 + (size_t)half * w;
#pragma GCC ivdep
                for (int x = 0; x < w; x++) {
                    /* first level: (0,1) and (2,3) with W_{2half}^j */
                    const float t1r = r1[x] * c1 - i1[x] * s1, t1i = r1[x] * s1 + i1[x] * c1;
                    const float t3r = r3[x] * c1 - i3[x] * s1, t3i = r3[x] * s1 + i3[x] * c1;
                    const float a0r = r0[x] + t1r, a0i = i0[x] + t1i, a1r = r0[x] - t1r, a1i = i0[x] - t1i;
                    const float a2r = r2[x] + t3r, a2i = i2[x] + t3i, a3r = r2[x] - t3r, a3i = i2[x] - t3i;
                    /* second level: (a0,a2) with W_{4half}^j, (a1,a3) with W_{4half}^{j+half} */
                    const float u2r = a2r * c2 - a2i * s2, u2i = a2r * s2 + a2i * c2;
                    const float u3r = a3r * c3 - a3i * s3, u3i = a3r * s3 + a3i * c3;
                    r0[x] = a0r + u2r; i0[x] = a0i + u2i;
                    r2[x] = a0r - u2r; i2[x] = a0i - u2i;
                    r1[x] = a1r + u3r; i1[x] = a1i + u3i;
                    r3[x] = a1r - u3r; i3[x] = a1i - u3i;
                }
            }
    }
}

typedef struct {
    float a_re[N], a_im[N], b_re[N], b_im[N];
    float prev_re[N], prev_im[N];
} fft_ws;

/* X = FFT2048(x): x in split form at ws->a (natural order), result in ws->b (natural order) */
static void fft2048(fft_ws *ws)
{
    /* pass A: 64 transforms of 32 points over n1 (rows of 64): bit-reverse the rows into b */
    for (int n1 = 0; n1 < N1; n1++) {
        memcpy(ws->b_re + bitrev32[n1] * N2, ws->a_re + n1 * N2, sizeof(float) * N2);
        memcpy(ws->b_im + bitrev32[n1] * N2, ws->a_im + n1 * N2, sizeof(float) * N2);
    }
    row_fft(ws->b_re, ws->b_im, N1, N2, tw32_re, tw32_im);
    /* twiddle W2048^(n2 k1), transpose [k1][n2] -> [n2][k1] with the rows bit-reversed for pass B */
    for (int k1 = 0; k1 < N1; k1++) {
        const float *restrict xr = ws->b_re + k1 * N2, *restrict xi = ws->b_im + k1 * N2;
        const float *restrict cr = twm_re + k1 * N2, *restrict ci = twm_im + k1 * N2;
        for (int n2 = 0; n2 < N2; n2++) {
            const float r = xr[n2] * cr[n2] - xi[n2] * ci[n2], i = xr[n2] * ci[n2] + xi[n2] * cr[n2];
            ws->a_re[bitrev64[n2] * N1 + k1] = r;
            ws->a_im[bitrev64[n2] * N1 + k1] = i;
        }
    }
    /* pass B: 32 transforms of 64 points over n2 (rows of 32); X[k1 + 32 k2] lands at [k2][k1] = natural order */
    row_fft(ws->a_re, ws->a_im, N2, N1, tw64_re, tw64_im);
    memcpy(ws->b_re, ws->a_re, sizeof(float) * N);
    memcpy(ws->b_im, ws->a_im, sizeof(float) * N);
}

void simd_ofdm_demod_frame(const float *iq, float freq_offset, int8_t *soft)
{
    pthread_once(&once, init_tables);
    fft_ws *ws = (fft_ws *)aligned_alloc(64, sizeof(fft_ws));
    const int32_t dphi = (int32_t)lrint((double)freq_offset * 4294967296.0);
    /* 16-sample phasor table and the block step */
    float t_re[16], t_im[16];
    for (int i = 0; i < 16; i++) {
        const double a = 2.0 * M_PI * (double)((uint32_t)i * (uint32_t)dphi) / 4294967296.0;
        t_re[i] = (float)cos(a); t_im[i] = (float)sin(a);
    }
    const double a16 = 2.0 * M_PI * (double)((uint32_t)16 * (uint32_t)dphi) / 4294967296.0;
    const float s_re = (float)cos(a16), s_im = (float)sin(a16);
    for (int l = 0; l < DAB_NB_FRAME_SYMBOLS; l++) {
        const float *x = iq + 2 * ((size_t)l * DAB_NB_SYM_PERIOD + DAB_NB_CP);
        /* phasor of the symbol's first useful sample from the exact phase, then the recurrence */
        const uint32_t n0 = (uint32_t)(l * DAB_NB_SYM_PERIOD + DAB_NB_CP);
        const double a0 = 2.0 * M_PI * (double)(n0 * (uint32_t)dphi) / 4294967296.0;
        float w_re = (float)cos(a0), w_im = (float)sin(a0);
        for (int blk = 0; blk < N / 16; blk++) {
            for (int i = 0; i < 16; i++) {
                const float pr = w_re * t_re[i] - w_im * t_im[i], pi = w_re * t_im[i] + w_im * t_re[i];
                const float xr = x[2 * (blk * 16 + i)], xi = x[2 * (blk * 16 + i) + 1];
                ws->a_re[blk * 16 + i] = xr * pr - xi * pi;
                ws->a_im[blk * 16 + i] = xr * pi + xi * pr;
            }
            const float nr = w_re * s_re - w_im * s_im, ni = w_re * s_im + w_im * s_re;
            w_re = nr; w_im = ni;
        }
        fft2048(ws);
        if (l > 0) {
            /* d = X_l conj X_{l-1} on every bin, then the de-interleaved carriers are quantised */
            float *restrict dr = ws->a_re, *restrict di = ws->a_im;
            for (int b = 0; b < N; b++) {
                dr[b] = ws->b_re[b] * ws->prev_re[b] + ws->b_im[b] * ws->prev_im[b];
                di[b] = ws->b_im[b] * ws->prev_re[b] - ws->b_re[b] * ws->prev_im[b];
            }
            int8_t *o = soft + (size_t)(l - 1) * DAB_NB_SYM_BITS;
            for (int n = 0; n < DAB_NB_CARRIERS; n++) {
                const float r = dr[bin_of_n[n]], i = di[bin_of_n[n]];
                const float ar = fabsf(r), ai = fabsf(i);
                const float m = ar > ai ? ar : ai;
                const float sc = m > 0.0f ? -127.0f / m : 0.0f;
                o[n] = (int8_t)(r * sc);
                o[DAB_NB_CARRIERS + n] = (int8_t)(i * sc);
            }
        }
        memcpy(ws->prev_re, ws->b_re, sizeof(float) * N);
        memcpy(ws->prev_im, ws->b_im, sizeof(float) * N);
    }
    free(ws);
}

/* ---- Viterbi: 64 x u16 path metrics, error metrics, decisions as 16-bit masks ---- */
#define VIT_MAX_STEPS 42000
typedef struct {
    uint16_t dec[2][32];            /* [input bit b][j]: 0xFFFF = the survivor into state 2j+b came from j + 32 */
} vit_dec;

void simd_viterbi(const int8_t *mother, int nsteps, uint8_t *out_bits, void *scratch)
{
    vit_dec *dec = (vit_dec *)scratch;
#if defined(__AVX2__)
    __m256i m[4];                                           /* states 0..15, 16..31, 32..47, 48..63 */
    m[0] = _mm256_set1_epi16(20000);
    m[0] = _mm256_insert_epi16(m[0], 0, 0);
    m[1] = m[2] = m[3] = _mm256_set1_epi16(20000);
    __m256i exp_v[4][2];                                    /* expected symbol (+127 / -127 as 0 / 254 offset) */
    for (int r = 0; r < 4; r++)
        for (int h = 0; h < 2; h++) {
            uint16_t e[16];
            for (int j = 0; j < 16; j++) e[j] = branch_exp[r][16 * h + j] ? 254 : 0;   /* soft + 127 in 0..254 */
            exp_v[r][h] = _mm256_loadu_si256((const __m256i *)e);
        }
    const __m256i maxerr = _mm256_set1_epi16(4 * 254);
    for (int t = 0; t < nsteps; t++) {
        __m256i e0 = _mm256_setzero_si256(), e1 = _mm256_setzero_si256();
        int erased = 0;
        for (int r = 0; r < 4; r++) {
            const int s = mother[4 * t + r];
            if (s == 0) { erased++; continue; }             /* punctured: the same cost on every branch */
            const __m256i sv = _mm256_set1_epi16((short)(s + 127));
            e0 = _mm256_add_epi16(e0, _mm256_abs_epi16(_mm256_sub_epi16(sv, exp_v[r][0])));
            e1 = _mm256_add_epi16(e1, _mm256_abs_epi16(_mm256_sub_epi16(sv, exp_v[r][1])));
        }
        const __m256i mx = _mm256_sub_epi16(maxerr, _mm256_set1_epi16((short)(254 * erased)));
        const __m256i c0 = _mm256_sub_epi16(mx, e0), c1 = _mm256_sub_epi16(mx, e1);
        /* j = 0..15 (h = 0) and 16..31 (h = 1); old states j (m[h]) and j + 32 (m[2 + h]) */
        __m256i a[2], b[2];
        const __m256i ee[2] = {e0, e1}, cc[2] = {c0, c1};
        for (int h = 0; h < 2; h++) {
            const __m256i m0 = _mm256_adds_epu16(m[h], ee[h]), m1 = _mm256_adds_epu16(m[2 + h], cc[h]);
            const __m256i m2 = _mm256_adds_epu16(m[h], cc[h]), m3 = _mm256_adds_epu16(m[2 + h], ee[h]);
            a[h] = _mm256_min_epu16(m0, m1);
            b[h] = _mm256_min_epu16(m2, m3);
            _mm256_storeu_si256((__m256i *)&dec[t].dec[0][16 * h], _mm256_cmpeq_epi16(a[h], m1));
            _mm256_storeu_si256((__m256i *)&dec[t].dec[1][16 * h], _mm256_cmpeq_epi16(b[h], m3));
        }
        /* new state 2j + b: interleave a (b = 0) and b (b = 1) */
        for (int h = 0; h < 2; h++) {
            const __m256i lo = _mm256_unpacklo_epi16(a[h], b[h]), hi = _mm256_unpackhi_epi16(a[h], b[h]);
            m[2 * h] = _mm256_permute2x128_si256(lo, hi, 0x20);
            m[2 * h + 1] = _mm256_permute2x128_si256(lo, hi, 0x31);
        }
        if ((t & 7) == 7) {                                 /* renormalise: subtract the smallest metric */
            __m256i mn = _mm256_min_epu16(_mm256_min_epu16(m[0], m[1]), _mm256_min_epu16(m[2], m[3]));
            __m128i q = _mm_min_epu16(_mm256_castsi256_si128(mn), _mm256_extracti128_si256(mn, 1));
            q = _mm_minpos_epu16(q);
            const __m256i sub = _mm256_set1_epi16((short)_mm_extract_epi16(q, 0));
            for (int i = 0; i < 4; i++) m[i] = _mm256_subs_epu16(m[i], sub);
        }
    }
#else
    uint16_t m[64], nm[64];
    for (int s = 0; s < 64; s++) m[s] = s ? 20000 : 0;
    for (int t = 0; t < nsteps; t++) {
        for (int j = 0; j < 32; j++) {
            int e = 0, mx = 0;
            for (int r = 0; r < 4; r++) {
                const int s = mother[4 * t + r];
                if (s == 0) continue;
                e += abs(s + 127 - (branch_exp[r][j] ? 254 : 0));
                mx += 254;
            }
            const int c = mx - e;
            const unsigned m0 = m[j] + e, m1 = m[j + 32] + c, m2 = m[j] + c, m3 = m[j + 32] + e;
            nm[2 * j] = (uint16_t)(m0 < m1 ? m0 : m1);
            nm[2 * j + 1] = (uint16_t)(m2 < m3 ? m2 : m3);
            dec[t].dec[0][j] = m1 <= m0 ? 0xFFFF : 0;
            dec[t].dec[1][j] = m3 <= m2 ? 0xFFFF : 0;
        }
        unsigned mn = 65535;
        for (int s = 0; s < 64; s++) if (nm[s] < mn) mn = nm[s];
        for (int s = 0; s < 64; s++) m[s] = (uint16_t)(nm[s] - mn);
    }
#endif
    /* traceback from state 0: the input bit of step t is the LSB of the state reached */
    int s = 0;
    for (int t = nsteps - 1; t >= 0; t--) {
        const int b = s & 1, j = s >> 1;
        if (t < nsteps - 6) out_bits[t] = (uint8_t)b;
        s = dec[t].dec[b][j] ? j + 32 : j;
    }
}

static void depuncture(const int8_t *punct, const uint8_t *mask, int n_mother, int8_t *mother)
{
    int k = 0;
    for (int i = 0; i < n_mother; i++) mother[i] = mask[i] ? punct[k++] : 0;
}

static void bits_to_bytes_descrambled(const uint8_t *bits, int nbits, uint8_t *out)
{
    for (int i = 0; i < nbits / 8; i++) {
        unsigned v = 0;
        for (int k = 0; k < 8; k++) v = (v << 1) | (unsigned)(bits[8 * i + k] ^ prbs_bits[8 * i + k]);
        out[i] = (uint8_t)v;
    }
}

void simd_fic_decode(const int8_t *soft9216, uint8_t *fib384, uint8_t *crc_ok12)
{
    pthread_once(&once, init_tables);
    int8_t mother[DAB_NB_FIC_MOTHER];
    uint8_t bits[DAB_NB_FIC_STEPS];
    void *scratch = malloc(sizeof(vit_dec) * DAB_NB_FIC_STEPS);
    for (int g = 0; g < DAB_NB_FIC_GROUPS; g++) {
        depuncture(soft9216 + g * DAB_NB_FIC_GROUP_BITS, fic_mask, DAB_NB_FIC_MOTHER, mother);
        simd_viterbi(mother, DAB_NB_FIC_STEPS, bits, scratch);
        bits_to_bytes_descrambled(bits, 768, fib384 + 96 * g);
    }
    for (int f = 0; f < DAB_NB_FIBS; f++) {
        const uint8_t *p = fib384 + 32 * f;
        crc_ok12[f] = (uint8_t)(oracle_crc16(p, 30) == (uint16_t)((p[30] << 8) | p[31]));
    }
    free(scratch);
}

void simd_msc_decode_lf(const int8_t *deint, const uint8_t *mask, int nsteps, uint8_t *out_bytes)
{
    pthread_once(&once, init_tables);
    int8_t *mother = (int8_t *)malloc((size_t)4 * nsteps);
    uint8_t *bits = (uint8_t *)malloc((size_t)nsteps);
    void *scratch = malloc(sizeof(vit_dec) * (size_t)nsteps);
    depuncture(deint, mask, 4 * nsteps, mother);
    simd_viterbi(mother, nsteps, bits, scratch);
    bits_to_bytes_descrambled(bits, nsteps - 6, out_bytes);
    free(mother); free(bits); free(scratch);
}

/* which instruction set this object was compiled for (reported beside the numbers) */
const char *simd_port_isa(void)
{
#if defined(__AVX512F__)
    return "avx512f (FFT loops) + avx2 (Viterbi)";
#elif defined(__AVX2__)
    return "avx2";
#else
    return "scalar (no AVX2 at compile time)";
#endif
}
