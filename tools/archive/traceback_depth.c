/* How often does a sliding-window (truncated) traceback differ from the full traceback?
 *
 * The channel decoder's survivor words make a round trip through HBM because its traceback starts at the end of the
 * codeword (state 0 after the six tail bits) -- the bytes the oracle defines.  A windowed traceback (start from the
 * best state L steps ahead, emit the bits older than that) would keep the survivors on chip, but it is only
 * bit-identical when every start state's path has merged with the final path within L steps.  This program counts how
 * often that fails, with the oracle's recursion (K = 7, rate 1/4, generators 133 171 145 133, int32 correlation
 * metric, ties to the lower predecessor) on (a) pure noise -- the inputs the GPU parity tests decode -- and
 * (b) transmitted codewords in noise, for several puncturing densities.
 *
 * build + run: gcc -O2 -o /tmp/traceback_depth tools/traceback_depth.c && /tmp/traceback_depth
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const unsigned POLY[4] = {109, 79, 83, 109};
static unsigned parity7(unsigned x) { x ^= x >> 4; x ^= x >> 2; x ^= x >> 1; return x & 1u; }

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 33); }
static double gauss(void) { double s = 0; for (int i = 0; i < 12; i++) s += rnd() / 2147483648.0; return s - 6.0; }

#define NSTEPS 1542                      /* one 64 kbit/s EEP 3-A codeword (the bench's sub-channel) */
static uint64_t dec[NSTEPS];
static int32_t metric_at[NSTEPS][64];   /* path metrics after every step (to find the best state) */

static void forward(const int8_t *soft)
{
    int32_t ma[64], mb[64], *old = ma, *cur = mb;
    int sgn[32][4];
    for (int j = 0; j < 32; j++)
        for (int p = 0; p < 4; p++) sgn[j][p] = parity7(((unsigned)j << 1) & POLY[p]) ? 1 : -1;
    old[0] = 0;
    for (int s = 1; s < 64; s++) old[s] = -8192;
    for (int t = 0; t < NSTEPS; t++) {
        const int8_t *s4 = soft + 4 * t;
        uint64_t d = 0;
        for (int j = 0; j < 32; j++) {
            const int32_t c = sgn[j][0] * s4[0] + sgn[j][1] * s4[1] + sgn[j][2] * s4[2] + sgn[j][3] * s4[3];
            const int32_t a0 = old[j] + c, a1 = old[j + 32] - c, b0 = old[j] - c, b1 = old[j + 32] + c;
            const int d0 = a1 > a0, d1 = b1 > b0;
            cur[2 * j] = d0 ? a1 : a0;
            cur[2 * j + 1] = d1 ? b1 : b0;
            d |= (uint64_t)d0 << (2 * j);
            d |= (uint64_t)d1 << (2 * j + 1);
        }
        dec[t] = d;
        memcpy(metric_at[t], cur, sizeof(ma));
        int32_t *tmp = old; old = cur; cur = tmp;
    }
}

/* bits of steps [t_lo, t_hi] on the path that is in state s after step t_hi; returns the state before step t_lo */
static unsigned trace(unsigned s, int t_hi, int t_lo, uint8_t *bits)
{
    for (int t = t_hi; t >= t_lo; t--) {
        if (bits) bits[t] = (uint8_t)(s & 1u);
        const unsigned h = (unsigned)((dec[t] >> s) & 1u);
        s = (s >> 1) | (h << 5);
    }
    return s;
}

int main(void)
{
    const int depths[] = {32, 48, 64, 96, 128, 192, 256};
    const int ND = (int)(sizeof(depths) / sizeof(depths[0]));
    const int W = 64;                      /* bits emitted per window */
    const int N = 20000;                   /* codewords per case */
    const double erased[] = {0.0, 0.5, 0.72};       /* share of mother bits punctured: rate 1/4, 1/2, 8/9 */
    static int8_t soft[4 * NSTEPS];
    static uint8_t full[NSTEPS], win[NSTEPS], msg[NSTEPS], enc[4 * NSTEPS];
    printf("K=7 rate-1/4 mother code, %d-step codewords, %d codewords per case, window of %d bits\n", NSTEPS, N, W);
    printf("columns: traceback depth L; cell: codewords whose windowed output differs from the full traceback (differing bits)\n");
    for (int kind = 0; kind < 3; kind++)                        /* 0: noise; 1: codeword at -3 dB Es/N0; 2: at +3 dB */
        for (int e = 0; e < 3; e++) {
            long bad_cw[16] = {0}, bad_bits[16] = {0};
            for (int n = 0; n < N; n++) {
                if (kind == 0) {
                    for (int i = 0; i < 4 * NSTEPS; i++) soft[i] = (int8_t)((int)(rnd() % 255) - 127);
                } else {
                    unsigned reg = 0;
                    for (int i = 0; i < NSTEPS; i++) {
                        msg[i] = i < NSTEPS - 6 ? (uint8_t)(rnd() & 1u) : 0;
                        reg = ((reg << 1) | msg[i]) & 0x7F;
                        for (int p = 0; p < 4; p++) enc[4 * i + p] = (uint8_t)parity7(reg & POLY[p]);
                    }
                    const double sigma = kind == 1 ? 1.0 : 0.5;      /* per-dimension noise for unit-energy symbols */
                    for (int i = 0; i < 4 * NSTEPS; i++) {
                        double v = (enc[i] ? 1.0 : -1.0) + sigma * gauss();
                        v *= 48.0;
                        soft[i] = (int8_t)(v > 127 ? 127 : v < -127 ? -127 : v);
                    }
                }
                if (erased[e] > 0.0)
                    for (int i = 0; i < 4 * (NSTEPS - 6); i++)
                        if (rnd() / 2147483648.0 < erased[e]) soft[i] = 0;
                forward(soft);
                trace(0u, NSTEPS - 1, 0, full);
                for (int d = 0; d < ND; d++) {
                    const int L = depths[d];
                    /* windows of W bits, oldest first; the last windows (less than L steps from the end) use the tail */
                    for (int t0 = 0; t0 < NSTEPS; t0 += W) {
                        const int t1 = t0 + W - 1 < NSTEPS - 1 ? t0 + W - 1 : NSTEPS - 1;
                        const int ts = t1 + L;                   /* step whose best state starts the traceback */
                        unsigned s;
                        if (ts >= NSTEPS - 1) {
                            s = trace(0u, NSTEPS - 1, t1 + 1, NULL);
                        } else {
                            int best = 0;
                            for (int q = 1; q < 64; q++) if (metric_at[ts][q] > metric_at[ts][best]) best = q;
                            s = trace((unsigned)best, ts, t1 + 1, NULL);
                        }
                        trace(s, t1, t0, win);
                    }
                    long nb = 0;
                    for (int t = 0; t < NSTEPS - 6; t++) nb += full[t] != win[t];
                    bad_cw[d] += nb != 0;
                    bad_bits[d] += nb;
                }
            }
            printf("%-26s %2.0f %% punctured:", kind == 0 ? "noise" : kind == 1 ? "codeword, sigma 1.0" : "codeword, sigma 0.5", 100 * erased[e]);
            for (int d = 0; d < ND; d++) printf("  L=%d: %ld (%ld)", depths[d], bad_cw[d], bad_bits[d]);
            printf("\n");
            fflush(stdout);
        }
    return 0;
}
