/*
 * oracle_bench.c -- multi-threaded timing harness around the CPU oracle (TEST INFRASTRUCTURE).
 * Used only by bench.py's cpu_baseline leg: every thread runs the whole per-frame hot path
 * (OFDM demod + FIC + 4 MSC logical frames of one subchannel) on its share of frames.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "dab_oracle.h"

/* The same harness times two implementations: the oracle (default) and, compiled a second time with -DBENCH_SIMD into
   libsimdport.so, the SIMD port of simd_port.c (entry points simd_bench_*). */
#ifdef BENCH_SIMD
void simd_ofdm_demod_frame(const float *iq, float freq_offset, int8_t *soft);
void simd_fic_decode(const int8_t *soft9216, uint8_t *fib384, uint8_t *crc_ok12);
void simd_msc_decode_lf(const int8_t *deint, const uint8_t *mask, int nsteps, uint8_t *out_bytes);
#define DEMOD(iq, fo, soft) simd_ofdm_demod_frame(iq, fo, soft)
#define FIC(soft, fib, ok) simd_fic_decode(soft, fib, ok)
#define MSC_LF(deint, mask, nsteps, out) simd_msc_decode_lf(deint, mask, nsteps, out)
#define NAME(x) simd_##x
#else
#define DEMOD(iq, fo, soft) oracle_ofdm_demod_frame(iq, fo, soft, NULL, NULL, NULL)
#define FIC(soft, fib, ok) oracle_fic_decode(soft, fib, ok)
#define MSC_LF(deint, mask, nsteps, out) oracle_msc_decode_lf(deint, mask, nsteps, out)
#define NAME(x) oracle_##x
#endif

typedef struct {
    const float *iq;
    size_t stride;          /* complex samples between frames */
    const float *fo;
    int n_frames;           /* distinct input frames (cycled) */
    int first, count;       /* work items of this thread (count < 0: run until *deadline) */
    double deadline;        /* CLOCK_MONOTONIC seconds; used when count < 0 */
    int done;               /* frames actually processed */
    const uint8_t *mask;
    int nsteps, sc_bits;
    unsigned checksum;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    int8_t *soft = (int8_t *)malloc(DAB_NB_FRAME_BITS);
    int8_t *deint = (int8_t *)malloc((size_t)j->sc_bits);
    uint8_t fib[384], ok[12];
    uint8_t *out = (uint8_t *)malloc((size_t)(j->nsteps - 6) / 8);
    for (int k = 0; j->count < 0 || k < j->count; k++) {
        if (j->count < 0 && (k & 3) == 0) {
            struct timespec now;
            clock_gettime(CLOCK_MONOTONIC, &now);
            if ((double)now.tv_sec + 1e-9 * (double)now.tv_nsec >= j->deadline) break;
        }
        const int f = (j->first + k) % j->n_frames;
        DEMOD(j->iq + 2 * (size_t)f * j->stride, j->fo ? j->fo[f] : 0.0f, soft);
        FIC(soft, fib, ok);
        for (int c = 0; c < DAB_NB_CIFS; c++) {
            /* the frame's own 4 CIFs stand in for the 16-CIF window: same arithmetic, same memory traffic */
            const int8_t *cifs[16];
            for (int i = 0; i < 16; i++)
                cifs[i] = soft + DAB_NB_FIC_BITS + (size_t)((c + i) & 3) * DAB_NB_CIF_BITS;
            oracle_time_deinterleave(cifs, j->sc_bits, deint);
            MSC_LF(deint, j->mask, j->nsteps, out);
            j->checksum += out[0];
        }
        j->checksum += fib[0] + ok[0];
        j->done++;
    }
    free(soft); free(deint); free(out);
    return NULL;
}

/* returns elapsed seconds for `total` frames over `threads` threads */
double NAME(bench_frames)(const float *iq, size_t stride, const float *fo, int n_frames, int total, int threads,
                           const uint8_t *mask, int nsteps, int sc_bits)
{
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    job_t *jobs = (job_t *)calloc((size_t)threads, sizeof(job_t));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int next = 0;
    for (int t = 0; t < threads; t++) {
        const int cnt = total / threads + (t < total % threads ? 1 : 0);
        jobs[t] = (job_t){iq, stride, fo, n_frames, next, cnt, 0.0, 0, mask, nsteps, sc_bits, 0};
        next += cnt;
        pthread_create(&tid[t], NULL, worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(tid); free(jobs);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* time-bounded variant: every thread processes frames until `seconds` have passed; returns elapsed seconds and
   stores the total number of frames completed in *frames_done */
double NAME(bench_frames_timed)(const float *iq, size_t stride, const float *fo, int n_frames, double seconds,
                                 int threads, const uint8_t *mask, int nsteps, int sc_bits, long *frames_done)
{
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    job_t *jobs = (job_t *)calloc((size_t)threads, sizeof(job_t));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const double deadline = (double)t0.tv_sec + 1e-9 * (double)t0.tv_nsec + seconds;
    for (int t = 0; t < threads; t++) {
        jobs[t] = (job_t){iq, stride, fo, n_frames, t, -1, deadline, 0, mask, nsteps, sc_bits, 0};
        pthread_create(&tid[t], NULL, worker, &jobs[t]);
    }
    long total = 0;
    for (int t = 0; t < threads; t++) { pthread_join(tid[t], NULL); total += jobs[t].done; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *frames_done = total;
    free(tid); free(jobs);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------------------------------------------
 * BASELINE config 1 ("OFDM demod only on the CPU path") and the plugin's deployment shape
 * (/root/reference/src/dab_module.cpp:92: Radio_Block(1, 1) = one OFDM thread; /root/reference/src/radio_block.cpp:23-44:
 * a 2-frame ThreadedRingBuffer feeding one decoder thread).
 * ------------------------------------------------------------------------------------------------ */

/* one thread, front end only; returns elapsed seconds, *frames_done frames */
double NAME(bench_ofdm_only_timed)(const float *iq, size_t stride, const float *fo, int n_frames, double seconds,
                                    long *frames_done)
{
    int8_t *soft = (int8_t *)malloc(DAB_NB_FRAME_BITS);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const double deadline = (double)t0.tv_sec + 1e-9 * (double)t0.tv_nsec + seconds;
    long k = 0;
    unsigned checksum = 0;
    for (;; k++) {
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((double)t1.tv_sec + 1e-9 * (double)t1.tv_nsec >= deadline) break;
        const int f = (int)(k % n_frames);
        DEMOD(iq + 2 * (size_t)f * stride, fo ? fo[f] : 0.0f, soft);
        checksum += (unsigned)soft[17];
    }
    *frames_done = k + (checksum == 0xFFFFFFFFu ? 1 : 0) * 0;
    free(soft);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

typedef struct {
    int8_t *slot[2];
    int head, tail, count, closed;      /* ring of two frames, as ThreadedRingBuffer(2 * nb_frame_bits) */
    pthread_mutex_t mu;
    pthread_cond_t not_full, not_empty;
    /* producer */
    const float *iq;
    size_t stride;
    const float *fo;
    int n_frames;
    double deadline;
    /* consumer */
    const uint8_t *mask;
    int nsteps, sc_bits;
    long decoded;
    unsigned checksum;
} pipe_t;

static void *pipe_ofdm_thread(void *arg)
{
    pipe_t *p = (pipe_t *)arg;
    int8_t *soft = (int8_t *)malloc(DAB_NB_FRAME_BITS);
    for (long k = 0;; k++) {
        struct timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        if ((double)now.tv_sec + 1e-9 * (double)now.tv_nsec >= p->deadline) break;
        const int f = (int)(k % p->n_frames);
        DEMOD(p->iq + 2 * (size_t)f * p->stride, p->fo ? p->fo[f] : 0.0f, soft);
        pthread_mutex_lock(&p->mu);
        while (p->count == 2) pthread_cond_wait(&p->not_full, &p->mu);     /* the writer blocks: back-pressure */
        memcpy(p->slot[p->head], soft, DAB_NB_FRAME_BITS);
        p->head ^= 1;
        p->count++;
        pthread_cond_signal(&p->not_empty);
        pthread_mutex_unlock(&p->mu);
    }
    pthread_mutex_lock(&p->mu);
    p->closed = 1;
    pthread_cond_signal(&p->not_empty);
    pthread_mutex_unlock(&p->mu);
    free(soft);
    return NULL;
}

static void *pipe_decoder_thread(void *arg)
{
    pipe_t *p = (pipe_t *)arg;
    int8_t *soft = (int8_t *)malloc(DAB_NB_FRAME_BITS);
    int8_t *deint = (int8_t *)malloc((size_t)p->sc_bits);
    uint8_t fib[384], ok[12];
    uint8_t *out = (uint8_t *)malloc((size_t)(p->nsteps - 6) / 8);
    for (;;) {
        pthread_mutex_lock(&p->mu);
        while (p->count == 0 && !p->closed) pthread_cond_wait(&p->not_empty, &p->mu);
        if (p->count == 0 && p->closed) { pthread_mutex_unlock(&p->mu); break; }
        memcpy(soft, p->slot[p->tail], DAB_NB_FRAME_BITS);
        p->tail ^= 1;
        p->count--;
        pthread_cond_signal(&p->not_full);
        pthread_mutex_unlock(&p->mu);
        FIC(soft, fib, ok);
        for (int c = 0; c < DAB_NB_CIFS; c++) {
            const int8_t *cifs[16];
            for (int i = 0; i < 16; i++)
                cifs[i] = soft + DAB_NB_FIC_BITS + (size_t)((c + i) & 3) * DAB_NB_CIF_BITS;
            oracle_time_deinterleave(cifs, p->sc_bits, deint);
            MSC_LF(deint, p->mask, p->nsteps, out);
            p->checksum += out[0];
        }
        p->checksum += fib[0] + ok[0];
        p->decoded++;
    }
    free(soft); free(deint); free(out);
    return NULL;
}

/* one OFDM thread -> 2-frame ring -> one decoder thread; returns elapsed seconds, *frames_done decoded frames */
double NAME(bench_pipeline_timed)(const float *iq, size_t stride, const float *fo, int n_frames, double seconds,
                                   const uint8_t *mask, int nsteps, int sc_bits, long *frames_done)
{
    pipe_t p;
    memset(&p, 0, sizeof(p));
    p.slot[0] = (int8_t *)malloc(DAB_NB_FRAME_BITS);
    p.slot[1] = (int8_t *)malloc(DAB_NB_FRAME_BITS);
    pthread_mutex_init(&p.mu, NULL);
    pthread_cond_init(&p.not_full, NULL);
    pthread_cond_init(&p.not_empty, NULL);
    p.iq = iq; p.stride = stride; p.fo = fo; p.n_frames = n_frames;
    p.mask = mask; p.nsteps = nsteps; p.sc_bits = sc_bits;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    p.deadline = (double)t0.tv_sec + 1e-9 * (double)t0.tv_nsec + seconds;
    pthread_t a, b;
    pthread_create(&a, NULL, pipe_ofdm_thread, &p);
    pthread_create(&b, NULL, pipe_decoder_thread, &p);
    pthread_join(a, NULL);
    pthread_join(b, NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *frames_done = p.decoded;
    free(p.slot[0]); free(p.slot[1]);
    pthread_mutex_destroy(&p.mu);
    pthread_cond_destroy(&p.not_full);
    pthread_cond_destroy(&p.not_empty);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
