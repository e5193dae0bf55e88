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
        oracle_ofdm_demod_frame(j->iq + 2 * (size_t)f * j->stride, j->fo ? j->fo[f] : 0.0f, soft, NULL, NULL, NULL);
        oracle_fic_decode(soft, fib, ok);
        for (int c = 0; c < DAB_NB_CIFS; c++) {
            /* the frame's own 4 CIFs stand in for the 16-CIF window: same arithmetic, same memory traffic */
            const int8_t *cifs[16];
            for (int i = 0; i < 16; i++)
                cifs[i] = soft + DAB_NB_FIC_BITS + (size_t)((c + i) & 3) * DAB_NB_CIF_BITS;
            oracle_time_deinterleave(cifs, j->sc_bits, deint);
            oracle_msc_decode_lf(deint, j->mask, j->nsteps, out);
            j->checksum += out[0];
        }
        j->checksum += fib[0] + ok[0];
        j->done++;
    }
    free(soft); free(deint); free(out);
    return NULL;
}

/* returns elapsed seconds for `total` frames over `threads` threads */
double oracle_bench_frames(const float *iq, size_t stride, const float *fo, int n_frames, int total, int threads,
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
double oracle_bench_frames_timed(const float *iq, size_t stride, const float *fo, int n_frames, double seconds,
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
