/* c_abi_smoke -- the C ABI from plain C99 (what a cgo / JNI / FFI binding links against): create a context, run
 * the front end and the FIC decoder on an all-zero frame and on a frame of noise, check the documented results. */
#include <stdio.h>
#include <stdlib.h>
#include "dabgpu.h"

int main(void) {
    dabgpu_cfg cfg = {0, 2, 1, 0, 0, {0, 0, 0}};
    dabgpu_ctx *ctx = NULL;
    int rc = dabgpu_create(&cfg, &ctx);
    if (rc != DABGPU_OK) { printf("create: %s\n", dabgpu_strerror(rc)); return rc == DABGPU_ERR_NODEVICE ? 77 : 1; }
    const size_t n = 76 * 2552;
    float *iq = (float *)dabgpu_host_alloc(2 * n * 2 * sizeof(float));       /* two frames, page-locked */
    int8_t *soft = (int8_t *)malloc(2 * 230400);
    uint8_t fib[2 * 12 * 32], ok[2 * 12];
    unsigned seed = 12345u;
    size_t i;
    int bad = 0;
    if (!iq || !soft) return 1;
    for (i = 0; i < 2 * n; i++) iq[i] = 0.0f;                                /* frame 0: silence */
    for (i = 2 * n; i < 4 * n; i++) { seed = seed * 1664525u + 1013904223u; iq[i] = (float)(seed >> 8) / 8388608.0f - 1.0f; }
    rc = dabgpu_ofdm_demod_frames(ctx, iq, n, 2, NULL, soft, NULL, NULL);
    if (rc) { printf("demod: %s\n", dabgpu_strerror(rc)); return 1; }
    for (i = 0; i < 230400; i++) bad += soft[i] != 0;                        /* silence -> erasures */
    rc = dabgpu_fic_decode(ctx, soft, 230400, 2, fib, ok);
    if (rc) { printf("fic: %s\n", dabgpu_strerror(rc)); return 1; }
    for (i = 0; i < 12; i++) bad += ok[12 + i] != 0;                         /* noise never passes a CRC16 (1 in 65536) */
    /* the per-frame call of the host mirror (ABI v4): a frame of noise is "not a PRS" -- nothing demodulated, one desync */
    {
        dabgpu_track_cfg tcfg;
        dabgpu_frame_result res;
        dabgpu_track_default_cfg(&tcfg);
        rc = dabgpu_streams_reset(ctx, 1);
        if (!rc) rc = dabgpu_ofdm_demod_stream_frame(ctx, 0, iq + 2 * n, 1, &tcfg, soft, NULL, &res);
        if (rc) { printf("frame call: %s\n", dabgpu_strerror(rc)); return 1; }
        bad += (res.flags & 1) != 0;
        bad += res.stats.total_frames_desync != 1 || res.stats.total_frames_read != 0;
        for (i = 0; i < 230400; i++) bad += soft[i] != 0;
    }
    printf("c abi ok: abi=%d erasures_nonzero=%d\n", dabgpu_abi_version(), bad);
    dabgpu_host_free(iq);
    free(soft);
    dabgpu_destroy(ctx);
    return bad ? 1 : 0;
}
