/*
 * dabgpu.h -- C ABI of the MI355X-native DAB Mode-I receive chain (libdabgpu.so).
 *
 * This is the drop-in boundary for the ONE hot path of the SDR++ DAB plugin: the
 * OFDM front end and the channel decoder that sit behind Radio_Block / OFDM_Demod /
 * BasicRadio.  Every entry point names the reference interface it replaces
 * (file:line under /root/reference).  The DSP those interfaces front lives in git
 * submodules that are empty in the reference snapshot, so the citations are the
 * plugin's call sites.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types, no exceptions.
 *   - return 0 (DABGPU_OK) or a negative dabgpu_status; dabgpu_strerror() explains.
 *   - `_dev` entry points take DEVICE pointers and a hipStream_t passed as void*
 *     (NULL = the context's own stream); they only enqueue work.  The variants
 *     without `_dev` take HOST pointers, copy, run and synchronise.
 *   - the context's own stream is NON-BLOCKING (hipStreamNonBlocking): it is not
 *     ordered behind the null stream or any other.  Device buffers handed to a
 *     `_dev` call with stream == NULL must already hold their data (and no fill of
 *     an output buffer may still be running); a caller that produces them on a
 *     stream of its own passes THAT stream.
 *   - the caller owns every buffer; a context is thread-compatible (one caller
 *     at a time per context), contexts are independent.
 *   - work enqueued through ONE context must be stream-ordered: the entry points
 *     share per-context work buffers (decoder scratch, the stream / tracked / frame
 *     calls' loop input, the stream states), so AT MOST ONE caller stream may be in
 *     flight per context -- two calls on different streams need an event between
 *     them (or two contexts, as the host mirror uses: one for OFDM_Demod, one for
 *     BasicRadio).
 *   - soft bits are int8: +127 = logical 1, -127 = logical 0, 0 = erased
 *     (`viterbi_bit_t`, /root/reference/src/radio_block.h:19).
 *   - the library REQUIRES a gfx950 device for everything except the table
 *     getters; there is no CPU fallback.
 */
#ifndef DABGPU_H
#define DABGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DABGPU_ABI_VERSION 6   /* 6: every closed-loop call defaults to the reference's cyclic-prefix estimator (dabgpu_track_cfg.decision_directed */
                               /*    defaults to 0 and the frame call honours it); 5: one frame-buffer allocator, host-fed ring, loop gate           */

typedef enum dabgpu_status {
    DABGPU_OK = 0,
    DABGPU_ERR_ARG = -1,       /* null / out-of-range argument              */
    DABGPU_ERR_HIP = -2,       /* a HIP runtime call failed                 */
    DABGPU_ERR_NOMEM = -3,     /* device or host allocation failed          */
    DABGPU_ERR_NODEVICE = -4,  /* no gfx950 device visible                  */
    DABGPU_ERR_PROFILE = -5,   /* unsupported transmission mode / profile   */
    DABGPU_ERR_CAPACITY = -6   /* request exceeds what the context was sized for */
} dabgpu_status;

/* ------------------------------------------------------------------------ */
/* A0: constant parameter blocks.                                            */
/* Replaces get_DAB_OFDM_params(mode)   /root/reference/src/radio_block.cpp:12 */
/*          get_dab_parameters(mode)    /root/reference/src/radio_block.cpp:13 */
/* Only transmission mode 1 is supported (radio_block.cpp:9).                 */
/* ------------------------------------------------------------------------ */
typedef struct dabgpu_ofdm_params {
    int32_t nb_frame_symbols;     /* 76  */
    int32_t nb_symbol_period;     /* 2552 */
    int32_t nb_null_period;       /* 2656 */
    int32_t nb_fft;               /* 2048 */
    int32_t nb_cyclic_prefix;     /* 504 */
    int32_t nb_data_carriers;     /* 1536 */
    int32_t freq_carrier_spacing; /* 1000 Hz */
    int32_t nb_frame_samples;     /* 196608 */
} dabgpu_ofdm_params;

typedef struct dabgpu_dab_params {
    int32_t nb_frame_bits;    /* 230400 */
    int32_t nb_symbols;       /* 75 data symbols */
    int32_t nb_fic_symbols;   /* 3 */
    int32_t nb_msc_symbols;   /* 72 */
    int32_t nb_sym_bits;      /* 3072 */
    int32_t nb_fic_bits;      /* 9216 */
    int32_t nb_msc_bits;      /* 221184 */
    int32_t nb_fibs;          /* 12 */
    int32_t nb_cifs;          /* 4 */
    int32_t nb_fib_bits;      /* 256 */
    int32_t nb_fib_cif_bits;  /* 2304: one punctured FIC group */
    int32_t nb_fibs_per_cif;  /* 3 */
    int32_t nb_cif_bits;      /* 55296 */
} dabgpu_dab_params;

int dabgpu_get_ofdm_params(int transmission_mode, dabgpu_ofdm_params *out);
int dabgpu_get_dab_params(int transmission_mode, dabgpu_dab_params *out);

/* A0': replaces get_DAB_PRS_reference(mode, span<complex<float>>[nb_fft])
 * /root/reference/src/radio_block.cpp:18-19.  out = nb_fft interleaved (re,im). */
int dabgpu_get_prs_reference(int transmission_mode, float *out_cf32, int nb_fft);
/* A0': replaces get_DAB_mapper_ref(span<int>[nb_data_carriers], nb_fft)
 * /root/reference/src/radio_block.cpp:20-21. */
int dabgpu_get_mapper_reference(int32_t *out, int nb_data_carriers, int nb_fft);

/* ------------------------------------------------------------------------ */
/* Context                                                                   */
/* ------------------------------------------------------------------------ */
typedef struct dabgpu_ctx dabgpu_ctx;

typedef struct dabgpu_cfg {
    int32_t device;          /* HIP device ordinal                                    */
    int32_t max_frames;      /* largest n_frames any call will pass (scratch sizing)  */
    int32_t transmission_mode; /* must be 1                                           */
    int32_t flags;           /* DABGPU_FLAG_*                                         */
    int32_t ofdm_symbol_runs;  /* front end: cut every frame into this many runs of    */
                             /* consecutive symbols (one wavefront each), 1..75;      */
                             /* 0 = chosen from the batch size                        */
    int32_t reserved[3];     /* must be 0                                             */
} dabgpu_cfg;

/* Which kernels decode (results are identical; the default picks by batch size).  Meant for tests and timing. */
#define DABGPU_FLAG_NONE 0
#define DABGPU_FLAG_VITERBI_WAVE  (1 << 0) /* channel decoder: one wavefront per codeword, always (codewords    */
                                           /* above ~680 kbit/s do not fit its LDS slab: those go per lane)      */
#define DABGPU_FLAG_VITERBI_LANE  (1 << 1) /* one codeword per lane wherever the length allows, any batch size */
#define DABGPU_FLAG_LANE_UNFUSED  (1 << 2) /* lane decoder: separate depuncture pass before the forward pass   */
/* Test hooks (tests/ only; they change no result, they force a branch that hardware takes rarely): */
#define DABGPU_FLAG_TEST_ONE_DOMAIN (1 << 30) /* dabgpu_alloc_frame_buffers(DABGPU_PLACE_DOMAINS): the check of the  */
                                           /* placed pair is taken as 1.00 (a box whose chunks share one domain)   */

/* Replaces the construction in Radio_Block::Radio_Block
 * (/root/reference/src/radio_block.cpp:11-22: params + PRS + mapper + OFDM_Demod). */
int  dabgpu_create(const dabgpu_cfg *cfg, dabgpu_ctx **out);
void dabgpu_destroy(dabgpu_ctx *ctx);
const char *dabgpu_strerror(int status);
int  dabgpu_abi_version(void);
/* Entry points make cfg.device current for the duration of the call and restore the caller's device afterwards: a
 * process may hold contexts on several GPUs. */
/* block until everything enqueued on the context's own stream has finished */
int  dabgpu_sync(dabgpu_ctx *ctx);
/* the context's own hipStream_t (as void*) */
void *dabgpu_stream(dabgpu_ctx *ctx);

/* Page-locked host memory for the host-pointer entry points: buffers obtained here are copied to and from the
 * device by DMA at the full link rate, pageable ones go through the runtime's bounce buffers (measured on the
 * pool's PCIe Gen5 link: see DESIGN.md section 5).  Any host pointer is accepted everywhere; this is an
 * optimisation for callers that own their buffers, e.g. the host mirror's frame and soft-bit buffers. */
void *dabgpu_host_alloc(size_t bytes);
void dabgpu_host_free(void *p);
/* (The one-frame calls -- dabgpu_ofdm_demod_stream_frame, dabgpu_decode_stream_frames, dabgpu_dabplus_superframes -- let
 * their kernels write results straight into page-locked caller buffers and end on a watched word instead of a stream
 * synchronisation ONLY for buffers obtained here, which are coherent; results written into any other page-locked memory
 * (hipHostRegister, non-coherent allocations) are followed by hipStreamSynchronize: HIP promises host visibility of kernel
 * writes to such memory only there.) */

/* Test hook (tests/ only; changes no result): the nth one-frame call of this context from now on
 * (dabgpu_ofdm_demod_stream_frame or dabgpu_decode_stream_frames, whichever the context serves) returns DABGPU_ERR_HIP
 * before any launch, once; 0 disarms.  What the host mirror must survive: /root/reference/src/radio_block.cpp:27, 37
 * (the two Process calls run on threads nobody can throw to). */
int dabgpu_test_fail_frame_call(dabgpu_ctx *ctx, int nth);

/* Device buffers for a batch user's IQ samples ([n_frames][frame_stride] cf32, frame_stride >= 196608 samples) and
 * soft bits ([n_frames][230400] int8).  Any device buffer is accepted by the _dev entry points; this call exists for
 * callers that own their buffers and want them placed for the front end.
 *   placement  DABGPU_PLACE_PLAIN    two hipMallocs.
 *              DABGPU_PLACE_DOMAINS  MI355X's HBM behaves as three domains of 96 GB, and a launch that reads its
 *                samples from the domain it writes its soft bits to runs up to ~10 % slower than one whose two streams
 *                lie apart (profiles/r02_hbm_domains.txt; against a plain pair the front end
 *                gained 0.8-3.9 % in six of six trials, profiles/r04_placement_ab.txt).  Physical memory is taken in
 *                chunks through the virtual-memory API (1 GiB for the samples, 256 MiB for the soft bits; never more
 *                than 1.5 x the pair's size held during set-up), every chunk's domain is found with a small data
 *                mover (two passes, ~40-110 ms), the IQ buffer is mapped over chunks of the most plentiful domain(s)
 *                and every 256 MiB of the soft-bit buffer over a chunk whose domain differs from the ~1.7 GiB of
 *                samples read WHILE it is written; the chunks left over go back.  All of it happens inside two
 *                address ranges per context (one to probe in, one for the pair), reserved by the first such call and
 *                released by dabgpu_destroy: one domain-aware pair per context at a time.  The call then CHECKS its
 *                own result (report->pair_over_same_domain) and, where the pair behaves as one domain -- a box whose
 *                virtual-memory chunks all come from one: the worst case, 2-9 % of the boxes seen --, gives it back
 *                and allocates a plain pair instead: the caller never has to compare.  That case, a second request
 *                while the first pair is alive, a request larger than the range was reserved for, buffers below
 *                ~4 GiB, a device without the virtual-memory API or without room, and any failure on the way all end
 *                in a PLAIN pair (report->method = 0, report->fallback_reason says why).  A set-up call: it
 *                synchronises, takes ~0.1-0.2 s and leaves noise in both buffers.
 * Returns an error only when the plain allocation fails too.  Release with dabgpu_free_frame_buffers (both pointers of
 * a domain-aware pair together). */
#define DABGPU_PLACE_PLAIN   0
#define DABGPU_PLACE_DOMAINS 1
/* report->fallback_reason */
#define DABGPU_PLAIN_REQUESTED    0  /* DABGPU_PLACE_PLAIN was asked for                                   */
#define DABGPU_PLAIN_SIZE         1  /* too small for the domains to matter, or more chunks than are handled */
#define DABGPU_PLAIN_NO_VMM       2  /* the virtual-memory API refused (reserve / map / set access)          */
#define DABGPU_PLAIN_NO_ROOM      3  /* free memory does not hold the chunks                                  */
#define DABGPU_PLAIN_ARENA_BUSY   4  /* the context's domain-aware pair is still alive                        */
#define DABGPU_PLAIN_ARENA_SMALL  5  /* larger than the address range the context reserved on its first call   */
#define DABGPU_PLAIN_PROBE_FAILED 6  /* a probe launch or its timing failed                                    */
#define DABGPU_PLAIN_ONE_DOMAIN   7  /* the placed pair's own check read >= 0.985 (it behaves as one domain: the   */
                                     /* worst case); it was given back and two plain allocations made instead;   */
                                     /* pair_over_same_domain, n_domains, domains describe the pair given back    */
typedef struct dabgpu_placement_report {
    int32_t method;             /* 0 = plain hipMalloc pair, 1 = domain-aware pair                              */
    int32_t fallback_reason;    /* method 0: DABGPU_PLAIN_*                                                     */
    int32_t n_chunks;           /* physical chunks taken during set-up                                         */
    int32_t iq_chunks, soft_chunks;   /* chunks (of either size) each buffer is mapped over                    */
    int32_t n_domains;          /* distinct HBM domains seen among the chunks (1..3)                           */
    int32_t conflicts;          /* per mille of the soft bits that are written beside reads from their own domain */
    int32_t runtime_error;      /* fallback after a failed runtime call: stage * 1000 + hipError_t (stage 1 reserve, */
                                /* 2 create, 3 map, 4 set access, 5 unmap, 6 memory info, 7 / 8 map / set access   */
                                /* of the pair); else 0                                                             */
    uint64_t chunk_bytes;
    uint64_t setup_peak_bytes;  /* device memory held at the peak of the set-up (<= 1.5 x the pair)            */
    float classify_ms;          /* time spent finding the domains                                              */
    float pair_over_same_domain; /* check of the result: a mover reading 3/4 of the samples' first chunk and writing */
                                /* the start of the soft-bit buffer, over the same mover writing into the last 1/4 */
                                /* of that chunk instead (same domain by construction): ~0.9 when the two buffers  */
                                /* lie apart, ~1.0 when they do not (0 = not measured).  A domain-aware pair is   */
                                /* only ever handed out below 0.985; at or above it the pair goes back and the    */
                                /* call ends in a plain pair (DABGPU_PLAIN_ONE_DOMAIN): the caller decides nothing */
    char domains[100];          /* one letter per chunk in allocation order: 'A' 'B' 'C' for the 1 GiB chunks, */
                                /* 'a' 'b' 'c' for the 256 MiB ones; NUL-terminated, cut at 95                  */
    char iq_map[72];            /* the chunks of the IQ buffer in address order, same letters                  */
    char soft_map[28];          /* the chunks of the soft-bit buffer in address order                          */
} dabgpu_placement_report;
int dabgpu_alloc_frame_buffers(dabgpu_ctx *ctx, int n_frames, size_t frame_stride, int placement, void **d_iq,
                               int8_t **d_soft, dabgpu_placement_report *report);
int dabgpu_free_frame_buffers(dabgpu_ctx *ctx, void *d_iq, int8_t *d_soft);

/* The ceiling the front end is measured against: a pure data mover of the front end's own geometry on the caller's
 * buffers -- per frame the useful 2048 samples of each of the 76 symbols (+ the PRS's prefix; with_prefixes != 0: the
 * whole 2552-sample periods, what a launch that produces the cyclic-prefix correlations reads) in, 230400 bytes out,
 * one wavefront per run of symbols cut exactly as dabgpu_ofdm_demod_frames_dev cuts the same batch, streaming
 * accesses, the same LDS footprint (occupancy) -- and no arithmetic.  Overwrites d_soft with meaningless bytes.
 * Time it with events on `stream`; bench.py reports it as roofline.mover_same_geometry_ms. */
int dabgpu_mover_frames_dev(dabgpu_ctx *ctx, const void *d_iq, size_t frame_stride, int n_frames, int8_t *d_soft,
                            int with_prefixes, void *stream);

/* ------------------------------------------------------------------------ */
/* A2..A6: OFDM front end on time-aligned frames.                             */
/* Replaces the READING_SYMBOLS work of OFDM_Demod::Process                    */
/*   /root/reference/src/dab_module.cpp:25 (call), src/radio_block.cpp:22 (ctor) */
/* up to the payload of the On_OFDM_Frame callback (src/radio_block.cpp:25).   */
/*                                                                            */
/* iq           cf32; frame f starts (first PRS sample, i.e. after the null    */
/*              symbol) at iq + f*frame_stride complex samples; 76*2552        */
/*              samples are read per frame.  Must be 16-byte aligned and        */
/*              frame_stride even.                                              */
/* freq_offset  [n_frames] correction in cycles/sample applied as               */
/*              x[n]*exp(+j*2*pi*f*n) (the sum the reference shows as           */
/*              GetNetFrequencyOffset(), src/render_radio_block.cpp:204).       */
/*              NULL = no correction.                                           */
/* soft         [n_frames][230400] int8 (frame bits in transmission order)      */
/* cyc          optional [n_frames][76] cf32: cyclic-prefix correlations        */
/*              sum conj(y[i])*y[i+2048]; their mean angle / (2*pi*2048) is the */
/*              fine frequency error (fine_freq_update_beta loop,               */
/*              src/render_radio_block.cpp:216).                                */
/* dqpsk        optional [n_frames][75][1536] cf32 differential symbols in      */
/*              carrier order -768..768 (GetFrameDataVec(),                     */
/*              src/render_radio_block.cpp:109).                                */
/* ------------------------------------------------------------------------ */
int dabgpu_ofdm_demod_frames_dev(dabgpu_ctx *ctx, const void *d_iq, size_t frame_stride, int n_frames,
                                 const float *d_freq_offset, int8_t *d_soft, void *d_cyc, void *d_dqpsk,
                                 void *stream);
int dabgpu_ofdm_demod_frames(dabgpu_ctx *ctx, const float *iq, size_t frame_stride, int n_frames,
                             const float *freq_offset, int8_t *soft, float *cyc, float *dqpsk);
/* (Serves the fine-frequency loop whose state the GUI reads -- GetFineFrequencyOffset, fine_freq_update_beta,
 * /root/reference/src/render_radio_block.cpp:202, :216 -- with an estimator of this library's own.)
 * The same front end reading ONE cyclic prefix per frame instead of 76 (the prefixes are 20 % of the samples, and the
 * kernel is bound by the bytes it moves), with the decision-directed frequency-error sums a loop needs instead of the
 * cyclic-prefix correlations:
 *   dd4  [n_frames][76] cf32; the SUM of entries 1..75 of a frame = sum over its 75 data symbols and 256 of each
 *        symbol's carriers (FFT bins v + 64 m, v < 64, m in {0, 1, 30, 31}, bin 0 replaced by 768) of u^4, u = the
 *        differential symbol X_l conj X_{l-1} divided by its magnitude (every term has magnitude 1 whatever the level of
 *        the input; |sum| / 19200 is a lock quality between 0 and 1).  How the sum is spread over the entries depends on
 *        how the launch cut the frame into runs: a run's total sits in the entry of its last symbol, its other entries
 *        are 0.  Whatever two bits a differential symbol carries, its fourth power is -exp(j 4 theta) with theta =
 *        2 pi * (residual offset, cycles per sample) * 2552: angle(-sum) / (4 * 2 pi * 2552) is the residual modulo
 *        1 / (4 * 2552) (0.2 carriers).
 *        Entry 0 of a frame = the cyclic-prefix correlation of its PRS (symbol 0; the one prefix that IS read, 0.3 % of
 *        the frame): angle / (2 pi 2048) is the same residual, coarser but unambiguous within half a carrier -- it picks
 *        the branch: residual = e_dd + k / (4 * 2552), k = round((e_cp - e_dd) * 4 * 2552).
 * OPT-IN everywhere: every closed-loop entry point runs the reference's loop (cyclic-prefix correlations) unless told
 * otherwise -- the tracked and frame calls through dabgpu_track_cfg.decision_directed = 1, the stream call and the ring after
 * dabgpu_set_stream_loop(..., decision_directed = 1) -- and only when the caller does not ask for the correlations
 * (d_cyc == NULL). */
int dabgpu_ofdm_demod_frames_dd_dev(dabgpu_ctx *ctx, const void *d_iq, size_t frame_stride, int n_frames,
                                    const float *d_freq_offset, int8_t *d_soft, void *d_dd4, void *stream);

/* ------------------------------------------------------------------------ */
/* Closed-loop front end: per-stream tracking state kept in device memory.    */
/* Replaces the fine-frequency loop of OFDM_Demod and the scalars the GUI      */
/* reads from it: GetFineFrequencyOffset / GetCoarseFrequencyOffset /          */
/* GetNetFrequencyOffset / GetSignalAverage / GetTotalFramesRead /             */
/* GetTotalFramesDesync and fine_freq_update_beta                              */
/*   (/root/reference/src/render_radio_block.cpp:202-207, :216).               */
/*                                                                            */
/* A context owns `n_streams` states (dabgpu_streams_reset).  The stream call  */
/* demodulates frames_per_stream consecutive frames of every stream with the   */
/* stream's current fine + coarse offset -- no host-supplied frequency -- and  */
/* then, on the device and in stream order, moves the fine offset by           */
/* -beta * (mean angle of the frames' cyclic-prefix correlations)/(2*pi*2048), */
/* kept within +-half a carrier; the next call uses the new value.  Frame      */
/* (s, f) starts at iq + (s*frames_per_stream + f)*frame_stride.               */
/* cyc may be NULL: the library keeps what the loop needs in its own scratch --  */
/* the correlations, or (dabgpu_set_stream_loop(.., decision_directed = 1)) the  */
/* dd4 sums, in which case only the PRS's cyclic prefix is read.                  */
/* ------------------------------------------------------------------------ */
typedef struct dabgpu_stream_state {      /* DEVICE memory, 64 bytes */
    float fine_freq_offset;               /* cycles/sample, within +-0.5/2048                     */
    float coarse_freq_offset;             /* cycles/sample (whole carriers: -k/2048)              */
    float signal_average;                 /* running mean of |re|+|im| per sample                 */
    float last_fine_error;                /* residual the most recent call measured, cycles/sample */
    int32_t total_frames_read;
    int32_t total_frames_desync;          /* frames lost: level below thresh_null_start x average, PRS not found, */
                                          /* or (tracked calls) begun before the capture did                      */
    int32_t tracking;                     /* tracked calls: 1 = next_frame_start is valid                         */
    int32_t last_time_offset;             /* tracked calls: where the most recent frame's impulse-response peak    */
                                          /* sat relative to the predicted position, samples                       */
    double next_frame_start;              /* first sample of the next frame (PRS prefix minus timing_margin),      */
                                          /* relative to the first sample of the NEXT capture                      */
    float drift;                          /* samples per frame the frame period differs from 196608                */
    float last_peak_to_mean;              /* impulse-response peak / mean of the most recent frame                 */
    int32_t loop_gated;                   /* decision-directed loop: calls whose fourth-power estimate was gated    */
                                          /* (quality below the gate: the PRS prefix alone was used; or a one-step  */
                                          /* branch departure held back) -- see dabgpu_set_loop_gate                */
    int32_t dd_branch;                    /* branch of the most recent accepted estimate, units of 0.2 carriers     */
    int32_t dd_pending;                   /* a one-step departure from branch 0 seen once (0x7fffffff: none)        */
    int32_t reserved;
} dabgpu_stream_state;

typedef struct dabgpu_stats {             /* HOST copy with the derived fields the GUI prints */
    int32_t state;                        /* OFDM_Demod::State value: 0 before the first frame (and after a tracked */
                                          /* stream lost every frame of a call), 4 = READING_SYMBOLS               */
    float fine_freq_offset;
    float coarse_freq_offset;
    float net_freq_offset;
    float signal_average;
    int32_t total_frames_read;
    int32_t total_frames_desync;
    float last_fine_error;
    int32_t tracking;
    int32_t last_time_offset;
    double next_frame_start;
    float drift;
    float last_peak_to_mean;
    int32_t loop_gated;                   /* see dabgpu_stream_state                                                */
    int32_t reserved;
} dabgpu_stats;

/* (re)create the context's stream states, all zero */
int dabgpu_streams_reset(dabgpu_ctx *ctx, int n_streams);
/* the states in device memory ([n_streams]), e.g. to seed them from a kernel of the caller's; NULL before a reset */
dabgpu_stream_state *dabgpu_stream_states(dabgpu_ctx *ctx);
/* host-side setter (synchronises the context stream): NULL = leave that offset as it is.  The host mirror stores the
 * coarse offset found at acquisition here (is_coarse_freq_correction, src/render_radio_block.cpp:215). */
int dabgpu_set_stream_offsets(dabgpu_ctx *ctx, int stream_index, const float *fine, const float *coarse);
int dabgpu_ofdm_demod_streams_dev(dabgpu_ctx *ctx, const void *d_iq, size_t frame_stride, int n_streams,
                                  int frames_per_stream, float fine_freq_update_beta, int8_t *d_soft, void *d_cyc,
                                  void *d_dqpsk, void *stream);
int dabgpu_ofdm_demod_streams(dabgpu_ctx *ctx, const float *iq, size_t frame_stride, int n_streams,
                              int frames_per_stream, float fine_freq_update_beta, int8_t *soft, float *cyc,
                              float *dqpsk);
/* the seven scalars of src/render_radio_block.cpp:192-207 for one stream.  Ordering: the states are read and written
 * by the stream calls on whatever stream those were given; dabgpu_get_stats, dabgpu_set_stream_offsets and
 * dabgpu_streams_reset first wait (on the host) for the most recent such call to finish, wherever it ran. */
int dabgpu_get_stats(dabgpu_ctx *ctx, int stream_index, dabgpu_stats *out);
/* signal_l1.update_beta (src/render_radio_block.cpp:235) and null_l1_search.thresh_null_start (:226-233) of the stream
 * calls' level average and desync count (defaults 0.95 / 0.35), and which estimator the fine-frequency loop of
 * dabgpu_ofdm_demod_streams_dev runs on when the caller passes no correlation buffer (d_cyc == NULL):
 *   decision_directed == 0 (default)  the cyclic-prefix correlations, kept in the library's scratch: pulls in from
 *                                     +-half a carrier, as the reference's loop does
 *   decision_directed != 0            the dd4 sums of dabgpu_ofdm_demod_frames_dd_dev: of the 76 cyclic prefixes of a frame
 *                                     only the PRS's is read (17 % fewer bytes); it resolves the sums' 0.2-carrier
 *                                     ambiguity, so this loop, too, pulls in from +-half a carrier and may run from
 *                                     the first call on.  Off by default: the drop-in path keeps the reference's
 *                                     data flow; this one is the library's own, opt-in. */
int dabgpu_set_stream_loop(dabgpu_ctx *ctx, float signal_update_beta, float thr_null_start, int decision_directed);
/* The decision-directed loop checks its own estimate before it moves (stream call here; tracked and frame calls:
 * dabgpu_track_cfg.dd_gate).  With S = the call's sum of unit fourth powers over n terms:
 *   quality  |S| < dd_gate * sqrt(n) -- what n random phases add up to, times the gate (2.5 by default): the sum carries no
 *            usable phase (single frames below ~3 dB SNR, nothing selected) and the call takes the cyclic-prefix
 *            estimate of its PRSs alone -- unbiased, coarser; 0 switches this gate off.
 *   branch   a stream whose previous residual lay inside +-0.1 carriers and whose PRS prefix now asks for the branch one
 *            step (0.2 carriers) away is believed only when the next call asks again; the first time the branch is held.
 * Either event counts in loop_gated (dabgpu_get_stats).  The loop on the cyclic-prefix correlations -- the reference's
 * estimator, fine_freq_update_beta at /root/reference/src/render_radio_block.cpp:216 -- has no such ambiguity and no gate. */
int dabgpu_set_loop_gate(dabgpu_ctx *ctx, float dd_gate);

/* Soft-bit selection (batch receivers that decode the FIC and a few sub-channels and never look at the rest of the
 * frame): from the next call on, dabgpu_ofdm_demod_frames[_dev], dabgpu_ofdm_demod_streams[_dev] and
 * dabgpu_ofdm_demod_acquired_dev of this context write only the listed parts of each frame's 230400 soft bits and leave the other bytes of `soft` untouched -- in
 * device memory and, for the host-pointer call, in the caller's host buffer (only the selected ranges are copied
 * back).  Symbols that carry no selected bit and are not the differential reference of one that does are not
 * transformed at all (their cyclic-prefix correlation is still produced when `cyc` is asked for).  The reference's
 * OFDM_Demod always emits whole frames (/root/reference/src/radio_block.cpp:25); its BasicRadio then reads the
 * FIC and the selected sub-channels only (:42) -- this moves that choice in front of the 230 kB store.
 *   ranges    frame-bit coordinates [first, first+count), both multiples of 16, inside 0..230400
 *   n_ranges  0 = write everything again (the default)
 * Calls with a constellation output (dqpsk != NULL) ignore the selection.  Not to be changed while a demodulation
 * call of this context is still running on some stream.
 * dabgpu_soft_selection writes the ranges of the FIC (with_fic != 0) and of the given sub-channels (4 CIFs each)
 * to `out` and returns how many there are (more than max_out = nothing written beyond max_out), or a negative
 * status for a bad sub-channel. */
struct dabgpu_subchannel;             /* defined with the MSC entry points below */
typedef struct dabgpu_bit_range {
    int32_t first;
    int32_t count;
} dabgpu_bit_range;
int dabgpu_ofdm_set_soft_selection(dabgpu_ctx *ctx, const dabgpu_bit_range *ranges, int n_ranges);
int dabgpu_soft_selection(const struct dabgpu_subchannel *subchannels, int n_subchannels, int with_fic,
                          dabgpu_bit_range *out, int max_out);

/* A2+A3 alone (the unfused "FFT stage"): frequency-corrected 2048-point forward
 * FFT of the useful part of each of the 76 symbols.  Replaces the FFTW3f plan the
 * reference links (/root/reference/CMakeLists.txt:55-64).
 * spectra  [n_frames][76][2048] cf32, unnormalised, natural bin order. */
int dabgpu_fft_symbols_dev(dabgpu_ctx *ctx, const void *d_iq, size_t frame_stride, int n_frames,
                           const float *d_freq_offset, void *d_spectra, void *stream);
int dabgpu_fft_symbols(dabgpu_ctx *ctx, const float *iq, size_t frame_stride, int n_frames,
                       const float *freq_offset, float *spectra);

/* ------------------------------------------------------------------------ */
/* Frame synchronisation on the phase reference symbol.      */
/* Replaces the RUNNING_COARSE_FREQ_SYNC and RUNNING_FINE_TIME_SYNC work of    */
/* OFDM_Demod (/root/reference/src/render_radio_block.cpp:195-196) and its     */
/* knobs is_coarse_freq_correction / max_coarse_freq_correction_norm /         */
/* impulse_peak_threshold_db (:213-231).                                       */
/*                                                                            */
/* iq            candidate f starts at iq + f*frame_stride: the caller's guess */
/*               of the FIRST sample of the PRS cyclic prefix; 2552 samples    */
/*               are read.  16-byte aligned, frame_stride even.                */
/* freq_offset   [n] correction applied before the search (NULL = none)        */
/* max_coarse    search range in carriers, 0..1023                             */
/* out[f].coarse_carriers  integer carrier offset k of the signal: apply       */
/*               -k/2048 cycles/sample as coarse correction                    */
/* out[f].time_offset      the PRS useful part starts at candidate + 504 +     */
/*               time_offset samples (signed, -1024..1023)                     */
/* out[f].peak_to_mean     impulse-response peak / mean power (threshold it    */
/*               like impulse_peak_threshold_db)                               */
/* out[f].coarse_peak_to_mean  same for the coarse correlation                 */
/* ------------------------------------------------------------------------ */
typedef struct dabgpu_sync_result {
    int32_t coarse_carriers;
    int32_t time_offset;
    float peak_to_mean;
    float coarse_peak_to_mean;
} dabgpu_sync_result;

int dabgpu_sync_prs_dev(dabgpu_ctx *ctx, const void *d_iq, size_t frame_stride, int n_frames,
                        const float *d_freq_offset, int max_coarse, dabgpu_sync_result *d_out, void *stream);
int dabgpu_sync_prs(dabgpu_ctx *ctx, const float *iq, size_t frame_stride, int n_frames, const float *freq_offset,
                    int max_coarse, dabgpu_sync_result *out);

/* ------------------------------------------------------------------------ */
/* Acquisition on unaligned captures: everything OFDM_Demod   */
/* does before READING_SYMBOLS -- FINDING_NULL_POWER_DIP, READING_NULL_AND_PRS, */
/* RUNNING_COARSE_FREQ_SYNC, RUNNING_FINE_TIME_SYNC                            */
/* (/root/reference/src/render_radio_block.cpp:193-196; knobs                  */
/* thresh_null_start/end, max_coarse_freq_correction_norm,                     */
/* impulse_peak_threshold_db at :213-235) -- for whole captures at once.        */
/*                                                                            */
/* iq       stream s = iq + s*stream_stride complex samples, n_samples each;    */
/*          any 8-byte aligned position (no frame alignment assumed)            */
/* out      [n_streams][max_frames]: one entry per null symbol found whose      */
/*          frame lies inside the capture, in time order; entries past          */
/*          counts[s] have flags 0 and start -1                                 */
/*   start            first sample of the PRS cyclic prefix minus               */
/*                    cfg.timing_margin, relative to the stream                 */
/*   freq_offset      correction to hand to the demodulator, cycles/sample      */
/*                    (= fine_offset - coarse_carriers/2048)                    */
/*   fine_offset      from the cyclic prefix of the PRS, (-0.5..0.5]/2048       */
/*   flags            bit 0: impulse peak_to_mean >= cfg.min_peak_to_mean       */
/*                    bit 1: all 76 symbols lie inside the capture              */
/* dabgpu_ofdm_demod_acquired_dev demodulates exactly those frames (A2..A6 as   */
/* dabgpu_ofdm_demod_frames): soft [n_streams*max_frames][230400]; entries      */
/* whose flags are not 3 give all-zero (erased) soft bits.                      */
/* ------------------------------------------------------------------------ */
typedef struct dabgpu_acquire_cfg {
    float thr_null_start;        /* dip begins below this fraction of the mean block L1 norm (0.35) */
    float thr_null_end;          /* and ends above this one (0.75)                                   */
    int32_t min_null_blocks;     /* shortest dip, in 64-sample blocks, taken for a null symbol (30)  */
    int32_t max_coarse_carriers; /* coarse search range (200)                                        */
    float min_peak_to_mean;      /* lock threshold on the impulse response (30 = 14.8 dB)            */
    int32_t timing_margin;       /* samples the FFT windows are kept inside the cyclic prefix (64)   */
    /* which tap of the channel impulse response a frame is aligned to:                                  */
    float impulse_peak_distance_probability; /* taps are scored |h|^2 w^2, w = 1 - (1 - p) |offset - expected| / 2552 */
                                 /* (src/render_radio_block.cpp:225; 0.15; 1 = unweighted)           */
    float first_path_rel;        /* > 0: the earliest tap within 504 samples before the scored peak  */
                                 /* with at least max(rel x peak power, 16 x mean) replaces it, so   */
                                 /* that a stronger LATE echo still falls inside the cyclic prefix   */
                                 /* (0.25 = -6 dB); 0 = the scored peak itself, the reference's rule */
    int32_t level_chunk_blocks;  /* the null thresholds follow the LOCAL level: mean block norm over    */
                                 /* chunks of this many 64-sample blocks, averaged over chunks c-2..c+2 */
                                 /* (power of two, 64..16384; 256 = 8 ms chunks, a 40 ms window) -- the   */
                                 /* batch counterpart of the reference's running average                */
                                 /* (signal_l1.update_beta, :235); 0 = the mean of the whole capture     */
    int32_t reserved;
} dabgpu_acquire_cfg;

typedef struct dabgpu_acquired_frame {
    int64_t start;
    float freq_offset;
    int32_t coarse_carriers;
    float fine_offset;
    float peak_to_mean;
    float coarse_peak_to_mean;
    int32_t flags;
} dabgpu_acquired_frame;

void dabgpu_acquire_default_cfg(dabgpu_acquire_cfg *cfg);
int dabgpu_acquire_dev(dabgpu_ctx *ctx, const void *d_iq, size_t stream_stride, int n_streams, int64_t n_samples,
                       const dabgpu_acquire_cfg *cfg, int max_frames, dabgpu_acquired_frame *d_out, int32_t *d_counts,
                       void *stream);
int dabgpu_acquire(dabgpu_ctx *ctx, const float *iq, size_t stream_stride, int n_streams, int64_t n_samples,
                   const dabgpu_acquire_cfg *cfg, int max_frames, dabgpu_acquired_frame *out, int32_t *counts);
int dabgpu_ofdm_demod_acquired_dev(dabgpu_ctx *ctx, const void *d_iq, size_t stream_stride, int n_streams,
                                   int max_frames, const dabgpu_acquired_frame *d_frames, int8_t *d_soft, void *d_cyc,
                                   void *d_dqpsk, void *stream);

/* ------------------------------------------------------------------------ */
/* Timing tracking: what OFDM_Demod's RUNNING_FINE_TIME_SYNC state does on     */
/* every frame once locked (/root/reference/src/render_radio_block.cpp:196;    */
/* knobs impulse_peak_threshold_db, impulse_peak_distance_probability,         */
/* :224-225) -- for batches of streams, with the position kept on the device.  */
/*                                                                            */
/* A stream is acquired ONCE (dabgpu_acquire_dev on its first capture, then     */
/* dabgpu_track_start_dev).  Every later capture goes through                  */
/* dabgpu_ofdm_demod_tracked_dev alone: per stream, the frames the state        */
/* predicts inside the capture (start_i = next_frame_start + i*(196608+drift)) */
/* are synchronised on their own PRS (impulse response -> exact start, lock),   */
/* demodulated where they lie with the stream's fine + coarse offset, and the   */
/* state is moved on: fine-frequency loop (on the cyclic-prefix correlations, as  */
/* the reference's; cfg.decision_directed = 1 and d_cyc == NULL: on the dd4 sums), */
/* next_frame_start and drift from a line through the measured starts.         */
/*   n_samples  samples per stream in this capture                             */
/*   advance    the next capture of every stream will begin this many samples   */
/*              after this one began.  Frames are only taken whole (+512        */
/*              samples), so consecutive captures must overlap by at least one  */
/*              frame + 512 + the drift: advance <= n_samples - 197632.         */
/*   frames     [n_streams][max_frames] as dabgpu_acquire_dev writes them        */
/*              (flags 3 = demodulated); counts[s] = frame slots of stream s     */
/*              that lay inside the capture.  soft / cyc / dqpsk as             */
/*              dabgpu_ofdm_demod_acquired_dev (slots without a locked frame:    */
/*              erased soft bits).  cyc may be NULL (cfg.decision_directed       */
/*              then chooses what the fine loop runs on).                        */
/* A stream none of whose frames locked in a call stops tracking (state 0 in     */
/* dabgpu_get_stats): acquire it again -- or set cfg.auto_acquire and the next   */
/* call does (then the very first call needs no dabgpu_acquire_dev either).      */
/* ------------------------------------------------------------------------ */
typedef struct dabgpu_track_cfg {
    float fine_freq_update_beta;              /* 0.9                                                   */
    float signal_update_beta;                 /* signal_l1.update_beta, 0.95                           */
    float thr_null_start;                     /* 0.35                                                  */
    float min_peak_to_mean;                   /* impulse_peak_threshold_db as a power ratio (100)      */
    float impulse_peak_distance_probability;  /* 0.15, see dabgpu_acquire_cfg                          */
    float first_path_rel;                     /* 0.25, see dabgpu_acquire_cfg                          */
    float drift_beta;                         /* share of a measured drift error taken per call (0.5)  */
    float coarse_freq_slow_beta;              /* frame call: see below (0.1)                           */
    int32_t timing_margin;                    /* 64 (batch calls); the frame call uses the host's own   */
    int32_t max_coarse_carriers;              /* frame call and auto-acquisition: whole-carrier search */
                                              /* range, 0 = off (204)                                  */
    int32_t decision_directed;                /* 0 (default) = the fine loop runs on the cyclic-prefix */
                                              /* correlations, the reference's estimator               */
                                              /* (fine_freq_update_beta, render_radio_block.cpp:216);   */
                                              /* 1 = tracked / frame call with d_cyc == NULL: on the    */
                                              /* dd4 sums, only the PRS's cyclic prefix is read -- this */
                                              /* library's own estimator, opt-in (ABI v6; v5: default 1) */
    int32_t auto_acquire;                     /* tracked call: != 0 = streams that are not tracking    */
                                              /* (never acquired, or lost) are ACQUIRED inside the     */
                                              /* call -- null-symbol search + PRS on their capture as  */
                                              /* dabgpu_acquire_dev does, their frames demodulated     */
                                              /* with the others', their tracking started; streams     */
                                              /* that are tracking cost nothing extra.  One call does  */
                                              /* everything from the first capture on (0)              */
    float dd_gate;                            /* decision-directed loop: quality gate, see             */
                                              /* dabgpu_set_loop_gate (2.5; 0 = off)                   */
    int32_t reserved;                         /* must be 0                                             */
} dabgpu_track_cfg;
void dabgpu_track_default_cfg(dabgpu_track_cfg *cfg);
/* only_lost != 0: streams that are tracking keep their state (re-acquisition of the lost ones beside them) */
int dabgpu_track_start_dev(dabgpu_ctx *ctx, const dabgpu_acquired_frame *d_frames, const int32_t *d_counts, int n_streams,
                           int max_frames, int64_t advance, int only_lost, void *stream);
int dabgpu_ofdm_demod_tracked_dev(dabgpu_ctx *ctx, const void *d_iq, size_t stream_stride, int n_streams,
                                  int64_t n_samples, int max_frames, int64_t advance, const dabgpu_track_cfg *cfg,
                                  int8_t *d_soft, void *d_cyc, void *d_dqpsk, dabgpu_acquired_frame *d_frames,
                                  int32_t *d_counts, void *stream);

/* One frame of one stream from host memory, everything in ONE call: what OFDM_Demod does between "76 symbols are in
 * the buffer" and the On_OFDM_Frame callback (RUNNING_COARSE_FREQ_SYNC / RUNNING_FINE_TIME_SYNC / READING_SYMBOLS,
 * /root/reference/src/render_radio_block.cpp:195-197; src/radio_block.cpp:25).  One upload of the frame; on the
 * device: synchronisation on the PRS, the coarse-offset update, demodulation with the stream's offsets, fine loop,
 * counters, level; one download of {soft bits, sync result, statistics}; one synchronisation.
 *   iq         76 * 2552 cf32 as the caller assembled them: iq[0] is its guess of the first PRS prefix sample,
 *              `cfg->timing_margin` samples early.  Any host memory; page-locked buffers (dabgpu_host_alloc) aligned to
 *              16 bytes -- `iq` and `soft` both -- are read / written by kernels in line with the others (no copy-engine
 *              hand-over, no copy by the CPU afterwards): ~10 % less time per call
 *   acquiring  != 0: first frame after a null-symbol detection -- the stream's fine offset is set from this PRS's own
 *              cyclic prefix (so that already this frame is demodulated with it) and the whole-carrier offset found on
 *              this PRS is STORED as its coarse offset (cfg->max_coarse_carriers > 0); 0: a residual of k carriers
 *              moves the coarse offset by coarse_freq_slow_beta * k
 *   result->flags  bit 0 = PRS found (peak_to_mean >= min_peak_to_mean), bit 1 = the FFT windows lie inside the
 *              cyclic prefix (0 <= sync.time_offset <= 488); only a frame with both is demodulated (others: erased
 *              soft bits, counted as desync)
 *   dqpsk      optional [75][1536] cf32 (GetFrameDataVec) */
typedef struct dabgpu_frame_result {
    dabgpu_sync_result sync;
    int32_t flags;
    int32_t reserved;
    dabgpu_stats stats;
} dabgpu_frame_result;
int dabgpu_ofdm_demod_stream_frame(dabgpu_ctx *ctx, int stream_index, const float *iq, int acquiring,
                                   const dabgpu_track_cfg *cfg, int8_t *soft, float *dqpsk, dabgpu_frame_result *result);

/* ------------------------------------------------------------------------ */
/* A7..A11: FIC.  Replaces the FIC branch of BasicRadio::Process              */
/*   /root/reference/src/radio_block.cpp:42 (call), :60 (ctor).               */
/* soft      frame f's bits start at soft + f*soft_stride; the first 9216 are  */
/*           the FIC (4 groups of 2304).                                       */
/* fib       [n_frames][12][32] bytes (30 data + CRC16), energy dispersal       */
/*           removed                                                           */
/* crc_ok    [n_frames][12] 1 = CRC matches                                    */
/* ------------------------------------------------------------------------ */
int dabgpu_fic_decode_dev(dabgpu_ctx *ctx, const int8_t *d_soft, size_t soft_stride, int n_frames,
                          uint8_t *d_fib, uint8_t *d_crc_ok, void *stream);
int dabgpu_fic_decode(dabgpu_ctx *ctx, const int8_t *soft, size_t soft_stride, int n_frames,
                      uint8_t *fib, uint8_t *crc_ok);

/* ------------------------------------------------------------------------ */
/* A12: one MSC subchannel.  Replaces the per-subchannel branch of             */
/* BasicRadio::Process (/root/reference/src/radio_block.cpp:42); the           */
/* descriptor mirrors the Subchannel entity the GUI prints                     */
/* (/root/reference/src/render_formatters.cpp:9-25).  Any size up to a         */
/* sub-channel that fills the CIF (864 capacity units).                        */
/* ------------------------------------------------------------------------ */
typedef struct dabgpu_subchannel {
    int32_t start_address;   /* first capacity unit (0..863)                  */
    int32_t length;          /* size in capacity units                        */
    int32_t is_uep;          /* 0 = EEP (long form), 1 = UEP (short form)     */
    int32_t eep_type;        /* EEP: 0 = A, 1 = B                             */
    int32_t protection_level;/* EEP 1..4; UEP 1..5                            */
    int32_t bitrate_kbps;    /* EEP: multiple of 8 (A) / 32 (B); UEP: one of  */
                             /* the 14 rates of the protection profile table  */
} dabgpu_subchannel;

/* UEP sub-channels are announced by an index into the standard's table of 64 protection profiles
 * (FIG 0/1 short form, `uep_prot_index` in the reference's Subchannel entity,
 * /root/reference/src/render_formatters.cpp:9-25): fills bit rate, level and size from it. */
int dabgpu_uep_subchannel(int table_index, int start_address, dabgpu_subchannel *out);

/* bytes one CIF of this subchannel decodes to (bitrate*3), or <0 */
int dabgpu_subchannel_bytes(const dabgpu_subchannel *sc);

/* n_streams independent ensembles, frames_per_stream consecutive frames each;
 * frame (s,f) is at soft + (s*frames_per_stream+f)*soft_stride.
 * history_in / history_out  [n_streams][15][length*64] int8: the subchannel's bits of the
 *           15 CIFs before / at the end of this call (time de-interleaver state, 16-CIF
 *           depth).  history_in may be NULL (treated as erasures); they must not alias.
 * out       [n_streams][frames_per_stream*4][bitrate*3] bytes; entry t is the logical
 *           frame completed by CIF t, i.e. the one transmitted 15 CIFs earlier. */
int dabgpu_msc_decode_dev(dabgpu_ctx *ctx, const dabgpu_subchannel *sc, const int8_t *d_soft,
                          size_t soft_stride, int n_streams, int frames_per_stream,
                          const int8_t *d_history_in, int8_t *d_history_out, uint8_t *d_out, void *stream);
int dabgpu_msc_decode(dabgpu_ctx *ctx, const dabgpu_subchannel *sc, const int8_t *soft,
                      size_t soft_stride, int n_streams, int frames_per_stream,
                      const int8_t *history_in, int8_t *history_out, uint8_t *out);

/* Whole-ensemble variant: decode `n_subchannels` subchannels of the same frames in one
 * call (all on one stream; EEP-A, EEP-B and UEP profiles).  Per subchannel i:
 *   history_in[i] / history_out[i]   as above, may be NULL pointers inside the arrays
 *   out[i]                           [n_streams][frames_per_stream*4][bitrate_i*3]
 * The three pointer arrays are HOST arrays of DEVICE pointers.  Subchannels must not overlap in the CIF. */
int dabgpu_msc_decode_multi_dev(dabgpu_ctx *ctx, const dabgpu_subchannel *sc, int n_subchannels,
                                const int8_t *d_soft, size_t soft_stride, int n_streams, int frames_per_stream,
                                const int8_t *const *d_history_in, int8_t *const *d_history_out,
                                uint8_t *const *d_out, void *stream);

/* A7..A12 in one call: what BasicRadio::Process does with a frame (/root/reference/src/radio_block.cpp:42) --
 * the FIC and every listed sub-channel -- for a whole batch of frames.  Arguments as dabgpu_fic_decode_dev and
 * dabgpu_msc_decode_multi_dev.  Large batches whose streams are a multiple of 16 frames long put the FIC and all
 * sub-channels through one pair of launches (their codewords then share the machine instead of queueing up behind
 * each other); any other shape is decoded part by part with the same results. */
int dabgpu_decode_frames_dev(dabgpu_ctx *ctx, const int8_t *d_soft, size_t soft_stride, int n_streams,
                             int frames_per_stream, uint8_t *d_fib, uint8_t *d_crc_ok, const dabgpu_subchannel *sc,
                             int n_subchannels, const int8_t *const *d_history_in, int8_t *const *d_history_out,
                             uint8_t *const *d_out, void *stream);

/* Host-pointer form for the plugin's one-frame-at-a-time use (BasicRadio::Process, src/radio_block.cpp:42): the
 * frames are uploaded ONCE, the FIC and every sub-channel are decoded from that copy, the results come back in one
 * batch of copies, one synchronisation.  history_in / history_out / out are HOST arrays of HOST pointers. */
int dabgpu_decode_frames(dabgpu_ctx *ctx, const int8_t *soft, size_t soft_stride, int n_streams, int frames_per_stream,
                         uint8_t *fib, uint8_t *crc_ok, const dabgpu_subchannel *sc, int n_subchannels,
                         const int8_t *const *history_in, int8_t *const *history_out, uint8_t *const *out);

/* The same for ONE stream fed frame after frame, as the plugin does: the time de-interleaver state of every listed
 * sub-channel stays on the device between calls (keyed by start address and size; a sub-channel seen for the first
 * time starts from erasures, and so does one that was left out of the previous call: a ring that misses a frame no
 * longer continues the stream and is dropped), so a call is one upload of the soft bits, the decode, one download of
 * all results.
 * dabgpu_decode_stream_reset drops the kept state (Radio_Block::reset_radio). */
int dabgpu_decode_stream_frames(dabgpu_ctx *ctx, const int8_t *soft, size_t soft_stride, int n_frames, uint8_t *fib,
                                uint8_t *crc_ok, const dabgpu_subchannel *sc, int n_subchannels, uint8_t *const *out);
int dabgpu_decode_stream_reset(dabgpu_ctx *ctx);

/* ------------------------------------------------------------------------ */
/* The host-fed ring: dabgpu_ofdm_demod_frames + dabgpu_decode_frames for a    */
/* caller whose samples start in HOST memory (files, a network), pipelined.    */
/* The reference runs these two stages on two threads with a 2-frame ring       */
/* between them (/root/reference/src/radio_block.cpp:23-44: the On_OFDM_Frame   */
/* callback writes, the radio thread reads and calls BasicRadio::Process); here  */
/* the ring is `slots` device staging sets and three engines work at once: the   */
/* upload of batch k+1, the kernels of batch k, the download of batch k-1.       */
/*                                                                            */
/* dabgpu_pipe_open   slots 2..8 (3 keeps all three engines busy); max_frames =   */
/*            the largest n_streams * frames_per_stream a submit will carry;       */
/*            frame_stride of the host IQ ([n][frame_stride] cf32, frame f's first  */
/*            PRS sample at iq + f*frame_stride, 76*2552 samples read).  One ring    */
/*            per context.                                                         */
/* dabgpu_pipe_submit enqueues ONE batch and returns at once: the samples go up,    */
/*            are demodulated -- with freq_offset[n_frames] as                       */
/*            dabgpu_ofdm_demod_frames does, or (freq_offset == NULL) closed loop on   */
/*            the context's stream states as dabgpu_ofdm_demod_streams does, cyclic-   */
/*            prefix or decision-directed per dabgpu_set_stream_loop -- and decoded     */
/*            as dabgpu_decode_frames does (FIC + the listed sub-channels); fib,         */
/*            crc_ok, out[i] ([n_streams][frames_per_stream*4][bitrate*3]) and, when      */
/*            soft != NULL, the soft bits come down.  Batches are processed in the        */
/*            order submitted and CONTINUE their streams: the time de-interleaver of       */
/*            every sub-channel stays on the device between submits (keyed by start         */
/*            address, size and n_streams; a sub-channel seen for the first time, or left    */
/*            out of the previous batch, starts from erasures; dabgpu_pipe_reset drops        */
/*            them all).  Every host buffer must stay valid and untouched until                */
/*            dabgpu_pipe_wait(ticket) returns; buffers from dabgpu_host_alloc move at          */
/*            the link rate (pageable memory works, through the runtime's bounce buffers         */
/*            and without overlap).  A submit that finds every slot busy first waits for           */
/*            the oldest batch to leave the device.                                                */
/* dabgpu_pipe_wait   blocks until that batch's results are in the caller's buffers.               */
/* Kernels run on the context's own stream: the synchronous entry points of the same context         */
/* stay ordered with the ring (and wait behind it).                                                   */
/* ------------------------------------------------------------------------ */
int dabgpu_pipe_open(dabgpu_ctx *ctx, int slots, int max_frames, size_t frame_stride);
int dabgpu_pipe_submit(dabgpu_ctx *ctx, const float *iq, int n_streams, int frames_per_stream, const float *freq_offset,
                       float fine_freq_update_beta, const dabgpu_subchannel *sc, int n_subchannels, int8_t *soft,
                       uint8_t *fib, uint8_t *crc_ok, uint8_t *const *out, int64_t *ticket);
int dabgpu_pipe_wait(dabgpu_ctx *ctx, int64_t ticket);
int dabgpu_pipe_reset(dabgpu_ctx *ctx);
int dabgpu_pipe_close(dabgpu_ctx *ctx);

/* ------------------------------------------------------------------------ */
/* DAB+ audio super-frame: Fire code, RS(120,110), AU CRC.    */
/* Replaces the checks the reference reports as the "Firecode / RS / AU"       */
/* flags and GetSuperFrameHeader()                                             */
/*   (/root/reference/src/render_radio_block.cpp:414-437).                     */
/*                                                                            */
/* in        super-frame f = 5 consecutive logical frames of a DAB+ subchannel */
/*           (the bytes dabgpu_msc_decode produces) = 15*bitrate bytes,        */
/*           starting at in + f*in_stride.  The caller finds the alignment by   */
/*           trying the 5 possible logical-frame offsets until firecode_ok.     */
/* out       [n][110*s] corrected data part (s = bitrate/8)                    */
/* status[f] firecode_ok (after RS), rs_corrected bytes, rs_uncorrectable       */
/*           codewords (of s), num_aus (0 when the Fire code fails),            */
/*           au_crc_mask (bit a = access unit a passes its CRC16),              */
/*           au_start[0..num_aus] byte offsets into `out`                       */
/* ------------------------------------------------------------------------ */
typedef struct dabgpu_superframe_status {
    int32_t firecode_ok;
    int32_t rs_corrected;
    int32_t rs_uncorrectable;
    int32_t num_aus;
    int32_t au_crc_mask;
    int32_t au_start[8];
    int32_t reserved[3];
} dabgpu_superframe_status;

int dabgpu_dabplus_superframes_dev(dabgpu_ctx *ctx, const uint8_t *d_in, size_t in_stride, int n_superframes,
                                   int bitrate_kbps, uint8_t *d_out, dabgpu_superframe_status *d_status,
                                   void *stream);
int dabgpu_dabplus_superframes(dabgpu_ctx *ctx, const uint8_t *in, size_t in_stride, int n_superframes,
                               int bitrate_kbps, uint8_t *out, dabgpu_superframe_status *status);

/* ------------------------------------------------------------------------ */
/* A9 on its own: batched punctured soft Viterbi (K=7, rate 1/4).             */
/* Replaces the `viterbi` package (/root/reference/CMakeLists.txt:53-54).     */
/* punct     [n_codewords][n_punct] int8 punctured soft bits                   */
/* mask      HOST pointer, 4*nsteps flags (1 = transmitted)                    */
/* out_bits  [n_codewords][(nsteps-6)/8] bytes, MSB first, NOT descrambled     */
/*                                                                            */
/* A DIAGNOSTIC entry point: no call of the reference's path reaches it (the  */
/* FIC and every sub-channel go through dabgpu_fic_decode* / dabgpu_msc_decode* */
/* / dabgpu_decode_*frames*, whose codewords all have nsteps = 96 k + 6 and run */
/* on viterbi_rot_kernel or the lane kernels).  Lengths that are NOT 96 k + 6  */
/* are decoded by a fourth implementation, viterbi_wave_kernel, which no DAB    */
/* profile uses and which exists for this call alone: it lets the tests hold    */
/* the decoder to an exhaustive maximum-likelihood search on short codewords    */
/* (tests/test_gpu_parity.py) and tools/decoder_fuzz.py hold all implementations */
/* to each other.  Not tuned, not part of any measured figure.                  */
/* ------------------------------------------------------------------------ */
int dabgpu_viterbi_dev(dabgpu_ctx *ctx, const int8_t *d_punct, int n_codewords, const uint8_t *mask,
                       int nsteps, uint8_t *d_out_bytes, void *stream);
int dabgpu_viterbi(dabgpu_ctx *ctx, const int8_t *punct, int n_codewords, const uint8_t *mask,
                   int nsteps, uint8_t *out_bytes);

/* ------------------------------------------------------------------------ */
/* Timing helper: duration (ms) of the launches of each kernel family,         */
/* measured with hipEvents on the launch stream, around the kernel launch       */
/* alone, while dabgpu_set_timing(ctx,1) is on (every call to it starts a new   */
/* measurement).  which: 0 = ofdm front end, 1 = fic (alone), 2 = msc / grouped */
/* decode (dabgpu_decode_frames*: the FIC is part of the grouped launch),        */
/* 3 = fft stage.  _last_ = the most recent launch; _mean_ = the mean over the  */
/* launches since timing was switched on (at most the last 32) and how many     */
/* that were.  Both wait for the launches they read.  _mean_ only: 4 / 5 / 6 =  */
/* the parts of 2 when it was the grouped codeword-per-lane launch (batches of  */
/* >= 24 576 codewords): forward pass | traceback | de-interleaver history copy */
/* (DABGPU_ERR_ARG when no timed call took that path).                          */
/* ------------------------------------------------------------------------ */
int dabgpu_set_timing(dabgpu_ctx *ctx, int enable);
int dabgpu_last_kernel_ms(dabgpu_ctx *ctx, int which, float *ms);
int dabgpu_mean_kernel_ms(dabgpu_ctx *ctx, int which, float *mean_ms, int *launches);

#ifdef __cplusplus
}
#endif
#endif
