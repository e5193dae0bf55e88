/*
 * dab_oracle.h -- CPU restatement (the ORACLE) of the DAB Mode-I receive chain.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / the CPU baseline, never as the thing shipped.
 *
 * PARITY UNPINNED.  The reference plugin (/root/reference) vendors its DSP as
 * git submodules (vendor/DAB-Radio, .../vendor/viterbi_decoder) that are EMPTY in
 * the snapshot, pins are unrecoverable (.gitmodules has URLs only), FFTW3f is
 * not installed, and the reference has no tests / golden vectors (SURVEY.md
 * section 0, 4, 8c).  This file therefore restates the published algorithm (ETSI
 * EN 300 401) anchored on the reference's own call sites:
 *   - get_DAB_OFDM_params / get_dab_parameters   src/radio_block.cpp:12-13
 *   - get_DAB_PRS_reference                      src/radio_block.cpp:18-19
 *   - get_DAB_mapper_ref                         src/radio_block.cpp:20-21
 *   - OFDM_Demod::Process -> On_OFDM_Frame       src/dab_module.cpp:25, src/radio_block.cpp:25
 *   - BasicRadio::Process                        src/radio_block.cpp:42
 * and is validated by known-answer round trips through an independent numpy
 * modulator (sdrplusplus-dab-radio-plugin_amd/dabgpu/synth.py) plus the standard's
 * structural invariants (tests/test_oracle_*.py).
 *
 * Conventions this oracle DEFINES (the HIP kernels must reproduce them):
 *   soft bit   int8, +127 = logical 1, -127 = logical 0, 0 = punctured/erased
 *   quantiser  soft = (int8)trunc(-127 * component / max(|re|,|im|))
 *   NCO        y[n] = x[n] * exp(+j*2*pi*phase(n)),  phase(n) = (uint32)(n*dphi)/2^32,
 *              dphi = (int32)lrint(freq_offset * 2^32), n = 0 at first PRS sample
 *   Viterbi    K=7, polys {109,79,83,109} (bit k of the mask taps a[i-k]),
 *              state = last six input bits, newest at LSB, exact integer
 *              correlation metric (maximise), start metric 0 for state 0 and
 *              -8192 otherwise, survivor = older-bit-1 predecessor only if its
 *              candidate is STRICTLY larger, traceback from state 0.
 *   bits->bytes  first bit is the MSB of byte 0.
 */
#ifndef DAB_ORACLE_H
#define DAB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Mode-I constants (ETSI EN 300 401 clause 14.2 / 14.3) ---- */
enum {
    DAB_NB_FFT            = 2048,
    DAB_NB_CP             = 504,
    DAB_NB_SYM_PERIOD     = 2552,
    DAB_NB_NULL_PERIOD    = 2656,
    DAB_NB_FRAME_SYMBOLS  = 76,    /* PRS + 75 data symbols */
    DAB_NB_CARRIERS       = 1536,
    DAB_NB_SYM_BITS       = 3072,
    DAB_NB_FRAME_BITS     = 230400,
    DAB_NB_FRAME_SAMPLES  = 196608,
    DAB_NB_FIC_BITS       = 9216,  /* 3 symbols */
    DAB_NB_FIC_GROUPS     = 4,
    DAB_NB_FIC_GROUP_BITS = 2304,  /* punctured */
    DAB_NB_FIC_MOTHER     = 3096,  /* 4*(768+6) */
    DAB_NB_FIC_STEPS      = 774,
    DAB_NB_FIBS           = 12,
    DAB_NB_CIFS           = 4,
    DAB_NB_CIF_BITS       = 55296, /* 864 CU * 64 */
    DAB_CU_BITS           = 64,
    DAB_TIME_DEPTH        = 16
};

/* ---- tables (A0, A0') ---- */
/* carrier mapper: out[n], n=0..1535 -> index into the 1536-long carrier vector
   ordered k=-768..-1,1..768.  Restates get_DAB_mapper_ref (radio_block.cpp:20). */
void oracle_get_mapper(int32_t *out1536);
/* FFT-bin of carrier-vector index i: i<768 -> 1280+i, else i-767 */
int  oracle_carrier_bin(int i);
/* PRS spectrum on 2048 FFT bins (interleaved re,im); zeros off the 1536 carriers.
   Restates get_DAB_PRS_reference (radio_block.cpp:18). */
void oracle_get_prs(float *out_cf32_2048);
/* puncture vector PI (1..24) as 32 flags */
void oracle_puncture_vector(int pi, uint8_t out32[32]);
/* FIC puncturing: flags over 3096 mother bits (1 = transmitted); returns kept count (2304) */
int  oracle_fic_puncture_mask(uint8_t *mask3096);
/* EEP puncturing for a subchannel: option 0 = A (8n kb/s), 1 = B (32n kb/s); level 1..4.
   Writes flags over (bitrate*24+6)*4 mother bits; returns kept count, or <0 if invalid.
   *out_cu receives the subchannel size in capacity units. */
int  oracle_eep_puncture_mask(int option, int level, int bitrate_kbps, uint8_t *mask, int *out_nsteps, int *out_cu);

/* ---- bit utilities ---- */
void     oracle_prbs(uint8_t *bits, int n);            /* x^9+x^5+1, all ones */
uint16_t oracle_crc16(const uint8_t *bytes, int n);   /* CCITT, init FFFF, complemented */
void     oracle_conv_encode(const uint8_t *bits, int nbits, uint8_t *out_mother /*4*(nbits+6)*/);

/* ---- channel decoder (A8..A12) ---- */
/* exact-integer Viterbi over `nsteps` trellis steps; mother_soft has 4*nsteps int8.
   out_bits gets nsteps-6 decoded bits (one per byte). */
void oracle_viterbi(const int8_t *mother_soft, int nsteps, uint8_t *out_bits);
/* depuncture: scatter punctured soft bits into mother positions (erasures 0) */
void oracle_depuncture(const int8_t *punct, const uint8_t *mask, int n_mother, int8_t *mother);
/* FIC of one frame: 9216 soft bits -> 12 FIBs x 32 bytes + 12 CRC flags */
void oracle_fic_decode(const int8_t *soft9216, uint8_t *fib384, uint8_t *crc_ok12);
/* time de-interleave one logical frame: cifs[k] points at the subchannel bits of
   CIF (t-15+k), k=0..15 (t = newest); out[i] = cifs[d(i%16)][i]. */
void oracle_time_deinterleave(const int8_t *const cifs[16], int nbits, int8_t *out);
/* one logical frame of a subchannel: deinterleaved punctured soft bits -> bytes */
void oracle_msc_decode_lf(const int8_t *deint, const uint8_t *mask, int nsteps, uint8_t *out_bytes);

/* UEP (clause 11.3.1, Table 8): table index 0..63 -> puncture mask (4*(24*bitrate+6) flags), trellis steps,
   transmitted bits WITHOUT the padding, size in CUs.  Returns 0, or -1 for a bad index. */
int oracle_uep_profile(int index, int *bitrate, int *level, int *size_cu);
int oracle_uep_puncture_mask(int index, uint8_t *mask, int *nsteps, int *n_kept, int *size_cu);

/* ---- OFDM front end (A2..A6) ---- */
void oracle_fft2048(const float *in_cf32, float *out_cf32);  /* forward, unnormalised, fp32 */
/* one aligned frame: iq = 76*2552 cf32 starting at the first PRS sample.
   soft: 230400 int8.  Optional outputs (may be NULL):
   spectra 76*2048 cf32, cyc 76 cf32 (cyclic-prefix correlations), dqpsk 75*1536 cf32
   (in carrier order -768..768 sans DC, before the mapper). */
void oracle_ofdm_demod_frame(const float *iq, float freq_offset, int8_t *soft,
                             float *spectra, float *cyc, float *dqpsk);
/* ... and dd4 (may be NULL) 76 cf32: entry l >= 1 = sum over 256 carriers (bins v + 64 m, v < 64, m in {0, 1, 30, 31}, DC
   replaced by bin 768) of u^4, u = (X_l conj X_{l-1}) / |X_l conj X_{l-1}| (taken from A6's scaling:
   no input level overflows the sum) -- the decision-directed frequency-error estimator: whatever
   two bits a differential symbol carries, its fourth power is -|d|^4 exp(j 4 theta), theta = 2 pi (residual offset)
   2552, unambiguous within +-1/(8 2552).  Entry 0 = the cyclic-prefix correlation of the PRS (symbol 0), the only prefix read:
   its angle / (2 pi 2048) is the same residual within +-1/4096 and picks the branch of the fourth-power estimate. */
void oracle_ofdm_demod_frame_dd(const float *iq, float freq_offset, int8_t *soft,
                                float *spectra, float *cyc, float *dqpsk, float *dd4);

/* ---- synchronisation on the phase reference symbol ----
 * Stands behind the RUNNING_COARSE_FREQ_SYNC and RUNNING_FINE_TIME_SYNC states of OFDM_Demod
 * (/root/reference/src/render_radio_block.cpp:195-196; knobs :213-231).
 * sym: 2552 cf32 starting at the CANDIDATE first sample of the PRS cyclic prefix; the FFT window is taken at
 * [504, 2552).  freq_offset is applied first (same NCO as the demodulator, n = 0 at the window start).
 *   coarse: D_k = sum over adjacent carrier pairs b of Q[b+k] * conj(S[b]),  Q[b] = X[b+1] conj X[b],
 *           S[b] = R[b+1] conj R[b];  *k = argmax |D_k|^2 over |k| <= max_coarse (first maximum scanning
 *           k = -max..+max)
 *   fine time: h = IFFT(X[b+k] conj R[b]);  *toff = argmax |h[n]|^2 as a signed offset (n >= 1024 -> n-2048):
 *           the PRS useful part really starts at window start + *toff
 *   *peak_to_mean = |h|^2 peak / mean,  *coarse_peak_to_mean = |D_k|^2 peak / mean over the scanned k */
void oracle_sync_prs(const float *sym, float freq_offset, int max_coarse, int32_t *k, int32_t *toff,
                     float *peak_to_mean, float *coarse_peak_to_mean);
/* The same with the tap choice of the GUI's "Impulse peak distance weight"
 * (impulse_peak_distance_probability, /root/reference/src/render_radio_block.cpp:225) [RECALL: the weighting lives in
 * the absent DAB-Radio sources; this is the form this oracle fixes]:
 *   score[n] = |h[n]|^2 * w^2,  w = 1 - (1 - distance_prob) * |t(n) - expected| / 2552,  t(n) the signed offset;
 *   the peak is the first maximum of the score; *peak_to_mean = |h[peak]|^2 / mean (unweighted).
 *   first_path_rel > 0: the EARLIEST tap within 504 samples before the peak with |h|^2 >= max(first_path_rel *
 *   |h[peak]|^2, 16 * mean) replaces it (alignment to the first significant path; 0 = the scored peak itself).
 * max_coarse = 0: no whole-carrier search (k = 0). */
void oracle_sync_prs_ex(const float *sym, float freq_offset, int max_coarse, int expected, float distance_prob,
                        float first_path_rel, int32_t *k, int32_t *toff, float *peak_to_mean,
                        float *coarse_peak_to_mean);

/* ---- acquisition on an unaligned capture (first half) ----
 * Stands behind the FINDING_NULL_POWER_DIP / READING_NULL_AND_PRS states of OFDM_Demod
 * (/root/reference/src/render_radio_block.cpp:193-194; knobs thresh_null_start / thresh_null_end,
 * :215-235) restated for a whole capture at once:
 *   l1[b]   = sum over the 64 samples of block b of |re| + |im|   (fixed summation tree, see the .c file)
 *   avg     = mean of l1[] (double accumulation, fixed order)
 *   a dip begins at the first block with l1 < thr_start*avg and ends at the first later block with
 *   l1 > thr_end*avg; it is a null symbol when it lasted min_blocks..83 blocks.  The candidate first sample of the
 *   PRS cyclic prefix is (end block)*64 - 48 (the boundary block is more often the one before); only candidates with a whole frame (+512 samples of slack) inside the
 *   capture are kept.  Returns the number of candidates written (<= max_out). */
void oracle_null_block_l1(const float *iq, int64_t n_samples, float *l1);
int  oracle_null_search(const float *iq, int64_t n_samples, float thr_start, float thr_end, int min_blocks,
                        int max_out, int64_t *cands);
/* level_chunk > 0: the thresholds follow the LOCAL signal level instead of the capture's mean -- the mean block norm of
 * every chunk of level_chunk blocks, averaged over the chunks c-2 .. c+2 (what the reference's running average
 * signal_l1.update_beta does for a stream: a slow fade must not look like a null symbol); 0 = the capture's mean. */
int  oracle_null_search_ex(const float *iq, int64_t n_samples, float thr_start, float thr_end, int min_blocks,
                           int level_chunk, int max_out, int64_t *cands);
/* One candidate -> frame: fractional frequency error from the cyclic prefix of the PRS (samples 64..439 of the
 * candidate's prefix against the samples 2048 later), then oracle_sync_prs with that correction.
 *   fine_offset     = -angle(sum conj(x[i]) x[i+2048]) / (2 pi 2048)   cycles/sample
 *   start           = cand + time_offset - margin   (first sample the demodulator should treat as PRS prefix)
 *   freq_offset     = fine_offset - coarse_carriers/2048  (what the demodulator should apply)
 *   flags bit 0     = peak_to_mean >= min_peak_to_mean (locked), bit 1 = the whole frame lies inside the capture */
typedef struct oracle_acquired_frame {
    int64_t start;
    float freq_offset;
    int32_t coarse_carriers;
    float fine_offset;
    float peak_to_mean;
    float coarse_peak_to_mean;
    int32_t flags;
} oracle_acquired_frame;
void oracle_acquire_candidate(const float *iq, int64_t n_samples, int64_t cand, int max_coarse,
                              float min_peak_to_mean, int margin, oracle_acquired_frame *out);
/* with the tap choice of oracle_sync_prs_ex (expected offset 0) */
void oracle_acquire_candidate_ex(const float *iq, int64_t n_samples, int64_t cand, int max_coarse,
                                 float min_peak_to_mean, int margin, float distance_prob, float first_path_rel,
                                 oracle_acquired_frame *out);

/* ---- DAB+ audio super-frame (dabplus_oracle.c) ---- */
uint16_t oracle_firecode(const uint8_t *bytes, int n);
void oracle_rs_encode(const uint8_t *data110, uint8_t *parity10);
int  oracle_rs_decode(uint8_t *cw120);           /* in place; corrected count or -1 */
void oracle_dabplus_superframe(uint8_t *sf, int s, int32_t status[5], int32_t au_start[8]);

#ifdef __cplusplus
}
#endif
#endif
