/*
 * ofdmrx.h -- C ABI of the MI355X-native OFDM receive path (libofdmrx.so).
 *
 * Drop-in boundary.  The reference has no plugin / FFI layer;
 * its only seams are
 *   (1) the process boundary   decode OUTPUT INPUT [SKIP]       decode.cc:559-563
 *   (2) the in-process seam    Decoder<value,cmplx,rate>(uint8_t *out,
 *           DSP::ReadPCM<value> *pcm, int skip_count)            decode.cc:375
 *       which pulls samples with pcm->read()/channels()/rate()   decode.cc:297-298,590
 *       and leaves 5380 payload bytes in `out`, which main() descrambles
 *       and writes                                               decode.cc:608-617
 * This library replaces seam (2) for batches of independent frames: the caller
 * (the `decode` CLI, a batch driver, or a binding) reads the WAV body into
 * memory and hands over raw PCM; the library returns payload bytes plus a
 * per-frame result struct carrying every diagnostic the reference prints to
 * stderr (decode.cc:400-401,438,446,502-503,517-519,555) and every failure
 * cause (decode.cc:393,419,430,435,440,543).
 *
 * Conventions: plain C, no exceptions across the ABI.  Return value 0 = ok,
 * negative = API / HIP error (ofdmrx_strerror).  A frame that fails to decode
 * is DATA (result.status), never an API error.  All buffers are caller-owned.
 * A handle is not thread-safe: one handle per (host thread, GPU).
 */
#ifndef OFDMRX_H
#define OFDMRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFDMRX_ABI_VERSION 1
/* Minor revisions keep every struct and signature of OFDMRX_ABI_VERSION 1 and add entry points or tighten a check:
 *   1: skip counts outside 0..OFDMRX_MAX_SKIP fail the call with OFDMRX_E_ARG (they used to be clamped)
 *   2: ofdmrx_set_esn0_rows, ofdmrx_list_decoded_frames, ofdmrx_debug_decode_cons, ofdmrx_config.flags bit 1 (OFDMRX_FLAG_SCL_ALWAYS);
 *      frames whose hard decisions already form a codeword are decided by a syndrome check (same outputs)
 *   3: ofdmrx_set_attempt_log (every preamble of a SKIP loop, decode.cc:390-448); frames the syndrome check leaves are
 *      list-decoded from a queue in full residencies of the decoder (same outputs); OFDMRX_TAP_CONS_RAW needs no flag, and the
 *      LLR / METRIC / LANE_MESG taps answer OFDMRX_E_UNSUPPORTED for a frame that never went through the list decoder
 *   4: ofdmrx_decode_batch_device delivers to pinned host memory when both output pointers are pinned host memory
 *   5: ofdmrx_sc_decided_frames, ofdmrx_get_sc_timing, ofdmrx_debug_sc_path, ofdmrx_last_chunk_first_frame, ofdmrx_config.flags bit 2
 *      (OFDMRX_FLAG_NO_SC); `samples` and the frame stride must be multiples of the sample FRAME size (2-channel input: of the I/Q pair);
 *      with pinned host outputs the optional Es/N0 rows and attempt log must be pinned host memory as well (else OFDMRX_E_ARG); frames
 *      with raw bit errors whose sign-following path provably is the list decoder's lane 0 are finished by a list-1 decode of
 *      that path (same outputs)
 *   6: ofdmrx_config.flags bit 3 (OFDMRX_FLAG_TWO_LANES): a device-entry call of four chunks or more runs its second half through a
 *      second pipeline beside the first (same outputs; the handle then holds the state of two pipelines); the list-1 pass takes whole
 *      residencies of its decoders and leaves the rest to the next chunk's run (same outputs) */
#define OFDMRX_ABI_MINOR 6

#define OFDMRX_PAYLOAD_BYTES 5380     /* decode.cc:587  data_len = 43040/8 */
#define OFDMRX_CODE_LEN 65536         /* decode.cc:309  code_order 16 */
#define OFDMRX_FRAME_SAMPLES 95200    /* one-frame file written by encode @ 8 kHz */
#define OFDMRX_MAX_LIST 8
#define OFDMRX_MAX_SKIP 64            /* largest SKIP count per frame (decode.cc:583-585,448); beyond it: OFDMRX_E_ARG */

/* sample formats of the PCM body (what DSP::ReadWAV accepts, decode.cc:576) */
enum { OFDMRX_FMT_S16 = 0, OFDMRX_FMT_U8 = 1, OFDMRX_FMT_F32 = 2 };

/* per-frame status: one value per exit of Decoder::Decoder */
enum {
	OFDMRX_OK = 0,
	OFDMRX_NO_SYNC = 1,        /* decode.cc:393-394  stream ended while searching */
	OFDMRX_OSD_ERROR = 2,      /* decode.cc:418-421 */
	OFDMRX_HEADER_CRC = 3,     /* decode.cc:429-432 */
	OFDMRX_BAD_MODE = 4,       /* decode.cc:434-437 (modes 6..13 are decoded) */
	OFDMRX_BAD_CALLSIGN = 5,   /* decode.cc:439-442 */
	OFDMRX_PAYLOAD_CRC = 6     /* decode.cc:542-545 */
};

/* API error codes */
enum {
	OFDMRX_E_ARG = -1, OFDMRX_E_NOMEM = -2, OFDMRX_E_HIP = -3, OFDMRX_E_NODEV = -4, OFDMRX_E_UNSUPPORTED = -5
};

#define OFDMRX_FLAG_KEEP_RAW_CONS 1
#define OFDMRX_FLAG_SCL_ALWAYS 2
#define OFDMRX_FLAG_NO_SC 4
#define OFDMRX_FLAG_TWO_LANES 8

typedef struct ofdmrx_handle ofdmrx_handle;

typedef struct {
	int32_t abi_version;       /* OFDMRX_ABI_VERSION */
	int32_t sample_rate;       /* 8000, 16000, 44100 or 48000: which Decoder<value,cmplx,rate> this handle is
	                            * (decode.cc:590-602); anything else: OFDMRX_E_UNSUPPORTED (decode.cc:603-605) */
	int32_t list_size;         /* SCL list = SIMD width of the reference build (decode.cc:164-169): 8 (AVX2, the
	                            * benchmarked configuration; 0 = 8) or 4 (the 128-bit build) */
	int32_t device;            /* HIP device ordinal */
	int32_t chunk_frames;      /* frames resident per pass (0 = default: 8192) */
	int32_t max_samples;       /* max samples per frame (0 = ofdmrx_frame_samples(sample_rate, 6)) */
	int32_t descramble;        /* 1 = XOR payload with Xorshift32 like main(), decode.cc:613-615 */
	int32_t flags;             /* bit 0 (OFDMRX_FLAG_KEEP_RAW_CONS, the name is historical): debug taps: run the list decoder for
	                            * every frame and keep its per-lane messages, so that OFDMRX_TAP_LLR / _METRIC / _LANE_MESG
	                            * exist for every frame with a header;
	                            * bit 1 (OFDMRX_FLAG_SCL_ALWAYS): run the list decoder for every frame.  Without either, a frame
	                            * whose channel hard decisions already form a codeword with a valid CRC-32 is finished by that
	                            * syndrome check - the list decoder's lane 0 provably is that codeword - with
	                            * identical payload, status, best_lane and bit_flips, and its LLRs are never written; a frame the
	                            * syndrome check leaves is decoded along its sign-following path alone (list size 1) and finished
	                            * there when that path provably is the list decoder's lane 0 (min over the information leaves of
	                            * fl(metric so far + |llr|) > the path's final metric) and its CRC-32 is zero - again
	                            * with identical payload, status, best_lane (0) and bit_flips; every other frame is list-decoded;
	                            * bit 2 (OFDMRX_FLAG_NO_SC): without that list-1 pass (syndrome check, then the list decoder);
	                            * bit 3 (OFDMRX_FLAG_TWO_LANES): cut a device-entry call of four chunks or more in two and run the second
	                            * half through a second pipeline beside the first (created on first use; it doubles the handle's device
	                            * state, about 18 GB at the default chunk).  For input in which most frames have raw bit errors (-20 dB:
	                            * +6 %, the README's multipath chain: +8 %; clean input: -1 %); wants GPU_MAX_HW_QUEUES=8 or more in the
	                            * environment before the HIP runtime starts.  OFDMRX_LANES=2 / =1 in the
	                            * environment overrides the flag */
	void *stream;              /* hipStream_t to run on, NULL = library-owned stream.  Batches longer than one chunk
	                            * also use library-owned streams for the list decoder and its finishing kernel (chunk
	                            * pipeline); the given stream waits for them, so work enqueued on `stream` after a
	                            * decode call sees the finished batch */
} ofdmrx_config;

/* mirrors the reference's stderr diagnostics */
typedef struct {
	int32_t status;            /* OFDMRX_OK ... */
	int32_t symbol_pos;        /* decode.cc:400 "symbol pos" (window coordinate) */
	int64_t sc_start;          /* stream index of the Schmidl-Cox symbol body, -1 if none */
	float cfo_rad;             /* decode.cc:399,401 coarse cfo, rad/sample */
	float cfo_fine;            /* decode.cc:501,503 finer cfo */
	float sfo_slope;           /* decode.cc:498 average Theil-Sen slope */
	int32_t oper_mode;         /* decode.cc:438 */
	uint64_t call_sign;        /* decode.cc:439-446, base-37 integer */
	int32_t best_lane;         /* decode.cc:532-541, -1 if no lane passed CRC-32 */
	int32_t bit_flips;         /* decode.cc:555: sign(LLR) != decoded bit over the payload positions.  LLRs are fp32 values
	                            * within the 1e-5 intermediate tolerance of a scalar build's; one that sits that close to
	                            * zero may carry either sign, so this diagnostic can differ by a count or two (observed:
	                            * +-2 in 0.3 % of the frames near the waterfall, never above it).  Far below the waterfall of
	                            * the HEADER (48 kHz frames with 5 % raw bit errors) one frame in 192 differed by 5: cfo_rad
	                            * differs in its last bits (1.5e-7 rad/sample), over 440 000 samples that is 0.07 rad of
	                            * carrier phase, which the Theil-Sen stage absorbs with different hard decisions for points
	                            * on a decision boundary; payload, lane and every other field were identical.  Round 4, 43 000
	                            * frames against the oracle: one frame (mode 7, -17 dB) differed by 9, its coarse cfo by
	                            * 2e-6 rad/sample - the same mechanism; mono input (its front end is a blocked scan, the
	                            * reference's a serial fp32 recurrence), 160 000 frames from -30 dB to the waterfall: beyond
	                            * +-2 in 0.05 % of the noisy frames, by up to 11 (21 where half the frames are lost);
	                            * nothing that is decided differed in any frame.  AT the waterfall (-15 / -14.5 dB, where
	                            * 38 % of the frames are lost) 4 of 65 536 frames differed from the scalar restatement in
	                            * something decided: sync position one sample apart (decode.cc:143's nearbyint on a
	                            * boundary), or the list decoder keeping / losing the right path an ulp apart.  The timing
	                            * tie also occurred once in 230 000 mono frames above the waterfall (same payload) */
	float esn0_db_last;        /* decode.cc:517-519, cumulative Es/N0 after the last row */
	int32_t n_sync_rejects;    /* falling edges rejected at decode.cc:140-145 */
} ofdmrx_frame_result;

/* hipEvent timings of the last decode call, milliseconds, summed over chunks.  In a pipelined (multi-chunk) call the
 * polar / finish spans run beside the Theil-Sen / LLR spans of the next chunk: the stage times then overlap and add up to
 * more than the wall time; TOTAL sums the per-chunk latencies (first kernel to last kernel of each chunk). */
enum {
	OFDMRX_T_FRONT = 0, OFDMRX_T_SYNC, OFDMRX_T_HEADER, OFDMRX_T_DEMOD, OFDMRX_T_THEILSEN,
	OFDMRX_T_LLR, OFDMRX_T_POLAR, OFDMRX_T_FINISH, OFDMRX_T_TOTAL, OFDMRX_T_COUNT
};
typedef struct {
	float ms[OFDMRX_T_COUNT];
	int32_t launches[OFDMRX_T_COUNT];   /* kernel launches per stage */
} ofdmrx_timing;

int ofdmrx_abi_version(void);
int ofdmrx_abi_minor(void);
const char *ofdmrx_strerror(int err);

/* replaces `new Decoder<value,cmplx,8000>` (decode.cc:592): allocates device
 * state, tables (frozen mask, twiddles, MLS kernels, BCH generator) once */
int ofdmrx_create(const ofdmrx_config *cfg, ofdmrx_handle **out);
void ofdmrx_destroy(ofdmrx_handle *h);

/*
 * Decode n_frames independent frames.  Frame f occupies
 * samples + f*frame_stride_bytes, samples_per_frame sample frames of
 * `channels` interleaved values (1 = real, 2 = analytic I/Q; decode.cc:578,298); `samples` and frame_stride_bytes are
 * multiples of the sample size (4 for 16-bit I/Q pairs), else OFDMRX_E_ARG.
 * skip_counts[f] (nullable) is decode.cc's SKIP argument (decode.cc:583-585,448): 0..OFDMRX_MAX_SKIP preambles to
 * pass over; a negative or larger count is OFDMRX_E_ARG (the reference would loop to the end of the stream).
 * payload_out: n_frames*5380 bytes, zeroed for failed frames (the reference
 * leaves them uninitialised, decode.cc:588).
 * HOST pointers; blocks until done.
 */
int ofdmrx_decode_batch(ofdmrx_handle *h, const void *samples, int sample_format, int channels,
	size_t samples_per_frame, size_t frame_stride_bytes, size_t n_frames,
	const int32_t *skip_counts, uint8_t *payload_out, ofdmrx_frame_result *results);

/* same with DEVICE pointers (inputs already resident in HBM); asynchronous on
 * the handle's stream.  d_payload_out / d_results are device buffers - or, both of them, PINNED HOST memory (hipHostMalloc, a
 * registered range; revision 1.4): then every chunk's payloads and records are copied out right behind the chunk, beside the
 * next chunk's kernels (frames the list decoder finishes in a later flush are delivered by its last kernel, straight into the
 * pinned arrays), and the batch is on the host when the handle's stream has drained - the host-pointer entry's output half
 * without its input half.  (Pageable host memory is refused: OFDMRX_E_ARG.)  d_skip_counts (nullable) is read back once on
 * the handle's stream before anything is enqueued (the counts steer the host loop), so it is ordered after earlier
 * work on that stream; that read-back is the call's only host synchronisation. */
int ofdmrx_decode_batch_device(ofdmrx_handle *h, const void *d_samples, int sample_format, int channels,
	size_t samples_per_frame, size_t frame_stride_bytes, size_t n_frames,
	const int32_t *d_skip_counts, uint8_t *d_payload_out, ofdmrx_frame_result *d_results);

int ofdmrx_synchronize(ofdmrx_handle *h);
int ofdmrx_get_timing(ofdmrx_handle *h, ofdmrx_timing *t);
int ofdmrx_chunk_frames(ofdmrx_handle *h);
/* A decode call runs its frames in chunks of at most ofdmrx_chunk_frames() frames - and a call whose outputs cross PCIe (the
 * host-pointer entry; the device entry with pinned host outputs) and whose 6144 or more frames fit ONE chunk runs as two halves
 * (the second half's kernels beside the first half's copies), unless the handle has OFDMRX_FLAG_KEEP_RAW_CONS.  The stage taps
 * below belong to the LAST chunk a call ran: this is the index, in that call, of the chunk's first frame (frame 0 of
 * ofdmrx_debug_dump).  Revision 1.5. */
long long ofdmrx_last_chunk_first_frame(ofdmrx_handle *h);
/* decode.cc:506-523 prints one Es/N0 value per constellation row.  rows = n_frames x OFDMRX_ROWS_MAX floats (dB; rows a frame's
 * mode does not have, and frames without a header: 0) in the memory space of the RESULTS of the decode calls that follow: a
 * device pointer for ofdmrx_decode_batch_device, a host pointer for ofdmrx_decode_batch.  NULL (the default) turns it off;
 * ofdmrx_frame_result.esn0_db_last always carries the last row's value. */
#define OFDMRX_ROWS_MAX 126    /* decode.cc:181 rows_max */
int ofdmrx_set_esn0_rows(ofdmrx_handle *h, float *rows);
/* frames of the last decode call that went through the list decoder; the rest were decided by the syndrome certificate
 * (see ofdmrx_config.flags).  -1 if the certificate is off for this handle.  Synchronises the handle's stream.
 * (The certificate is adaptive: after a chunk in which it was tried for 64 frames or more and finished fewer than one in twenty it
 * is tried for a sample of one frame in sixteen only, until a fifth of the sample - summed over chunks until eight frames have been
 * tried - passes again; a frame it was not tried for is list-decoded, with the same outputs.  Every call starts with the certificate on.) */
long long ofdmrx_list_decoded_frames(ofdmrx_handle *h);
/* frames of the last decode call that the list-1 pass finished (neither the syndrome check nor the list decoder); -1 if that
 * pass is off for this handle (OFDMRX_FLAG_KEEP_RAW_CONS / _SCL_ALWAYS / _NO_SC).  Synchronises the handle's stream.  The pass is
 * adaptive like the syndrome check: after a run of 64 entries or more of which it finished fewer than an eighth only a probe sample
 * goes through it - one frame in sixteen of every FOURTH chunk - until an eighth of the sample (summed over chunks until eight frames
 * have been tried) is finished again; the others go straight to the list decoder.  Default layout: one codeword per wave, ten resident
 * decoders per CU (OFDMRX_SC_LB / OFDMRX_SC_WPC in the environment change it); a run takes whole residencies of them and leaves the rest
 * to the next chunk's run (revision 1.6), the last run of a call takes everything. */
long long ofdmrx_sc_decided_frames(ofdmrx_handle *h);
/* hipEvent time and launches of that pass (k_sc + k_sc_finish) in the last decode call, like ofdmrx_timing's stages (which keep
 * their layout); either pointer may be NULL */
int ofdmrx_get_sc_timing(ofdmrx_handle *h, float *ms, int32_t *launches);
/* decode.cc:390-448 prints "symbol pos" / "coarse cfo" and the header's outcome for EVERY preamble the SKIP loop examines, not
 * only for the last one (which ofdmrx_frame_result describes).  log = n_frames x (OFDMRX_MAX_SKIP + 1) records, counts =
 * n_frames numbers of records written (0: the stream ended before any preamble), both in the memory space of the RESULTS of the
 * decode calls that follow (see ofdmrx_set_esn0_rows).  NULL, NULL (the default) turns it off. */
typedef struct {
	int32_t status;            /* OFDMRX_OK, or OFDMRX_OSD_ERROR .. OFDMRX_BAD_CALLSIGN: what decode.cc:417-442 made of this preamble */
	int32_t symbol_pos;        /* decode.cc:400 */
	float cfo_rad;             /* decode.cc:401 */
	int32_t oper_mode;         /* decode.cc:433 (valid from OFDMRX_BAD_MODE on) */
	uint64_t call_sign;        /* decode.cc:439 */
} ofdmrx_attempt;
int ofdmrx_set_attempt_log(ofdmrx_handle *h, ofdmrx_attempt *log, int32_t *counts);

/* ---- stage taps for parity tests (host destination buffers) --------------
 * Valid for frames of the LAST chunk processed (frame index relative to that
 * chunk's first frame, ofdmrx_last_chunk_first_frame()).  The rotated constellation is made on demand (the pipeline never stores it); LLR / METRIC /
 * LANE_MESG exist for frames that went through the list decoder (every frame with a header when the handle was created with
 * OFDMRX_FLAG_KEEP_RAW_CONS or OFDMRX_FLAG_SCL_ALWAYS; LANE_MESG needs the former), otherwise: OFDMRX_E_UNSUPPORTED. */
enum {
	OFDMRX_TAP_HDR_SOFT = 1,   /* int8  [255]      decode.cc:413-416 */
	OFDMRX_TAP_CONS_RAW = 2,   /* cf32  [cons_cnt <= 32400] decode.cc:464-477 (21600 in mode 6) */
	OFDMRX_TAP_CONS_ROT = 3,   /* cf32  [cons_cnt]          decode.cc:481-495 */
	OFDMRX_TAP_SLOPE = 4,      /* f32   [rows <= 126]       (50 in mode 6) */
	OFDMRX_TAP_YINT = 5,       /* f32   [rows] */
	OFDMRX_TAP_PRECISION = 6,  /* f32   [rows]              decode.cc:517 */
	OFDMRX_TAP_LLR = 7,        /* f32   [65536]    decode.cc:529 */
	OFDMRX_TAP_METRIC = 8,     /* f32   [8] */
	OFDMRX_TAP_LANE_MESG = 9,  /* u8    [8][5476]  systematic message bits per lane, LE packed */
	OFDMRX_TAP_ANALYTIC = 10   /* cf32  [samples]  after D1 (mono only) */
};
int ofdmrx_debug_dump(ofdmrx_handle *h, int tap, size_t frame, void *dst, size_t dst_bytes);

/* ---- single-stage entry points for parity tests (HOST pointers) ---------- */
/* D9+D10: CODE::PolarListDecoder + systematic() (decode.cc:530-531) */
int ofdmrx_debug_polar(ofdmrx_handle *h, const float *llr /*n*65536*/, size_t n,
	uint8_t *lane_mesg /*n*8*5476*/, float *metric /*n*8*/);
/* the sign-following path of the list decoder alone (k_sc): n LLR vectors, vector i of the code of mode oper_modes[i] (NULL: all
 * mode 6) -> its re-encoded codeword (bit i = bit i % 8 of byte i / 8), the hard decisions of the LLRs packed alike, its path
 * metric, min over the information leaves of fl(metric so far + |llr|), and whether the rule "min_fork > metric, every |llr| <
 * 6e29" holds.  Any output may be NULL. */
int ofdmrx_debug_sc_path(ofdmrx_handle *h, const float *llr /*n*65536*/, size_t n, const int32_t *oper_modes /*n*/,
	uint8_t *codeword /*n*8192*/, uint8_t *hard /*n*8192*/, float *metric /*n*/, float *min_fork /*n*/, int32_t *rule_ok /*n*/);
/* D5 output (n x 21600 rotated constellation points of mode-6 frames, cf32) -> payloads + results through D6-D10 as the pipeline
 * chains them: use_cert 0 = the list decoder for every frame, 1 = the syndrome certificate in front of it, 2 = syndrome certificate,
 * list-1 pass, list decoder (the default chain), 3 = list-1 pass, list decoder; cert_out (nullable): 1 = the frame was finished
 * by the syndrome certificate, 2 = by the list-1 pass, 0 = by the list decoder (decode.cc:505-555) */
int ofdmrx_debug_decode_cons(ofdmrx_handle *h, const float *cons /*n*21600*2*/, size_t n, int use_cert,
	uint8_t *payload /*n*5380*/, ofdmrx_frame_result *results /*n*/, int32_t *cert_out /*n*/);
/* DSP::TheilSenEstimator::compute on rows of y[cols], x = i - cols/2 (decode.cc:488) */
int ofdmrx_debug_theil_sen(ofdmrx_handle *h, const float *y, size_t rows, int cols,
	float *slope, float *yint);
/* CODE::OrderedStatisticsDecoder<255,71,4> (decode.cc:417): soft n*255 -> hard n*32 (BE bits), unique n */
int ofdmrx_debug_osd(ofdmrx_handle *h, const int8_t *soft, size_t n, uint8_t *hard, int32_t *unique);
/* DSP::FastFourierTransform<1280|640,cmplx,-1|+1> (decode.cc:191,43-44): n transforms */
int ofdmrx_debug_fft(ofdmrx_handle *h, const float *in, size_t n, int len, int sign, float *out);

/* ---- build-owned channel model on device (aicodix/disorders is absent) ----
 * out frame f = base frame (f % n_base) + complex AWGN of power 10^(noise_db/10)
 * (re/im split equally), counter-based RNG keyed by (seed, first_frame+f).
 * 2-channel int16 in and out, DEVICE pointers, samples_per_frame each. */
int ofdmrx_util_awgn_tile(ofdmrx_handle *h, const int16_t *d_base, size_t n_base,
	int16_t *d_out, size_t n_out, size_t samples_per_frame, float noise_db,
	uint64_t seed, uint64_t first_frame);

/* deterministic part of the README.md:49 chain (multipath | cfo | sfo), applied to n 2-channel int16 frames on the
 * device (n <= 65535); ofdmrx_util_awgn_tile then adds the independent noise.  Definitions: DESIGN.md / oracle/channel.c */
typedef struct {
	float cfo_hz;              /* carrier frequency offset, Hz (at the handle's sample rate) */
	float sfo_ppm;             /* sampling frequency offset, ppm */
	int32_t ntaps;             /* multipath taps (0..8); 0 = pass-through */
	int32_t delays[8];         /* samples */
	float gains_re[8], gains_im[8];
} ofdmrx_channel;
/* delays must lie in [0, samples_per_frame); d_in and d_out must not overlap (else OFDMRX_E_ARG) */
int ofdmrx_util_channel(ofdmrx_handle *h, const int16_t *d_in, int16_t *d_out, size_t n_frames,
	size_t samples_per_frame, const ofdmrx_channel *ch);

/* ---- N2: the transmitter on the device (replaces Encoder<value,cmplx,rate>(pcm, inp, count=1, freq_off,
 * call_sign, oper_mode), encode.cc:271,424-436, for batches; rate = the handle's sample_rate).  d_payload:
 * n_frames x 5380 UNSCRAMBLED bytes (what main() reads from the input files, encode.cc:414); d_pcm: n_frames x
 * ofdmrx_frame_samples(rate, mode) x channels int16, exactly the body of the WAV
 * `encode OUT RATE 16 CHANNELS OFFSET MODE CALLSIGN file` writes.  DEVICE pointers.
 * ofdmrx_frame_samples: sample frames of that one-payload file = 2 x rate of silence (encode.cc:423,441) +
 * (rows + 5) symbols; ofdmrx_tx_frame_samples(mode) is the 8 kHz value. */
long ofdmrx_frame_samples(int sample_rate, int oper_mode);
long ofdmrx_tx_frame_samples(int oper_mode);
/* Streams of `count` payloads, as `encode OUT RATE BITS CHANNELS OFFSET MODE CALLSIGN file1 .. fileN` writes them
 * (encode.cc:288-313: pilot | count x (S&C, meta, pilot, rows) | zero symbol, `rate` samples of silence either side),
 * bits = 8 (unsigned, offset 128) or 16.  ofdmrx_stream_samples: sample frames of one such stream.
 * _device: n_streams x count x 5380 payload bytes in, n_streams x samples x channels PCM out, DEVICE pointers;
 * asynchronous on the handle's stream like the decode entry (scratch is kept in the handle: no allocation per call).
 * ofdmrx_tx_encode_stream: the same for ONE stream with HOST pointers (what the `encode` CLI calls). */
long ofdmrx_stream_samples(int sample_rate, int oper_mode, int count);
/* call sign -> the base-37 integer of the header (encode.cc:320-335: ' ' = 0, digits 1..10, letters of either case
 * 11..36), -1 if the string holds any other character.  Valid call signs are 0 < value < 129961739795077
 * (encode.cc:358, decode.cc:439). */
long long ofdmrx_callsign_value(const char *call_sign);
int ofdmrx_tx_encode_stream_device(ofdmrx_handle *h, const uint8_t *d_payload, size_t n_streams, int count,
	int oper_mode, int freq_off, const char *call_sign, int channels, int bits, void *d_pcm);
int ofdmrx_tx_encode_stream(ofdmrx_handle *h, const uint8_t *payload, int count, int oper_mode, int freq_off,
	const char *call_sign, int channels, int bits, void *pcm);
int ofdmrx_tx_encode_device(ofdmrx_handle *h, const uint8_t *d_payload, size_t n_frames, int oper_mode,
	int freq_off, const char *call_sign, int channels, int16_t *d_pcm);

#ifdef __cplusplus
}
#endif
#endif
