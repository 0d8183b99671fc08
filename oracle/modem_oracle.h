/*
 * modem_oracle.h -- CPU restatement ("oracle") of the aicodix/modem mode-6..13
 * OFDM transmit + receive chain.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load, link or execute anything under oracle/.  The product library
 * (modem_amd/csrc) never includes this header and never links this code.
 *
 * PARITY STATUS: "parity unpinned" at the third-party boundary.  The
 * reference (decode.cc / encode.cc) takes almost all of its arithmetic from
 * the un-vendored, un-pinned sibling checkouts aicodix/dsp and aicodix/code
 * (Makefile:2 "-I../dsp -I../code").  Those headers are absent, so the
 * reference cannot be compiled and has no golden vectors of its own.  What
 * IS pinned against real reference material:
 *   - psk.hh (8PSK/QPSK map/hard/soft): checked against the real header via
 *     oracle/_ref and the committed tests/golden/psk_vectors.json
 *   - polar_tables.hh: the frozen masks are regenerated from freezer.cc's
 *     recipe and pinned by SHA-256 of the reference's own table
 *   - every constant / index rule of decode.cc and encode.cc (cited inline)
 * Everything taken from the absent headers is restated from the call-site
 * contract in decode.cc/encode.cc.
 *
 * All citations "decode.cc:N", "encode.cc:N", "psk.hh:N", "freezer.cc:N" are
 * file:line into the reference tree.
 */
#ifndef MODEM_ORACLE_H
#define MODEM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } orc_cf;

/* ---- constants, mode 6 @ 8 kHz (decode.cc:171-189, 305-312) ------------- */
enum {
	ORC_RATE = 8000,
	ORC_SYMBOL_LEN = 1280,          /* decode.cc:171 */
	ORC_GUARD_LEN = 160,            /* decode.cc:173 */
	ORC_FILTER_LEN = 21,            /* decode.cc:172 */
	ORC_DATA_BITS = 43040,          /* decode.cc:174 */
	ORC_DATA_BYTES = 5380,
	ORC_CRC_BITS = 43072,           /* decode.cc:175 */
	ORC_CODE_ORDER = 16,
	ORC_CODE_LEN = 65536,
	ORC_MLS0_LEN = 127,             /* decode.cc:182 */
	ORC_MLS0_POLY = 0x89,           /* 0b10001001, decode.cc:184 */
	ORC_MLS1_LEN = 255,             /* decode.cc:185 */
	ORC_MLS1_POLY = 0x12b,          /* 0b100101011, decode.cc:187 */
	ORC_MLS2_POLY = 0x951,          /* 0b100101010001, encode.cc:39 */
	ORC_BUFFER_LEN = 8640,          /* decode.cc:188 */
	ORC_SEARCH_POS = 2880,          /* decode.cc:189 */
	ORC_BCH_N = 255, ORC_BCH_K = 71, ORC_OSD_ORDER = 4, /* decode.cc:199 */
	ORC_MAX_LIST = 8,
	ORC_CONS_MAX = 32400,           /* decode.cc:178 */
	ORC_ROWS_MAX = 126,             /* decode.cc:181 */
	ORC_COLS_MAX = 512,
	ORC_FRAME_SAMPLES = 95200       /* one-frame file from encode @8k */
};

/* the four sample rates main() instantiates (decode.cc:590-602, encode.cc:424-436) and
 * their derived lengths (decode.cc:171-173,188-189).  The plain entry points
 * below are the 8 kHz instantiation; the *_rate variants take the rate. */
typedef struct { int rate, symbol_len, guard_len, filter_len, buffer_len, search_pos; } orc_rate_cfg;
int orc_rate_lookup(int rate, orc_rate_cfg *c);   /* 1 = supported */
size_t orc_frame_samples(int rate, int oper_mode, int count);   /* sample frames of encode's output file */

/* sample formats of the in-memory stream handed to the decoder */
enum { ORC_FMT_S16 = 0, ORC_FMT_U8 = 1, ORC_FMT_F32 = 2 };

/* per-frame status; mirrors every failure exit of Decoder::Decoder */
enum {
	ORC_OK = 0,
	ORC_NO_SYNC = 1,        /* decode.cc:393-394 pcm ran dry while searching */
	ORC_OSD_ERROR = 2,      /* decode.cc:418-421 */
	ORC_HEADER_CRC = 3,     /* decode.cc:429-432 */
	ORC_BAD_MODE = 4,       /* decode.cc:434-437 */
	ORC_BAD_CALLSIGN = 5,   /* decode.cc:439-442 */
	ORC_PAYLOAD_CRC = 6     /* decode.cc:542-545 */
};

typedef struct {
	int oper_mode, cons_cols, cons_rows, mod_bits, cons_bits, mesg_bits, cons_cnt;
	int table;              /* 0: frozen_64800_43072, 1: frozen_64512_43072 */
	int band_width;
} orc_mode;

/* decode.cc:302-374 / encode.cc:197-270,363-387 ; returns 0 if unsupported */
int orc_mode_lookup(int oper_mode, orc_mode *m);

/* ---- small exactness-critical primitives -------------------------------- */
/* CODE::Xorshift32 (decode.cc:613, encode.cc:417) */
typedef struct { uint32_t y; } orc_xorshift32;
void orc_xorshift32_init(orc_xorshift32 *s);
uint32_t orc_xorshift32_next(orc_xorshift32 *s);
void orc_scramble(uint8_t *buf, int len);  /* XOR with the byte stream */

/* CODE::MLS (decode.cc:238,407; encode.cc:134,144,165) */
typedef struct { int poly, test, reg; } orc_mls;
void orc_mls_init(orc_mls *m, int poly);
int orc_mls_next(orc_mls *m);

/* CODE::CRC<T>(poly), reflected, init 0 (decode.cc:197-198) */
uint16_t orc_crc16_u64(uint16_t poly, uint64_t data);         /* crc0(md<<9) */
uint32_t orc_crc32_bit(uint32_t poly, uint32_t crc, int bit); /* crc1(bool) */
uint32_t orc_crc32_bytes(uint32_t poly, const uint8_t *p, int n);

int orc_get_be_bit(const uint8_t *buf, int i);
void orc_set_be_bit(uint8_t *buf, int i, int v);
int orc_get_le_bit(const uint8_t *buf, int i);
void orc_set_le_bit(uint8_t *buf, int i, int v);

long long orc_base37_encode(const char *str);             /* encode.cc:320-335 */
void orc_base37_decode(char *str, long long val, int len);/* decode.cc:155-159 */

/* ---- FFT: DSP::FastFourierTransform<N,cmplx,SIGN> contract -------------- */
/* out-of-place, unnormalised, natural order, sign=-1 forward / +1 backward */
void orc_fft(orc_cf *out, const orc_cf *in, int n, int sign);

/* ---- PSK (psk.hh) -------------------------------------------------------- */
void orc_psk8_hard(float *b, orc_cf c);                    /* psk.hh:118-123 */
void orc_psk8_soft(float *b, orc_cf c, float precision);   /* psk.hh:125-130 */
orc_cf orc_psk8_map(const float *b);                       /* psk.hh:132-139 */
void orc_psk4_hard(float *b, orc_cf c);                    /* psk.hh:70-74 */
void orc_psk4_soft(float *b, orc_cf c, float precision);   /* psk.hh:76-80 */
orc_cf orc_psk4_map(const float *b);                       /* psk.hh:82-85 */

/* ---- polar code ---------------------------------------------------------- */
/* freezer.cc:15-32 recipe; table 0 = (64800,43072), 1 = (64512,43072).
 * Writes 2048 words, bit i of word i/32 set = frozen. */
void orc_frozen_table(int table, uint32_t *frozen);
const uint32_t *orc_frozen_get(int table);                 /* cached */
/* CODE::PolarSysEnc<int8_t> (encode.cc:302): NRZ in/out */
void orc_polar_sysenc(int8_t *code, const int8_t *mesg, const uint32_t *frozen, int level);
/* CODE::PolarEncoder (decode.cc:256): non-systematic, one lane, NRZ */
void orc_polar_enc(int8_t *code, const int8_t *mesg, const uint32_t *frozen, int level);
/* CODE::PolarListDecoder<SIMD<float,L>,16> (decode.cc:530).
 * llr[1<<level]; mesg_out[count*L] (+1/-1, lane-minor, final lane order);
 * metric_out[L]; returns count of unfrozen leaves. */
int orc_polar_list_decode(float *metric_out, int8_t *mesg_out, const float *llr,
	const uint32_t *frozen, int level, int L);

int orc_polar_lane_mesg(const float *llr, const uint32_t *frozen, int level, int L,
	uint8_t *lane_mesg, int mesg_bytes, float *metric);
/* NOT reference code: the checker of the build's "SC dominance" certificate (polar.c).  The sign-following
 * path of the list decoder alone, with one lane's arithmetic of orc_polar_list_decode: hard[1<<level] = its re-encoded
 * codeword (+1/-1), *metric = its path metric, *min_fork = min over the information leaves of fl(metric so far + |llr|).
 * min_fork > metric  =>  that path is lane 0 of orc_polar_list_decode for every list size.  Returns the number of
 * information leaves. */
int orc_polar_sc_path(const float *llr, const uint32_t *frozen, int level, int8_t *hard, float *metric, float *min_fork);

/* ---- BCH(255,71) + OSD --------------------------------------------------- */
void orc_bch_encode(const uint8_t *data /*9 B*/, uint8_t *parity /*23 B*/); /* encode.cc:164 */
void orc_bch_genmat(int8_t *genmat /*255*71*/);                              /* decode.cc:378-384 */
/* CODE::OrderedStatisticsDecoder<255,71,4> (decode.cc:417) */
int orc_osd_decode(uint8_t *hard /*32 B, BE bits*/, const int8_t *soft /*255*/, const int8_t *genmat);

/* ---- DSP pieces ---------------------------------------------------------- */
/* DSP::TheilSenEstimator (decode.cc:488-494) */
void orc_theil_sen(const float *x, const float *y, int n, float *slope, float *yint);
/* DSP::Hilbert<cmplx,21> coefficients: reco + 5 imco */
void orc_hilbert_coeffs(float *reco, float *imco /*5*/);
void orc_hilbert_coeffs_n(int taps, float *reco, float *imco /*(taps-1)/4*/);
/* front end: D0/D1.  raw interleaved samples -> complex stream z[n_frames] */
void orc_front_end(const void *samples, int fmt, int channels, size_t n, orc_cf *z);
void orc_front_end_rate(int rate, const void *samples, int fmt, int channels, size_t n, orc_cf *z);

/* ---- WAV ----------------------------------------------------------------- */
typedef struct {
	int rate, bits, channels;
	size_t frames;
	void *data;      /* raw PCM body as stored (u8 / s16 / s24 / s32), malloc'd */
	int fmt;         /* ORC_FMT_* if directly usable, else -1 */
} orc_wav;
int orc_wav_read(const char *name, orc_wav *w);   /* 0 ok */
void orc_wav_free(orc_wav *w);
/* DSP::WriteWAV semantic: clamp, nearbyint(v*(2^(bits-1)-1)), 8-bit offset 128 */
int orc_wav_write(const char *name, int rate, int bits, int channels,
	const orc_cf *z, size_t frames);

/* ---- encoder (encode.cc:27-318), test-vector source ---------------------- */
/* Produces the complex baseband-at-offset stream exactly as Encoder writes it:
 * leading pilot | per payload: S&C, meta, pilot, rows data | trailing zero symbol.
 * 'inp' = count*5380 bytes ALREADY scrambled (encode.cc:415-419 done by caller
 * or by orc_encode_stream's scramble flag).  Returns number of complex samples
 * written to out (caller provides capacity >= (2+count*(3+rows))*1440). */
size_t orc_encode(orc_cf *out, const uint8_t *inp, int count, int freq_off,
	uint64_t call_sign, int oper_mode, int papr);
/* whole-file helper: rate silence + frames + rate silence, quantised like
 * WriteWAV to 'bits' and returned as raw PCM (u8 or s16), interleaved
 * channels.  payload = count*5380 UNSCRAMBLED bytes.  Returns sample frames. */
size_t orc_encode_pcm(void *pcm, int bits, int channels, const uint8_t *payload,
	int count, int freq_off, const char *call_sign, int oper_mode);
size_t orc_encode_rate(int rate, orc_cf *out, const uint8_t *inp, int count, int freq_off,
	uint64_t call_sign, int oper_mode, int papr);
size_t orc_encode_pcm_rate(int rate, void *pcm, int bits, int channels, const uint8_t *payload,
	int count, int freq_off, const char *call_sign, int oper_mode);

/* ---- decoder (decode.cc:161-557) ----------------------------------------- */
typedef struct {
	int32_t status;
	int32_t symbol_pos;       /* decode.cc:400 (window coordinate) */
	int64_t sc_start;         /* stream index of the S&C symbol body start */
	float cfo_rad;            /* decode.cc:399 coarse */
	float cfo_fine;           /* decode.cc:501 */
	float sfo_slope;          /* avg Theil-Sen slope (decode.cc:496) */
	int32_t oper_mode;
	uint64_t call_sign;
	int32_t best_lane;
	int32_t bit_flips;        /* decode.cc:555 */
	float esn0_db_last;       /* decode.cc:518, last row */
	int32_t n_sync_rejects;   /* correlator falling edges rejected (decode.cc:140-145) */
} orc_result;

/* optional stage taps; any pointer may be NULL */
typedef struct {
	int8_t *hdr_soft;      /* [255]    decode.cc:413-416 */
	orc_cf *cons_raw;      /* [cons_cnt] after decode.cc:464-477 */
	orc_cf *cons_rot;      /* [cons_cnt] after decode.cc:481-495 */
	float *slope, *yint;   /* [rows] */
	float *precision;      /* [rows]   decode.cc:517 */
	float *llr;            /* [65536]  after lengthen, decode.cc:529 */
	float *metric;         /* [L] */
	uint8_t *lane_mesg;    /* [L][5512] systematic message bits (<= 44096), LE packed, per lane */
} orc_taps;

/* samples: interleaved raw stream (the WAV body). payload: 5380 B, descrambled
 * iff descramble!=0 (main(), decode.cc:613-615).  On failure payload is zeroed
 * (documented deviation from decode.cc:588 F9). */
int orc_decode(const void *samples, int fmt, int channels, size_t n_frames,
	int skip_count, int list_size, int descramble,
	uint8_t *payload, orc_result *res, orc_taps *taps);

/* same but from an already-conditioned complex stream (after D1) */
int orc_decode_cf(const orc_cf *z, size_t n, int skip_count, int list_size,
	int descramble, uint8_t *payload, orc_result *res, orc_taps *taps);

int orc_decode_rate(int rate, const void *samples, int fmt, int channels, size_t n_frames,
	int skip_count, int list_size, int descramble,
	uint8_t *payload, orc_result *res, orc_taps *taps);
int orc_decode_cf_rate(int rate, const orc_cf *z, size_t n, int skip_count, int list_size,
	int descramble, uint8_t *payload, orc_result *res, orc_taps *taps);

/* ---- numerics the reference leaves open (-Ofast, absent headers): switches to the PLAIN forms ----------------
 * The default restatement fixes a few fp32 evaluation orders so that the GPU kernels can reproduce them bit for
 * bit.  Each flag below switches one of them back to the plain form a scalar build of the
 * reference would most likely use; tests/test_oracle_numerics.py shows that payload, status, sync position and
 * header fields do not depend on any of them (the winning lane index may, with ORC_NUM_SURVIVORS_UNSORTED).
 * Process-wide, not thread-safe against concurrent decodes: set it before a batch. */
enum {
	ORC_NUM_RATE0_LEAFWALK = 1,      /* all-frozen nodes are walked leaf by leaf: each frozen leaf adds max(0,-llr) to the
	                                  * metric when it is reached (default: nodes of 2..128 leaves charged once, butterfly order) */
	ORC_NUM_SURVIVORS_UNSORTED = 2,  /* the L survivors of a fork stay in candidate order 2k+u (what a partition such as
	                                  * std::nth_element may leave) instead of rank order */
	ORC_NUM_SMA_TREE = 4,            /* sliding sums of decode.cc:86-90 as a fp32 ring of leaves under a binary add tree
	                                  * (default: differences of double prefix sums) */
	ORC_NUM_PHASOR_RECURSIVE = 8,    /* NCO as prev *= delta; prev /= |prev| in fp32 (default: closed form, phase in double) */
	ORC_NUM_SNR_FP32 = 16,           /* sp/np of decode.cc:507-517 accumulated term by term in fp32 (default: per-row double) */
	ORC_NUM_BLOCKDC_FP32 = 32        /* the DC blocker of decode.cc:299 as an fp32 recurrence (default: state in double, outputs rounded once) */
};
void orc_set_numerics(unsigned flags);
unsigned orc_get_numerics(void);

/* batch helper for the cpu_baseline: n frames at fixed stride, OpenMP over
 * frames when built with -fopenmp. Returns threads used. */
int orc_decode_batch(const void *samples, int fmt, int channels, size_t frames_per,
	size_t stride_bytes, int n, int list_size, uint8_t *payload /*n*5380*/,
	orc_result *res /*n*/, int threads);
/* test helper: the same, plus per frame { list decoder's lane-0 metric, M*, min_fork of orc_polar_sc_path on the frame's LLRs,
 * 1 if the payload decoder ran }: what the build's SC-dominance certificate is checked against */
int orc_decode_batch_sc(const void *samples, int fmt, int channels, size_t frames_per,
	size_t stride_bytes, int n, int list_size, uint8_t *payload, orc_result *res, float *sc, int threads);

/* ---- build-owned channel models (aicodix/disorders is absent) - */
/* counter-based RNG: splitmix64(seed,frame,index) -> Box-Muller. noise_db is a
 * LEVEL: per-complex-sample noise power 10^(noise_db/10) rel. full scale 1.0 */
void orc_chan_awgn(orc_cf *z, size_t n, float noise_db, uint64_t seed, uint64_t frame);
void orc_chan_cfo(orc_cf *z, size_t n, float hz, int rate);
/* resample by (1+ppm*1e-6) with 8-tap windowed-sinc; out has same length */
void orc_chan_sfo(orc_cf *out, const orc_cf *in, size_t n, float ppm);
void orc_chan_multipath(orc_cf *out, const orc_cf *in, size_t n,
	const int *delays, const orc_cf *gains, int ntaps);
/* quantise complex stream to PCM as WriteWAV does */
void orc_quantise(void *pcm, int bits, int channels, const orc_cf *z, size_t n);

#ifdef __cplusplus
}
#endif
#endif
