/*
 * oracle/decoder.c -- TEST INFRASTRUCTURE (see modem_oracle.h header).
 *
 * Restatement of SchmidlCox<> (decode.cc:37-153) and Decoder<float,
 * Complex<float>,8000> (decode.cc:161-557).  The reference pulls one sample at
 * a time through a BipBuffer; here the whole conditioned stream z[] is resident
 * and the 8640-sample window at "time n" (n = index of the newest consumed
 * sample) is samples[j] = z[n-8639+j], zero for negative indices.
 *
 * Numerics the reference leaves open and how they are fixed here
 * (reference is built -Ofast, Makefile:2, so its own fp32 intermediates are not
 * reproducible; north_star: bits exact, intermediates within 1e-5):
 *  - SMA4 sliding sums (sma.hh ABSENT): "sum of the last N inputs".  Computed
 *    as differences of double-precision prefix sums of the exact (double)
 *    products, then rounded once to fp32.
 *  - Phasor (phasor.hh ABSENT, recursive NCO): closed form, phase in double.
 *  - running sp/np of decode.cc:507-517: accumulated in double per row, rounded
 *    once per row.
 */
#include "modem_oracle.h"
#include <malloc.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* constants of the Decoder<value,cmplx,rate> instantiation in use (decode.cc:171-173,188-189,590-602) */
static _Thread_local orc_rate_cfg RC = { ORC_RATE, ORC_SYMBOL_LEN, ORC_GUARD_LEN, ORC_FILTER_LEN, 8640, 2880 };
#define SL (RC.symbol_len)
#define GL (RC.guard_len)
#define HS (SL / 2)                      /* correlator symbol_len, decode.cc:196 */
#define MATCH_LEN (GL | 1)               /* decode.cc:41 */
#define MATCH_DEL ((MATCH_LEN - 1) / 2)  /* decode.cc:42 */
#define BUFFER_LEN (RC.buffer_len)
#define SEARCH_POS (RC.search_pos)
enum { HS_MAX = 3840 };
static const float TWO_PI = 6.28318530717958647692f, PI_F = 3.14159265358979323846f;

static inline int bin(int carrier) { return (carrier + SL) % SL; }       /* decode.cc:219-222 */
static inline int binh(int carrier) { return (carrier + HS) % HS; }      /* decode.cc:58-61 */
static inline int nrz(int bit) { return 1 - 2 * bit; }                   /* decode.cc:223-226 */
static inline float cnorm(orc_cf a) { return a.re * a.re + a.im * a.im; }
static inline orc_cf cmul(orc_cf a, orc_cf b)
{
	orc_cf r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re };
	return r;
}
static inline orc_cf cconj(orc_cf a) { orc_cf r = { a.re, -a.im }; return r; }
/* DSP::Complex operator/: a*conj(b)/norm(b) */
static inline orc_cf cdiv(orc_cf a, orc_cf b)
{
	orc_cf n = cmul(a, cconj(b));
	float d = cnorm(b);
	orc_cf r = { n.re / d, n.im / d };
	return r;
}
/* decode.cc:62-70 == decode.cc:227-235 */
static inline orc_cf demod_or_erase(orc_cf curr, orc_cf prev)
{
	orc_cf zero = { 0.f, 0.f };
	if (!(cnorm(prev) > 0.f))
		return zero;
	orc_cf cons = cdiv(curr, prev);
	if (!(cnorm(cons) <= 4.f))
		return zero;
	return cons;
}
static inline orc_cf zat(const orc_cf *z, size_t n, long i)
{
	orc_cf zero = { 0.f, 0.f };
	return (i >= 0 && (size_t)i < n) ? z[i] : zero;
}
/* unit phasor e^{j*omega*k}: closed form of DSP::Phasor (omega in fp32 as stored) */
static inline orc_cf phasor(float omega, long k)
{
	double a = (double)omega * (double)k;
	orc_cf r = { (float)cos(a), (float)sin(a) };
	return r;
}

/* DSP::Phasor (phasor.hh ABSENT): osc() returns the current unit phasor and advances it by omega.  Default: closed
 * form e^{j omega n} with the phase in double.  ORC_NUM_PHASOR_RECURSIVE: the recursive NCO in fp32,
 * prev *= delta; prev /= |prev| after every call. */
typedef struct { float omega; long n; orc_cf prev, delta; int recursive; } osc_t;
static void osc_init(osc_t *o, float omega)
{
	o->omega = omega;
	o->n = 0;
	o->prev.re = 1.f; o->prev.im = 0.f;
	o->delta.re = cosf(omega); o->delta.im = sinf(omega);
	o->recursive = (orc_get_numerics() & ORC_NUM_PHASOR_RECURSIVE) != 0;
}
static inline orc_cf osc_next(osc_t *o)
{
	if (!o->recursive)
		return phasor(o->omega, o->n++);
	orc_cf tmp = o->prev;
	orc_cf p = cmul(o->prev, o->delta);
	float a = sqrtf(cnorm(p));
	o->prev.re = p.re / a;
	o->prev.im = p.im / a;
	return tmp;
}
static inline void osc_skip(osc_t *o, int count)
{
	if (!o->recursive) { o->n += count; return; }
	for (int i = 0; i < count; ++i)
		(void)osc_next(o);
}

/* DSP::SMA4 (sma.hh / swa.hh ABSENT) in its plain fp32 form: a ring of NUM leaves under a binary add tree stored
 * heap-style in tree[1 .. 2 NUM), parent = left + right; the window sum is tree[1] (ORC_NUM_SMA_TREE) */
typedef struct { float *tree; int num, leaf; } swa_t;
static void swa_init(swa_t *w, int num)
{
	w->tree = (float *)calloc((size_t)2 * num, sizeof(float));
	w->num = num;
	w->leaf = num;
}
static inline float swa_push(swa_t *w, float v)
{
	w->tree[w->leaf] = v;
	for (int child = w->leaf, parent = child / 2; parent; child = parent, parent /= 2)
		w->tree[parent] = w->tree[child & ~1] + w->tree[child | 1];
	if (++w->leaf >= 2 * w->num)
		w->leaf = w->num;
	return w->tree[1];
}

/* ---- SchmidlCox ---------------------------------------------------------- */
typedef struct {
	orc_cf kern[HS_MAX]; /* decode.cc:80-82 */
	double *Sc_re, *Sc_im, *Sp, *Sm;   /* prefix sums, index shifted by 1 */
	float *timing;
	float *argP;         /* ORC_NUM_SMA_TREE: arg(P) for every t (the tree form is streamed once), else NULL */
	size_t n;
	const orc_cf *z;
} sc_t;

/* decode.cc:236-244 mls0_seq + decode.cc:76-83 ctor */
static void sc_init_kern(orc_cf *kern)
{
	orc_cf seq[HS], tmp[HS];
	orc_mls seq0;
	orc_mls_init(&seq0, ORC_MLS0_POLY);
	memset(seq, 0, sizeof(orc_cf) * (size_t)HS);
	const int mls0_off = -ORC_MLS0_LEN + 1;   /* decode.cc:183 */
	for (int i = 0; i < ORC_MLS0_LEN; ++i) {
		seq[(i + mls0_off / 2 + HS) % HS].re = (float)nrz(orc_mls_next(&seq0));
	}
	orc_fft(tmp, seq, HS, -1);
	for (int i = 0; i < HS; ++i) {
		kern[i].re = tmp[i].re / (float)HS;
		kern[i].im = -tmp[i].im / (float)HS;
	}
}

/* sliding quantities of decode.cc:86-91 for every n */
static void sc_prepare(sc_t *s, const orc_cf *z, size_t n)
{
	s->z = z;
	s->n = n;
	/* c[u] = z[u]*conj(z[u+640]) (decode.cc:86), p[u] = |z[u]|^2 (decode.cc:87) */
	s->Sc_re = (double *)malloc(sizeof(double) * (n + 1) * 4);
	s->Sc_im = s->Sc_re + (n + 1);
	s->Sp = s->Sc_im + (n + 1);
	s->Sm = s->Sp + (n + 1);
	s->timing = (float *)malloc(sizeof(float) * (n + 1));
	s->argP = NULL;
	if (orc_get_numerics() & ORC_NUM_SMA_TREE) {
		/* decode.cc:86-91 sample by sample: cor / pwr / match are fp32 tree sums over rings of HS, 2 HS, MATCH_LEN inputs */
		swa_t cre, cim, pw, mt;
		swa_init(&cre, HS); swa_init(&cim, HS); swa_init(&pw, 2 * HS); swa_init(&mt, MATCH_LEN);
		s->argP = (float *)malloc(sizeof(float) * (n + 1));
		for (size_t t = 0; t < n; ++t) {
			long u = (long)t - (BUFFER_LEN - 1) + SEARCH_POS + HS;   /* samples[search_pos+symbol_len] */
			orc_cf a = zat(z, n, u), b = zat(z, n, u + HS);
			orc_cf c = cmul(a, cconj(b));
			float Pre = swa_push(&cre, c.re), Pim = swa_push(&cim, c.im);
			float R = 0.5f * swa_push(&pw, cnorm(b));
			float min_R = 0.0001f * HS;
			R = fmaxf(R, min_R);
			s->timing[t] = swa_push(&mt, (Pre * Pre + Pim * Pim) / (R * R));
			s->argP[t] = atan2f(Pim, Pre);
		}
		free(cre.tree); free(cim.tree); free(pw.tree); free(mt.tree);
		return;
	}
	s->Sc_re[0] = s->Sc_im[0] = s->Sp[0] = s->Sm[0] = 0.0;
	for (size_t u = 0; u < n; ++u) {
		double ar = z[u].re, ai = z[u].im, br = 0.0, bi = 0.0;
		if (u + HS < n) { br = z[u + HS].re; bi = z[u + HS].im; }
		s->Sc_re[u + 1] = s->Sc_re[u] + (ar * br + ai * bi);
		s->Sc_im[u + 1] = s->Sc_im[u] + (ai * br - ar * bi);
		s->Sp[u + 1] = s->Sp[u] + (ar * ar + ai * ai);
	}
	#define PFX(S, i) ((i) <= 0 ? 0.0 : (S)[(size_t)(i) > n ? n : (size_t)(i)])
	for (size_t t = 0; t < n; ++t) {
		/* P sums c[u] for u in (t-5119-640, t-5119]; R sums p[u] for u in (t-4479-1280, t-4479] */
		long hc = (long)t - (BUFFER_LEN - 1 - (SEARCH_POS + HS)) + 1;   /* exclusive end */
		long hp = (long)t - (BUFFER_LEN - 1 - (SEARCH_POS + 2 * HS)) + 1;
		float Pre = (float)(PFX(s->Sc_re, hc) - PFX(s->Sc_re, hc - HS));
		float Pim = (float)(PFX(s->Sc_im, hc) - PFX(s->Sc_im, hc - HS));
		float R = 0.5f * (float)(PFX(s->Sp, hp) - PFX(s->Sp, hp - 2 * HS));
		float min_R = 0.0001f * HS;                      /* decode.cc:88 */
		R = fmaxf(R, min_R);
		double m = (double)((Pre * Pre + Pim * Pim) / (R * R));   /* decode.cc:90: the fp32 expression; only the moving sum is double */
		s->Sm[t + 1] = s->Sm[t] + m;
		s->timing[t] = (float)(s->Sm[t + 1] - PFX(s->Sm, (long)t + 1 - MATCH_LEN));
	}
}
/* arg(P) at time t (decode.cc:91 feeds arg(P) into an 80-deep delay) */
static float sc_phase(const sc_t *s, long t)
{
	size_t n = s->n;
	if (t < 0)
		return 0.f;
	if (s->argP)
		return s->argP[t];
	long hc = t - (BUFFER_LEN - 1 - (SEARCH_POS + HS)) + 1;
	float Pre = (float)(PFX(s->Sc_re, hc) - PFX(s->Sc_re, hc - HS));
	float Pim = (float)(PFX(s->Sc_im, hc) - PFX(s->Sc_im, hc - HS));
	return atan2f(Pim, Pre);
}
static void sc_free(sc_t *s)
{
	free(s->Sc_re);
	free(s->timing);
	free(s->argP);
}

/* decode.cc:110-151: the work done on the falling edge at time t.
 * symbol_pos is the window coordinate; returns 1 on accept. */
static int sc_process(const sc_t *s, long t, int index_max, float phase_max,
	int *symbol_pos_out, float *cfo_rad_out)
{
	orc_cf tmp0[HS], tmp1[HS], tmp2[HS];
	float frac_cfo = phase_max / (float)HS;                 /* decode.cc:110 */
	int symbol_pos = SEARCH_POS - index_max;            /* decode.cc:114 */
	long base = t - (BUFFER_LEN - 1);
	osc_t osc;                                              /* decode.cc:112-113 */
	osc_init(&osc, frac_cfo);
	for (int i = 0; i < HS; ++i)                            /* decode.cc:117-118 */
		tmp1[i] = cmul(zat(s->z, s->n, base + i + symbol_pos + HS), osc_next(&osc));
	orc_fft(tmp0, tmp1, HS, -1);
	for (int i = 0; i < HS; ++i)                            /* decode.cc:120-121 */
		tmp1[i] = demod_or_erase(tmp0[i], tmp0[binh(i - 1)]);
	orc_fft(tmp0, tmp1, HS, -1);
	for (int i = 0; i < HS; ++i)
		tmp0[i] = cmul(tmp0[i], s->kern[i]);
	orc_fft(tmp2, tmp0, HS, +1);
	int shift = 0;
	float peak = 0.f, next = 0.f;
	for (int i = 0; i < HS; ++i) {                          /* decode.cc:127-139 */
		float power = cnorm(tmp2[i]);
		if (power > peak) {
			next = peak;
			peak = power;
			shift = i;
		} else if (power > next) {
			next = power;
		}
	}
	if (peak <= next * 4.f)                                 /* decode.cc:140-141 */
		return 0;
	int pos_err = (int)nearbyintf(atan2f(tmp2[shift].im, tmp2[shift].re) * (float)HS / TWO_PI);
	if (abs(pos_err) > GL / 2)                              /* decode.cc:144-145 */
		return 0;
	symbol_pos -= pos_err;
	float cfo_rad = (float)shift * (TWO_PI / (float)HS) - frac_cfo;   /* decode.cc:148 */
	if (cfo_rad >= PI_F)
		cfo_rad -= TWO_PI;
	*symbol_pos_out = symbol_pos;
	*cfo_rad_out = cfo_rad;
	return 1;
}

/* state of the per-sample trigger logic, decode.cc:93-108 */
typedef struct {
	int collect;          /* SchmittTrigger state */
	float timing_max, phase_max;
	int index_max;
	long t;               /* next sample index to consume */
} trig_t;

/* run the correlator from tr->t until it returns true (decode.cc:392-396).
 * Returns 1 with *t_hit = time of the accepted falling edge, 0 when the
 * stream ends (pcm->good() false). */
static int sc_search(const sc_t *s, trig_t *tr, int *symbol_pos, float *cfo_rad, int *rejects)
{
	const float lo = (float)(0.17 * MATCH_LEN), hi = (float)(0.19 * MATCH_LEN);   /* decode.cc:76 */
	for (; (size_t)tr->t < s->n; ) {
		long t = tr->t++;
		float timing = s->timing[t];
		int prev = tr->collect;
		if (timing > hi) tr->collect = 1;          /* SchmittTrigger */
		else if (timing < lo) tr->collect = 0;
		int process = prev && !tr->collect;        /* FallingEdgeTrigger */
		if (!tr->collect && !process)
			continue;
		if (tr->timing_max < timing) {             /* decode.cc:99-105 */
			tr->timing_max = timing;
			tr->phase_max = sc_phase(s, t - MATCH_DEL);
			tr->index_max = MATCH_DEL;
		} else if (tr->index_max < HS + GL + MATCH_DEL) {
			++tr->index_max;
		}
		if (!process)
			continue;
		int index_max = tr->index_max;
		float phase_max = tr->phase_max;
		tr->index_max = 0;                         /* decode.cc:115-116 */
		tr->timing_max = 0.f;
		if (sc_process(s, t, index_max, phase_max, symbol_pos, cfo_rad))
			return 1;
		++*rejects;
	}
	return 0;
}

/* ---- Decoder --------------------------------------------------------------- */
static void zero_result(orc_result *r)
{
	memset(r, 0, sizeof(*r));
	r->best_lane = -1;
	r->sc_start = -1;
}

int orc_decode_cf(const orc_cf *z, size_t n, int skip_count, int list_size,
	int descramble, uint8_t *payload, orc_result *res, orc_taps *taps)
{
	return orc_decode_cf_rate(ORC_RATE, z, n, skip_count, list_size, descramble, payload, res, taps);
}

int orc_decode_cf_rate(int rate, const orc_cf *z, size_t n, int skip_count, int list_size,
	int descramble, uint8_t *payload, orc_result *res, orc_taps *taps)
{
	orc_result rr;
	zero_result(&rr);
	memset(payload, 0, ORC_DATA_BYTES);
	if (!orc_rate_lookup(rate, &RC)) {                     /* decode.cc:603-605 */
		orc_rate_lookup(ORC_RATE, &RC);
		rr.status = ORC_NO_SYNC;
		*res = rr;
		return -1;
	}
	const int L = (list_size == 4) ? 4 : 8;
	sc_t sc;
	sc_init_kern(sc.kern);
	sc_prepare(&sc, z, n);
	static _Thread_local int8_t genmat[ORC_BCH_N * ORC_BCH_K];
	static _Thread_local int have_genmat;
	if (!have_genmat) {          /* decode.cc:378-384 */
		orc_bch_genmat(genmat);
		have_genmat = 1;
	}
	trig_t tr = { 0, 0.f, 0.f, 0, 0 };
	orc_mode md;
	memset(&md, 0, sizeof(md));
	int okay, symbol_pos = 0;
	float cfo_rad = 0.f;
	long t_hit = 0;
	osc_t osc;                   /* decode.cc:403: keeps its phase from the header symbol through the payload symbols */
	osc_init(&osc, 0.f);
	orc_cf fdom[SL], tdom[SL];
	do {                                                   /* decode.cc:390-448 */
		okay = 0;
		if (!sc_search(&sc, &tr, &symbol_pos, &cfo_rad, &rr.n_sync_rejects)) {
			rr.status = rr.status ? rr.status : ORC_NO_SYNC;
			sc_free(&sc);
			*res = rr;
			return rr.status;
		}
		t_hit = tr.t - 1;
		rr.symbol_pos = symbol_pos;
		rr.cfo_rad = cfo_rad;
		rr.sc_start = t_hit - (BUFFER_LEN - 1) + symbol_pos;
		long base = t_hit - (BUFFER_LEN - 1);
		/* the Phasor keeps its phase across omega() changes; only continuity
		 * within one frame matters (differential demodulation), so the phase
		 * origin is restarted at each header attempt */
		osc_init(&osc, -cfo_rad);
		for (int i = 0; i < SL; ++i)                       /* decode.cc:403-405 */
			tdom[i] = cmul(zat(z, n, base + i + symbol_pos + (SL + GL)), osc_next(&osc));
		orc_fft(fdom, tdom, SL, -1);
		orc_mls seq1;
		orc_mls_init(&seq1, ORC_MLS1_POLY);
		const int mls1_off = -ORC_MLS1_LEN / 2;            /* decode.cc:186 */
		for (int i = 0; i < ORC_MLS1_LEN; ++i) {
			float sgn = (float)nrz(orc_mls_next(&seq1));
			fdom[bin(i + mls1_off)].re *= sgn;
			fdom[bin(i + mls1_off)].im *= sgn;
		}
		int8_t soft[ORC_MLS1_LEN];
		uint8_t data[32];
		for (int i = 0; i < ORC_MLS1_LEN; ++i) {           /* decode.cc:412-416 */
			float v = nearbyintf(127.f * demod_or_erase(fdom[bin(i + mls1_off)], fdom[bin(i - 1 + mls1_off)]).re);
			soft[i] = (int8_t)fminf(fmaxf(v, -128.f), 127.f);
		}
		if (taps && taps->hdr_soft)
			memcpy(taps->hdr_soft, soft, ORC_MLS1_LEN);
		int unique = orc_osd_decode(data, soft, genmat);   /* decode.cc:417 */
		if (!unique) {
			rr.status = ORC_OSD_ERROR;
			continue;
		}
		uint64_t mdw = 0;
		for (int i = 0; i < 55; ++i)
			mdw |= (uint64_t)orc_get_be_bit(data, i) << i;
		uint16_t cs = 0;
		for (int i = 0; i < 16; ++i)
			cs |= (uint16_t)(orc_get_be_bit(data, i + 55) << i);
		if (orc_crc16_u64(0xA8F4, mdw << 9) != cs) {       /* decode.cc:428-432 */
			rr.status = ORC_HEADER_CRC;
			continue;
		}
		rr.oper_mode = (int)(mdw & 255);
		if (!orc_mode_lookup(rr.oper_mode, &md)) {         /* decode.cc:433-437 */
			rr.status = ORC_BAD_MODE;
			continue;
		}
		rr.call_sign = mdw >> 8;
		if ((mdw >> 8) == 0 || (mdw >> 8) >= 129961739795077ULL) {   /* decode.cc:439-442 */
			rr.status = ORC_BAD_CALLSIGN;
			continue;
		}
		rr.status = ORC_OK;
		okay = 1;
	} while (skip_count--);
	if (!okay) {                                           /* decode.cc:450-451 */
		sc_free(&sc);
		*res = rr;
		return rr.status;
	}
	sc_free(&sc);

	const int cons_rows = md.cons_rows, cons_cols = md.cons_cols, mod_bits = md.mod_bits;
	const int code_off = -cons_cols / 2;                   /* decode.cc:454 */
	const uint32_t *frozen = orc_frozen_get(md.table);
	orc_cf *cons = (orc_cf *)malloc(sizeof(orc_cf) * ORC_CONS_MAX);
	float *code = (float *)malloc(sizeof(float) * ORC_CODE_LEN);
	orc_cf prev[ORC_COLS_MAX];
	/* decode.cc:456-462: pilot symbol body sits at sc_start + 2*1440 */
	long body = rr.sc_start + 2 * (SL + GL);
	for (int i = 0; i < SL; ++i)
		tdom[i] = cmul(zat(z, n, body + i), osc_next(&osc));
	osc_skip(&osc, GL);
	orc_fft(fdom, tdom, SL, -1);
	for (int j = 0; j < cons_rows; ++j) {                  /* decode.cc:464-477 */
		body += SL + GL;
		for (int i = 0; i < SL; ++i)
			tdom[i] = cmul(zat(z, n, body + i), osc_next(&osc));
		osc_skip(&osc, GL);
		for (int i = 0; i < cons_cols; ++i)
			prev[i] = fdom[bin(i + code_off)];
		orc_fft(fdom, tdom, SL, -1);
		for (int i = 0; i < cons_cols; ++i)
			cons[cons_cols * j + i] = demod_or_erase(fdom[bin(i + code_off)], prev[i]);
	}
	if (taps && taps->cons_raw)
		memcpy(taps->cons_raw, cons, sizeof(orc_cf) * (size_t)md.cons_cnt);
	{                                                      /* decode.cc:479-504 */
		float index[ORC_COLS_MAX], phase[ORC_COLS_MAX];
		float sum_slope = 0.f, sum_yint = 0.f;
		for (int j = 0; j < cons_rows; ++j) {
			for (int i = 0; i < cons_cols; ++i) {
				float tmp[3];
				orc_cf c = cons[cons_cols * j + i], m;
				if (mod_bits == 3) { orc_psk8_hard(tmp, c); m = orc_psk8_map(tmp); }
				else { orc_psk4_hard(tmp, c); m = orc_psk4_map(tmp); }
				index[i] = (float)(i + code_off);
				orc_cf d = cmul(c, cconj(m));
				phase[i] = atan2f(d.im, d.re);
			}
			float slope, yint;
			orc_theil_sen(index, phase, cons_cols, &slope, &yint);
			if (taps && taps->slope) taps->slope[j] = slope;
			if (taps && taps->yint) taps->yint[j] = yint;
			sum_slope += slope;
			sum_yint += yint;
			for (int i = 0; i < cons_cols; ++i) {
				float a = -(yint + slope * (float)(i + code_off));   /* -tse(i+code_off) */
				orc_cf rot = { cosf(a), sinf(a) };                   /* DSP::polar(1, a) */
				cons[cons_cols * j + i] = cmul(cons[cons_cols * j + i], rot);
			}
		}
		float avg_slope = sum_slope / (float)cons_rows;
		float avg_yint = sum_yint / (float)cons_rows;
		rr.sfo_slope = avg_slope;
		rr.cfo_fine = cfo_rad + avg_yint / (float)(SL + GL);   /* decode.cc:501 */
	}
	if (taps && taps->cons_rot)
		memcpy(taps->cons_rot, cons, sizeof(orc_cf) * (size_t)md.cons_cnt);
	{                                                      /* decode.cc:505-523 */
		float sp = 0.f, np = 0.f;
		const int snr_fp32 = (orc_get_numerics() & ORC_NUM_SNR_FP32) != 0;
		for (int j = 0; j < cons_rows; ++j) {
			double dsp = 0.0, dnp = 0.0;
			for (int i = 0; i < cons_cols; ++i) {
				float tmp[3];
				orc_cf c = cons[cons_cols * j + i], h;
				if (mod_bits == 3) { orc_psk8_hard(tmp, c); h = orc_psk8_map(tmp); }
				else { orc_psk4_hard(tmp, c); h = orc_psk4_map(tmp); }
				if (snr_fp32) {                            /* decode.cc:512-515 as written */
					orc_cf e = { c.re - h.re, c.im - h.im };
					sp += cnorm(h);
					np += cnorm(e);
					continue;
				}
				double er = (double)c.re - h.re, ei = (double)c.im - h.im;
				dsp += (double)h.re * h.re + (double)h.im * h.im;
				dnp += er * er + ei * ei;
			}
			if (!snr_fp32) {
				sp = (float)((double)sp + dsp);
				np = (float)((double)np + dnp);
			}
			float precision = sp / np;
			if (taps && taps->precision) taps->precision[j] = precision;
			rr.esn0_db_last = 10.f * log10f(precision);    /* DSP::decibel */
			for (int i = 0; i < cons_cols; ++i) {
				float *b = code + mod_bits * (cons_cols * j + i);
				if (mod_bits == 3) orc_psk8_soft(b, cons[cons_cols * j + i], precision);
				else orc_psk4_soft(b, cons[cons_cols * j + i], precision);
			}
		}
	}
	/* lengthen(): decode.cc:245-253 */
	for (int i = ORC_CODE_LEN - 1, j = md.cons_bits - 1, k = md.mesg_bits - 1; i >= 0; --i) {
		if (((frozen[i / 32] >> (i % 32)) & 1) || k-- < ORC_CRC_BITS)
			code[i] = code[j--];
		else
			code[i] = 9000.f;    /* PolarHelper<float>::quant(9000) */
	}
	if (taps && taps->llr)
		memcpy(taps->llr, code, sizeof(float) * ORC_CODE_LEN);
	int8_t *mesg = (int8_t *)malloc((size_t)44096 * L);
	int8_t *lane_u = (int8_t *)malloc(44096);
	int8_t *mess = (int8_t *)malloc(ORC_CODE_LEN);
	float metric[ORC_MAX_LIST];
	int count = orc_polar_list_decode(metric, mesg, code, frozen, ORC_CODE_ORDER, L);   /* decode.cc:530 */
	(void)count;
	if (taps && taps->metric)
		memcpy(taps->metric, metric, sizeof(float) * (size_t)L);
	/* systematic(): decode.cc:254-261, per lane */
	for (int k = 0; k < L; ++k) {
		for (int i = 0; i < md.mesg_bits; ++i)
			lane_u[i] = mesg[(size_t)i * L + k];
		orc_polar_enc(mess, lane_u, frozen, ORC_CODE_ORDER);
		for (int i = 0, j = 0; i < ORC_CODE_LEN && j < md.mesg_bits; ++i)
			if (!((frozen[i / 32] >> (i % 32)) & 1))
				mesg[(size_t)(j++) * L + k] = mess[i];
	}
	if (taps && taps->lane_mesg) {
		memset(taps->lane_mesg, 0, (size_t)L * 5512);
		for (int k = 0; k < L; ++k)
			for (int i = 0; i < md.mesg_bits; ++i)
				orc_set_le_bit(taps->lane_mesg + (size_t)k * 5512, i, mesg[(size_t)i * L + k] < 0);
	}
	int best = -1;
	for (int k = 0; k < L; ++k) {                          /* decode.cc:532-541 */
		uint32_t crc = 0;
		for (int i = 0; i < ORC_CRC_BITS; ++i)
			crc = orc_crc32_bit(0xD419CC15u, crc, mesg[(size_t)i * L + k] < 0);
		if (crc == 0) {
			best = k;
			break;
		}
	}
	rr.best_lane = best;
	if (best < 0) {
		rr.status = ORC_PAYLOAD_CRC;                       /* decode.cc:542-545 */
	} else {
		int flips = 0;
		for (int i = 0, j = 0; i < ORC_DATA_BITS; ++i, ++j) {   /* decode.cc:546-554 */
			while ((frozen[j / 32] >> (j % 32)) & 1)
				++j;
			int received = code[j] < 0.f;
			int decoded = mesg[(size_t)i * L + best] < 0;
			flips += received != decoded;
			orc_set_le_bit(payload, i, decoded);
		}
		rr.bit_flips = flips;
		if (descramble)
			orc_scramble(payload, ORC_DATA_BYTES);         /* decode.cc:613-615 */
	}
	free(mesg);
	free(lane_u);
	free(mess);
	free(cons);
	free(code);
	*res = rr;
	return rr.status;
}

int orc_decode_rate(int rate, const void *samples, int fmt, int channels, size_t n_frames,
	int skip_count, int list_size, int descramble,
	uint8_t *payload, orc_result *res, orc_taps *taps)
{
	orc_cf *z = (orc_cf *)malloc(sizeof(orc_cf) * (n_frames ? n_frames : 1));
	orc_front_end_rate(rate, samples, fmt, channels, n_frames, z);
	int r = orc_decode_cf_rate(rate, z, n_frames, skip_count, list_size, descramble, payload, res, taps);
	free(z);
	return r;
}

int orc_decode(const void *samples, int fmt, int channels, size_t n_frames,
	int skip_count, int list_size, int descramble,
	uint8_t *payload, orc_result *res, orc_taps *taps)
{
	return orc_decode_rate(ORC_RATE, samples, fmt, channels, n_frames, skip_count, list_size, descramble, payload, res, taps);
}

int orc_decode_batch(const void *samples, int fmt, int channels, size_t frames_per,
	size_t stride_bytes, int n, int list_size, uint8_t *payload, orc_result *res, int threads)
{
	int used = 1;
	(void)threads;
	/* every frame allocates and frees a few MB of work arrays: keep them in the threads' malloc arenas instead of one
	 * mmap / munmap (and a page-fault storm) per array - with tens of threads that serialises in the kernel and the
	 * batch scales 13x on 64 threads instead of ~50x */
	mallopt(M_MMAP_THRESHOLD, 512 << 20);
	mallopt(M_TRIM_THRESHOLD, 1 << 30);
	#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
	for (int f = 0; f < n; ++f)
		orc_decode((const uint8_t *)samples + (size_t)f * stride_bytes, fmt, channels, frames_per,
			0, list_size, 1, payload + (size_t)f * ORC_DATA_BYTES, &res[f], NULL);
#ifdef _OPENMP
	used = threads > 0 ? threads : 1;
#endif
	return used;
}

/* test helper (NOT reference code): orc_decode_batch that also leaves, per frame, what the build's SC-dominance certificate is
 * checked against - sc[f] = { lane 0's metric of the list decoder, M* and min_fork of the sign-following path on the same LLRs
 * (orc_polar_sc_path), 1 if the frame reached the payload decoder at all } */
int orc_decode_batch_sc(const void *samples, int fmt, int channels, size_t frames_per,
	size_t stride_bytes, int n, int list_size, uint8_t *payload, orc_result *res, float *sc, int threads)
{
	mallopt(M_MMAP_THRESHOLD, 512 << 20);
	mallopt(M_TRIM_THRESHOLD, 1 << 30);
	#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
	for (int f = 0; f < n; ++f) {
		float *llr = (float *)malloc(sizeof(float) * ORC_CODE_LEN);
		int8_t *hard = (int8_t *)malloc(ORC_CODE_LEN);
		float metric[ORC_MAX_LIST];
		orc_taps taps;
		memset(&taps, 0, sizeof(taps));
		taps.llr = llr;
		taps.metric = metric;
		for (int i = 0; i < ORC_CODE_LEN; ++i)
			llr[i] = NAN;
		orc_decode((const uint8_t *)samples + (size_t)f * stride_bytes, fmt, channels, frames_per,
			0, list_size, 1, payload + (size_t)f * ORC_DATA_BYTES, &res[f], &taps);
		float *o = sc + (size_t)f * 4;
		o[0] = o[1] = o[2] = o[3] = 0.f;
		if (res[f].status == 0 || res[f].status == 6) {
			o[0] = metric[0];
			orc_polar_sc_path(llr, orc_frozen_get(res[f].oper_mode >= 10), 16, hard, &o[1], &o[2]);
			o[3] = 1.f;
		}
		free(llr);
		free(hard);
	}
	return threads > 0 ? threads : 1;
}
