/*
 * oracle/encoder.c -- TEST INFRASTRUCTURE (see modem_oracle.h header).
 * Restatement of Encoder<value,cmplx,8000> (encode.cc:27-318) and of main()'s
 * payload handling (encode.cc:399-441).  The transmitter is NOT on the
 * accelerated path; it exists because it is the only source of test input.
 */
#include "modem_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* symbol_len / guard_len of the Encoder<value,cmplx,rate> instantiation in use (encode.cc:31-32) */
static _Thread_local int SL = ORC_SYMBOL_LEN, GL = ORC_GUARD_LEN;
enum { SL_MAX = 7680, GL_MAX = 960 };

typedef struct {
	orc_mode md;
	const uint32_t *frozen;
	int code_off, mls0_off, mls1_off;
	orc_cf fdom[SL_MAX], tdom[SL_MAX], temp[SL_MAX], guard[GL_MAX];
	orc_cf fdom4[4 * SL_MAX], tdom4[4 * SL_MAX];
	orc_cf *out;
	size_t pos;
	int papr;
} enc_t;

static inline int bin(int carrier) { return (carrier + SL) % SL; }              /* encode.cc:68-71 */
static inline int bin4(int carrier) { return (carrier + 4 * SL) % (4 * SL); }  /* encode.cc:72-75 */
static inline int nrz(int bit) { return 1 - 2 * bit; }                         /* encode.cc:76-79 */
static inline float cnorm(orc_cf a) { return a.re * a.re + a.im * a.im; }

/* encode.cc:80-100 */
static void improve_papr(enc_t *e)
{
	for (int i = 0; i < 4 * SL; ++i)
		e->fdom4[i].re = e->fdom4[i].im = 0.f;
	for (int i = -SL / 2; i < SL / 2; ++i)
		e->fdom4[bin4(i)] = e->fdom[bin(i)];
	orc_fft(e->tdom4, e->fdom4, 4 * SL, +1);
	const float s4 = sqrtf((float)(4 * SL));
	for (int i = 0; i < 4 * SL; ++i) {
		e->tdom4[i].re /= s4;
		e->tdom4[i].im /= s4;
	}
	for (int i = 0; i < 4 * SL; ++i) {
		float amp = fmaxf(fabsf(e->tdom4[i].re), fabsf(e->tdom4[i].im));
		if (amp > 1.f) {
			e->tdom4[i].re /= amp;
			e->tdom4[i].im /= amp;
		}
	}
	orc_fft(e->fdom4, e->tdom4, 4 * SL, -1);
	for (int i = -SL / 2; i < SL / 2; ++i) {
		if (cnorm(e->temp[bin(i)]) != 0.f) {
			e->temp[bin(i)].re = e->fdom4[bin4(i)].re / s4;
			e->temp[bin(i)].im = e->fdom4[bin4(i)].im / s4;
		} else {
			e->temp[bin(i)].re = e->temp[bin(i)].im = 0.f;
		}
	}
}

/* encode.cc:101-131 */
static void symbol(enc_t *e, int papr_reduction)
{
	for (int i = 0; i < SL; ++i)
		e->temp[i] = e->fdom[i];
	if (papr_reduction && e->papr)
		improve_papr(e);
	orc_fft(e->tdom, e->temp, SL, +1);
	const float s8 = sqrtf((float)(8 * SL));
	for (int i = 0; i < SL; ++i) {
		e->tdom[i].re /= s8;
		e->tdom[i].im /= s8;
	}
	for (int i = 0; i < GL; ++i) {
		float x = (float)i / (float)(GL - 1);
		x = 0.5f * (1.f - cosf((float)M_PI * x));
		orc_cf a = e->guard[i], b = e->tdom[i + SL - GL];
		e->guard[i].re = (1.f - x) * a.re + x * b.re;   /* DSP::lerp */
		e->guard[i].im = (1.f - x) * a.im + x * b.im;
	}
	memcpy(e->out + e->pos, e->guard, sizeof(orc_cf) * GL);      /* encode.cc:127 */
	e->pos += GL;
	memcpy(e->out + e->pos, e->tdom, sizeof(orc_cf) * SL);       /* encode.cc:128 */
	e->pos += SL;
	for (int i = 0; i < GL; ++i)
		e->guard[i] = e->tdom[i];
}

/* encode.cc:132-141 */
static void pilot_block(enc_t *e)
{
	orc_mls seq2;
	orc_mls_init(&seq2, ORC_MLS2_POLY);
	float code_fac = sqrtf((float)SL / (float)e->md.cons_cols);
	memset(e->fdom, 0, sizeof(orc_cf) * (size_t)SL);
	for (int i = e->code_off; i < e->code_off + e->md.cons_cols; ++i) {
		e->fdom[bin(i)].re = code_fac * (float)nrz(orc_mls_next(&seq2));
		e->fdom[bin(i)].im = 0.f;
	}
	symbol(e, 1);
}

/* encode.cc:142-154 */
static void schmidl_cox(enc_t *e)
{
	orc_mls seq0;
	orc_mls_init(&seq0, ORC_MLS0_POLY);
	float mls0_fac = sqrtf((float)(2 * SL) / (float)ORC_MLS0_LEN);
	memset(e->fdom, 0, sizeof(orc_cf) * (size_t)SL);
	e->fdom[bin(e->mls0_off - 2)].re = mls0_fac;
	for (int i = 0; i < ORC_MLS0_LEN; ++i)
		e->fdom[bin(2 * i + e->mls0_off)].re = (float)nrz(orc_mls_next(&seq0));
	for (int i = 0; i < ORC_MLS0_LEN; ++i) {
		orc_cf *a = &e->fdom[bin(2 * i + e->mls0_off)];
		orc_cf b = e->fdom[bin(2 * (i - 1) + e->mls0_off)];
		orc_cf r = { a->re * b.re - a->im * b.im, a->re * b.im + a->im * b.re };
		*a = r;
	}
	symbol(e, 0);
}

/* encode.cc:155-179 */
static void meta_data(enc_t *e, uint64_t md)
{
	uint8_t data[9] = { 0 }, parity[23] = { 0 };
	for (int i = 0; i < 55; ++i)
		orc_set_be_bit(data, i, (int)((md >> i) & 1));
	uint16_t cs = orc_crc16_u64(0xA8F4, md << 9);
	for (int i = 0; i < 16; ++i)
		orc_set_be_bit(data, i + 55, (cs >> i) & 1);
	orc_bch_encode(data, parity);
	orc_mls seq4;
	orc_mls_init(&seq4, ORC_MLS1_POLY);
	float mls1_fac = sqrtf((float)SL / (float)ORC_MLS1_LEN);
	memset(e->fdom, 0, sizeof(orc_cf) * (size_t)SL);
	e->fdom[bin(e->mls1_off - 1)].re = mls1_fac;
	for (int i = 0; i < 71; ++i)
		e->fdom[bin(i + e->mls1_off)].re = (float)nrz(orc_get_be_bit(data, i));
	for (int i = 71; i < ORC_MLS1_LEN; ++i)
		e->fdom[bin(i + e->mls1_off)].re = (float)nrz(orc_get_be_bit(parity, i - 71));
	for (int i = 0; i < ORC_MLS1_LEN; ++i) {
		orc_cf *a = &e->fdom[bin(i + e->mls1_off)];
		orc_cf b = e->fdom[bin(i - 1 + e->mls1_off)];
		orc_cf r = { a->re * b.re - a->im * b.im, a->re * b.im + a->im * b.re };
		*a = r;
	}
	for (int i = 0; i < ORC_MLS1_LEN; ++i) {
		float s = (float)nrz(orc_mls_next(&seq4));
		e->fdom[bin(i + e->mls1_off)].re *= s;
		e->fdom[bin(i + e->mls1_off)].im *= s;
	}
	symbol(e, 1);
}

size_t orc_encode_rate(int rate, orc_cf *out, const uint8_t *inp, int count, int freq_off,
	uint64_t call_sign, int oper_mode, int papr)
{
	orc_rate_cfg rc;
	if (!orc_rate_lookup(rate, &rc))                /* encode.cc:424-439 */
		return 0;
	SL = rc.symbol_len;
	GL = rc.guard_len;
	enc_t *e = (enc_t *)calloc(1, sizeof(enc_t));
	if (!orc_mode_lookup(oper_mode, &e->md)) {   /* encode.cc:281-282 */
		free(e);
		return 0;
	}
	e->frozen = orc_frozen_get(e->md.table);
	e->out = out;
	e->papr = papr;
	int offset = (freq_off * SL) / rate;            /* encode.cc:283 */
	e->code_off = offset - e->md.cons_cols / 2;     /* encode.cc:284 */
	e->mls0_off = offset - ORC_MLS0_LEN + 1;        /* encode.cc:285 */
	e->mls1_off = offset - ORC_MLS1_LEN / 2;        /* encode.cc:286 */
	int8_t *code = (int8_t *)malloc(ORC_CODE_LEN);
	int8_t *mesg = (int8_t *)malloc(44096);
	const int mod_bits = e->md.mod_bits;
	pilot_block(e);                                  /* encode.cc:288 */
	for (int k = 0; k < count; ++k) {
		schmidl_cox(e);
		meta_data(e, (call_sign << 8) | (uint64_t)oper_mode);
		pilot_block(e);
		const uint8_t *p = inp + (size_t)k * ORC_DATA_BYTES;
		for (int i = 0; i < ORC_DATA_BITS; ++i)
			mesg[i] = (int8_t)nrz(orc_get_le_bit(p, i));
		uint32_t crc = orc_crc32_bytes(0xD419CC15u, p, ORC_DATA_BYTES);
		for (int i = 0; i < 32; ++i)
			mesg[i + ORC_DATA_BITS] = (int8_t)nrz((crc >> i) & 1);
		for (int i = ORC_CRC_BITS; i < e->md.mesg_bits; ++i)
			mesg[i] = 1;
		orc_polar_sysenc(code, mesg, e->frozen, ORC_CODE_ORDER);
		/* shorten(): encode.cc:180-186 */
		for (int i = 0, j = 0, kk = 0; i < ORC_CODE_LEN; ++i)
			if (((e->frozen[i / 32] >> (i % 32)) & 1) || kk++ < ORC_CRC_BITS)
				code[j++] = code[i];
		for (int j = 0; j < e->md.cons_rows; ++j) {
			for (int i = 0; i < e->md.cons_cols; ++i) {
				float b[3];
				const int8_t *c = code + mod_bits * (e->md.cons_cols * j + i);
				for (int t = 0; t < mod_bits; ++t)
					b[t] = (float)c[t];
				orc_cf m = mod_bits == 3 ? orc_psk8_map(b) : orc_psk4_map(b);
				orc_cf *a = &e->fdom[bin(i + e->code_off)];
				orc_cf r = { a->re * m.re - a->im * m.im, a->re * m.im + a->im * m.re };
				*a = r;
			}
			symbol(e, 1);
		}
	}
	memset(e->fdom, 0, sizeof(orc_cf) * (size_t)SL);
	symbol(e, 1);                                    /* encode.cc:311-313 */
	size_t n = e->pos;
	free(code);
	free(mesg);
	free(e);
	return n;
}

size_t orc_encode(orc_cf *out, const uint8_t *inp, int count, int freq_off,
	uint64_t call_sign, int oper_mode, int papr)
{
	return orc_encode_rate(ORC_RATE, out, inp, count, freq_off, call_sign, oper_mode, papr);
}

/* main(): encode.cc:399-441 */
size_t orc_encode_pcm_rate(int rate, void *pcm, int bits, int channels, const uint8_t *payload,
	int count, int freq_off, const char *call_sign, int oper_mode)
{
	orc_mode md;
	orc_rate_cfg rc;
	if (!orc_mode_lookup(oper_mode, &md) || !orc_rate_lookup(rate, &rc))
		return 0;
	long long cs = orc_base37_encode(call_sign);
	if (cs <= 0 || cs >= 129961739795077LL)      /* encode.cc:358 */
		return 0;
	uint8_t *inp = (uint8_t *)malloc((size_t)count * ORC_DATA_BYTES);
	memcpy(inp, payload, (size_t)count * ORC_DATA_BYTES);
	for (int j = 0; j < count; ++j)
		orc_scramble(inp + (size_t)j * ORC_DATA_BYTES, ORC_DATA_BYTES);   /* encode.cc:417-419 */
	size_t syms = 2 + (size_t)count * (3 + (size_t)md.cons_rows);
	size_t total = 2 * (size_t)rate + syms * (size_t)(rc.symbol_len + rc.guard_len);   /* encode.cc:423,441 */
	orc_cf *z = (orc_cf *)calloc(total, sizeof(orc_cf));
	size_t n = orc_encode_rate(rate, z + rate, inp, count, freq_off, (uint64_t)cs, oper_mode, 1);
	(void)n;
	orc_quantise(pcm, bits, channels, z, total);
	free(z);
	free(inp);
	return total;
}

size_t orc_encode_pcm(void *pcm, int bits, int channels, const uint8_t *payload,
	int count, int freq_off, const char *call_sign, int oper_mode)
{
	return orc_encode_pcm_rate(ORC_RATE, pcm, bits, channels, payload, count, freq_off, call_sign, oper_mode);
}
