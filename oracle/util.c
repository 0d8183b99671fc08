/*
 * oracle/util.c -- TEST INFRASTRUCTURE (see modem_oracle.h header).
 * Small exactness-critical primitives taken by the reference from the absent
 * aicodix/code headers; restated from the call-site contracts.
 */
#include "modem_oracle.h"
#include <string.h>

/* ---- mode table: decode.cc:302-374 == encode.cc:197-270; bandwidths encode.cc:363-387 */
int orc_mode_lookup(int oper_mode, orc_mode *m)
{
	static const struct { int mode, cols, bits, cons_bits, mesg_bits, table, bw; } T[] = {
		{ 6, 432, 3, 64800, 43808, 0, 2700 },   /* decode.cc:305-312 */
		{ 7, 400, 3, 64800, 43808, 0, 2500 },   /* decode.cc:313-320 */
		{ 8, 400, 2, 64800, 43808, 0, 2500 },   /* decode.cc:321-328 */
		{ 9, 360, 2, 64800, 43808, 0, 2250 },   /* decode.cc:329-336 */
		{ 10, 512, 3, 64512, 44096, 1, 3200 },  /* decode.cc:337-344 */
		{ 11, 384, 3, 64512, 44096, 1, 2400 },  /* decode.cc:345-352 */
		{ 12, 384, 2, 64512, 44096, 1, 2400 },  /* decode.cc:353-360 */
		{ 13, 256, 2, 64512, 44096, 1, 1600 },  /* decode.cc:361-368 */
	};
	for (unsigned i = 0; i < sizeof(T) / sizeof(T[0]); ++i) {
		if (T[i].mode != oper_mode)
			continue;
		m->oper_mode = oper_mode;
		m->cons_cols = T[i].cols;
		m->mod_bits = T[i].bits;
		m->cons_bits = T[i].cons_bits;
		m->mesg_bits = T[i].mesg_bits;
		m->table = T[i].table;
		m->band_width = T[i].bw;
		m->cons_cnt = m->cons_bits / m->mod_bits;     /* decode.cc:372 */
		m->cons_rows = m->cons_cnt / m->cons_cols;    /* decode.cc:453 */
		return 1;
	}
	return 0;
}

/* ---- rates: decode.cc:590-602 / encode.cc:424-436 and decode.cc:171-173,188-189 */
int orc_rate_lookup(int rate, orc_rate_cfg *c)
{
	if (rate != 8000 && rate != 16000 && rate != 44100 && rate != 48000)
		return 0;
	c->rate = rate;
	c->symbol_len = (1280 * rate) / 8000;                       /* decode.cc:171 */
	c->filter_len = (((21 * rate) / 8000) & ~3) | 1;            /* decode.cc:172 */
	c->guard_len = c->symbol_len / 8;                           /* decode.cc:173 */
	c->buffer_len = 6 * (c->symbol_len + c->guard_len);         /* decode.cc:188 */
	c->search_pos = c->buffer_len - 4 * (c->symbol_len + c->guard_len);   /* decode.cc:189 */
	return 1;
}

size_t orc_frame_samples(int rate, int oper_mode, int count)
{
	orc_rate_cfg rc;
	orc_mode md;
	if (!orc_rate_lookup(rate, &rc) || !orc_mode_lookup(oper_mode, &md))
		return 0;
	return 2 * (size_t)rate + (2 + (size_t)count * (3 + (size_t)md.cons_rows)) * (size_t)(rc.symbol_len + rc.guard_len);
}

/* ---- CODE::Xorshift32: Marsaglia xorshift32, default seed (decode.cc:613) */
void orc_xorshift32_init(orc_xorshift32 *s) { s->y = 2463534242u; }
uint32_t orc_xorshift32_next(orc_xorshift32 *s)
{
	uint32_t y = s->y;
	y ^= y << 13;
	y ^= y >> 17;
	y ^= y << 5;
	return s->y = y;
}
/* decode.cc:613-615 / encode.cc:417-419: byte ^= (uint8_t)scrambler() */
void orc_scramble(uint8_t *buf, int len)
{
	orc_xorshift32 s;
	orc_xorshift32_init(&s);
	for (int i = 0; i < len; ++i)
		buf[i] ^= (uint8_t)orc_xorshift32_next(&s);
}

/* ---- CODE::MLS: Galois LFSR, reg=1 start (decode.cc:238,407) */
static int hibit(unsigned n)
{
	n |= n >> 1; n |= n >> 2; n |= n >> 4; n |= n >> 8; n |= n >> 16;
	return (int)(n ^ (n >> 1));
}
void orc_mls_init(orc_mls *m, int poly)
{
	m->poly = poly;
	m->test = hibit((unsigned)poly) >> 1;
	m->reg = 1;
}
int orc_mls_next(orc_mls *m)
{
	int fb = (m->reg & m->test) != 0;
	m->reg <<= 1;
	m->reg ^= fb * m->poly;
	return fb;
}

/* ---- CODE::CRC<T>: reflected (right-shifting), init 0, no final xor ------ */
uint32_t orc_crc32_bit(uint32_t poly, uint32_t crc, int bit)
{
	uint32_t tmp = crc ^ (uint32_t)(bit & 1);
	return (crc >> 1) ^ ((tmp & 1) * poly);
}
uint32_t orc_crc32_bytes(uint32_t poly, const uint8_t *p, int n)
{
	/* operator()(uint8_t): identical to feeding the 8 bits LSB first
	 * (encode.cc:297 bytes vs decode.cc:536 bits) */
	uint32_t crc = 0;
	for (int i = 0; i < n; ++i)
		for (int b = 0; b < 8; ++b)
			crc = orc_crc32_bit(poly, crc, (p[i] >> b) & 1);
	return crc;
}
/* crc0(md << 9) with the uint64_t overload: low byte first (decode.cc:428-429) */
uint16_t orc_crc16_u64(uint16_t poly, uint64_t data)
{
	uint16_t crc = 0;
	for (int i = 0; i < 64; ++i) {
		uint16_t tmp = crc ^ (uint16_t)((data >> i) & 1);
		crc = (uint16_t)((crc >> 1) ^ ((tmp & 1) * poly));
	}
	return crc;
}

/* ---- CODE::get/set_be_bit, get/set_le_bit (decode.cc:424,427,553) -------- */
int orc_get_be_bit(const uint8_t *buf, int i) { return (buf[i / 8] >> (7 - i % 8)) & 1; }
void orc_set_be_bit(uint8_t *buf, int i, int v)
{
	uint8_t m = (uint8_t)(0x80 >> (i % 8));
	buf[i / 8] = (uint8_t)((buf[i / 8] & ~m) | (v ? m : 0));
}
int orc_get_le_bit(const uint8_t *buf, int i) { return (buf[i / 8] >> (i % 8)) & 1; }
void orc_set_le_bit(uint8_t *buf, int i, int v)
{
	uint8_t m = (uint8_t)(1 << (i % 8));
	buf[i / 8] = (uint8_t)((buf[i / 8] & ~m) | (v ? m : 0));
}

/* ---- base37: encode.cc:320-335, decode.cc:155-159 ------------------------ */
long long orc_base37_encode(const char *str)
{
	/* ' ' = 0, '0'..'9' = 1..10, 'A'..'Z' / 'a'..'z' = 11..36; anything else is rejected */
	static const char alphabet[] = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ";
	long long acc = 0;
	for (; *str; ++str) {
		int c = (*str >= 'a' && *str <= 'z') ? *str - 'a' + 'A' : *str;
		const char *hit = c ? strchr(alphabet, c) : NULL;
		if (!hit)
			return -1;
		acc = acc * 37 + (hit - alphabet);
	}
	return acc;
}
void orc_base37_decode(char *str, long long val, int len)
{
	for (int i = len - 1; i >= 0; --i, val /= 37)
		str[i] = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ"[val % 37];
}
