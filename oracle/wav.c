/*
 * oracle/wav.c -- TEST INFRASTRUCTURE (see modem_oracle.h header).
 * DSP::ReadWAV / DSP::WriteWAV contract (wav.hh ABSENT; call sites
 * decode.cc:576-578,590 and encode.cc:422-423,441): canonical RIFF/WAVE PCM,
 * 8-bit unsigned offset 128, 16/24/32-bit signed little endian, samples scaled
 * by 1/(2^(bits-1)-1); writer clamps to [-1,1] and rounds to nearest.
 */
#include "modem_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

int orc_wav_read(const char *name, orc_wav *w)
{
	memset(w, 0, sizeof(*w));
	FILE *f = fopen(name, "rb");
	if (!f)
		return -1;
	uint8_t *buf = NULL;
	size_t len = 0, cap = 0;
	for (;;) {
		if (len + 65536 > cap) {
			cap = cap ? cap * 2 : 1 << 20;
			buf = (uint8_t *)realloc(buf, cap);
		}
		size_t n = fread(buf + len, 1, 65536, f);
		len += n;
		if (n < 65536)
			break;
	}
	fclose(f);
	if (len < 12 || memcmp(buf, "RIFF", 4) || memcmp(buf + 8, "WAVE", 4)) {
		free(buf);
		return -2;
	}
	size_t pos = 12;
	int have_fmt = 0;
	while (pos + 8 <= len) {
		uint32_t sz = rd32(buf + pos + 4);
		const uint8_t *body = buf + pos + 8;
		if (!memcmp(buf + pos, "fmt ", 4) && sz >= 16) {
			int tag = rd16(body);
			w->channels = rd16(body + 2);
			w->rate = (int)rd32(body + 4);
			w->bits = rd16(body + 14);
			if (tag != 1 && tag != 0xfffe) {
				free(buf);
				return -3;
			}
			have_fmt = 1;
		} else if (!memcmp(buf + pos, "data", 4) && have_fmt) {
			size_t avail = len - (pos + 8);
			if (sz > avail)   /* streamed files carry 0xffffffff / short sizes */
				sz = (uint32_t)avail;
			int bytes = w->bits / 8;
			if (bytes < 1 || bytes > 4 || w->channels < 1) {
				free(buf);
				return -4;
			}
			w->frames = sz / (size_t)(bytes * w->channels);
			size_t cnt = w->frames * (size_t)w->channels;
			if (bytes == 1) {
				w->fmt = ORC_FMT_U8;
				w->data = malloc(cnt ? cnt : 1);
				memcpy(w->data, body, cnt);
			} else if (bytes == 2) {
				w->fmt = ORC_FMT_S16;
				int16_t *d = (int16_t *)malloc((cnt ? cnt : 1) * 2);
				for (size_t i = 0; i < cnt; ++i)
					d[i] = (int16_t)rd16(body + 2 * i);
				w->data = d;
			} else {
				w->fmt = ORC_FMT_F32;
				float *d = (float *)malloc((cnt ? cnt : 1) * 4);
				float factor = (float)((1u << (w->bits - 1)) - 1);
				for (size_t i = 0; i < cnt; ++i) {
					int32_t v = 0;
					for (int b = 0; b < bytes; ++b)
						v |= (int32_t)((uint32_t)body[bytes * i + b] << (8 * b + 8 * (4 - bytes)));
					v >>= 8 * (4 - bytes);
					d[i] = (float)v / factor;
				}
				w->data = d;
			}
			free(buf);
			return 0;
		}
		pos += 8 + (size_t)sz + (sz & 1);
	}
	free(buf);
	return -5;
}

void orc_wav_free(orc_wav *w)
{
	free(w->data);
	memset(w, 0, sizeof(*w));
}

/* WritePCM::write(buf, frames, stride=2): writes channels() of the two
 * interleaved values (mono => real part only), encode.cc:127-128 */
void orc_quantise(void *pcm, int bits, int channels, const orc_cf *z, size_t n)
{
	const float factor = (float)((1u << (bits - 1)) - 1);
	for (size_t i = 0; i < n; ++i) {
		for (int c = 0; c < channels; ++c) {
			float v = c ? z[i].im : z[i].re;
			v = fminf(fmaxf(v, -1.f), 1.f);
			int q = (int)nearbyintf(factor * v);
			if (bits == 8)
				((uint8_t *)pcm)[i * (size_t)channels + c] = (uint8_t)(q + 128);
			else
				((int16_t *)pcm)[i * (size_t)channels + c] = (int16_t)q;
		}
	}
}

int orc_wav_write(const char *name, int rate, int bits, int channels, const orc_cf *z, size_t frames)
{
	if (bits != 8 && bits != 16)
		return -1;
	FILE *f = fopen(name, "wb");
	if (!f)
		return -2;
	int bytes = bits / 8;
	uint32_t data_len = (uint32_t)(frames * (size_t)(bytes * channels));
	uint8_t h[44];
	memcpy(h, "RIFF", 4);
	uint32_t riff = 36 + data_len;
	memcpy(h + 4, &riff, 4);
	memcpy(h + 8, "WAVEfmt ", 8);
	uint32_t v32 = 16; memcpy(h + 16, &v32, 4);
	uint16_t v16 = 1; memcpy(h + 20, &v16, 2);
	v16 = (uint16_t)channels; memcpy(h + 22, &v16, 2);
	v32 = (uint32_t)rate; memcpy(h + 24, &v32, 4);
	v32 = (uint32_t)(rate * bytes * channels); memcpy(h + 28, &v32, 4);
	v16 = (uint16_t)(bytes * channels); memcpy(h + 32, &v16, 2);
	v16 = (uint16_t)bits; memcpy(h + 34, &v16, 2);
	memcpy(h + 36, "data", 4);
	memcpy(h + 40, &data_len, 4);
	fwrite(h, 1, 44, f);
	void *pcm = malloc(data_len ? data_len : 1);
	orc_quantise(pcm, bits, channels, z, frames);
	fwrite(pcm, 1, data_len, f);
	free(pcm);
	fclose(f);
	return 0;
}
