/*
 * oracle/main_encode.c -- TEST INFRASTRUCTURE.  CLI with the argv of the
 * reference's encode (encode.cc:337-445):
 *   encode OUTPUT RATE BITS CHANNELS OFFSET MODE CALLSIGN INPUT..
 */
#include "modem_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
	if (argc < 9) {
		fprintf(stderr, "usage: %s OUTPUT RATE BITS CHANNELS OFFSET MODE CALLSIGN INPUT..\n", argv[0]);
		return 1;
	}
	const char *output_name = argv[1];
	if (!strcmp(output_name, "-"))
		output_name = "/dev/stdout";
	int output_rate = atoi(argv[2]), output_bits = atoi(argv[3]), output_chan = atoi(argv[4]);
	int freq_off = atoi(argv[5]), oper_mode = atoi(argv[6]);
	orc_mode md;
	if (oper_mode < 6 || oper_mode > 13 || !orc_mode_lookup(oper_mode, &md)) {
		fprintf(stderr, "Unsupported operation mode.\n");
		return 1;
	}
	long long cs = orc_base37_encode(argv[7]);
	if (cs <= 0 || cs >= 129961739795077LL) {
		fprintf(stderr, "Unsupported call sign.\n");
		return 1;
	}
	/* encode.cc:389-397 */
	if ((output_chan == 1 && freq_off < md.band_width / 2) || freq_off < md.band_width / 2 - output_rate / 2
		|| freq_off > output_rate / 2 - md.band_width / 2) {
		fprintf(stderr, "Unsupported frequency offset.\n");
		return 1;
	}
	if (freq_off % 50) {
		fprintf(stderr, "Frequency offset must be divisible by 50.\n");
		return 1;
	}
	orc_rate_cfg rc;
	if (!orc_rate_lookup(output_rate, &rc)) {             /* encode.cc:424-439 */
		fprintf(stderr, "Unsupported sample rate.\n");
		return 1;
	}
	int count = argc - 8;
	uint8_t *payload = (uint8_t *)calloc((size_t)count, ORC_DATA_BYTES);
	for (int j = 0; j < count; ++j) {
		const char *name = argv[j + 8];
		if (argc == 9 && !strcmp(name, "-"))
			name = "/dev/stdin";
		FILE *f = fopen(name, "rb");
		if (!f) {
			fprintf(stderr, "Couldn't open file \"%s\" for reading.\n", name);
			return 1;
		}
		for (int i = 0; i < ORC_DATA_BYTES; ++i)
			payload[(size_t)j * ORC_DATA_BYTES + i] = (uint8_t)fgetc(f);   /* encode.cc:414 */
		fclose(f);
	}
	size_t total = orc_frame_samples(output_rate, oper_mode, count);
	void *pcm = malloc(total * (size_t)output_chan * (size_t)(output_bits / 8));
	size_t n = orc_encode_pcm_rate(output_rate, pcm, output_bits, output_chan, payload, count, freq_off, argv[7], oper_mode);
	/* write via the WAV writer from already-quantised data: re-expand to cf */
	orc_cf *z = (orc_cf *)calloc(n, sizeof(orc_cf));
	float factor = (float)((1u << (output_bits - 1)) - 1);
	for (size_t i = 0; i < n; ++i) {
		for (int c = 0; c < output_chan; ++c) {
			float v = output_bits == 8 ? (float)((int)((uint8_t *)pcm)[i * output_chan + c] - 128) / factor
				: (float)((int16_t *)pcm)[i * output_chan + c] / factor;
			if (c) z[i].im = v; else z[i].re = v;
		}
	}
	int r = orc_wav_write(output_name, output_rate, output_bits, output_chan, z, n);
	free(z);
	free(pcm);
	free(payload);
	return r ? 1 : 0;
}
