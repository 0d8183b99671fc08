/*
 * oracle/main_decode.c -- TEST INFRASTRUCTURE.  CLI with the argv of the
 * reference's decode (decode.cc:559-620):  decode OUTPUT INPUT [SKIP]
 */
#include "modem_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

int main(int argc, char **argv)
{
	if (argc < 3 || argc > 4) {
		fprintf(stderr, "usage: %s OUTPUT INPUT [SKIP]\n", argv[0]);
		return 1;
	}
	const char *output_name = argv[1];
	if (!strcmp(output_name, "-"))
		output_name = "/dev/stdout";
	const char *input_name = argv[2];
	if (!strcmp(input_name, "-"))
		input_name = "/dev/stdin";
	orc_wav w;
	if (orc_wav_read(input_name, &w)) {
		fprintf(stderr, "Couldn't open file \"%s\" for reading.\n", input_name);
		return 1;
	}
	if (w.channels < 1 || w.channels > 2) {
		fprintf(stderr, "Only real or analytic signal (one or two channels) supported.\n");
		return 1;
	}
	int skip_count = argc > 3 ? atoi(argv[3]) : 0;
	orc_rate_cfg rc;
	if (!orc_rate_lookup(w.rate, &rc)) {                  /* decode.cc:590-605 */
		fprintf(stderr, "Unsupported sample rate.\n");
		return 1;
	}
	uint8_t out[ORC_DATA_BYTES];
	orc_result r;
	orc_decode_rate(w.rate, w.data, w.fmt, w.channels, w.frames, skip_count, 8, 1, out, &r, NULL);
	static const char *msg[] = { "", "", "OSD error.", "header CRC error.", "operation mode unsupported.",
		"call sign unsupported.", "payload decoding error." };
	if (r.sc_start >= 0) {
		fprintf(stderr, "symbol pos: %d\n", r.symbol_pos);
		fprintf(stderr, "coarse cfo: %g Hz \n", r.cfo_rad * ((float)w.rate / 6.28318530717958647692f));
	}
	if (r.status >= ORC_OSD_ERROR && r.status <= ORC_PAYLOAD_CRC)
		fprintf(stderr, "%s\n", msg[r.status]);
	if (r.status == ORC_OK || r.status == ORC_PAYLOAD_CRC) {
		char cs[10];
		orc_base37_decode(cs, (long long)r.call_sign, 9);
		cs[9] = 0;
		fprintf(stderr, "oper mode: %d\ncall sign: %s\n", r.oper_mode, cs);
		fprintf(stderr, "finer cfo: %g Hz \n", r.cfo_fine * ((float)w.rate / 6.28318530717958647692f));
		fprintf(stderr, "Es/N0 (dB): ... %g\n", r.esn0_db_last);
	}
	if (r.status == ORC_OK)
		fprintf(stderr, "bit flips: %d\n", r.bit_flips);
	FILE *f = fopen(output_name, "wb");
	if (!f) {
		fprintf(stderr, "Couldn't open file \"%s\" for writing.\n", output_name);
		return 1;
	}
	fwrite(out, 1, ORC_DATA_BYTES, f);
	fclose(f);
	orc_wav_free(&w);
	return 0;
}
