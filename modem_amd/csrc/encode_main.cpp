// encode_main.cpp -- `encode OUTPUT RATE BITS CHANNELS OFFSET MODE CALLSIGN INPUT..`: the reference's transmitter command
// line (encode.cc:337-443) on top of the C ABI (ofdmrx_tx_encode_stream).  Same argv, same checks and messages, same
// "-" = /dev/stdout | /dev/stdin handling, same exit codes; the WAV it writes is the reference's stream:
// RATE samples of silence, pilot | per input file: Schmidl-Cox, meta data, pilot, payload rows | zero symbol, silence.
#include "../../include/ofdmrx.h"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static void put16(uint8_t *p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }

int main(int argc, char **argv)
{
	if (argc < 9) {
		std::fprintf(stderr, "usage: %s OUTPUT RATE BITS CHANNELS OFFSET MODE CALLSIGN INPUT..\n", argv[0]);
		return 1;
	}
	const char *output_name = argv[1];
	if (output_name[0] == '-' && output_name[1] == 0)
		output_name = "/dev/stdout";
	const int output_rate = std::atoi(argv[2]), output_bits = std::atoi(argv[3]), output_chan = std::atoi(argv[4]);
	const int freq_off = std::atoi(argv[5]), oper_mode = std::atoi(argv[6]);
	if (oper_mode < 6 || oper_mode > 13) {
		std::fprintf(stderr, "Unsupported operation mode.\n");
		return 1;
	}
	const long long call_sign = ofdmrx_callsign_value(argv[7]);   // encode.cc:357
	if (call_sign <= 0 || call_sign >= 129961739795077LL) {
		std::fprintf(stderr, "Unsupported call sign.\n");
		return 1;
	}
	static const int bw[14] = { 0, 0, 0, 0, 0, 0, 2700, 2500, 2500, 2250, 3200, 2400, 2400, 1600 };   // encode.cc:363-387
	const int band_width = bw[oper_mode];
	if ((output_chan == 1 && freq_off < band_width / 2) || freq_off < band_width / 2 - output_rate / 2 ||
		freq_off > output_rate / 2 - band_width / 2) {
		std::fprintf(stderr, "Unsupported frequency offset.\n");
		return 1;
	}
	if (freq_off % 50) {
		std::fprintf(stderr, "Frequency offset must be divisible by 50.\n");
		return 1;
	}
	const int input_count = argc - 8;
	std::vector<uint8_t> input_data((size_t)OFDMRX_PAYLOAD_BYTES * input_count);
	for (int j = 0; j < input_count; ++j) {
		const char *input_name = argv[j + 8];
		if (argc == 9 && input_name[0] == '-' && input_name[1] == 0)
			input_name = "/dev/stdin";
		FILE *f = std::fopen(input_name, "rb");
		if (!f) {
			std::fprintf(stderr, "Couldn't open file \"%s\" for reading.\n", input_name);
			return 1;
		}
		for (int i = 0; i < OFDMRX_PAYLOAD_BYTES; ++i)        // encode.cc:414: get() past the end yields 0xff
			input_data[(size_t)j * OFDMRX_PAYLOAD_BYTES + i] = (uint8_t)std::fgetc(f);
		std::fclose(f);
	}
	if ((output_bits != 8 && output_bits != 16) || output_chan < 1 || output_chan > 2) {
		std::fprintf(stderr, "Unsupported sample format.\n");
		return 1;
	}
	ofdmrx_config cfg{};
	cfg.abi_version = OFDMRX_ABI_VERSION;
	cfg.sample_rate = output_rate;
	cfg.list_size = 8;
	cfg.chunk_frames = 1;
	cfg.descramble = 1;
	ofdmrx_handle *h = nullptr;
	int r = ofdmrx_create(&cfg, &h);
	if (r == OFDMRX_E_UNSUPPORTED) {
		std::fprintf(stderr, "Unsupported sample rate.\n");   // encode.cc:437-439
		return 1;
	}
	if (r) {
		std::fprintf(stderr, "ofdmrx_create: %s\n", ofdmrx_strerror(r));
		return 1;
	}
	const long frames = ofdmrx_stream_samples(output_rate, oper_mode, input_count);
	const size_t bytes = (size_t)frames * (size_t)output_chan * (size_t)(output_bits / 8);
	std::vector<uint8_t> pcm(bytes);
	r = ofdmrx_tx_encode_stream(h, input_data.data(), input_count, oper_mode, freq_off, argv[7], output_chan, output_bits, pcm.data());
	ofdmrx_destroy(h);
	if (r) {
		std::fprintf(stderr, "ofdmrx_tx_encode_stream: %s\n", ofdmrx_strerror(r));
		return 1;
	}
	FILE *o = std::fopen(output_name, "wb");
	if (!o) {
		std::fprintf(stderr, "Couldn't open file \"%s\" for writing.\n", output_name);
		return 1;
	}
	uint8_t hd[44];
	const int fb = output_chan * (output_bits / 8);
	std::memcpy(hd, "RIFF", 4); put32(hd + 4, (uint32_t)(36 + bytes)); std::memcpy(hd + 8, "WAVEfmt ", 8);
	put32(hd + 16, 16); put16(hd + 20, 1); put16(hd + 22, (uint16_t)output_chan); put32(hd + 24, (uint32_t)output_rate);
	put32(hd + 28, (uint32_t)(output_rate * fb)); put16(hd + 32, (uint16_t)fb); put16(hd + 34, (uint16_t)output_bits);
	std::memcpy(hd + 36, "data", 4); put32(hd + 40, (uint32_t)bytes);
	std::fwrite(hd, 1, 44, o);
	std::fwrite(pcm.data(), 1, bytes, o);
	std::fclose(o);
	return 0;
}
