// decode_main.cpp -- `decode OUTPUT INPUT [SKIP]`, the reference's CLI (decode.cc:559-620)
// on top of the C ABI: read the WAV body into memory, decode it as a batch of one frame on
// the GPU, descramble, write 5380 bytes.  Same argv, same "-" handling, same exit codes
// (0 even when decoding fails, decode.cc:542-545,619); on failure the output is zeros
// (the reference writes an uninitialised buffer, decode.cc:588).
#include "../../include/ofdmrx.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

struct Wav { int rate = 0, bits = 0, channels = 0, fmt = -1; size_t frames = 0; std::vector<uint8_t> pcm; };

// DSP::ReadWAV contract (decode.cc:576-578,590): RIFF/WAVE PCM, 8-bit unsigned, 16/24/32-bit signed LE
static bool read_wav(const char *name, Wav &w)
{
	FILE *f = std::fopen(name, "rb");
	if (!f)
		return false;
	std::vector<uint8_t> buf;
	uint8_t tmp[65536];
	size_t n;
	while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0)
		buf.insert(buf.end(), tmp, tmp + n);
	std::fclose(f);
	if (buf.size() < 12 || std::memcmp(buf.data(), "RIFF", 4) || std::memcmp(buf.data() + 8, "WAVE", 4))
		return false;
	size_t pos = 12;
	bool have_fmt = false;
	while (pos + 8 <= buf.size()) {
		uint32_t sz = rd32(&buf[pos + 4]);
		const uint8_t *body = &buf[pos + 8];
		if (!std::memcmp(&buf[pos], "fmt ", 4) && sz >= 16) {
			w.channels = rd16(body + 2);
			w.rate = (int)rd32(body + 4);
			w.bits = rd16(body + 14);
			have_fmt = true;
		} else if (!std::memcmp(&buf[pos], "data", 4) && have_fmt) {
			size_t avail = buf.size() - (pos + 8);
			if (sz > avail)
				sz = (uint32_t)avail;
			int bytes = w.bits / 8;
			if (bytes < 1 || bytes > 4 || w.channels < 1)
				return false;
			w.frames = sz / (size_t)(bytes * w.channels);
			size_t cnt = w.frames * (size_t)w.channels;
			if (bytes == 1) {
				w.fmt = OFDMRX_FMT_U8;
				w.pcm.assign(body, body + cnt);
			} else if (bytes == 2) {
				w.fmt = OFDMRX_FMT_S16;
				w.pcm.assign(body, body + 2 * cnt);   // little endian host
			} else {
				w.fmt = OFDMRX_FMT_F32;
				w.pcm.resize(4 * cnt);
				float *d = (float *)w.pcm.data();
				float factor = (float)((1u << (w.bits - 1)) - 1);
				for (size_t i = 0; i < cnt; ++i) {
					int32_t v = 0;
					for (int b = 0; b < bytes; ++b)
						v |= (int32_t)((uint32_t)body[bytes * i + b] << (8 * b + 8 * (4 - bytes)));
					v >>= 8 * (4 - bytes);
					d[i] = (float)v / factor;
				}
			}
			return true;
		}
		pos += 8 + (size_t)sz + (sz & 1);
	}
	return false;
}

int main(int argc, char **argv)
{
	if (argc < 3 || argc > 4) {
		std::fprintf(stderr, "usage: %s OUTPUT INPUT [SKIP]\n", argv[0]);
		return 1;
	}
	const char *output_name = argv[1];
	if (!std::strcmp(output_name, "-"))
		output_name = "/dev/stdout";
	const char *input_name = argv[2];
	if (!std::strcmp(input_name, "-"))
		input_name = "/dev/stdin";
	Wav w;
	if (!read_wav(input_name, w)) {
		std::fprintf(stderr, "Couldn't open file \"%s\" for reading.\n", input_name);
		return 1;
	}
	if (w.channels < 1 || w.channels > 2) {
		std::fprintf(stderr, "Only real or analytic signal (one or two channels) supported.\n");
		return 1;
	}
	int32_t skip_count = argc > 3 ? std::atoi(argv[3]) : 0;
	if (w.rate != 8000 && w.rate != 16000 && w.rate != 44100 && w.rate != 48000) {   // decode.cc:590-605
		std::fprintf(stderr, "Unsupported sample rate.\n");
		return 1;
	}
	ofdmrx_config cfg{};
	cfg.abi_version = OFDMRX_ABI_VERSION;
	cfg.sample_rate = w.rate;
	cfg.list_size = 8;
	cfg.device = 0;
	cfg.chunk_frames = 1;
	cfg.max_samples = (int32_t)w.frames;
	cfg.descramble = 1;
	ofdmrx_handle *h = nullptr;
	int r = ofdmrx_create(&cfg, &h);
	if (r) {
		std::fprintf(stderr, "ofdmrx_create: %s\n", ofdmrx_strerror(r));
		return 1;
	}
	std::vector<uint8_t> out(OFDMRX_PAYLOAD_BYTES, 0);
	ofdmrx_frame_result res{};
	size_t bps = w.fmt == OFDMRX_FMT_S16 ? 2 : w.fmt == OFDMRX_FMT_U8 ? 1 : 4;
	size_t stride = (w.frames * bps * (size_t)w.channels + 3) & ~(size_t)3;
	w.pcm.resize(stride);
	std::vector<ofdmrx_attempt> attempts(OFDMRX_MAX_SKIP + 1);
	int32_t n_attempts = 0;
	ofdmrx_set_attempt_log(h, attempts.data(), &n_attempts);
	r = ofdmrx_decode_batch(h, w.pcm.data(), w.fmt, w.channels, w.frames, stride, 1, &skip_count, out.data(), &res);
	if (r) {
		std::fprintf(stderr, "ofdmrx_decode_batch: %s\n", ofdmrx_strerror(r));
		return 1;
	}
	const float hz = (float)w.rate / 6.28318530717958647692f;   // decode.cc:401,503
	auto call_sign_text = [](unsigned long long v, char *cs) {   // base37_decoder, decode.cc:155-159
		for (int i = 8; i >= 0; --i, v /= 37)
			cs[i] = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ"[v % 37];
		cs[9] = 0;
	};
	// decode.cc:390-448: one block of lines per preamble the SKIP loop examined, the last one included
	for (int a = 0; a < n_attempts; ++a) {
		const ofdmrx_attempt &at = attempts[a];
		std::fprintf(stderr, "symbol pos: %d\n", at.symbol_pos);
		std::fprintf(stderr, "coarse cfo: %g Hz \n", at.cfo_rad * hz);
		if (at.status == OFDMRX_OSD_ERROR) {
			std::fprintf(stderr, "OSD error.\n");
		} else if (at.status == OFDMRX_HEADER_CRC) {
			std::fprintf(stderr, "header CRC error.\n");
		} else if (at.status == OFDMRX_BAD_MODE) {
			std::fprintf(stderr, "operation mode %d unsupported.\n", at.oper_mode);   // decode.cc:435
		} else {
			std::fprintf(stderr, "oper mode: %d\n", at.oper_mode);
			if (at.status == OFDMRX_BAD_CALLSIGN) {
				std::fprintf(stderr, "call sign unsupported.\n");
			} else {
				char cs[10];
				call_sign_text(at.call_sign, cs);
				std::fprintf(stderr, "call sign: %s\n", cs);
			}
		}
	}
	if (res.status == OFDMRX_OK || res.status == OFDMRX_PAYLOAD_CRC) {
		// rows of the mode (decode.cc:302-374, 453): cons_bits / mod_bits / cols
		static const int cols[14] = { 0, 0, 0, 0, 0, 0, 432, 400, 400, 360, 512, 384, 384, 256 };
		static const int bits[14] = { 0, 0, 0, 0, 0, 0, 3, 3, 2, 2, 3, 3, 2, 2 };
		const int rows = ((res.oper_mode >= 10 ? 64512 : 64800) / bits[res.oper_mode]) / cols[res.oper_mode];
		std::fprintf(stderr, "demod ");                     // decode.cc:463,476-478: one dot per payload symbol
		for (int j = 0; j < rows; ++j)
			std::fprintf(stderr, ".");
		std::fprintf(stderr, " done\n");
		// decode.cc:500,502: sfo_rad -= avg_slope * symbol_len / (symbol_len + guard_len).  The reference never initialises
		// sfo_rad; this prints the value for a start from zero
		const float sfo_rad = -res.sfo_slope * 8.f / 9.f;
		std::fprintf(stderr, "coarse sfo: %g ppm\n", 1000000.f * sfo_rad / 6.28318530717958647692f);
		std::fprintf(stderr, "finer cfo: %g Hz \n", res.cfo_fine * hz);
		// decode.cc:506-523: the cumulative Es/N0 after every row = 10 log10 of the precision the soft demapper used
		std::vector<float> prec((size_t)rows, 0.f);
		std::fprintf(stderr, "Es/N0 (dB):");
		if (ofdmrx_debug_dump(h, OFDMRX_TAP_PRECISION, 0, prec.data(), prec.size() * sizeof(float)) == 0)
			for (int j = 0; j < rows; ++j)
				std::fprintf(stderr, " %g", 10.f * std::log10(prec[j]));
		else
			std::fprintf(stderr, " %g", res.esn0_db_last);
		std::fprintf(stderr, "\n");
	}
	if (res.status == OFDMRX_PAYLOAD_CRC)
		std::fprintf(stderr, "payload decoding error.\n");         // decode.cc:543
	if (res.status == OFDMRX_OK)
		std::fprintf(stderr, "bit flips: %d\n", res.bit_flips);
	ofdmrx_destroy(h);
	FILE *f = std::fopen(output_name, "wb");
	if (!f) {
		std::fprintf(stderr, "Couldn't open file \"%s\" for writing.\n", output_name);
		return 1;
	}
	std::fwrite(out.data(), 1, out.size(), f);
	std::fclose(f);
	return 0;
}
