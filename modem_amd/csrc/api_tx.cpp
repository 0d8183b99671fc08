// api_tx.cpp -- channel models (N3) and the device transmitter (N2) behind include/ofdmrx.h (encode.cc:205-291, 399-441).
#include "api_internal.h"

extern "C" int ofdmrx_util_awgn_tile(ofdmrx_handle *h, const int16_t *d_base, size_t n_base, int16_t *d_out, size_t n_out,
	size_t spf, float noise_db, uint64_t seed, uint64_t first_frame)
{
	if (!h || !d_base || !d_out || !n_base || !n_out || !spf)
		return OFDMRX_E_ARG;
	HIP_OK(hipSetDevice(h->cfg.device));
	const float sigma = std::sqrt(0.5f * std::pow(10.f, noise_db / 10.f));
	launch_awgn_tile(h->stream, d_base, n_base, d_out, n_out, spf, sigma, seed, first_frame);
	HIP_OK(hipGetLastError());
	return 0;
}

extern "C" int ofdmrx_util_channel(ofdmrx_handle *h, const int16_t *d_in, int16_t *d_out, size_t n_frames, size_t spf,
	const ofdmrx_channel *ch)
{
	if (!h || !d_in || !d_out || !n_frames || n_frames > 65535 || !spf || !ch || ch->ntaps < 0 || ch->ntaps > 8)
		return OFDMRX_E_ARG;
	for (int i = 0; i < ch->ntaps; ++i)
		if (ch->delays[i] < 0 || (size_t)ch->delays[i] >= spf)
			return OFDMRX_E_ARG;
	{
		const char *a = (const char *)d_in, *b = (const char *)d_out;
		const size_t bytes = n_frames * spf * 2 * sizeof(int16_t);
		if (a < b + bytes && b < a + bytes)                   // the resampler reads neighbours of what other blocks write
			return OFDMRX_E_ARG;
	}
	HIP_OK(hipSetDevice(h->cfg.device));
	struct { float cfo_hz, sfo_ppm; int ntaps; int delays[8]; float gre[8], gim[8]; } cp;
	cp.cfo_hz = ch->cfo_hz;
	cp.sfo_ppm = ch->sfo_ppm;
	cp.ntaps = ch->ntaps;
	for (int i = 0; i < 8; ++i) { cp.delays[i] = ch->delays[i]; cp.gre[i] = ch->gains_re[i]; cp.gim[i] = ch->gains_im[i]; }
	if (cp.ntaps == 0) { cp.ntaps = 1; cp.delays[0] = 0; cp.gre[0] = 1.f; cp.gim[0] = 0.f; }
	launch_channel(h->stream, h->rate, d_in, d_out, n_frames, spf, &cp);
	HIP_OK(hipGetLastError());
	return 0;
}

// ---- N2: transmitter on the device (Encoder<value,cmplx,rate>, encode.cc:271-317) -------------------
extern "C" long long ofdmrx_callsign_value(const char *call_sign) { return call_sign ? callsign_value(call_sign) : -1; }

extern "C" long ofdmrx_stream_samples(int sample_rate, int oper_mode, int count)
{
	if (oper_mode < 6 || oper_mode > 13 || !rate_supported(sample_rate) || count < 1 || count > 4096)
		return OFDMRX_E_ARG;
	ModeDesc md = mode_desc(oper_mode);
	const long stride = rate_symbol_len(sample_rate) + rate_symbol_len(sample_rate) / 8;
	// silence | pilot | count x (S&C, meta, pilot, rows) | zero symbol | silence  (encode.cc:288-313,423,441)
	return 2L * sample_rate + (2 + (long)count * (3 + md.rows)) * stride;
}

extern "C" long ofdmrx_frame_samples(int sample_rate, int oper_mode) { return ofdmrx_stream_samples(sample_rate, oper_mode, 1); }

extern "C" long ofdmrx_tx_frame_samples(int oper_mode) { return ofdmrx_frame_samples(8000, oper_mode); }

extern "C" int ofdmrx_tx_encode_stream_device(ofdmrx_handle *h, const uint8_t *d_payload, size_t n_streams, int count,
	int oper_mode, int freq_off, const char *call_sign, int channels, int bits, void *d_pcm)
{
	if (!h || !d_payload || !d_pcm || !n_streams || !call_sign || channels < 1 || channels > 2 || (bits != 8 && bits != 16))
		return OFDMRX_E_ARG;
	if (oper_mode < 6 || oper_mode > 13 || freq_off % 50 || count < 1 || count > 4096)   // encode.cc:353,394
		return OFDMRX_E_ARG;
	long long cs = callsign_value(call_sign);
	if (cs <= 0 || cs >= 129961739795077LL)               // encode.cc:358
		return OFDMRX_E_ARG;
	HIP_OK(hipSetDevice(h->cfg.device));
	struct { int oper_mode, offset, channels, nsym; unsigned long long md; long frame_samples; int count, bits, symbol_len; } tp;
	ModeDesc md = mode_desc(oper_mode);
	const int SL = rate_symbol_len(h->rate);
	tp.oper_mode = oper_mode;
	tp.offset = (freq_off * SL) / h->rate;                // encode.cc:283
	tp.channels = channels;
	tp.nsym = 2 + count * (3 + md.rows);
	tp.md = ((unsigned long long)cs << 8) | (unsigned)oper_mode;
	tp.frame_samples = ofdmrx_stream_samples(h->rate, oper_mode, count);
	tp.count = count;
	tp.bits = bits;
	tp.symbol_len = SL;
	// streams per launch: bounded scratch (44.1 / 48 kHz keep the 4x PAPR buffers in global scratch)
	const size_t budget = h->rate <= 16000 ? 1024 : 128;
	const size_t chunk = std::max<size_t>(1, budget / (size_t)count);
	// scratch lives in the handle and grows on demand: the call only enqueues kernels on the handle's stream
	DevBuf &code = h->tx_code, &rowsym = h->tx_rowsym, &tdom = h->tx_tdom, &big = h->tx_big;
	const size_t nc = std::min(chunk, n_streams);
	const bool grow = code.bytes < nc * (size_t)count * 2048 * sizeof(uint32_t) || rowsym.bytes < nc * (size_t)count * CONS_MAX * sizeof(cf)
		|| tdom.bytes < nc * (size_t)tp.nsym * SL * sizeof(cf) || big.bytes < tx_big_scratch_bytes(h->rate, (int)nc, tp.nsym);
	if (grow)
		HIP_OK(hipStreamSynchronize(h->stream));              // a buffer about to be replaced may still be read by an earlier call
	int r = code.ensure(nc * (size_t)count * 2048 * sizeof(uint32_t));
	r = r ? r : rowsym.ensure(nc * (size_t)count * CONS_MAX * sizeof(cf));
	r = r ? r : tdom.ensure(nc * (size_t)tp.nsym * SL * sizeof(cf));
	if (tx_big_scratch_bytes(h->rate, (int)nc, tp.nsym))
		r = r ? r : big.ensure(tx_big_scratch_bytes(h->rate, (int)nc, tp.nsym));
	if (r)
		return r;
	const size_t out_stride = (size_t)tp.frame_samples * channels * (bits / 8);
	for (size_t f0 = 0; f0 < n_streams; f0 += chunk) {
		int n = (int)std::min(chunk, n_streams - f0);
		launch_tx(h->stream, h->rate, n, d_payload + f0 * (size_t)count * PAYLOAD_BYTES, h->dev, &tp, h->dev.tw_sym4,
			code.as<uint32_t>(), rowsym.as<cf>(), tdom.as<cf>(), big.as<cf>(), (char *)d_pcm + f0 * out_stride);
	}
	HIP_OK(hipGetLastError());
	return 0;
}

extern "C" int ofdmrx_tx_encode_device(ofdmrx_handle *h, const uint8_t *d_payload, size_t n_frames, int oper_mode,
	int freq_off, const char *call_sign, int channels, int16_t *d_pcm)
{
	return ofdmrx_tx_encode_stream_device(h, d_payload, n_frames, 1, oper_mode, freq_off, call_sign, channels, 16, d_pcm);
}

// host-pointer convenience for the `encode` CLI: payloads up, one stream down
extern "C" int ofdmrx_tx_encode_stream(ofdmrx_handle *h, const uint8_t *payload, int count, int oper_mode, int freq_off,
	const char *call_sign, int channels, int bits, void *pcm)
{
	if (!h || !payload || !pcm || count < 1)
		return OFDMRX_E_ARG;
	const long spf = ofdmrx_stream_samples(h->rate, oper_mode, count);
	if (spf < 0 || channels < 1 || channels > 2 || (bits != 8 && bits != 16))
		return OFDMRX_E_ARG;
	HIP_OK(hipSetDevice(h->cfg.device));
	DevBuf dp, dx;
	const size_t out_bytes = (size_t)spf * channels * (bits / 8);
	int r = dp.ensure((size_t)count * PAYLOAD_BYTES);
	r = r ? r : dx.ensure(out_bytes);
	if (!r) {
		hipError_t e = hipMemcpy(dp.p, payload, (size_t)count * PAYLOAD_BYTES, hipMemcpyHostToDevice);
		if (e != hipSuccess) { g_last_error = hipGetErrorString(e); r = OFDMRX_E_HIP; }
	}
	r = r ? r : ofdmrx_tx_encode_stream_device(h, dp.as<uint8_t>(), 1, count, oper_mode, freq_off, call_sign, channels, bits, dx.p);
	if (!r) {
		hipError_t e = hipStreamSynchronize(h->stream);       // the device entry only enqueues
		e = e == hipSuccess ? hipMemcpy(pcm, dx.p, out_bytes, hipMemcpyDeviceToHost) : e;
		if (e != hipSuccess) { g_last_error = hipGetErrorString(e); r = OFDMRX_E_HIP; }
	}
	dp.release();
	dx.release();
	return r;
}
