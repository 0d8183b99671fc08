// api_create.cpp -- host side of libofdmrx.so: the C ABI of include/ofdmrx.h (with api_pipeline.cpp, api_debug.cpp, api_tx.cpp).
// Replaces the in-process seam Decoder<value,cmplx,rate>(out, pcm, skip) (decode.cc:375)
// for batches of independent frames.  One handle = one GPU + one stream; frames are
// processed in resident chunks (device state for chunk_frames frames is allocated once and
// reused).  There is NO CPU fallback: every stage is a HIP kernel, errors are returned.
// This file: the handle's life, its device state, counters and timing.
#include "api_internal.h"

thread_local std::string g_last_error;

extern "C" int ofdmrx_abi_version(void) { return OFDMRX_ABI_VERSION; }
extern "C" int ofdmrx_abi_minor(void) { return OFDMRX_ABI_MINOR; }

extern "C" const char *ofdmrx_strerror(int err)
{
	switch (err) {
	case 0: return "ok";
	case OFDMRX_E_ARG: return "invalid argument (a null pointer, a count out of range, or samples / frame stride that are not multiples of one sample frame: bytes per sample x channels)";
	case OFDMRX_E_NOMEM: return g_last_error.empty() ? "out of device memory" : g_last_error.c_str();
	case OFDMRX_E_HIP: return g_last_error.empty() ? "HIP error" : g_last_error.c_str();
	case OFDMRX_E_NODEV: return "no usable HIP device (the receive path has no CPU fallback)";
	case OFDMRX_E_UNSUPPORTED: return "unsupported configuration (sample rate 8000/16000/44100/48000, list size 4 or 8)";
	default: return "unknown error";
	}
}

extern "C" int ofdmrx_create(const ofdmrx_config *cfg, ofdmrx_handle **out)
{
	if (!cfg || !out || cfg->abi_version != OFDMRX_ABI_VERSION)
		return OFDMRX_E_ARG;
	if (!rate_supported(cfg->sample_rate) || (cfg->list_size != 0 && cfg->list_size != 8 && cfg->list_size != 4))   // decode.cc:590-605,164-169
		return OFDMRX_E_UNSUPPORTED;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
		return OFDMRX_E_NODEV;
	HIP_OK(hipSetDevice(cfg->device));
	ofdmrx_handle *h = new (std::nothrow) ofdmrx_handle;
	if (!h)
		return OFDMRX_E_NOMEM;
	h->cfg = *cfg;
	h->rate = cfg->sample_rate;
	h->list = cfg->list_size == 4 ? 4 : 8;
	h->cert_mode = !(cfg->flags & (OFDMRX_FLAG_KEEP_RAW_CONS | OFDMRX_FLAG_SCL_ALWAYS)) && !std::getenv("OFDMRX_NO_CERT");   // (the rule holds for any list size)
	h->sc_mode = !(cfg->flags & (OFDMRX_FLAG_KEEP_RAW_CONS | OFDMRX_FLAG_SCL_ALWAYS | OFDMRX_FLAG_NO_SC)) && !std::getenv("OFDMRX_NO_SC");   // (so does this one)
	// default chunk: 8192 frames at every rate: the per-frame decoder state does not grow with the rate, and the two-stream
	// schedule wants a few thousand codewords per polar launch (44.1 / 48 kHz: 121 k / 125 k frames/s against 111 k / 112 k
	// with 4096).  What does grow is the per-chunk input: a 48 kHz frame is 4.2 MB of int16 pairs (34.6 GB per 8192 frames; the
	// host-pointer entry stages two such chunks, mono input adds a 69 GB analytic-signal buffer) - buffers are sized to
	// min(batch, chunk), so only a large batch pays that; cfg.chunk_frames lowers it.
	h->chunk = cfg->chunk_frames > 0 ? cfg->chunk_frames : 8192;
	h->max_samples = cfg->max_samples > 0 ? cfg->max_samples : ofdmrx_frame_samples(cfg->sample_rate, 6);
	if (cfg->chunk_frames <= 0) {
		// the DEFAULT chunk also has to fit what is free on the device right now (a second handle, a smaller part): per frame
		// about 3.2 MB of decoder state (both parities), the resident input (two staged chunks for the host entry) and, for
		// mono input, the analytic copy.  Halved until it fits 60 % of the free memory; an explicit cfg.chunk_frames is taken as is.
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b) {
			const double per_frame = 3.2e6 + (double)h->max_samples * (2.0 * 4.0 + 8.0);
			const double fixed = 4096.0 * 8.0 * CODE_LEN * 4.0 + 1.0e9;      // level stores of the resident list decoders + tables, staging
			while (h->chunk > 1024 && fixed + per_frame * h->chunk > 0.6 * (double)free_b)
				h->chunk /= 2;
		}
	}
	if (cfg->stream) {
		h->stream = (hipStream_t)cfg->stream;
	} else {
		hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
		if (e != hipSuccess) {
			g_last_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
			delete h;
			return OFDMRX_E_HIP;
		}
		h->own_stream = true;
	}
	// (stream_c, the host-pointer entry's copy stream, is made by its first call: HIP maps streams onto a few hardware queues, and
	// every stream that exists takes part in that)
	for (hipStream_t *sx : { &h->stream_b, &h->stream_fin }) {
		hipError_t e = hipStreamCreateWithFlags(sx, hipStreamNonBlocking);
		if (e != hipSuccess) {
			g_last_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
			ofdmrx_destroy(h);
			return OFDMRX_E_HIP;
		}
	}
	{
		int cus = 0;
		(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device);
		int wpc = 16;             // resident list decoders per CU: one wave each, 96 VGPRs + 8 KB of LDS (k_polar.hip)
		if (const char *e2 = std::getenv("OFDMRX_POLAR_WPC"))
			wpc = std::max(1, std::atoi(e2));
		h->polar_grid = wpc * std::max(cus, 1);
		int swpc = 8;             // resident SC decoders per CU: one wave each, two per SIMD (239 VGPRs) + 16 KB of LDS (k_sc.hip)
		if (const char *e3 = std::getenv("OFDMRX_SC_WPC"))
			swpc = std::max(1, std::atoi(e3));
		h->sc_grid = swpc * std::max(cus, 1);
		h->sc_grid6 = swpc * std::max(cus, 1);                   // one codeword per wave: the same eight per CU (256 VGPRs; k_sc.hip SC6_WAVES)
		if (const char *e6 = std::getenv("OFDMRX_SC_TOP"))
			h->sc_top = std::atoi(e6) != 0;
		if (const char *e4 = std::getenv("OFDMRX_SC_LB"))
			h->sc_lb = std::atoi(e4) == 5 ? 5 : (std::atoi(e4) == 0 ? 0 : 6);
		h->lanes = (cfg->flags & OFDMRX_FLAG_TWO_LANES) ? 2 : 1;
		if (const char *e5 = std::getenv("OFDMRX_LANES"))
			h->lanes = std::atoi(e5) == 2 ? 2 : 1;
		// handles with debug taps have one pipeline; the list decoder for every frame runs at the memory system's rate whatever is
		// beside it
		if (cfg->flags & (OFDMRX_FLAG_KEEP_RAW_CONS | OFDMRX_FLAG_SCL_ALWAYS))
			h->lanes = 1;
	}
	build_tables(h->host, h->rate);
	int r = 0;
	r = r ? r : upload(h, h->host.tw_sym, &h->dev.tw_sym);
	r = r ? r : upload(h, h->host.sc_kern, &h->dev.sc_kern);
	r = r ? r : upload(h, h->host.mls1_nrz, &h->dev.mls1_nrz);
	r = r ? r : upload(h, h->host.mls0_nrz, &h->dev.mls0_nrz);
	r = r ? r : upload(h, h->host.mls2_nrz, &h->dev.mls2_nrz);
	r = r ? r : upload(h, h->host.tw_sym4, &h->dev.tw_sym4);
	r = r ? r : upload(h, h->host.tw_symc, &h->dev.tw_symc);
	r = r ? r : upload(h, h->host.frozen, &h->dev.frozen);
	r = r ? r : upload(h, h->host.info_pos, &h->dev.info_pos);
	r = r ? r : upload(h, h->host.info_compress, &h->dev.info_compress);
	r = r ? r : upload(h, h->host.node_lev, &h->dev.node_lev);
	r = r ? r : upload(h, h->host.node_lev64, &h->dev.node_lev64);
	r = r ? r : upload(h, h->host.node_lev32, &h->dev.node_lev32);
	r = r ? r : upload(h, h->host.frozen_t, &h->dev.frozen_t);
	r = r ? r : upload(h, h->host.genmat_bits, &h->dev.genmat_bits);
	r = r ? r : upload(h, h->host.osd_pairs, &h->dev.osd_pairs);
	r = r ? r : upload(h, h->host.osd_triples, &h->dev.osd_triples);
	r = r ? r : upload(h, h->host.crc32_tab, &h->dev.crc32_tab);
	r = r ? r : upload(h, h->host.crc32_shift168, &h->dev.crc32_shift168);
	r = r ? r : upload(h, h->host.crc32_adv, &h->dev.crc32_adv);
	r = r ? r : upload(h, h->host.scramble, &h->dev.scramble);
	if (r) {
		ofdmrx_destroy(h);
		return r;
	}
	*out = h;
	return 0;
}

extern "C" void ofdmrx_destroy(ofdmrx_handle *h)
{
	if (!h)
		return;
	(void)hipSetDevice(h->cfg.device);
	if (h->lane2)
		ofdmrx_destroy(h->lane2);
	h->lane2 = nullptr;
	for (hipEvent_t e : { h->ev_lane_in, h->ev_lane_done })
		if (e)
			(void)hipEventDestroy(e);

	if (h->stream)
		(void)hipStreamSynchronize(h->stream);
	for (hipStream_t sx : { h->stream_b, h->stream_fin, h->stream_c })
		if (sx) {
			(void)hipStreamSynchronize(sx);
			(void)hipStreamDestroy(sx);
		}
	for (DevBuf *b : { &h->st, &h->hdr_soft, &h->cons, &h->slope, &h->yint, &h->precision, &h->slot_of, &h->res, &h->payload, &h->payload2, &h->res2,
			&h->chunk_flags, &h->soft, &h->s_ctl, &h->s_slots, &h->s_llr, &h->s_cw, &h->s_xw, &h->s_stat, &h->sc_soft, &h->q_ctl, &h->q_slots, &h->q_llr, &h->q_hard, &h->q_metric, &h->q_lane_mesg, &h->rot_tap, &h->tx_code, &h->tx_rowsym,
			&h->tx_tdom, &h->tx_big, &h->esn0_dev, &h->esn0_dev2, &h->att_dev, &h->att_dev2, &h->attc_dev, &h->attc_dev2, &h->dc, &h->z, &h->in_stage, &h->in_stage2, &h->skip_stage, &h->carr, &h->sc_scratch })
		b->release();
	for (void *p : h->table_allocs)
		(void)hipFree(p);
	for (void *p : h->out_stage)
		if (p)
			(void)hipHostFree(p);
	for (hipEvent_t e : h->ev_pool)
		(void)hipEventDestroy(e);
	if (h->own_stream && h->stream)
		(void)hipStreamDestroy(h->stream);
	delete h;
}

extern "C" int ofdmrx_chunk_frames(ofdmrx_handle *h) { return h ? h->chunk : OFDMRX_E_ARG; }
// index (in the last decode call) of the first frame of the LAST chunk that call ran: what frame 0 of ofdmrx_debug_dump is
extern "C" long long ofdmrx_last_chunk_first_frame(ofdmrx_handle *h)
{
	if (!h)
		return OFDMRX_E_ARG;
	return (h->split_at && h->lane2) ? (long long)(h->split_at + h->lane2->last_first) : (long long)h->last_first;
}

// decode.cc:517-519 prints one Es/N0 value per constellation row; a batch caller gets them here: rows = n_frames x
// OFDMRX_ROWS_MAX floats (dB; rows a frame's mode does not have, and frames without a header: 0) in the memory space of
// the results of the decode calls that follow (device pointer for ofdmrx_decode_batch_device, host pointer for
// ofdmrx_decode_batch).  NULL turns the output off.
extern "C" int ofdmrx_set_esn0_rows(ofdmrx_handle *h, float *rows)
{
	if (!h)
		return OFDMRX_E_ARG;
	h->esn0_user = rows;
	return 0;
}

// decode.cc:400-447 prints one block of lines per preamble of the SKIP loop; a batch caller gets them here (see ofdmrx.h)
extern "C" int ofdmrx_set_attempt_log(ofdmrx_handle *h, ofdmrx_attempt *log, int32_t *counts)
{
	if (!h || (log == nullptr) != (counts == nullptr))
		return OFDMRX_E_ARG;
	h->att_user = log;
	h->att_counts_user = counts;
	return 0;
}

// frames of the last decode call that went through the list decoder (the others were decided by the syndrome
// certificate); -1: the certificate is off for this handle (every frame with a header is list-decoded)
extern "C" long long ofdmrx_list_decoded_frames(ofdmrx_handle *h)
{
	if (!h)
		return OFDMRX_E_ARG;
	if (!h->cert_mode)
		return -1;
	if (!h->q_ctl.p)
		return 0;
	if (hipSetDevice(h->cfg.device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
		return OFDMRX_E_HIP;
	ListQueue q;
	if (hipMemcpyAsync(&q, h->q_ctl.p, sizeof(q), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
		return OFDMRX_E_HIP;
	long long more = 0;
	if (h->split_at && h->lane2 && (more = ofdmrx_list_decoded_frames(h->lane2)) < 0)
		return more;
	return (long long)q.tail + more;                          // entries queued since the call began (both lanes)
}

// frames of the last decode call that the list-1 pass finished; -1: that pass is off for this handle
extern "C" long long ofdmrx_sc_decided_frames(ofdmrx_handle *h)
{
	if (!h)
		return OFDMRX_E_ARG;
	if (!h->sc_mode)
		return -1;
	if (!h->s_ctl.p)
		return 0;
	if (hipSetDevice(h->cfg.device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
		return OFDMRX_E_HIP;
	ListQueue q;
	if (hipMemcpyAsync(&q, h->s_ctl.p, sizeof(q), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
		return OFDMRX_E_HIP;
	long long more = 0;
	if (h->split_at && h->lane2 && (more = ofdmrx_sc_decided_frames(h->lane2)) < 0)
		return more;
	return (long long)q.done_total + more;
}

// device state for chunks of up to n frames; the list decoder's queue for calls whose chunks have up to n frames
int ensure_capacity(ofdmrx_handle *h, int n, bool mono, long samples)
{
	int r = 0;
	if (n > h->cap) {
		const size_t N = (size_t)n;
		r = r ? r : h->st.ensure(N * sizeof(SyncState));
		r = r ? r : h->hdr_soft.ensure(N * 256);
		r = r ? r : h->cons.ensure(N * CONS_MAX * sizeof(cf));
		r = r ? r : h->slope.ensure(N * ROWS_MAX * sizeof(float));
		r = r ? r : h->yint.ensure(N * ROWS_MAX * sizeof(float));
		r = r ? r : h->precision.ensure(N * ROWS_MAX * sizeof(float));
		r = r ? r : h->slot_of.ensure(N * sizeof(int));
		r = r ? r : h->chunk_flags.ensure(256);
		r = r ? r : h->res.ensure(N * sizeof(Result));
		r = r ? r : h->payload.ensure(N * PAYLOAD_BYTES);
		r = r ? r : h->rot_tap.ensure(CONS_MAX * sizeof(cf));
		// one 2 MiB level store per RESIDENT decoder, not per frame
		r = r ? r : h->soft.ensure((size_t)(std::min<long>((long)N, (long)h->polar_grid) + 8) * 8 * CODE_LEN * sizeof(float));
		// The queue: while the flush of chunk c - 1 reads its entries (fewer than flush_unit left over + one chunk) k_back of
		// chunk c adds one chunk at most, and k_back of chunk c + 1 waits for that flush to end (run_pipeline)
		h->flush_unit = (unsigned)std::max<long>(1, std::min<long>((long)N, (long)h->polar_grid));
		h->q_cap = (unsigned)(2 * N + h->flush_unit + 8);
		const size_t Q = h->q_cap;
		r = r ? r : h->q_ctl.ensure(sizeof(ListQueue));
		r = r ? r : h->q_slots.ensure(Q * sizeof(ListSlot));
		r = r ? r : h->q_llr.ensure(Q * CODE_LEN * sizeof(float));
		r = r ? r : h->q_hard.ensure(Q * CODE_LEN);
		r = r ? r : h->q_metric.ensure(Q * LIST * sizeof(float));
		if (h->cfg.flags & 1)                                     // (the per-lane messages are a debug tap)
			r = r ? r : h->q_lane_mesg.ensure(Q * LIST * MESG_BYTES);
		if (h->sc_mode) {                                         // the SC ring: a run behind k_back of every chunk takes whole residencies of
			// k_sc's decoders and leaves less than one of them to the next chunk (k_sc_plan): one chunk of slots + that
			const int sc_waves = h->sc_lb == 5 ? h->sc_grid * sc_codewords_per_wave(5) : h->sc_grid6;
			h->sc_unit = (unsigned)std::max<long>(1, std::min<long>((long)N, (long)sc_waves));
			h->s_cap = (unsigned)(N + h->sc_unit + 8);
			const size_t S = h->s_cap;
			r = r ? r : h->s_ctl.ensure(sizeof(ListQueue));
			r = r ? r : h->s_slots.ensure(S * sizeof(ListSlot));
			r = r ? r : h->s_llr.ensure(S * CODE_LEN * sizeof(float));
			r = r ? r : h->s_cw.ensure(S * (CODE_LEN / 8));
			r = r ? r : h->s_xw.ensure(S * (CODE_LEN / 8));
			r = r ? r : h->s_stat.ensure(S * sizeof(ScStat));
			r = r ? r : h->sc_soft.ensure((size_t)(std::min<long>((long)N, (long)std::max(h->sc_grid, h->sc_grid6)) + 1) * sc_store_bytes(0));
		}
		if (!demod_forms_cons(h->rate))                       // (the carriers go through HBM only when k_theil_sen forms the rows)
			r = r ? r : h->carr.ensure(N * CARR_MAX * sizeof(cf));
#ifndef SYNC_FFT_IN_LDS
#define SYNC_FFT_IN_LDS 1
#endif
		if (h->rate != 8000 || !SYNC_FFT_IN_LDS)
			r = r ? r : h->sc_scratch.ensure(N * (size_t)rate_symbol_len(h->rate) * sizeof(cf));
		if (r)
			return r;
		h->cap = n;
	}
	if (mono) {
		size_t need = (size_t)std::max(n, h->cap) * (size_t)samples;
		r = r ? r : h->z.ensure(need * sizeof(cf));           // scratch: only the windows the sync / header kernels read are ever written
		r = r ? r : h->dc.ensure((size_t)std::max(n, h->cap) * (size_t)mono_ck_per_frame(samples) * sizeof(double));
	}
	return r;
}

// the host waits for everything the handle has enqueued
int host_wait(ofdmrx_handle *h)
{
	HIP_OK(hipStreamSynchronize(h->stream));
	if (h->lane2)
		HIP_OK(hipStreamSynchronize(h->lane2->stream));
	return 0;
}
extern "C" int ofdmrx_synchronize(ofdmrx_handle *h)
{
	if (!h)
		return OFDMRX_E_ARG;
	return host_wait(h);
}

extern "C" int ofdmrx_get_timing(ofdmrx_handle *h, ofdmrx_timing *t)
{
	if (!h || !t)
		return OFDMRX_E_ARG;
	if (int r = host_wait(h))
		return r;
	std::memset(&h->timing, 0, sizeof(h->timing));
	h->sc_ms = 0.f;
	h->sc_launches = 0;
	for (const auto &sp : h->spans) {
		if (sp.a == (size_t)-1 || sp.b == (size_t)-1)
			continue;
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, h->ev_pool[sp.a], h->ev_pool[sp.b]) == hipSuccess) {
			if (sp.stage == T_SC) {
				h->sc_ms += ms;
				h->sc_launches += 1;
			} else {
				h->timing.ms[sp.stage] += ms;
				h->timing.launches[sp.stage] += 1;
			}
		}
	}
	if (h->split_at && h->lane2) {                                // the second lane's spans of the same call
		ofdmrx_timing t2;
		if (int r = ofdmrx_get_timing(h->lane2, &t2))
			return r;
		for (int i = 0; i < OFDMRX_T_COUNT; ++i) {
			h->timing.ms[i] += t2.ms[i];
			h->timing.launches[i] += t2.launches[i];
		}
		h->sc_ms += h->lane2->sc_ms;
		h->sc_launches += h->lane2->sc_launches;
	}
	*t = h->timing;
	return 0;
}

extern "C" int ofdmrx_get_sc_timing(ofdmrx_handle *h, float *ms, int32_t *launches)
{
	ofdmrx_timing t;
	if (int r = ofdmrx_get_timing(h, &t))
		return r;
	if (ms)
		*ms = h->sc_ms;
	if (launches)
		*launches = h->sc_launches;
	return 0;
}
