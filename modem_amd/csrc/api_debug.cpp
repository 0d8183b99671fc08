// api_debug.cpp -- stage taps and single-stage entry points of include/ofdmrx.h (the parity tests' way in).
#include <cstdlib>
#include "api_internal.h"

// Stage taps of the LAST chunk the handle decoded.  CONS_RAW is the constellation as the demodulator left it (decode.cc:474-475),
// CONS_ROT what decode.cc:494 makes of it - produced on demand, the pipeline itself never stores it.  LLR / METRIC / LANE_MESG
// belong to frames that went through the list decoder: a frame without a queue slot (no header; finished by the syndrome
// certificate - create the handle with OFDMRX_FLAG_KEEP_RAW_CONS or OFDMRX_FLAG_SCL_ALWAYS to list-decode every frame) has
// none, and the call says so (OFDMRX_E_UNSUPPORTED) instead of returning stale memory; LANE_MESG needs
// OFDMRX_FLAG_KEEP_RAW_CONS.
extern "C" int ofdmrx_debug_dump(ofdmrx_handle *h, int tap, size_t frame, void *dst, size_t dst_bytes)
{
	if (!h || !dst)
		return OFDMRX_E_ARG;
	if (h->split_at && h->lane2)                                  // the call's last chunk went through the second lane
		return ofdmrx_debug_dump(h->lane2, tap, frame, dst, dst_bytes);
	if (frame >= (size_t)h->last_n)
		return OFDMRX_E_ARG;
	HIP_OK(hipSetDevice(h->cfg.device));
	if (int r = host_wait(h))
		return r;
	const void *src = nullptr;
	size_t cap = 0;   // bytes available per frame; the copy is min(dst_bytes, cap)
	int slot = -1;
	if (tap == OFDMRX_TAP_LLR || tap == OFDMRX_TAP_METRIC || tap == OFDMRX_TAP_LANE_MESG) {
		HIP_OK(hipMemcpy(&slot, h->slot_of.as<int>() + frame, sizeof(int), hipMemcpyDeviceToHost));
		if (slot <= -2 && tap == OFDMRX_TAP_LLR) {                // finished by the list-1 pass: its LLRs are still in the SC ring
			HIP_OK(hipMemcpy(dst, h->s_llr.as<float>() + (size_t)(-2 - slot) * CODE_LEN, std::min<size_t>(dst_bytes, CODE_LEN * 4), hipMemcpyDeviceToHost));
			return 0;
		}
		if (slot < 0 || (tap == OFDMRX_TAP_LANE_MESG && !h->q_lane_mesg.p))
			return OFDMRX_E_UNSUPPORTED;
	}
	switch (tap) {
	case OFDMRX_TAP_HDR_SOFT: src = h->hdr_soft.as<int8_t>() + frame * 256; cap = 255; break;
	case OFDMRX_TAP_CONS_RAW: src = h->cons.as<cf>() + frame * CONS_MAX; cap = CONS_MAX * sizeof(cf); break;
	case OFDMRX_TAP_CONS_ROT:
		launch_rotate_tap(h->stream, h->st.as<SyncState>() + frame, h->cons.as<cf>() + frame * CONS_MAX, h->slope.as<float>() + frame * ROWS_MAX,
			h->yint.as<float>() + frame * ROWS_MAX, h->rot_tap.as<cf>());
		HIP_OK(hipGetLastError());
		HIP_OK(hipStreamSynchronize(h->stream));
		src = h->rot_tap.p; cap = CONS_MAX * sizeof(cf); break;
	case OFDMRX_TAP_SLOPE: src = h->slope.as<float>() + frame * ROWS_MAX; cap = ROWS_MAX * 4; break;
	case OFDMRX_TAP_YINT: src = h->yint.as<float>() + frame * ROWS_MAX; cap = ROWS_MAX * 4; break;
	case OFDMRX_TAP_PRECISION: src = h->precision.as<float>() + frame * ROWS_MAX; cap = ROWS_MAX * 4; break;
	case OFDMRX_TAP_LLR: src = h->q_llr.as<float>() + (size_t)slot * CODE_LEN; cap = CODE_LEN * 4; break;
	case OFDMRX_TAP_METRIC: src = h->q_metric.as<float>() + (size_t)slot * LIST; cap = LIST * 4; break;
	case OFDMRX_TAP_LANE_MESG: src = h->q_lane_mesg.as<uint8_t>() + (size_t)slot * LIST * MESG_BYTES; cap = LIST * MESG_BYTES; break;
	case OFDMRX_TAP_ANALYTIC:
		if (!h->last_mono)
			return OFDMRX_E_ARG;
		{   // the pipeline never forms the whole analytic signal (mono_front.h): this does, for the one frame, from the last chunk's
			// samples (the caller's buffer for the device entry: it must still be there) and kept states
			FrameBatch fb1 = h->last_fb;
			fb1.samples = (const char *)fb1.samples + frame * fb1.frame_stride_bytes;
			const int ckpf = mono_ck_per_frame(fb1.samples_per_frame);
			launch_front_end(h->stream, h->rate, 1, fb1, mono_args(h->host.front, h->dc.as<double>() + frame * (size_t)ckpf, ckpf),
				h->z.as<cf>() + frame * (size_t)h->last_spf);
			HIP_OK(hipGetLastError());
			HIP_OK(hipStreamSynchronize(h->stream));
		}
		src = h->z.as<cf>() + frame * (size_t)h->last_spf;
		cap = (size_t)h->last_spf * sizeof(cf);
		break;
	default: return OFDMRX_E_ARG;
	}
	HIP_OK(hipMemcpy(dst, src, dst_bytes < cap ? dst_bytes : cap, hipMemcpyDeviceToHost));
	return 0;
}

// ---- single-stage entry points ------------------------------------------------------
// n mode-6 codewords straight into the list decoder's queue (slot i = codeword i), one forced flush
static int queue_run_all(ofdmrx_handle *h, int list)
{
	launch_queue_plan(h->stream, h->queue(), 0, 1, 1);
	launch_polar(h->stream, list, std::min(h->polar_grid, h->cap), h->queue(), 0, h->q_slots.as<ListSlot>(), h->q_llr.as<float>(), h->soft.as<float>(),
		h->q_hard.as<uint8_t>(), h->dev, h->q_metric.as<float>());
	return 0;
}
extern "C" int ofdmrx_debug_polar(ofdmrx_handle *h, const float *llr, size_t n, uint8_t *lane_mesg, float *metric)
{
	if (!h || !llr || !n || n > (size_t)h->chunk)
		return OFDMRX_E_ARG;
	HIP_OK(hipSetDevice(h->cfg.device));
	int r = ensure_capacity(h, (int)n, false, 0);
	if (r)
		return r;
	DevBuf lm;                                                // the per-lane messages: a buffer of this call's own
	r = lm.ensure((size_t)h->q_cap * LIST * MESG_BYTES);
	if (r)
		return r;
	if ((r = host_wait(h)))
		return r;
	HIP_OK(hipMemcpy(h->q_llr.p, llr, n * CODE_LEN * sizeof(float), hipMemcpyHostToDevice));
	HIP_OK(hipMemsetAsync(h->res.p, 0, n * sizeof(Result), h->stream));
	launch_queue_reset(h->stream, h->queue(), h->q_cap);
	launch_queue_fill(h->stream, h->queue(), h->q_slots.as<ListSlot>(), (int)n, h->payload.as<uint8_t>(), h->res.as<Result>(), 6);
	queue_run_all(h, h->list);
	launch_finish(h->stream, h->list, (int)n, h->queue(), 0, h->q_slots.as<ListSlot>(), h->q_llr.as<float>(), h->q_hard.as<uint8_t>(), h->dev, 0,
		lm.as<uint8_t>());
	hipError_t e = hipGetLastError();
	e = e == hipSuccess ? hipStreamSynchronize(h->stream) : e;
	if (e == hipSuccess && lane_mesg)
		e = hipMemcpy(lane_mesg, lm.p, n * LIST * MESG_BYTES, hipMemcpyDeviceToHost);
	if (e == hipSuccess && metric)
		e = hipMemcpy(metric, h->q_metric.p, n * LIST * sizeof(float), hipMemcpyDeviceToHost);
	lm.release();
	if (e != hipSuccess) {
		g_last_error = hipGetErrorString(e);
		return OFDMRX_E_HIP;
	}
	h->last_n = 0;
	return 0;
}

// the sign-following path alone: n LLR vectors -> k_sc's outputs (codeword, hard decisions, metric, min_fork, rule)
extern "C" int ofdmrx_debug_sc_path(ofdmrx_handle *h, const float *llr, size_t n, const int32_t *oper_modes, uint8_t *codeword, uint8_t *hard,
	float *metric, float *min_fork, int32_t *rule_ok)
{
	if (!h || !llr || !n || n > (size_t)h->chunk)
		return OFDMRX_E_ARG;
	for (size_t i = 0; oper_modes && i < n; ++i)
		if (oper_modes[i] < 6 || oper_modes[i] > 13)
			return OFDMRX_E_ARG;
	HIP_OK(hipSetDevice(h->cfg.device));
	int r = ensure_capacity(h, (int)n, false, 0);
	if (r)
		return r;
	// buffers of this call's own: the handle may have been created without the pass
	DevBuf ctl, slots, dl, cw, xw, stat, soft;
	int grid = (int)std::min<size_t>(n, (size_t)std::max(h->sc_grid, h->sc_grid6));
	if (const char *e = std::getenv("OFDMRX_SC_DECODERS"))          // (tests: a few decoders take the codewords one after the other - what a decoder
		grid = std::max(1, std::min(grid, std::atoi(e)));          // carries from one codeword to the next, k_sc.hip look_first, is exercised)
	r = r ? r : ctl.ensure(sizeof(ListQueue));
	r = r ? r : slots.ensure(n * sizeof(ListSlot));
	r = r ? r : dl.ensure(n * CODE_LEN * sizeof(float));
	r = r ? r : cw.ensure(n * (CODE_LEN / 8));
	r = r ? r : xw.ensure(n * (CODE_LEN / 8));
	r = r ? r : stat.ensure(n * sizeof(ScStat));
	r = r ? r : soft.ensure((size_t)(grid + 1) * sc_store_bytes(0));
	if (!r && (r = host_wait(h)) == 0) {
		hipError_t e = hipMemcpy(dl.p, llr, n * CODE_LEN * sizeof(float), hipMemcpyHostToDevice);
		launch_queue_reset(h->stream, ctl.as<ListQueue>(), (unsigned)n);
		launch_queue_fill(h->stream, ctl.as<ListQueue>(), slots.as<ListSlot>(), (int)n, h->payload.as<uint8_t>(), h->res.as<Result>(), 6);
		if (oper_modes && e == hipSuccess) {                      // (the slots' modes decide the frozen table and who sits beside whom)
			e = hipStreamSynchronize(h->stream);
			std::vector<ListSlot> ls(n);
			e = e == hipSuccess ? hipMemcpy(ls.data(), slots.p, n * sizeof(ListSlot), hipMemcpyDeviceToHost) : e;
			for (size_t i = 0; i < n; ++i)
				ls[i].oper_mode = oper_modes[i];
			e = e == hipSuccess ? hipMemcpy(slots.p, ls.data(), n * sizeof(ListSlot), hipMemcpyHostToDevice) : e;
		}
		launch_sc_plan(h->stream, ctl.as<ListQueue>());
		launch_sc(h->stream, h->sc_lb ? h->sc_lb : 6, grid, grid, ctl.as<ListQueue>(), slots.as<ListSlot>(), dl.as<float>(), soft.as<float>(), cw.as<unsigned long long>(),
			xw.as<unsigned long long>(), stat.as<ScStat>(), h->dev, h->sc_top);
		e = e == hipSuccess ? hipGetLastError() : e;
		e = e == hipSuccess ? hipStreamSynchronize(h->stream) : e;
		if (e == hipSuccess && codeword)
			e = hipMemcpy(codeword, cw.p, n * (CODE_LEN / 8), hipMemcpyDeviceToHost);
		if (e == hipSuccess && hard)
			e = hipMemcpy(hard, xw.p, n * (CODE_LEN / 8), hipMemcpyDeviceToHost);
		std::vector<ScStat> st(n);
		if (e == hipSuccess)
			e = hipMemcpy(st.data(), stat.p, n * sizeof(ScStat), hipMemcpyDeviceToHost);
		if (e != hipSuccess) {
			g_last_error = hipGetErrorString(e);
			r = OFDMRX_E_HIP;
		} else
			for (size_t i = 0; i < n; ++i) {
				if (metric) metric[i] = st[i].metric;
				if (min_fork) min_fork[i] = st[i].min_fork;
				if (rule_ok) rule_ok[i] = st[i].ok;
			}
	}
	for (DevBuf *b : { &ctl, &slots, &dl, &cw, &xw, &stat, &soft })
		b->release();
	h->last_n = 0;
	return r;
}

// D5 output -> payload: ROTATED constellation rows of mode-6 frames through D6-D10 exactly as the pipeline chains them (the rows'
// Theil-Sen lines are set to zero, so k_back's rotation is the identity), with the syndrome certificate (use_cert != 0: tried for
// every frame, the list decoder only for the frames it leaves) or without (the list decoder for every frame); cert_out
// (nullable) receives the certificate's verdict per frame (1 = finished by it)
extern "C" int ofdmrx_debug_decode_cons(ofdmrx_handle *h, const float *cons, size_t n, int use_cert, uint8_t *payload,
	ofdmrx_frame_result *results, int32_t *cert_out)
{
	if (!h || !cons || !n || n > (size_t)h->chunk || !payload || !results || h->list != 8 || use_cert < 0 || use_cert > 3)
		return OFDMRX_E_ARG;
	const bool with_sc = use_cert >= 2;                           // 2: syndrome certificate, list-1 pass, list decoder (the default chain); 3: without the first
	if (with_sc && !h->sc_mode)
		return OFDMRX_E_UNSUPPORTED;
	HIP_OK(hipSetDevice(h->cfg.device));
	int r = ensure_capacity(h, (int)n, false, 0);
	if (r)
		return r;
	std::vector<SyncState> st(n);
	std::memset(st.data(), 0, n * sizeof(SyncState));
	for (auto &s : st) { s.okay = 1; s.oper_mode = 6; }
	if ((r = host_wait(h)))
		return r;
	HIP_OK(hipMemcpy(h->st.p, st.data(), n * sizeof(SyncState), hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy2D(h->cons.p, CONS_MAX * sizeof(cf), cons, 21600 * sizeof(cf), 21600 * sizeof(cf), n, hipMemcpyHostToDevice));
	HIP_OK(hipMemsetAsync(h->res.p, 0, n * sizeof(Result), h->stream));
	HIP_OK(hipMemsetAsync(h->slope.p, 0, n * ROWS_MAX * sizeof(float), h->stream));
	HIP_OK(hipMemsetAsync(h->yint.p, 0, n * ROWS_MAX * sizeof(float), h->stream));
	launch_queue_reset(h->stream, h->queue(), h->q_cap);
	if (with_sc)
		launch_queue_reset(h->stream, h->sc_queue(), h->s_cap);
	launch_back(h->stream, h->rate, (int)n, (use_cert == 1 || use_cert == 2) ? 1 : 0, h->st.as<SyncState>(), h->cons.as<cf>(), h->slope.as<float>(), h->yint.as<float>(),
		h->precision.as<float>(), h->res.as<Result>(), nullptr, h->dev, h->cfg.descramble, h->payload.as<uint8_t>(), h->queue(),
		h->q_slots.as<ListSlot>(), h->q_llr.as<float>(), h->slot_of.as<int>(), nullptr, nullptr, with_sc ? h->sc_ring() : ScRing{ nullptr, nullptr, nullptr });
	if (with_sc)
		run_sc_pass(h, h->stream, (int)n);
	launch_queue_snap(h->stream, h->queue(), 0);
	queue_run_all(h, 8);
	launch_finish(h->stream, 8, (int)n, h->queue(), 0, h->q_slots.as<ListSlot>(), h->q_llr.as<float>(), h->q_hard.as<uint8_t>(), h->dev,
		h->cfg.descramble, nullptr);
	HIP_OK(hipGetLastError());
	HIP_OK(hipStreamSynchronize(h->stream));
	HIP_OK(hipMemcpy(payload, h->payload.p, n * PAYLOAD_BYTES, hipMemcpyDeviceToHost));
	HIP_OK(hipMemcpy(results, h->res.p, n * sizeof(Result), hipMemcpyDeviceToHost));
	if (cert_out) {
		std::vector<int> slot(n);
		HIP_OK(hipMemcpy(slot.data(), h->slot_of.p, n * sizeof(int), hipMemcpyDeviceToHost));
		for (size_t i = 0; i < n; ++i)
			cert_out[i] = slot[i] == -1 ? 1 : (slot[i] <= -2 ? 2 : 0);   // finished by the syndrome certificate / the list-1 pass / the list decoder
	}
	h->last_n = (int)n;
	return 0;
}

extern "C" int ofdmrx_debug_theil_sen(ofdmrx_handle *h, const float *y, size_t rows, int cols, float *slope, float *yint)
{
	if (!h || !y || !rows || cols < 2 || cols > 512 || !slope || !yint)
		return OFDMRX_E_ARG;
	HIP_OK(hipSetDevice(h->cfg.device));
	DevBuf dy, ds, di;
	int r = dy.ensure(rows * cols * 4);
	r = r ? r : ds.ensure(rows * 4);
	r = r ? r : di.ensure(rows * 4);
	if (!r) {
		hipError_t e = hipMemcpy(dy.p, y, rows * cols * 4, hipMemcpyHostToDevice);
		launch_theil_sen_raw(h->stream, (int)rows, cols, dy.as<float>(), ds.as<float>(), di.as<float>());
		e = e == hipSuccess ? hipStreamSynchronize(h->stream) : e;
		e = e == hipSuccess ? hipMemcpy(slope, ds.p, rows * 4, hipMemcpyDeviceToHost) : e;
		e = e == hipSuccess ? hipMemcpy(yint, di.p, rows * 4, hipMemcpyDeviceToHost) : e;
		if (e != hipSuccess) {
			g_last_error = hipGetErrorString(e);
			r = OFDMRX_E_HIP;
		}
	}
	dy.release(); ds.release(); di.release();
	return r;
}

extern "C" int ofdmrx_debug_osd(ofdmrx_handle *h, const int8_t *soft, size_t n, uint8_t *hard, int32_t *unique)
{
	if (!h || !soft || !n || !hard || !unique)
		return OFDMRX_E_ARG;
	HIP_OK(hipSetDevice(h->cfg.device));
	DevBuf dsf, dh, du;
	int r = dsf.ensure(n * 255);
	r = r ? r : dh.ensure(n * 32);
	r = r ? r : du.ensure(n * 4);
	if (!r) {
		hipError_t e = hipMemcpy(dsf.p, soft, n * 255, hipMemcpyHostToDevice);
		launch_osd_only(h->stream, (int)n, h->dev, dsf.as<int8_t>(), dh.as<uint8_t>(), du.as<int32_t>());
		e = e == hipSuccess ? hipStreamSynchronize(h->stream) : e;
		e = e == hipSuccess ? hipMemcpy(hard, dh.p, n * 32, hipMemcpyDeviceToHost) : e;
		e = e == hipSuccess ? hipMemcpy(unique, du.p, n * 4, hipMemcpyDeviceToHost) : e;
		if (e != hipSuccess) {
			g_last_error = hipGetErrorString(e);
			r = OFDMRX_E_HIP;
		}
	}
	dsf.release(); dh.release(); du.release();
	return r;
}

extern "C" int ofdmrx_debug_fft(ofdmrx_handle *h, const float *in, size_t n, int len, int sign, float *out)
{
	if (!h || !in || !out || !n || (len != rate_symbol_len(h->rate) && len != rate_symbol_len(h->rate) / 2) || (sign != 1 && sign != -1))
		return OFDMRX_E_ARG;
	HIP_OK(hipSetDevice(h->cfg.device));
	DevBuf di, dout;
	size_t bytes = n * (size_t)len * sizeof(cf);
	int r = di.ensure(bytes);
	r = r ? r : dout.ensure(bytes);
	if (!r) {
		hipError_t e = hipMemcpy(di.p, in, bytes, hipMemcpyHostToDevice);
		launch_fft_debug(h->stream, h->rate, (int)n, len, sign, di.as<cf>(), dout.as<cf>(), h->dev);
		e = e == hipSuccess ? hipStreamSynchronize(h->stream) : e;
		e = e == hipSuccess ? hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost) : e;
		if (e != hipSuccess) {
			g_last_error = hipGetErrorString(e);
			r = OFDMRX_E_HIP;
		}
	}
	di.release(); dout.release();
	return r;
}
