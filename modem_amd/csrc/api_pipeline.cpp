// api_pipeline.cpp -- the chunk pipeline behind both decode entries of include/ofdmrx.h (decode.cc:387-556 for batches of frames).
#include "api_internal.h"

// Events come from a pool that is sized BEFORE a call enqueues anything (ensure_events), so recording one can never
// allocate and an index is always valid; a failed hipEventRecord is remembered and fails the call at its end.
static int ensure_events(ofdmrx_handle *h, size_t need)
{
	while (h->ev_pool.size() < need) {
		hipEvent_t e;
		HIP_OK(hipEventCreate(&e));
		h->ev_pool.push_back(e);
	}
	return 0;
}
static size_t mark(ofdmrx_handle *h, hipStream_t on = nullptr)
{
	if (h->ev_used >= h->ev_pool.size()) {                // cannot happen: the pool was sized for the whole call
		h->sticky = hipErrorOutOfMemory;
		return 0;
	}
	size_t i = h->ev_used++;
	hipError_t e = hipEventRecord(h->ev_pool[i], on ? on : h->stream);
	if (e != hipSuccess && h->sticky == hipSuccess)
		h->sticky = e;
	return i;
}
static size_t events_per_chunk(int max_skip) { return 32 + 16 * (size_t)(max_skip + 1); }

// One resident chunk = every stage of SURVEY 8(a) D1..D8 as kernels on ONE stream, in three pieces so that the pipeline can put
// events between them:
//   front1  front end (mono), rounds of sync + header/OSD (decode.cc:390-448 do { } while (skip_count--)), demod
//   front2  Theil-Sen
//   back    k_back: rotation, SNR, certificate / LLRs into the list decoder's queue; k_queue_snap
// D9 + D10 for the queued frames = a flush (run_flush).
// wait_before_sync (event index or -1): the first sync launch waits for it; *ev_after_sync (nullable) receives the event
// recorded right after that launch - the pipeline gives the scan a slot of its own between two list-decoder launches.
static int run_front1(ofdmrx_handle *h, hipStream_t s, FrameBatch fb, int n, const int32_t *d_skip, int max_skip,
	size_t *t_begin, Attempt *d_att, int32_t *d_att_counts, size_t wait_before_sync = (size_t)-1, size_t *ev_after_sync = nullptr)
{
	const bool mono = fb.channels == 1;
	SyncState *st = h->st.as<SyncState>();
	cf *z = mono ? h->z.as<cf>() : nullptr;
	const MonoArgs ma = mono_args(h->host.front, mono ? h->dc.as<double>() : nullptr, mono_ck_per_frame(fb.samples_per_frame));
	size_t e0 = mark(h, s);
	if (mono) {                                               // D1: the DC blocker's states; the rest of it happens in the consumers
		Range r("ofdmrx:front_end");
		launch_mono_carries(s, h->rate, n, fb, h->host.front, h->dc.as<double>());
		if (!mono_fused(h->rate))                             // (the other rates: the whole analytic signal, read like 2-channel input)
			launch_front_end(s, h->rate, n, fb, ma, z);
	}
	size_t e1 = mark(h, s);
	launch_init_sync(s, n, st, d_skip, h->chunk_flags.as<int>(), d_att_counts);
	size_t last = e1;
	for (int round = 0; round <= max_skip; ++round) {         // decode.cc:390-448
		if (round == 0 && wait_before_sync != (size_t)-1)
			HIP_OK(hipStreamWaitEvent(s, h->ev_pool[wait_before_sync], 0));
		size_t a = mark(h, s);
		{
			Range r("ofdmrx:sync");
			launch_sync(s, h->rate, n, fb, z, h->dev, st, h->sc_scratch.as<cf>(), ma);
		}
		size_t b = mark(h, s);
		if (round == 0 && ev_after_sync)
			*ev_after_sync = b;
		{
			Range r("ofdmrx:header_osd");
			launch_header(s, h->rate, n, fb, z, ma, h->dev, st, h->hdr_soft.as<int8_t>(), d_att, d_att_counts);
		}
		size_t c = mark(h, s);
		h->spans.push_back({ OFDMRX_T_SYNC, a, b });
		h->spans.push_back({ OFDMRX_T_HEADER, b, c });
		last = c;
	}
	{
		Range r("ofdmrx:demod");
		launch_demod(s, h->rate, n, fb, z, ma, h->dev, st, h->cons.as<cf>(), h->carr.as<cf>());
	}
	size_t d = mark(h, s);
	h->spans.push_back({ OFDMRX_T_DEMOD, last, d });
	h->spans.push_back({ OFDMRX_T_FRONT, e0, e1 });
	*t_begin = e0;
	HIP_OK(hipGetLastError());
	h->last_n = n;
	h->last_mono = mono;
	h->last_spf = fb.samples_per_frame;
	h->last_fb = fb;
	return 0;
}

static int run_front2(ofdmrx_handle *h, hipStream_t s, int n)
{
	size_t e4 = mark(h, s);
	{
		Range r("ofdmrx:theil_sen");
		launch_theil_sen(s, n, h->st.as<SyncState>(), h->cons.as<cf>(), demod_forms_cons(h->rate) ? nullptr : h->carr.as<cf>(),
			h->slope.as<float>(), h->yint.as<float>(), h->chunk_flags.as<int>());
	}
	size_t e5 = mark(h, s);
	h->spans.push_back({ OFDMRX_T_THEILSEN, e4, e5 });
	HIP_OK(hipGetLastError());
	return 0;
}

// The list-1 pass on what k_back put into the SC ring (k_sc.hip): frames it decides are finished, the rest move on to the list
// decoder's queue - all in stream order with k_back's own entries there, so the snapshot behind it sees complete entries only
void run_sc_pass(ofdmrx_handle *h, hipStream_t s, int n, bool force, int chunk_seq)
{
	Range r("ofdmrx:sc_path");
	launch_sc_plan(s, h->sc_queue(), h->sc_unit, force ? 1 : 0);
	launch_sc(s, h->sc_lb, std::min(h->sc_grid, (n + 1) / 2), std::min(h->sc_grid6, n), h->sc_queue(), h->s_slots.as<ListSlot>(), h->s_llr.as<float>(),
		h->sc_soft.as<float>(), h->s_cw.as<unsigned long long>(), h->s_xw.as<unsigned long long>(), h->s_stat.as<ScStat>(), h->dev, h->sc_top);
	launch_sc_finish(s, (int)std::min<unsigned>(h->s_cap, (unsigned)n + h->sc_unit), h->sc_queue(), h->s_slots.as<ListSlot>(), h->s_llr.as<float>(), h->s_cw.as<unsigned long long>(),
		h->s_xw.as<unsigned long long>(), h->s_stat.as<ScStat>(), h->dev, h->cfg.descramble, h->queue(), h->q_slots.as<ListSlot>(),
		h->q_llr.as<float>(), h->slot_of.as<int>(), chunk_seq);
	launch_sc_adapt(s, h->sc_queue());
}

// D5's rotation + D6-D8 + the certificate: frames it finishes get payload + result here, the others a queue slot and their LLRs
// sc_force: the list-1 pass takes everything that waits in its ring (else whole residencies); chunk_seq: which chunk of its call
static int run_back(ofdmrx_handle *h, hipStream_t s, int par, int n, Result *d_res, float *d_esn0, uint8_t *d_payload,
	uint8_t *payload_later = nullptr, Result *res_later = nullptr, bool sc_force = true, int chunk_seq = 0)
{
	size_t e5 = mark(h, s);
	{
		Range r("ofdmrx:back");
		launch_back(s, h->rate, n, h->cert_mode, h->st.as<SyncState>(), h->cons.as<cf>(), h->slope.as<float>(), h->yint.as<float>(),
			h->precision.as<float>(), d_res, d_esn0, h->dev, h->cfg.descramble, d_payload, h->queue(), h->q_slots.as<ListSlot>(),
			h->q_llr.as<float>(), h->slot_of.as<int>(), payload_later, res_later, h->sc_ring(), chunk_seq);
	}
	size_t e6 = mark(h, s);
	h->spans.push_back({ OFDMRX_T_LLR, e5, e6 });
	if (h->sc_mode) {
		run_sc_pass(h, s, n, sc_force, chunk_seq);
		size_t e7 = mark(h, s);
		h->spans.push_back({ T_SC, e6, e7 });
	}
	launch_queue_snap(s, h->queue(), par);
	HIP_OK(hipGetLastError());
	return 0;
}

// A flush of the list decoder's queue: plan (what it takes: nothing until one residency waits, unless forced) | k_polar on
// s_polar, then k_finish on s_fin.  *ev_polar receives the event behind k_polar.
static int run_flush(ofdmrx_handle *h, hipStream_t s_polar, hipStream_t s_fin, int par, bool force, size_t t_begin, size_t *ev_polar)
{
	size_t e6 = mark(h, s_polar);
	{
		Range r("ofdmrx:polar_scl");
		launch_queue_plan(s_polar, h->queue(), par, h->flush_unit, force ? 1 : 0);
		launch_polar(s_polar, h->list, std::min(h->polar_grid, h->cap), h->queue(), par, h->q_slots.as<ListSlot>(), h->q_llr.as<float>(),
			h->soft.as<float>(), h->q_hard.as<uint8_t>(), h->dev, h->q_metric.as<float>());
	}
	size_t e7 = mark(h, s_polar);
	h->spans.push_back({ OFDMRX_T_POLAR, e6, e7 });
	if (ev_polar)
		*ev_polar = e7;
	if (s_fin != s_polar)
		HIP_OK(hipStreamWaitEvent(s_fin, h->ev_pool[e7], 0));
	size_t e8 = mark(h, s_fin);
	{
		Range r("ofdmrx:finish");
		launch_finish(s_fin, h->list, (int)h->q_cap, h->queue(), par, h->q_slots.as<ListSlot>(), h->q_llr.as<float>(), h->q_hard.as<uint8_t>(),
			h->dev, h->cfg.descramble, h->q_lane_mesg.as<uint8_t>());
	}
	size_t e9 = mark(h, s_fin);
	h->spans.push_back({ OFDMRX_T_FINISH, e8, e9 });
	h->spans.push_back({ OFDMRX_T_TOTAL, t_begin, e9 });
	HIP_OK(hipGetLastError());
	return 0;
}

static int check_args(ofdmrx_handle *h, const void *samples, int fmt, int channels, size_t spf, size_t stride,
	size_t n, const void *payload, const void *results)
{
	if (!h || !samples || !payload || !results || n == 0)
		return OFDMRX_E_ARG;
	if (fmt < OFDMRX_FMT_S16 || fmt > OFDMRX_FMT_F32 || channels < 1 || channels > 2)   // decode.cc:578
		return OFDMRX_E_ARG;
	size_t bps = fmt == OFDMRX_FMT_S16 ? 2 : fmt == OFDMRX_FMT_U8 ? 1 : 4;
	if (spf == 0 || spf > (size_t)0x7fffffff / 2 || stride < spf * bps * (size_t)channels)
		return OFDMRX_E_ARG;
	const size_t frame_bytes = bps * (size_t)channels;        // one sample frame: the kernels load I/Q pairs with one access
	if ((stride % frame_bytes) || ((size_t)samples % frame_bytes))   // whole sample frames between the frames, frames on such a boundary
		return OFDMRX_E_ARG;
	return 0;
}

// decode.cc:583-585,448: SKIP = number of preambles to pass over.  0..OFDMRX_MAX_SKIP per frame; anything else is an
// argument error (the reference would loop until the stream ends).  Returns the largest count or a negative error.
static int max_skip_of(const int32_t *skip, size_t n)
{
	int m = 0;
	for (size_t i = 0; i < n; ++i) {
		if (skip[i] < 0 || skip[i] > OFDMRX_MAX_SKIP)
			return OFDMRX_E_ARG;
		m = std::max(m, (int)skip[i]);
	}
	return m;
}

// How a call's frames are cut into pipeline stages: the handle's chunk, uniformly
struct ChunkPlan {
	std::vector<size_t> start;                                // n_chunks + 1 frame indices
	size_t count() const { return start.size() - 1; }
	size_t first(size_t c) const { return start[c]; }
	size_t size(size_t c) const { return start[c + 1] - start[c]; }
	size_t largest() const
	{
		size_t m = 0;
		for (size_t c = 0; c < count(); ++c)
			m = std::max(m, size(c));
		return m;
	}
};
// host_side: the call's outputs (and, for the host entry, its samples) cross PCIe.  A batch that fits one chunk then runs as two
// halves when it is large enough for half-sized kernels to fill the machine: the second half's kernels run beside the first half's
// copies (8192 frames: 1.49 -> 1.57 M frames/s; four quarters: 1.40 M, profiles/r04_v25_one_chunk_split.txt).  With the outputs
// left in HBM one chunk is the faster form (1.86 against 1.77 M).
static ChunkPlan plan_chunks(const ofdmrx_handle *h, size_t n_frames, bool host_side = false)
{
	ChunkPlan p;
	size_t step = (size_t)h->chunk;
	if (host_side && n_frames <= step && n_frames >= 6144 && !(h->cfg.flags & OFDMRX_FLAG_KEEP_RAW_CONS))   // (a handle with debug taps keeps
		step = (n_frames + 1) / 2;                                                                     // one chunk: its taps index the call's frames)
	for (size_t f = 0; f < n_frames; f += step)
		p.start.push_back(f);
	// the last chunk of a longer call likewise: nothing runs beside the copies of the call's last chunk, so it is the smaller the better
	// (headline: 1.845 -> 1.855 M frames/s; three or four parts: 1.84 / 1.81 M - OFDMRX_NO_TAIL_SPLIT=1 keeps the chunk whole)
	if (host_side && p.start.size() > 1 && n_frames - p.start.back() >= 6144 && !(h->cfg.flags & OFDMRX_FLAG_KEEP_RAW_CONS) && !std::getenv("OFDMRX_NO_TAIL_SPLIT"))
		p.start.push_back(p.start.back() + (n_frames - p.start.back() + 1) / 2);
	p.start.push_back(n_frames);
	return p;
}

// The chunk pipeline behind both entry points.  Chunk c's samples are at src(c) on the device when front1(c) runs
// (`ready` = event to wait for, or -1) and its payloads / results go to dst(c) (device buffers).
// after_front1(c) / after_flush(c) let the host-pointer entry hang its copies on the same queues.
struct PipeHooks {
	virtual ~PipeHooks() {}
	virtual int before_front1(size_t c, FrameBatch *fb, size_t *ready) = 0;   // fill fb.samples; ready = event index or -1
	virtual void dst(size_t c, uint8_t **payload, Result **res) = 0;
	virtual float *esn0(size_t) { return nullptr; }       // device destination of chunk c's per-row Es/N0 values (decode.cc:517-519), or null
	virtual void attempts(size_t, Attempt **log, int32_t **counts) { *log = nullptr; *counts = nullptr; }   // ... of its attempt log
	virtual int after_front1(size_t, size_t /*event*/) { return 0; }
	virtual int after_flush(size_t, hipStream_t /*the stream k_finish ran on*/) { return 0; }
	// dst(c) is a staging buffer that leaves before_flush(c): k_finish then delivers what the queue held back to dst_later(c) itself
	virtual void dst_later(size_t, uint8_t **payload, Result **res) { *payload = nullptr; *res = nullptr; }
	virtual int before_flush(size_t, hipStream_t /*the stream k_finish will run on*/, hipEvent_t /*behind k_back of that chunk*/) { return 0; }
	virtual bool outputs_leave_by_chunk() { return false; }   // every flush takes everything: chunk c is complete behind flush(c)
};

// Three queues, each chunk passes through all of them:
//   A (the handle's stream):  sync | header+OSD | demod | Theil-Sen | k_back    of chunk c, chunk after chunk
//   B:                        flush(c - 1) = plan | k_polar                     the list decoder, for what the queue holds
//   C:                        k_finish of flush(c - 1) (+ the host entry's output copies)
// * A kernel launched beside the resident list decoders makes no progress until they drain, so the scan - the
//   first kernel of a chunk - gets a slot of its own: sync(c) waits for polar(c - 2) to end and polar(c - 1) waits for sync(c).
//   Everything else on A runs beside polar(c - 1).
// * A flush takes nothing until one full residency of the list decoder waits in the queue, and then whole residencies; the
//   last flush of a call takes everything (so does every flush of a call whose outputs leave chunk by chunk).  At noise levels
//   where the certificate leaves a few frames per chunk the list decoder therefore runs once per several chunks, full.
// * k_back(c) waits for flush(c - 2) to have ended, copies included: that bounds the queue (ensure_capacity) and frees the host
//   entry's output staging of that parity.
// A call of one chunk (and OFDMRX_NO_OVERLAP=1, the profiler's setting: every kernel alone on the machine) runs all of it on A.
static int run_pipeline(ofdmrx_handle *h, PipeHooks &hooks, const ChunkPlan &plan, int fmt, int channels, size_t spf, size_t stride,
	const int32_t *d_skip, int max_skip)
{
	const size_t n_chunks = plan.count(), NONE = (size_t)-1;
	int r = ensure_events(h, h->ev_used + n_chunks * events_per_chunk(max_skip) + 8);
	r = r ? r : ensure_capacity(h, (int)plan.largest(), channels == 1, (long)spf);
	if (r)
		return r;
	const bool overlap = n_chunks > 1 && !std::getenv("OFDMRX_NO_OVERLAP");
	hipStream_t sa = h->stream, sb = overlap ? h->stream_b : sa, sc = overlap ? h->stream_fin : sa;
	const bool every = hooks.outputs_leave_by_chunk();
	// (The scan's slot of its own - sync(c) waits for polar(c - 2), polar(c - 1) for sync(c) - also keeps the list decoder's mostly
	// EMPTY launches of the default path in a fixed place between the front kernels: without the two waits the headline loses 2 - 3 %,
	// profiles/r06_hw_queues_and_two_lanes.txt)
	const bool scan_slot = overlap;
	std::vector<size_t> ev_back(n_chunks, NONE), ev_polar(n_chunks, NONE), ev_fin(n_chunks, NONE), ev_out(n_chunks, NONE), t0s(n_chunks, 0);
	launch_queue_reset(sa, h->queue(), h->q_cap);
	if (h->sc_mode)
		launch_queue_reset(sa, h->sc_queue(), h->s_cap);
	auto flush = [&](size_t p, size_t ev_sync_next) -> int {
		const int par = (int)(p & 1);
		if (overlap) {
			HIP_OK(hipStreamWaitEvent(sb, h->ev_pool[ev_back[p]], 0));
			if (ev_sync_next != NONE && scan_slot)
				HIP_OK(hipStreamWaitEvent(sb, h->ev_pool[ev_sync_next], 0));
			if (p >= 2)
				HIP_OK(hipStreamWaitEvent(sb, h->ev_pool[ev_fin[p - 2]], 0));   // k_finish(p - 2) has read the run of this parity
		}
		int rr = hooks.before_flush(p, sc, h->ev_pool[ev_back[p]]);
		ev_out[p] = mark(h, sc);                                  // (the chunk's own arrays have left for the host, if that is where they go)
		rr = rr ? rr : run_flush(h, sb, sc, par, every || p + 1 == n_chunks, t0s[p], &ev_polar[p]);
		rr = rr ? rr : hooks.after_flush(p, sc);
		ev_fin[p] = mark(h, sc);
		return rr;
	};
	for (size_t c = 0; c < n_chunks && !r; ++c) {
		const int n = (int)plan.size(c);
		FrameBatch fb{ nullptr, stride, (long)spf, fmt, channels };
		size_t ready = NONE, ev_sync = NONE;
		uint8_t *pay;
		Result *res;
		r = hooks.before_front1(c, &fb, &ready);
		if (r)
			break;
		hooks.dst(c, &pay, &res);
		Attempt *att;
		int32_t *att_counts;
		hooks.attempts(c, &att, &att_counts);
		if (ready != NONE)
			HIP_OK(hipStreamWaitEvent(sa, h->ev_pool[ready], 0));
		if (att && overlap && c >= 2)                             // (host entry: the log's staging of this parity has left with chunk c - 2)
			HIP_OK(hipStreamWaitEvent(sa, h->ev_pool[ev_fin[c - 2]], 0));
		r = run_front1(h, sa, fb, n, d_skip ? d_skip + plan.first(c) : nullptr, max_skip, &t0s[c], att, att_counts,
			(scan_slot && c >= 2) ? ev_polar[c - 2] : NONE, &ev_sync);
		h->last_first = plan.first(c);
		if (!r && overlap && c >= 1)                              // flush(c - 1): its LLRs are in the queue, sync(c) is on its way
			r = flush(c - 1, ev_sync);
		r = r ? r : hooks.after_front1(c, mark(h, sa));
		r = r ? r : run_front2(h, sa, n);
		if (r)
			break;
		if (overlap && c >= 2)
			HIP_OK(hipStreamWaitEvent(sa, h->ev_pool[ev_fin[c - 2]], 0));
		uint8_t *pay_later;
		Result *res_later;
		hooks.dst_later(c, &pay_later, &res_later);
		// a frame the list-1 pass left over from chunk c - 1 is finished by this chunk's run and delivered where the list decoder's
		// frames are (pay_later: the caller's arrays): behind the copy that took chunk c - 1's own arrays there
		if (overlap && c >= 1 && h->sc_mode && ev_out[c - 1] != NONE)
			HIP_OK(hipStreamWaitEvent(sa, h->ev_pool[ev_out[c - 1]], 0));
		r = run_back(h, sa, (int)(c & 1), n, res, hooks.esn0(c), pay, pay_later, res_later, every || c + 1 == n_chunks, (int)c);
		ev_back[c] = mark(h, sa);
		if (!r && !overlap)
			r = flush(c, NONE);
	}
	if (!r && overlap) {
		r = flush(n_chunks - 1, NONE);
		if (!r)                                                   // the caller's stream sees the finished batch (C is in order)
			HIP_OK(hipStreamWaitEvent(sa, h->ev_pool[ev_fin[n_chunks - 1]], 0));
	}
	return r;
}

// 1: pinned host memory, 0: device (or managed) memory, -1: the runtime does not know the pointer (pageable host memory)
static int host_pinned(const void *p)
{
	hipPointerAttribute_t a;
	if (hipPointerGetAttributes(&a, p) != hipSuccess) {
		(void)hipGetLastError();
		return -1;
	}
	if (a.type == hipMemoryTypeUnregistered)
		return -1;
	return a.type == hipMemoryTypeHost ? 1 : 0;
}

static int finish_call(ofdmrx_handle *h, int r)
{
	if (!r && h->sticky != hipSuccess) {
		g_last_error = std::string("hipEventRecord: ") + hipGetErrorString(h->sticky);
		r = OFDMRX_E_HIP;
	}
	h->sticky = hipSuccess;
	return r;
}

static int decode_device_lane(ofdmrx_handle *h, const void *d_samples, int fmt, int channels,
	size_t spf, size_t stride, size_t n_frames, const int32_t *d_skip, uint8_t *d_payload, ofdmrx_frame_result *d_results)
{
	int r = 0;
	int max_skip = 0;
	if (d_skip) {
		// the counts steer the host loop (rounds of sync + header): fetched on the handle's stream, so they are ordered
		// after whatever produced them there
		std::vector<int32_t> hs(n_frames);
		HIP_OK(hipMemcpyAsync(hs.data(), d_skip, n_frames * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
		HIP_OK(hipStreamSynchronize(h->stream));
		max_skip = max_skip_of(hs.data(), n_frames);
		if (max_skip < 0)
			return max_skip;
	}
	h->ev_used = 0;
	h->spans.clear();
	const int out_kind = host_pinned(d_payload), res_kind = host_pinned(d_results);
	if (out_kind < 0 || res_kind < 0 || out_kind != res_kind)
		return OFDMRX_E_ARG;
	// the optional outputs live in the memory space of the results: pinned host memory too, then
	if (out_kind == 1 && ((h->esn0_user && host_pinned(h->esn0_user) != 1) ||
			(h->att_user && (host_pinned(h->att_user) != 1 || host_pinned(h->att_counts_user) != 1))))
		return OFDMRX_E_ARG;
	const ChunkPlan plan = plan_chunks(h, n_frames, out_kind == 1);
	struct Dev : PipeHooks {
		const ChunkPlan *plan; const char *samples; size_t stride; uint8_t *pay; Result *res;
		int before_front1(size_t c, FrameBatch *fb, size_t *) override { fb->samples = samples + plan->first(c) * stride; return 0; }
		ofdmrx_handle *h = nullptr;
		bool host_out = false;                                    // pay / res are pinned host memory: per-chunk device buffers + copies
		void dst(size_t c, uint8_t **p, Result **r) override
		{
			if (host_out) {
				*p = ((c & 1) ? h->payload2 : h->payload).as<uint8_t>();
				*r = ((c & 1) ? h->res2 : h->res).as<Result>();
			} else {
				*p = pay + plan->first(c) * PAYLOAD_BYTES;
				*r = res + plan->first(c);
			}
		}
		// host outputs: the chunk's staging leaves right behind its k_back (certified frames complete, queued frames with their
		// preliminary record and a zeroed payload); what the list decoder finishes later k_finish writes into the pinned host arrays
		// itself, over PCIe - so the queue keeps working across chunks on this route too
		void dst_later(size_t c, uint8_t **p, Result **r) override
		{
			*p = host_out ? pay + plan->first(c) * PAYLOAD_BYTES : nullptr;
			*r = host_out ? res + plan->first(c) : nullptr;
		}
		int before_flush(size_t c, hipStream_t s, hipEvent_t back_done) override
		{
			if (!host_out)
				return 0;
			uint8_t *p;
			Result *rs;
			dst(c, &p, &rs);
			HIP_OK(hipStreamWaitEvent(s, back_done, 0));
			HIP_OK(hipMemcpyAsync(pay + plan->first(c) * PAYLOAD_BYTES, p, plan->size(c) * PAYLOAD_BYTES, hipMemcpyDeviceToHost, s));
			HIP_OK(hipMemcpyAsync(res + plan->first(c), rs, plan->size(c) * sizeof(Result), hipMemcpyDeviceToHost, s));
			return 0;
		}
		float *rows = nullptr;
		float *esn0(size_t c) override { return rows ? rows + plan->first(c) * ROWS_MAX : nullptr; }
		Attempt *att = nullptr;
		int32_t *attc = nullptr;
		void attempts(size_t c, Attempt **l, int32_t **n) override
		{
			*l = att ? att + plan->first(c) * ATTEMPTS_MAX : nullptr;
			*n = att ? attc + plan->first(c) : nullptr;
		}
	} hooks;
	hooks.rows = h->esn0_user;
	hooks.att = (Attempt *)h->att_user;
	hooks.attc = h->att_counts_user;
	hooks.plan = &plan;
	hooks.samples = (const char *)d_samples;
	hooks.stride = stride;
	hooks.pay = d_payload;
	hooks.res = (Result *)d_results;
	// Outputs in pinned HOST memory (hipHostMalloc / a registered range): every chunk's payloads and records leave for them on the
	// copy queue right behind the chunk's flush - beside the next chunk's kernels - instead of one copy of the whole batch that the
	// caller hangs behind the call.  (Samples stay where they are: in HBM.)
	hooks.host_out = out_kind == 1;
	if (hooks.host_out) {
		hooks.h = h;
		const size_t nc = plan.largest();
		r = h->payload.ensure(nc * PAYLOAD_BYTES);
		r = r ? r : h->res.ensure(nc * sizeof(Result));
		if (plan.count() > 1) {
			r = r ? r : h->payload2.ensure(nc * PAYLOAD_BYTES);
			r = r ? r : h->res2.ensure(nc * sizeof(Result));
		}
		if (r)
			return r;
	}
	return finish_call(h, run_pipeline(h, hooks, plan, fmt, channels, spf, stride, d_skip, max_skip));
}

extern "C" int ofdmrx_decode_batch_device(ofdmrx_handle *h, const void *d_samples, int fmt, int channels,
	size_t spf, size_t stride, size_t n_frames, const int32_t *d_skip, uint8_t *d_payload, ofdmrx_frame_result *d_results)
{
	int r = check_args(h, d_samples, fmt, channels, spf, stride, n_frames, d_payload, d_results);
	if (r)
		return r;
	HIP_OK(hipSetDevice(h->cfg.device));
	// two lanes: whole chunks to each, the second half through lane2 (see ofdmrx_handle).  Not for calls with SKIP counts (their
	// rounds are steered from the host), nor below four chunks of at least 1024 frames: a pipeline that short has nothing to share
	const size_t chunk = (size_t)h->chunk;
	const bool could_split = h->lanes == 2 && !d_skip && chunk >= 1024 && n_frames >= 4 * chunk;
	h->split_at = 0;
	if (!could_split)
		return decode_device_lane(h, d_samples, fmt, channels, spf, stride, n_frames, d_skip, d_payload, d_results);
	if (!h->lane2) {
		ofdmrx_config c2 = h->cfg;
		c2.stream = nullptr;                                      // a stream of its own
		c2.chunk_frames = h->chunk;
		c2.flags &= ~OFDMRX_FLAG_TWO_LANES;
		r = ofdmrx_create(&c2, &h->lane2);
		if (r) {                                                  // (no room for a second pipeline: one lane)
			h->lane2 = nullptr;
			h->lanes = 1;
			return decode_device_lane(h, d_samples, fmt, channels, spf, stride, n_frames, d_skip, d_payload, d_results);
		}
		h->lane2->cert_mode = h->cert_mode;
		h->lane2->sc_mode = h->sc_mode;
		HIP_OK(hipEventCreateWithFlags(&h->ev_lane_in, hipEventDisableTiming));
		HIP_OK(hipEventCreateWithFlags(&h->ev_lane_done, hipEventDisableTiming));
	}
	ofdmrx_handle *g = h->lane2;
	const size_t n1 = ((n_frames / chunk + 1) / 2) * chunk, n2 = n_frames - n1;
	// whatever the caller's stream has enqueued so far (the samples' producer) comes first for the second lane too
	HIP_OK(hipEventRecord(h->ev_lane_in, h->stream));
	HIP_OK(hipStreamWaitEvent(g->stream, h->ev_lane_in, 0));
	g->esn0_user = h->esn0_user ? h->esn0_user + n1 * ROWS_MAX : nullptr;
	g->att_user = h->att_user ? h->att_user + n1 * ATTEMPTS_MAX : nullptr;
	g->att_counts_user = h->att_counts_user ? h->att_counts_user + n1 : nullptr;
	r = decode_device_lane(h, d_samples, fmt, channels, spf, stride, n1, nullptr, d_payload, d_results);
	if (r)
		return r;
	r = decode_device_lane(g, (const char *)d_samples + n1 * stride, fmt, channels, spf, stride, n2, nullptr, d_payload + n1 * PAYLOAD_BYTES, d_results + n1);
	h->split_at = n1;
	// the caller's stream sees the finished batch
	HIP_OK(hipEventRecord(h->ev_lane_done, g->stream));
	HIP_OK(hipStreamWaitEvent(h->stream, h->ev_lane_done, 0));
	return r;
}

// Host-pointer entry: the same chunk pipeline with three copies hung on its events.  Chunk c+1 is copied in on a copy
// stream while chunk c runs (from pageable memory that call blocks the host thread - which is exactly the time the GPU
// needs for chunk c; from pinned memory it is asynchronous); the staging buffer of chunk c is free once front1(c) has
// read it (Theil-Sen, LLRs, polar work on the carriers).  Payloads and results leave through pinned staging buffers
// right behind the back half of their chunk.
extern "C" int ofdmrx_decode_batch(ofdmrx_handle *h, const void *samples, int fmt, int channels,
	size_t spf, size_t stride, size_t n_frames, const int32_t *skip, uint8_t *payload_out, ofdmrx_frame_result *results)
{
	int r = check_args(h, samples, fmt, channels, spf, stride, n_frames, payload_out, results);
	if (r)
		return r;
	HIP_OK(hipSetDevice(h->cfg.device));
	int max_skip = 0;
	if (skip) {
		max_skip = max_skip_of(skip, n_frames);
		if (max_skip < 0)
			return max_skip;
	}
	h->ev_used = 0;
	h->spans.clear();
	h->split_at = 0;
	if (!h->stream_c)
		HIP_OK(hipStreamCreateWithFlags(&h->stream_c, hipStreamNonBlocking));
	const ChunkPlan plan = plan_chunks(h, n_frames, true);
	const size_t n_chunks = plan.count(), nc = plan.largest();
	r = ensure_events(h, n_chunks * (events_per_chunk(max_skip) + 4) + 8);
	r = r ? r : h->in_stage.ensure(nc * stride);
	if (n_chunks > 1) {
		r = r ? r : h->in_stage2.ensure(nc * stride);
		r = r ? r : h->payload2.ensure(nc * PAYLOAD_BYTES);
		r = r ? r : h->res2.ensure(nc * sizeof(Result));
	}
	r = r ? r : h->payload.ensure(nc * PAYLOAD_BYTES);
	r = r ? r : h->res.ensure(nc * sizeof(Result));
	if (skip)
		r = r ? r : h->skip_stage.ensure(n_frames * sizeof(int32_t));
	if (r)
		return r;
	const size_t esn0_bytes = h->esn0_user ? nc * ROWS_MAX * sizeof(float) : 0;
	if (esn0_bytes) {
		r = h->esn0_dev.ensure(esn0_bytes);
		if (!r && n_chunks > 1)
			r = h->esn0_dev2.ensure(esn0_bytes);
		if (r)
			return r;
	}
	const size_t att_bytes = h->att_user ? nc * ATTEMPTS_MAX * sizeof(Attempt) : 0, attc_bytes = h->att_user ? nc * sizeof(int32_t) : 0;
	if (att_bytes) {
		r = h->att_dev.ensure(att_bytes);
		r = r ? r : h->attc_dev.ensure(attc_bytes);
		if (!r && n_chunks > 1) {
			r = h->att_dev2.ensure(att_bytes);
			r = r ? r : h->attc_dev2.ensure(attc_bytes);
		}
		if (r)
			return r;
	}
	const size_t out_bytes = nc * (PAYLOAD_BYTES + sizeof(Result)) + esn0_bytes + att_bytes + attc_bytes;
	for (int q = 0; q < (n_chunks > 1 ? 2 : 1); ++q)
		if (h->out_stage_cap[q] < out_bytes) {
			if (h->out_stage[q])
				(void)hipHostFree(h->out_stage[q]);
			h->out_stage[q] = nullptr;
			h->out_stage_cap[q] = 0;
			HIP_OK(hipHostMalloc(&h->out_stage[q], out_bytes, hipHostMallocDefault));
			h->out_stage_cap[q] = out_bytes;
		}
	if (skip) {
		HIP_OK(hipMemcpyAsync(h->skip_stage.p, skip, n_frames * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
		HIP_OK(hipStreamSynchronize(h->stream));         // `skip` may be pageable and go out of scope
	}
	struct Host : PipeHooks {
		ofdmrx_handle *h; const ChunkPlan *plan; const char *samples; size_t stride, n_chunks, nc;
		uint8_t *payload_out; ofdmrx_frame_result *results;
		std::vector<size_t> ev_in, ev_f1, ev_out;
		size_t copied_out = 0;
		size_t n_of(size_t c) const { return plan->size(c); }
		void *stage(size_t c) const { return (c & 1) ? h->in_stage2.p : h->in_stage.p; }
		int copy_in(size_t c)
		{
			if (c >= 2 && ev_f1[c - 2] != (size_t)-1)    // the staging buffer was last read by front1(c-2)
				HIP_OK(hipStreamWaitEvent(h->stream_c, h->ev_pool[ev_f1[c - 2]], 0));
			HIP_OK(hipMemcpyAsync(stage(c), samples + plan->first(c) * stride, n_of(c) * stride, hipMemcpyHostToDevice, h->stream_c));
			ev_in[c] = mark(h, h->stream_c);
			return 0;
		}
		int copy_out(size_t c)                             // pinned staging -> the caller's arrays, once chunk c has left the device
		{
			HIP_OK(hipEventSynchronize(h->ev_pool[ev_out[c]]));
			const char *src = (const char *)h->out_stage[c & 1];
			std::memcpy(payload_out + plan->first(c) * PAYLOAD_BYTES, src, n_of(c) * PAYLOAD_BYTES);
			std::memcpy(results + plan->first(c), src + nc * PAYLOAD_BYTES, n_of(c) * sizeof(Result));
			if (h->esn0_user)
				std::memcpy(h->esn0_user + plan->first(c) * ROWS_MAX, src + nc * (PAYLOAD_BYTES + sizeof(Result)), n_of(c) * ROWS_MAX * sizeof(float));
			if (h->att_user) {
				std::memcpy(h->att_user + plan->first(c) * ATTEMPTS_MAX, src + att_off, n_of(c) * ATTEMPTS_MAX * sizeof(Attempt));
				std::memcpy(h->att_counts_user + plan->first(c), src + att_off + nc * ATTEMPTS_MAX * sizeof(Attempt), n_of(c) * sizeof(int32_t));
			}
			return 0;
		}
		size_t att_off = 0;                                // where the attempt log starts in the pinned staging
		void attempts(size_t c, Attempt **l, int32_t **n) override
		{
			*l = h->att_user ? ((c & 1) ? h->att_dev2 : h->att_dev).as<Attempt>() : nullptr;
			*n = h->att_user ? ((c & 1) ? h->attc_dev2 : h->attc_dev).as<int32_t>() : nullptr;
		}
		int before_front1(size_t c, FrameBatch *fb, size_t *ready) override
		{
			if (c == 0) {
				int r = copy_in(0);
				if (r)
					return r;
			}
			fb->samples = stage(c);
			*ready = ev_in[c];
			return 0;
		}
		int after_front1(size_t c, size_t ev) override
		{
			ev_f1[c] = ev;
			return c + 1 < n_chunks ? copy_in(c + 1) : 0;    // chunk c+1 travels while chunk c is decoded
		}
		void dst(size_t c, uint8_t **p, Result **r) override
		{
			*p = ((c & 1) ? h->payload2 : h->payload).as<uint8_t>();
			*r = ((c & 1) ? h->res2 : h->res).as<Result>();
		}
		float *esn0(size_t c) override { return h->esn0_user ? ((c & 1) ? h->esn0_dev2 : h->esn0_dev).as<float>() : nullptr; }
		bool outputs_leave_by_chunk() override { return true; }
		int after_flush(size_t c, hipStream_t s) override
		{
			// out_stage[c & 1] still holds chunk c-2 until the host has copied it out
			while (copied_out + 2 <= c) {
				int r = copy_out(copied_out++);
				if (r)
					return r;
			}
			uint8_t *p;
			Result *rs;
			dst(c, &p, &rs);
			char *d = (char *)h->out_stage[c & 1];
			HIP_OK(hipMemcpyAsync(d, p, n_of(c) * PAYLOAD_BYTES, hipMemcpyDeviceToHost, s));
			HIP_OK(hipMemcpyAsync(d + nc * PAYLOAD_BYTES, rs, n_of(c) * sizeof(Result), hipMemcpyDeviceToHost, s));
			if (h->esn0_user)
				HIP_OK(hipMemcpyAsync(d + nc * (PAYLOAD_BYTES + sizeof(Result)), esn0(c), n_of(c) * ROWS_MAX * sizeof(float), hipMemcpyDeviceToHost, s));
			if (h->att_user) {
				Attempt *al;
				int32_t *an;
				attempts(c, &al, &an);
				HIP_OK(hipMemcpyAsync(d + att_off, al, n_of(c) * ATTEMPTS_MAX * sizeof(Attempt), hipMemcpyDeviceToHost, s));
				HIP_OK(hipMemcpyAsync(d + att_off + nc * ATTEMPTS_MAX * sizeof(Attempt), an, n_of(c) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
			}
			ev_out[c] = mark(h, s);
			return 0;
		}
	} hooks;
	hooks.h = h;
	hooks.samples = (const char *)samples;
	hooks.stride = stride;
	hooks.plan = &plan;
	hooks.n_chunks = n_chunks;
	hooks.nc = nc;
	hooks.payload_out = payload_out;
	hooks.results = results;
	hooks.att_off = nc * (PAYLOAD_BYTES + sizeof(Result)) + esn0_bytes;
	hooks.ev_in.assign(n_chunks, (size_t)-1);
	hooks.ev_f1.assign(n_chunks, (size_t)-1);
	hooks.ev_out.assign(n_chunks, (size_t)-1);
	r = run_pipeline(h, hooks, plan, fmt, channels, spf, stride, skip ? h->skip_stage.as<int32_t>() : nullptr, max_skip);
	while (!r && hooks.copied_out < n_chunks)
		r = hooks.copy_out(hooks.copied_out++);
	if (!r)
		HIP_OK(hipStreamSynchronize(h->stream));
	return finish_call(h, r);
}
