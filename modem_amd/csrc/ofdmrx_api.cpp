// ofdmrx_api.cpp -- host side of libofdmrx.so: the C ABI of include/ofdmrx.h.
// Replaces the in-process seam Decoder<value,cmplx,rate>(out, pcm, skip) (decode.cc:375)
// for batches of independent frames.  One handle = one GPU + one stream; frames are
// processed in resident chunks (device state for chunk_frames frames is allocated once and
// reused).  There is NO CPU fallback: every stage is a HIP kernel, errors are returned.
#include "../../include/ofdmrx.h"
#include "kernels.h"
#include "tables.h"
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace rx;

static_assert(sizeof(Result) == sizeof(ofdmrx_frame_result), "Result must mirror ofdmrx_frame_result");
static_assert(sizeof(Attempt) == sizeof(ofdmrx_attempt) && ATTEMPTS_MAX == OFDMRX_MAX_SKIP + 1, "Attempt must mirror ofdmrx_attempt");

namespace {

thread_local std::string g_last_error;

struct DevBuf {
	void *p = nullptr;
	size_t bytes = 0;
	int ensure(size_t need)
	{
		if (need <= bytes)
			return 0;
		if (p)
			(void)hipFree(p);
		p = nullptr;
		bytes = 0;
		hipError_t e = hipMalloc(&p, need);
		if (e != hipSuccess) {
			g_last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
			return OFDMRX_E_NOMEM;
		}
		bytes = need;
		return 0;
	}
	void release()
	{
		if (p)
			(void)hipFree(p);
		p = nullptr;
		bytes = 0;
	}
	template <typename T> T *as() const { return (T *)p; }
};

// ---- optional roctx ranges around the stage launches (OFDMRX_ROCTX=1): markers for rocprofv3 --marker-trace.
// The library is looked up at run time, so libofdmrx.so keeps its single dependency (libamdhip64).
struct Roctx {
	int (*push)(const char *) = nullptr;
	int (*pop)() = nullptr;
	Roctx()
	{
		if (!std::getenv("OFDMRX_ROCTX"))
			return;
		for (const char *name : { "librocprofiler-sdk-roctx.so", "libroctx64.so" }) {
			if (void *lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
				push = (int (*)(const char *))dlsym(lib, "roctxRangePushA");
				pop = (int (*)())dlsym(lib, "roctxRangePop");
				if (push && pop)
					return;
			}
		}
		push = nullptr;
		pop = nullptr;
	}
};
struct Range {                     // RAII: one named range per stage of a chunk
	static Roctx &api() { static Roctx r; return r; }
	explicit Range(const char *name) { if (api().push) api().push(name); }
	~Range() { if (api().pop) api().pop(); }
};

// call sign -> base-37 integer (the encoding of encode.cc:320-335: ' ' = 0, '0'..'9' = 1..10, letters of either case
// = 11..36); -1 for any other character
long long callsign_value(const char *str)
{
	static const std::array<int8_t, 256> digit = [] {
		std::array<int8_t, 256> t{};
		t.fill(-1);
		t[(unsigned char)' '] = 0;
		for (int i = 0; i < 10; ++i)
			t[(unsigned char)('0' + i)] = (int8_t)(1 + i);
		for (int i = 0; i < 26; ++i)
			t[(unsigned char)('A' + i)] = t[(unsigned char)('a' + i)] = (int8_t)(11 + i);
		return t;
	}();
	long long acc = 0;
	for (; *str; ++str) {
		const int d = digit[(unsigned char)*str];
		if (d < 0)
			return -1;
		acc = acc * 37 + d;
	}
	return acc;
}

}  // namespace

static constexpr int T_SC = OFDMRX_T_COUNT;   // internal stage index of the list-1 pass

struct ofdmrx_handle {
	ofdmrx_config cfg;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	int chunk = 0;
	long max_samples = 0;
	int rate = 8000;
	int list = 8;             // SCL list size: 8 (AVX2 build of the reference) or 4 (decode.cc:164-169)
	HostTables host;
	Tables dev{};
	std::vector<void *> table_allocs;
	// per-chunk device state: one buffer each - every stage from the scan to k_back runs on the handle's stream, chunk after
	// chunk; what crosses to the list decoder's streams goes through the queue below
	int cap = 0;              // frames the buffers below are sized for
	long cap_samples = 0;     // samples per frame the mono buffers are sized for
	DevBuf st, hdr_soft, cons, slope, yint, precision, slot_of, res, payload;
	DevBuf payload2, res2;    // second parity of the device-side output staging (host-pointer entry)
	DevBuf chunk_flags;       // per-chunk device flags (k_theil_sen: the largest row count met)
	DevBuf soft;              // the level stores of the resident list decoders (2 MiB each)
	// the list decoder's work queue (kernels.h: ListQueue): control block + one slot per entry
	DevBuf q_ctl, q_slots, q_llr, q_hard, q_metric, q_lane_mesg;
	unsigned q_cap = 0;       // slots
	// the SC ring in front of it (k_sc.hip): control block, one slot per frame of a chunk (LLRs, P*'s codeword, the channel's hard
	// decisions, ScStat), one level store per resident decoder
	DevBuf s_ctl, s_slots, s_llr, s_cw, s_xw, s_stat, sc_soft;
	unsigned s_cap = 0;
	unsigned sc_unit = 1;     // entries a run of the list-1 pass takes at a time: one residency of its decoders (the last run of a call: everything)
	int sc_mode = 1;          // 1: the list-1 pass (adaptive) in front of the list decoder, 0: off
	int sc_grid = 0, sc_grid6 = 0;   // resident SC decoders (waves): two codewords per wave / one (k_sc.hip)
	int sc_lb = 6;            // one codeword per wave (6, the default: with 64 loads in flight it is the faster layout at every run length, and
	                          // it moves 2.1 MB per codeword against 2.7), two (5), or 0: the run's length picks on the device (OFDMRX_SC_LB)
	ListQueue *sc_queue() const { return s_ctl.as<ListQueue>(); }
	ScRing sc_ring() const { return sc_mode ? ScRing{ s_ctl.as<ListQueue>(), s_slots.as<ListSlot>(), s_llr.as<float>() } : ScRing{ nullptr, nullptr, nullptr }; }
	unsigned flush_unit = 1;  // entries a flush takes at a time (one residency of the list decoder) unless it is forced
	DevBuf rot_tap;           // OFDMRX_TAP_CONS_ROT: the rotated rows of one frame, made on demand
	DevBuf tx_code, tx_rowsym, tx_tdom, tx_big;   // transmitter scratch, kept between calls (no allocation, no synchronisation per call)
	hipStream_t stream_b = nullptr;     // the list decoder (k_polar) of chunk c - 1 runs here beside the front stages of chunk c
	hipStream_t stream_fin = nullptr;   // k_finish (+ the host entry's output copies) of chunk c - 2
	hipStream_t stream_c = nullptr;     // host-pointer entry: host-to-device copies of the next chunk
	hipError_t sticky = hipSuccess;   // first failed hipEventRecord of the running call
	int polar_grid = 0;       // resident list decoders
	int cert_mode = 1;        // 1: syndrome certificate (adaptive), 0: every frame with a header is list-decoded
	float *esn0_user = nullptr;   // ofdmrx_set_esn0_rows: n x OFDMRX_ROWS_MAX floats in the memory space of the results (NULL = off)
	DevBuf esn0_dev, esn0_dev2;   // host-pointer entry: per-chunk device staging of the row values, by parity
	ofdmrx_attempt *att_user = nullptr;   // ofdmrx_set_attempt_log: n x (OFDMRX_MAX_SKIP + 1) records and n counts, same memory space (NULL = off)
	int32_t *att_counts_user = nullptr;
	DevBuf att_dev, att_dev2, attc_dev, attc_dev2;   // host-pointer entry: their device staging, by parity
	ListQueue *queue() const { return q_ctl.as<ListQueue>(); }
	DevBuf dc, z;             // mono front end only
	long last_spf = 0;
	DevBuf in_stage, in_stage2, skip_stage;
	void *out_stage[2] = { nullptr, nullptr };   // pinned host staging of payloads + results (host-pointer entry)
	size_t out_stage_cap[2] = { 0, 0 };
	DevBuf carr;                   // payload carriers of every symbol (demod -> Theil-Sen) at the rates whose demodulator does not form the rows
	DevBuf sc_scratch;             // rates above 8 kHz: 2 x symbol_len/2 cf per frame for the S&C trigger part
	int last_n = 0;           // frames in the last chunk (for taps)
	size_t last_first = 0;    // index of that chunk's first frame in its call (ofdmrx_last_chunk_first_frame)
	bool last_mono = false;
	FrameBatch last_fb{};     // the last chunk's samples (the ANALYTIC tap forms its frame's analytic signal from them)
	// timing
	std::vector<hipEvent_t> ev_pool;
	size_t ev_used = 0;
	struct Span { int stage; size_t a, b; };
	std::vector<Span> spans;
	ofdmrx_timing timing{};
	float sc_ms = 0.f;        // the list-1 pass (stage T_SC: ofdmrx_timing keeps its layout, ofdmrx_get_sc_timing reports it)
	int sc_launches = 0;
	// Two lanes (round 6, opt-in: OFDMRX_FLAG_TWO_LANES / OFDMRX_LANES=2): a device-entry call of four chunks or more is cut in two,
	// the second half runs through a second pipeline of the same configuration - `lane2`, a handle of its own with its own streams
	// and state - beside the first.  Kernels of the two lanes fill each other's gaps - the tail of a k_sc run in which most
	// persistent decoders have run out of codewords, the dispatch ramp of every launch: -20 dB 816 -> 850 - 865 k frames/s,
	// configs[3] 892 -> 950 - 990 k, -26 dB 1.42 -> 1.56 M.  Where the syndrome certificate finishes every frame there is nothing
	// to share (the front kernels are bound by vector issue): 1.82 M with one lane, 1.79 - 1.80 M with two - hence opt-in.  It needs
	// hardware queues of its own: HIP maps streams onto GPU_MAX_HW_QUEUES of them (default 4) and kernels of streams that share
	// one run in order (with 4 the lanes mostly alternate: 794 k at -20 dB).  profiles/r06_hw_queues_and_two_lanes.txt
	ofdmrx_handle *lane2 = nullptr;
	int lanes = 1;
	size_t split_at = 0;      // frames of the last call that went through this handle's own pipeline (0: all of them)
	hipEvent_t ev_lane_in = nullptr, ev_lane_done = nullptr;
};

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
	g_last_error = std::string(#call) + ": " + hipGetErrorString(e_); return OFDMRX_E_HIP; } } while (0)

template <typename T>
static int upload(ofdmrx_handle *h, const std::vector<T> &v, const T **out)
{
	void *p = nullptr;
	HIP_OK(hipMalloc(&p, v.size() * sizeof(T)));
	HIP_OK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
	h->table_allocs.push_back(p);
	*out = (const T *)p;
	return 0;
}

extern "C" int ofdmrx_abi_version(void) { return OFDMRX_ABI_VERSION; }
extern "C" int ofdmrx_abi_minor(void) { return OFDMRX_ABI_MINOR; }

extern "C" const char *ofdmrx_strerror(int err)
{
	switch (err) {
	case 0: return "ok";
	case OFDMRX_E_ARG: return "invalid argument";
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
		h->sc_grid6 = (std::getenv("OFDMRX_SC_WPC") ? swpc : 10) * std::max(cus, 1);   // 168 VGPRs, 16 KB of LDS: ten of these per CU
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
	r = r ? r : upload(h, h->host.node_lev, &h->dev.node_lev);
	r = r ? r : upload(h, h->host.node_lev64, &h->dev.node_lev64);
	r = r ? r : upload(h, h->host.node_lev32, &h->dev.node_lev32);
	r = r ? r : upload(h, h->host.genmat_bits, &h->dev.genmat_bits);
	r = r ? r : upload(h, h->host.osd_pairs, &h->dev.osd_pairs);
	r = r ? r : upload(h, h->host.osd_triples, &h->dev.osd_triples);
	r = r ? r : upload(h, h->host.crc32_tab, &h->dev.crc32_tab);
	r = r ? r : upload(h, h->host.crc32_shift168, &h->dev.crc32_shift168);
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
static int ensure_capacity(ofdmrx_handle *h, int n, bool mono, long samples)
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
static void run_sc_pass(ofdmrx_handle *h, hipStream_t s, int n, bool force = true, int chunk_seq = 0)
{
	Range r("ofdmrx:sc_path");
	launch_sc_plan(s, h->sc_queue(), h->sc_unit, force ? 1 : 0);
	launch_sc(s, h->sc_lb, std::min(h->sc_grid, (n + 1) / 2), std::min(h->sc_grid6, n), h->sc_queue(), h->s_slots.as<ListSlot>(), h->s_llr.as<float>(),
		h->sc_soft.as<float>(), h->s_cw.as<unsigned long long>(), h->s_xw.as<unsigned long long>(), h->s_stat.as<ScStat>(), h->dev);
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

// the host waits for everything the handle has enqueued
static int host_wait(ofdmrx_handle *h)
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
	const int grid = (int)std::min<size_t>(n, (size_t)std::max(h->sc_grid, h->sc_grid6));
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
			xw.as<unsigned long long>(), stat.as<ScStat>(), h->dev);
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
