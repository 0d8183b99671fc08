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
	// per-chunk device state
	int cap = 0;              // frames the buffers below are sized for
	long cap_samples = 0;     // samples per frame the mono buffers are sized for
	DevBuf st, hdr_soft, cons, slope, yint, precision, llr, soft, hard, metric, lane_mesg, res, payload;
	DevBuf st2, llr2;         // second parity of the two buffers that cross from the front stages to the polar stage
	DevBuf cons2, slope2, yint2, precision2;   // ... and of what the LLR kernel reads: it runs on the back stream, ahead of the list decoder
	DevBuf payload2, res2;    // second parity of the device-side output staging (host-pointer entry)
	DevBuf work_counter;      // k_polar's shared codeword counter (zeroed on the stream before every launch)
	DevBuf tx_code, tx_rowsym, tx_tdom, tx_big;   // transmitter scratch, kept between calls (no allocation, no synchronisation per call)
	hipStream_t stream_b = nullptr;   // polar + finish of chunk c run here while the front stages of chunk c+1 run on `stream`
	hipStream_t stream_c = nullptr;   // host-pointer entry: host-to-device copies of the next chunk
	hipStream_t stream_f[2] = { nullptr, nullptr };   // sync / header / demod of a chunk run as four sub-batches over three streams
	hipError_t sticky = hipSuccess;   // first failed hipEventRecord of the running call
	int polar_grid = 0;       // resident polar decoders while overlapping (0 = one per codeword)
	int last_par = 0;         // parity used by the last chunk (taps)
	DevBuf cert, cert2;       // syndrome certificate: verdict per frame (+ one flag), by parity
	bool use_cert = true;     // no debug taps, not switched off
	float *esn0_user = nullptr;   // ofdmrx_set_esn0_rows: n x OFDMRX_ROWS_MAX floats in the memory space of the results (NULL = off)
	DevBuf esn0_dev, esn0_dev2;   // host-pointer entry: per-chunk device staging of the row values, by parity
	DevBuf cert_log;          // per chunk of the last call: frames the certificate left to the list decoder
	int cert_chunks = 0;
	int *cert_of(int par) { return use_cert ? (par ? cert2 : cert).as<int>() : nullptr; }
	DevBuf hard2;             // second parity of the list decoder's output: finish(c) reads its own while polar(c+1) writes
	uint8_t *hard_of(int par) { return (par ? hard2 : hard).as<uint8_t>(); }
	DevBuf st3;               // third SyncState array: finish(c-1) still reads its own while sync / header of chunk c+1 write theirs
	SyncState *st_of(int i) { return (i == 0 ? st : i == 1 ? st2 : st3).as<SyncState>(); }
	cf *cons_of(int par) { return (par ? cons2 : cons).as<cf>(); }
	float *slope_of(int par) { return (par ? slope2 : slope).as<float>(); }
	float *yint_of(int par) { return (par ? yint2 : yint).as<float>(); }
	float *precision_of(int par) { return (par ? precision2 : precision).as<float>(); }
	float *llr_of(int par) { return (par ? llr2 : llr).as<float>(); }
	DevBuf dc, z;             // mono front end only
	DevBuf cons_raw;          // only with cfg.flags & 1 (keep the pre-rotation constellation for taps)
	long last_spf = 0;
	DevBuf in_stage, in_stage2, skip_stage;
	void *out_stage[2] = { nullptr, nullptr };   // pinned host staging of payloads + results (host-pointer entry)
	size_t out_stage_cap[2] = { 0, 0 };
	DevBuf carr;                   // 8 kHz: payload carriers of every symbol (demod -> Theil-Sen)
	DevBuf sc_scratch;             // rates above 8 kHz: 2 x symbol_len/2 cf per frame for the S&C trigger part
	int last_n = 0;           // frames in the last chunk (for taps)
	bool last_mono = false;
	// timing
	std::vector<hipEvent_t> ev_pool;
	size_t ev_used = 0;
	struct Span { int stage; size_t a, b; };
	std::vector<Span> spans;
	ofdmrx_timing timing{};
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
	h->use_cert = !(cfg->flags & (OFDMRX_FLAG_KEEP_RAW_CONS | OFDMRX_FLAG_SCL_ALWAYS)) && !std::getenv("OFDMRX_NO_CERT");   // (the rule holds for any list size)
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
	{
		// OFDMRX_POLAR_PRIO=n (experiments): stream priority of the list decoder's queue (numerically larger = lower)
		int prio_lo = 0, prio_hi = 0;
		(void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
		const char *pe = std::getenv("OFDMRX_POLAR_PRIO");
		hipError_t e = pe ? hipStreamCreateWithPriority(&h->stream_b, hipStreamNonBlocking, std::max(prio_hi, std::min(prio_lo, std::atoi(pe))))
			: hipStreamCreateWithFlags(&h->stream_b, hipStreamNonBlocking);
		if (e != hipSuccess) {
			g_last_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
			ofdmrx_destroy(h);
			return OFDMRX_E_HIP;
		}
		e = hipStreamCreateWithFlags(&h->stream_c, hipStreamNonBlocking);
		if (e != hipSuccess) {
			g_last_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
			ofdmrx_destroy(h);
			return OFDMRX_E_HIP;
		}
		for (hipStream_t &sf : h->stream_f) {
			e = hipStreamCreateWithFlags(&sf, hipStreamNonBlocking);
			if (e != hipSuccess) {
				g_last_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
				ofdmrx_destroy(h);
				return OFDMRX_E_HIP;
			}
		}
		int cus = 0;
		(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device);
		int wpc = 16;             // resident list decoders per CU (the front stages of the next chunks share the machine with them)
		if (const char *e2 = std::getenv("OFDMRX_POLAR_WPC"))
			wpc = std::atoi(e2);
		h->polar_grid = wpc > 0 && cus > 0 ? wpc * cus : 0;
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
	if (h->stream)
		(void)hipStreamSynchronize(h->stream);
	for (hipStream_t sx : { h->stream_b, h->stream_c, h->stream_f[0], h->stream_f[1] })
		if (sx) {
			(void)hipStreamSynchronize(sx);
			(void)hipStreamDestroy(sx);
		}
	for (DevBuf *b : { &h->st, &h->hdr_soft, &h->cons, &h->slope, &h->yint, &h->precision, &h->llr, &h->soft, &h->hard,
			&h->metric, &h->lane_mesg, &h->res, &h->payload, &h->dc, &h->z, &h->cons_raw, &h->in_stage, &h->in_stage2, &h->skip_stage, &h->sc_scratch, &h->st2, &h->llr2, &h->carr, &h->payload2, &h->res2, &h->tx_code, &h->tx_rowsym, &h->tx_tdom, &h->tx_big, &h->work_counter, &h->cons2, &h->slope2, &h->yint2, &h->precision2, &h->st3, &h->hard2, &h->cert, &h->cert2, &h->cert_log, &h->esn0_dev, &h->esn0_dev2 })
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

// frames of the last decode call that went through the list decoder (the others were decided by the syndrome
// certificate); -1: the certificate is off for this handle (every frame with a header is list-decoded)
extern "C" long long ofdmrx_list_decoded_frames(ofdmrx_handle *h)
{
	if (!h)
		return OFDMRX_E_ARG;
	if (!h->use_cert)
		return -1;
	if (hipSetDevice(h->cfg.device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
		return OFDMRX_E_HIP;
	std::vector<int> v((size_t)h->cert_chunks);
	if (!v.empty() && hipMemcpy(v.data(), h->cert_log.p, v.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
		return OFDMRX_E_HIP;
	long long sum = 0;
	for (int x : v)
		sum += x;
	return sum;
}

constexpr int CERT_LOG_MAX = 4096;   // chunks per call whose list-decoder counts are kept (ofdmrx_list_decoded_frames)
static int ensure_capacity(ofdmrx_handle *h, int n, bool mono, long samples, bool two_parities = false)
{
	int r = 0;
	if (two_parities) {
		const size_t N2 = (size_t)std::max(n, h->cap);
		r = r ? r : h->st2.ensure(N2 * sizeof(SyncState));
		r = r ? r : h->st3.ensure(N2 * sizeof(SyncState));
		r = r ? r : h->llr2.ensure(N2 * CODE_LEN * sizeof(float));
		r = r ? r : h->cons2.ensure(N2 * CONS_MAX * sizeof(cf));
		r = r ? r : h->slope2.ensure(N2 * ROWS_MAX * sizeof(float));
		r = r ? r : h->yint2.ensure(N2 * ROWS_MAX * sizeof(float));
		r = r ? r : h->precision2.ensure(N2 * ROWS_MAX * sizeof(float));
		r = r ? r : h->hard2.ensure(N2 * CODE_LEN);
		r = r ? r : h->cert2.ensure((N2 + 2) * sizeof(int));
		if (r)
			return r;
	}
	if (n > h->cap) {
		const size_t N = (size_t)n;
		r = r ? r : h->st.ensure(N * sizeof(SyncState));
		r = r ? r : h->hdr_soft.ensure(N * 256);
		r = r ? r : h->cons.ensure(N * CONS_MAX * sizeof(cf));
		r = r ? r : h->slope.ensure(N * ROWS_MAX * sizeof(float));
		r = r ? r : h->yint.ensure(N * ROWS_MAX * sizeof(float));
		r = r ? r : h->precision.ensure(N * ROWS_MAX * sizeof(float));
		r = r ? r : h->llr.ensure(N * CODE_LEN * sizeof(float));
		// one 2 MiB level store per RESIDENT decoder (launches never use more than polar_grid of them), not per frame
		r = r ? r : h->soft.ensure((size_t)(std::min<long>((long)N, h->polar_grid > 0 ? h->polar_grid : (long)N) + 8) * 8 * CODE_LEN * sizeof(float));
		r = r ? r : h->hard.ensure(N * CODE_LEN);
		r = r ? r : h->metric.ensure(N * LIST * sizeof(float));
		r = r ? r : h->work_counter.ensure(256);
		r = r ? r : h->cert.ensure((N + 2) * sizeof(int));
		r = r ? r : h->cert_log.ensure(CERT_LOG_MAX * sizeof(int));
		r = r ? r : h->lane_mesg.ensure(N * LIST * MESG_BYTES);
		r = r ? r : h->res.ensure(N * sizeof(Result));
		r = r ? r : h->payload.ensure(N * PAYLOAD_BYTES);
		if (h->cfg.flags & 1)
			r = r ? r : h->cons_raw.ensure(N * CONS_MAX * sizeof(cf));
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
		r = r ? r : h->z.ensure(need * sizeof(cf));
		r = r ? r : h->dc.ensure(front_end_scratch_bytes(h->rate, std::max(n, h->cap), samples));
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

// One resident chunk = every stage of SURVEY 8(a) D1..D10 as kernels, in two halves:
//   front (D1..D8: front end, sync/header rounds, demod, Theil-Sen, LLRs) -> st[par], llr[par]
//   back  (D9, D10: polar list decoder, systematic bits / CRC / pack)     <- st[par], llr[par]
// A one-chunk call runs both on the handle's stream.  A longer batch is pipelined: back(c) runs on the
// second stream while front(c+1) runs on the handle's stream.
// wait_before_sync (event index or -1): the first sync launch waits for it; *ev_after_sync (nullable) receives the event
// recorded right after that launch - the three-queue schedule gives the scan a slot of its own between two polar launches.
static int run_front1(ofdmrx_handle *h, hipStream_t s, int par, int sti, FrameBatch fb, int n, const int32_t *d_skip, int max_skip,
	size_t *t_begin, size_t wait_before_sync = (size_t)-1, size_t *ev_after_sync = nullptr, int excl_level = 1)
{
	const bool mono = fb.channels == 1;
	SyncState *st = h->st_of(sti);
	const cf *z = mono ? h->z.as<cf>() : nullptr;
	size_t e0 = mark(h, s);
	if (mono) {
		Range r("ofdmrx:front_end");
		launch_front_end(s, h->rate, n, fb, h->host.front, h->dc.as<double>(), h->z.as<cf>());
	}
	size_t e1 = mark(h, s);
	// sync, header and demod are short, latency-bound kernels with different shapes (1 wave x 224 VGPRs, 4 waves x 216, 4 waves
	// x 128).  OFDMRX_FRONT_SPLIT=1 sends a chunk through them as four sub-batches over three streams, so that the sync of
	// one sub-batch runs beside the header / demod of another (frames are independent; every per-frame array is offset).
	// Measured neutral to slightly negative (147.9 k against 148.7 k frames/s), so it is off by default.
	static const bool split_on = std::getenv("OFDMRX_FRONT_SPLIT") != nullptr;
	const int parts = (n >= 2048 && split_on && h->stream_f[0] && h->stream_f[1]) ? 4 : 1;
	const int per = (n + parts - 1) / parts;
	const size_t sl = (size_t)rate_symbol_len(h->rate);
	size_t e3 = e1;
	for (int q = 0; q < parts; ++q) {
		const int f0 = q * per, nq = std::min(per, n - f0);
		if (nq <= 0)
			break;
		hipStream_t sq = q % 3 == 0 ? s : h->stream_f[q % 3 - 1];
		if (sq != s)
			HIP_OK(hipStreamWaitEvent(sq, h->ev_pool[e1], 0));
		FrameBatch fbq = fb;
		fbq.samples = (const char *)fb.samples + (size_t)f0 * fb.frame_stride_bytes;
		SyncState *stq = st + f0;
		const cf *zq = z ? z + (size_t)f0 * (size_t)fb.samples_per_frame : nullptr;
		cf *scq = h->sc_scratch.p ? h->sc_scratch.as<cf>() + (size_t)f0 * sl : nullptr;
		launch_init_sync(sq, nq, stq, d_skip ? d_skip + f0 : nullptr, h->work_counter.as<int>() + 16 + par);   // the chunk's flags, by parity
		size_t last = e1;
		for (int round = 0; round <= max_skip; ++round) {      // decode.cc:390-448 do { } while (skip_count--)
			if (round == 0 && q == 0 && wait_before_sync != (size_t)-1)
				HIP_OK(hipStreamWaitEvent(sq, h->ev_pool[wait_before_sync], 0));
			size_t a = mark(h, sq);
			{
				Range r("ofdmrx:sync");
				launch_sync(sq, h->rate, nq, fbq, zq, h->dev, stq, scq);
			}
			size_t b = mark(h, sq);
			if (round == 0 && q == 0 && ev_after_sync && excl_level == 1)
				*ev_after_sync = b;
			{
				Range r("ofdmrx:header_osd");
				launch_header(sq, h->rate, nq, fbq, zq, h->dev, stq, h->hdr_soft.as<int8_t>() + (size_t)f0 * 256);
			}
			size_t c = mark(h, sq);
			if (round == 0 && q == 0 && ev_after_sync && excl_level == 2)
				*ev_after_sync = c;
			h->spans.push_back({ OFDMRX_T_SYNC, a, b });
			h->spans.push_back({ OFDMRX_T_HEADER, b, c });
			last = c;
		}
		{
			Range r("ofdmrx:demod");
			launch_demod(sq, h->rate, nq, fbq, zq, h->dev, stq, h->cons_of(par) + (size_t)f0 * CONS_MAX,
				h->carr.p ? h->carr.as<cf>() + (size_t)f0 * CARR_MAX : nullptr);
		}
		size_t d = mark(h, sq);
		if (q == 0 && ev_after_sync && excl_level == 3)
			*ev_after_sync = d;
		h->spans.push_back({ OFDMRX_T_DEMOD, last, d });
		if (sq != s)
			HIP_OK(hipStreamWaitEvent(s, h->ev_pool[d], 0));
		e3 = d;
	}
	if ((h->cfg.flags & 1) && demod_forms_cons(h->rate))
		HIP_OK(hipMemcpyAsync(h->cons_raw.p, h->cons_of(par), (size_t)n * CONS_MAX * sizeof(cf), hipMemcpyDeviceToDevice, s));
	(void)e3;
	h->spans.push_back({ OFDMRX_T_FRONT, e0, e1 });
	*t_begin = e0;
	HIP_OK(hipGetLastError());
	h->last_n = n;
	h->last_mono = mono;
	h->last_spf = fb.samples_per_frame;
	h->last_par = par;
	return 0;
}

// front2 = the Theil-Sen stage (stream A).  The LLR kernel that follows it (D6-D8) is the first kernel of the back half:
// it is short and HBM-bound, and every millisecond on stream A is on the critical path of a chunk.
static int run_front2(ofdmrx_handle *h, hipStream_t s, int par, int sti, int n, Result *)
{
	SyncState *st = h->st_of(sti);
	size_t e4 = mark(h, s);
	const bool from_carr = !demod_forms_cons(h->rate);
	{
		Range r("ofdmrx:theil_sen");
		launch_theil_sen(s, n, st, h->cons_of(par), from_carr ? h->carr.as<cf>() : nullptr,
			(from_carr && (h->cfg.flags & 1)) ? h->cons_raw.as<cf>() : nullptr, h->slope_of(par), h->yint_of(par), h->work_counter.as<int>() + 16 + par);
	}
	size_t e5 = mark(h, s);
	h->spans.push_back({ OFDMRX_T_THEILSEN, e4, e5 });
	HIP_OK(hipGetLastError());
	return 0;
}

// The back half in two pieces, so that the pipeline can put an event between them: the LLR kernel (short, needs only the
// Theil-Sen results) and polar + finish; *ev_polar (nullable) receives the event recorded right after the polar kernel.
// D6-D8.  With the syndrome certificate on this is k_back (k_finish.hip): it also FINISHES the frames the certificate decides
// (payload + result), so it needs the payload destination.
static int run_llr(ofdmrx_handle *h, hipStream_t s, int par, int sti, int n, Result *d_res, float *d_esn0, uint8_t *d_payload)
{
	size_t e5 = mark(h, s);
	{
		Range r("ofdmrx:llr");
		if (h->use_cert) {
			int *log = (h->cert_chunks < CERT_LOG_MAX && h->cert_log.p) ? h->cert_log.as<int>() + h->cert_chunks : nullptr;
			launch_back(s, h->rate, n, h->st_of(sti), h->cons_of(par), h->slope_of(par), h->yint_of(par), h->precision_of(par),
				h->llr_of(par), d_res, d_esn0, h->dev, h->cfg.descramble, d_payload, h->cert_of(par), log);
			if (h->cert_chunks < CERT_LOG_MAX)
				++h->cert_chunks;
		} else {
			launch_llr(s, h->rate, n, h->st_of(sti), h->cons_of(par), h->slope_of(par), h->yint_of(par), h->precision_of(par),
				h->llr_of(par), d_res, d_esn0);
		}
	}
	size_t e6 = mark(h, s);
	h->spans.push_back({ OFDMRX_T_LLR, e5, e6 });
	HIP_OK(hipGetLastError());
	return 0;
}

static int run_polar(ofdmrx_handle *h, hipStream_t s, int par, int sti, int n, int grid, size_t *ev_begin, size_t *ev_end)
{
	size_t e6 = mark(h, s);
	{
		Range r("ofdmrx:polar_scl");
		launch_polar(s, h->list, n, grid, h->st_of(sti), h->llr_of(par), h->soft.as<float>(), h->hard_of(par), h->dev, h->metric.as<float>(), h->work_counter.as<int>(),
			h->cert_of(par));
	}
	size_t e7 = mark(h, s);
	h->spans.push_back({ OFDMRX_T_POLAR, e6, e7 });
	if (ev_begin)
		*ev_begin = e6;
	if (ev_end)
		*ev_end = e7;
	HIP_OK(hipGetLastError());
	return 0;
}
static int run_finish(ofdmrx_handle *h, hipStream_t s, int par, int sti, int n, uint8_t *d_payload, Result *d_res, bool want_lane_mesg,
	size_t t_begin)
{
	size_t e7 = mark(h, s);
	{
		Range r("ofdmrx:finish");
		launch_finish(s, h->list, n, h->st_of(sti), h->llr_of(par), h->hard_of(par), h->dev, h->cfg.descramble,
			want_lane_mesg ? h->lane_mesg.as<uint8_t>() : nullptr, d_payload, d_res, h->cert_of(par));
	}
	size_t e8 = mark(h, s);
	h->spans.push_back({ OFDMRX_T_FINISH, e7, e8 });
	h->spans.push_back({ OFDMRX_T_TOTAL, t_begin, e8 });
	HIP_OK(hipGetLastError());
	return 0;
}
static int run_polar_finish(ofdmrx_handle *h, hipStream_t s, int par, int sti, int n, int grid, uint8_t *d_payload, Result *d_res,
	bool want_lane_mesg, size_t t_begin, size_t *ev_polar)
{
	int r = run_polar(h, s, par, sti, n, grid, nullptr, ev_polar);
	return r ? r : run_finish(h, s, par, sti, n, d_payload, d_res, want_lane_mesg, t_begin);
}

static int run_back(ofdmrx_handle *h, hipStream_t s, int par, int sti, int n, int grid, uint8_t *d_payload, Result *d_res,
	bool want_lane_mesg, size_t t_begin, float *d_esn0 = nullptr)
{
	int r = run_llr(h, s, par, sti, n, d_res, d_esn0, d_payload);
	return r ? r : run_polar_finish(h, s, par, sti, n, grid, d_payload, d_res, want_lane_mesg, t_begin, nullptr);
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
	if (fmt == OFDMRX_FMT_S16 && channels == 2 && (stride & 3))
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

// How a call's frames are cut into pipeline stages: the handle's chunk, uniformly; a batch that fits one chunk runs every
// kernel back to back on the handle's stream.  OFDMRX_SPLIT_SMALL=1 cuts such a batch (>= 2048 frames) in two halves so
// that the two-stream overlap engages - the round-1 verdict asked for that (8192 frames per GPU when 65536 are sharded
// over 8), but it measures SLOWER than the plain sequence: 8192 frames in one call 132.6 k frames/s back to back, 129.4 k
// as 4096 + 4096, 123.7 k as 3328 + 4864 (one whole round of the resident polar grid, then the rest): with two chunks
// the pipeline is all fill and drain, and each polar launch pays its own ragged tail.  So it is off by default.
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
static ChunkPlan plan_chunks(const ofdmrx_handle *h, size_t n_frames)
{
	ChunkPlan p;
	const size_t chunk = (size_t)h->chunk;
	const bool split_small = std::getenv("OFDMRX_SPLIT_SMALL") != nullptr;
	if (split_small && n_frames <= chunk && n_frames >= 2048 && !std::getenv("OFDMRX_NO_OVERLAP")) {
		p.start = { 0, ((n_frames + 1) / 2 + 63) & ~(size_t)63, n_frames };
		return p;
	}
	// a short first chunk shortens the pipeline's fill (its sync / header / demod / Theil-Sen run with no polar stage beside them)
	size_t first = chunk;
	if (const char *e = std::getenv("OFDMRX_FIRST_CHUNK"))
		first = std::min(chunk, std::max<size_t>(64, (size_t)std::atol(e)));
	size_t f = 0;
	if (n_frames > chunk && first < chunk) {
		p.start.push_back(0);
		f = first;
	}
	for (; f < n_frames; f += chunk)
		p.start.push_back(f);
	p.start.push_back(n_frames);
	return p;
}

// The chunk pipeline behind both entry points.  Chunk c's samples are at src(c) on the device when front1(c) runs
// (`ready` = event to wait for, or -1) and its payloads / results go to dst(c) (device buffers).
//   A (the handle's stream):  front1(c)  sync | header+OSD | demod
//   B (second stream):                 back(c-1)  polar | finish      beside
//   A:                                 front2(c)  Theil-Sen | LLRs
// after_front1(c) / after_back(c) let the host-pointer entry hang its copies on the same events.
struct PipeHooks {
	virtual ~PipeHooks() {}
	virtual int before_front1(size_t c, FrameBatch *fb, size_t *ready) = 0;   // fill fb.samples; ready = event index or -1
	virtual void dst(size_t c, uint8_t **payload, Result **res) = 0;
	virtual float *esn0(size_t) { return nullptr; }       // device destination of chunk c's per-row Es/N0 values (decode.cc:517-519), or null
	virtual int after_front1(size_t, size_t /*event*/) { return 0; }
	virtual int after_back(size_t, size_t /*event*/, hipStream_t /*stream the back half ran on*/) { return 0; }
};

static int run_pipeline(ofdmrx_handle *h, PipeHooks &hooks, const ChunkPlan &plan, int fmt, int channels, size_t spf, size_t stride,
	const int32_t *d_skip, int max_skip)
{
	const size_t n_chunks = plan.count(), chunk_max = plan.largest();
	int r = ensure_events(h, h->ev_used + n_chunks * events_per_chunk(max_skip) + 8);
	if (r)
		return r;
	auto n_of = [&](size_t c) { return (int)plan.size(c); };
	if (n_chunks == 1 || !h->stream_b || std::getenv("OFDMRX_NO_OVERLAP")) {
		r = ensure_capacity(h, (int)chunk_max, channels == 1, (long)spf);
		for (size_t c = 0; c < n_chunks && !r; ++c) {
			FrameBatch fb{ nullptr, stride, (long)spf, fmt, channels };
			size_t ready = (size_t)-1, t0 = 0;
			uint8_t *pay;
			Result *res;
			r = hooks.before_front1(c, &fb, &ready);
			if (r)
				break;
			if (ready != (size_t)-1)
				HIP_OK(hipStreamWaitEvent(h->stream, h->ev_pool[ready], 0));
			hooks.dst(c, &pay, &res);
			r = run_front1(h, h->stream, 0, 0, fb, n_of(c), d_skip ? d_skip + plan.first(c) : nullptr, max_skip, &t0);
			r = r ? r : hooks.after_front1(c, mark(h, h->stream));
			r = r ? r : run_front2(h, h->stream, 0, 0, n_of(c), res);
			r = r ? r : run_back(h, h->stream, 0, 0, n_of(c), h->polar_grid, pay, res, true, t0, hooks.esn0(c));
			r = r ? r : hooks.after_back(c, mark(h, h->stream), h->stream);
		}
		return r;
	}
	r = ensure_capacity(h, (int)chunk_max, channels == 1, (long)spf, true);
	if (r)
		return r;
	const size_t NONE = (size_t)-1;
	static const bool sched_r2 = std::getenv("OFDMRX_SCHED_R2") != nullptr;   // the two-stream schedule of round 2 (below), for A/B runs
	if (!sched_r2 && h->stream_f[0]) {
		// Three queues, each chunk passes through all of them (round 3: since the Theil-Sen stage went from 15 ms to 3 ms per
		// chunk the list decoder IS the period, so its stream should wait for nothing but its own input):
		//   A (the handle's stream):  sync | header+OSD | demod | Theil-Sen | LLRs   of chunk c
		//   B:                        polar(c - 1), back to back
		//   C:                        finish(c - 2)
		// Header, demodulator, Theil-Sen and LLR kernels run beside the resident decoders at 1.1-1.6x their time alone.  The
		// scan does not: its 20 KB of LDS fit once per CU beside sixteen decoders and the lone wave starves (30 ms instead of
		// 0.85).  So sync(c) gets a slot of its own: it waits for polar(c-2) to end and polar(c-1) waits for it
		// (OFDMRX_SYNC_SHARED=1 drops both waits).
		// Buffers: llr / cons / slope / yint / precision / hard by parity c & 1, SyncState by c mod 3.  LLRs(c) overwrite what
		// polar(c-2) and finish(c-2) read; init_sync(c) overwrites what finish(c-3) read.
		static const int excl = std::getenv("OFDMRX_EXCL") ? std::atoi(std::getenv("OFDMRX_EXCL")) : 1;   // 0 none, 1 sync, 2 + header, 3 + demod
		static const bool sync_shared = excl == 0;
		hipStream_t sa = h->stream, sb = h->stream_b, sc = h->stream_f[0];
		std::vector<size_t> ev_done(n_chunks, NONE), ev_polar(n_chunks, NONE), ev_llr(n_chunks, NONE), t0s(n_chunks, 0);
		auto enqueue_polar = [&](size_t p, size_t ev_sync_next) -> int {   // polar(p) on B, finish(p) on C
			const int par = (int)(p & 1), sti = (int)(p % 3);
			uint8_t *pay;
			Result *res;
			hooks.dst(p, &pay, &res);
			HIP_OK(hipStreamWaitEvent(sb, h->ev_pool[ev_llr[p]], 0));
			if (ev_sync_next != NONE)
				HIP_OK(hipStreamWaitEvent(sb, h->ev_pool[ev_sync_next], 0));
			if (p >= 2)
				HIP_OK(hipStreamWaitEvent(sb, h->ev_pool[ev_done[p - 2]], 0));   // hard[par] is free
			int rr = run_polar(h, sb, par, sti, n_of(p), h->polar_grid, nullptr, &ev_polar[p]);
			if (rr)
				return rr;
			HIP_OK(hipStreamWaitEvent(sc, h->ev_pool[ev_polar[p]], 0));
			rr = run_finish(h, sc, par, sti, n_of(p), pay, res, true, t0s[p]);
			if (rr)
				return rr;
			ev_done[p] = mark(h, sc);
			return hooks.after_back(p, ev_done[p], sc);
		};
		for (size_t c = 0; c < n_chunks; ++c) {
			const int par = (int)(c & 1), sti = (int)(c % 3);
			FrameBatch fb{ nullptr, stride, (long)spf, fmt, channels };
			size_t ready = NONE, ev_sync = NONE;
			uint8_t *pay;
			Result *res;
			r = hooks.before_front1(c, &fb, &ready);
			if (r)
				return r;
			hooks.dst(c, &pay, &res);
			if (ready != NONE)
				HIP_OK(hipStreamWaitEvent(sa, h->ev_pool[ready], 0));
			if (c >= 3)
				HIP_OK(hipStreamWaitEvent(sa, h->ev_pool[ev_done[c - 3]], 0));
			r = run_front1(h, sa, par, sti, fb, n_of(c), d_skip ? d_skip + plan.first(c) : nullptr, max_skip, &t0s[c],
				(c >= 2 && !sync_shared) ? ev_polar[c - 2] : NONE, &ev_sync, excl);
			if (r)
				return r;
			if (c >= 1) {                                         // polar(c-1): its LLRs are on their way, sync(c) is in the queue
				r = enqueue_polar(c - 1, sync_shared ? NONE : ev_sync);
				if (r)
					return r;
			}
			r = hooks.after_front1(c, mark(h, sa));
			r = r ? r : run_front2(h, sa, par, sti, n_of(c), res);
			if (r)
				return r;
			if (c >= 2)
				HIP_OK(hipStreamWaitEvent(sa, h->ev_pool[ev_done[c - 2]], 0));
			r = run_llr(h, sa, par, sti, n_of(c), res, hooks.esn0(c), pay);
			if (r)
				return r;
			ev_llr[c] = mark(h, sa);
		}
		r = enqueue_polar(n_chunks - 1, NONE);
		if (r)
			return r;
		for (size_t c = n_chunks >= 3 ? n_chunks - 3 : 0; c < n_chunks; ++c)   // the caller's stream sees the finished batch
			HIP_OK(hipStreamWaitEvent(sa, h->ev_pool[ev_done[c]], 0));
		return 0;
	}
	std::vector<size_t> ev_back(n_chunks, NONE), ev_polar(n_chunks, NONE), ev_f2(n_chunks, NONE), t0s(n_chunks, 0);
	// Order of a period (chunk c on stream A, chunk c-1 on stream B):
	//   A: front1(c) = sync / header / demod, alone on the device
	//   A: front2(c) = Theil-Sen                      B: back(c-1) = llr, polar, finish
	// back(c-1) is launched once front1(c) is through, and front1(c+1) waits for back(c-1): the three front kernels are
	// short and latency-bound, and although each fits on a CU beside the resident polar grid (12 decoders x 96 VGPRs / 8 KB)
	// they starve there - measured: 134 k frames/s with the whole front inside the polar phase (OFDMRX_FRONT_OVERLAP=1)
	// against 158 k with this order.  Even the two light HBM-bound kernels of the back half disturb them: llr(c-1) beside
	// front1(c) (OFDMRX_LLR_EARLY=1) costs sync 1.25 -> 1.9 ms: 154.6 k; front1(c+1) beside finish(c-1)
	// (OFDMRX_FINISH_LATE=1; SyncState has three buffers, c mod 3, to allow it) is neutral: 158.7 k against 158.5 k.
	static const bool front_exclusive = std::getenv("OFDMRX_FRONT_OVERLAP") == nullptr;
	// experiments: the LLR kernel of chunk c-1 beside front1(c) / front1(c+1) not waiting for finish(c-1)
	static const bool llr_early = std::getenv("OFDMRX_LLR_EARLY") != nullptr, finish_late = std::getenv("OFDMRX_FINISH_LATE") != nullptr;
	auto enqueue_back = [&](size_t p, size_t ev_f1, bool last) -> int {
		uint8_t *pay;
		Result *res;
		hooks.dst(p, &pay, &res);
		const int par = (int)(p & 1), sti = (int)(p % 3);
		HIP_OK(hipStreamWaitEvent(h->stream_b, h->ev_pool[ev_f2[p]], 0));
		if (ev_f1 != NONE && !llr_early)
			HIP_OK(hipStreamWaitEvent(h->stream_b, h->ev_pool[ev_f1], 0));
		int rr = run_llr(h, h->stream_b, par, sti, n_of(p), res, hooks.esn0(p), pay);
		if (rr)
			return rr;
		if (ev_f1 != NONE && llr_early)
			HIP_OK(hipStreamWaitEvent(h->stream_b, h->ev_pool[ev_f1], 0));
		rr = run_polar_finish(h, h->stream_b, par, sti, n_of(p), h->polar_grid, pay, res, true, t0s[p], &ev_polar[p]);
		if (rr)
			return rr;
		ev_back[p] = mark(h, h->stream_b);
		return hooks.after_back(p, ev_back[p], h->stream_b);
	};
	for (size_t c = 0; c <= n_chunks; ++c) {
		const int par = (int)(c & 1), sti = (int)(c % 3);
		size_t ev_f1 = NONE;
		if (c >= 1 && !front_exclusive) {                    // nothing left to share the machine with after the last chunk: all decoders resident
			r = enqueue_back(c - 1, NONE, c == n_chunks);
			if (r)
				return r;
		}
		if (c < n_chunks) {
			FrameBatch fb{ nullptr, stride, (long)spf, fmt, channels };
			size_t ready = NONE;
			r = hooks.before_front1(c, &fb, &ready);
			if (r)
				return r;
			if (ready != NONE)
				HIP_OK(hipStreamWaitEvent(h->stream, h->ev_pool[ready], 0));
			// cons / slope / yint [par] are free once llr(c-2) has read them, st[sti] once finish(c-3) is done: both precede
			// polar(c-2) on stream B
			const std::vector<size_t> &gate = (front_exclusive && finish_late) ? ev_polar : ev_back;
			if (c >= 2 && gate[c - 2] != NONE)
				HIP_OK(hipStreamWaitEvent(h->stream, h->ev_pool[gate[c - 2]], 0));
			r = run_front1(h, h->stream, par, sti, fb, n_of(c), d_skip ? d_skip + plan.first(c) : nullptr, max_skip, &t0s[c]);
			if (r)
				return r;
			ev_f1 = mark(h, h->stream);
			r = hooks.after_front1(c, ev_f1);
			if (r)
				return r;
		}
		if (c >= 1 && front_exclusive) {
			r = enqueue_back(c - 1, ev_f1, c == n_chunks);
			if (r)
				return r;
		}
		if (c < n_chunks) {
			uint8_t *pay;
			Result *res;
			hooks.dst(c, &pay, &res);
			r = run_front2(h, h->stream, par, sti, n_of(c), res);
			if (r)
				return r;
			ev_f2[c] = mark(h, h->stream);
		}
	}
	for (size_t c = n_chunks >= 2 ? n_chunks - 2 : 0; c < n_chunks; ++c)   // the caller's stream sees the finished batch
		if (ev_back[c] != NONE)
			HIP_OK(hipStreamWaitEvent(h->stream, h->ev_pool[ev_back[c]], 0));
	return 0;
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

extern "C" int ofdmrx_decode_batch_device(ofdmrx_handle *h, const void *d_samples, int fmt, int channels,
	size_t spf, size_t stride, size_t n_frames, const int32_t *d_skip, uint8_t *d_payload, ofdmrx_frame_result *d_results)
{
	int r = check_args(h, d_samples, fmt, channels, spf, stride, n_frames, d_payload, d_results);
	if (r)
		return r;
	HIP_OK(hipSetDevice(h->cfg.device));
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
	h->cert_chunks = 0;
	h->spans.clear();
	const ChunkPlan plan = plan_chunks(h, n_frames);
	struct Dev : PipeHooks {
		const ChunkPlan *plan; const char *samples; size_t stride; uint8_t *pay; Result *res;
		int before_front1(size_t c, FrameBatch *fb, size_t *) override { fb->samples = samples + plan->first(c) * stride; return 0; }
		void dst(size_t c, uint8_t **p, Result **r) override { *p = pay + plan->first(c) * PAYLOAD_BYTES; *r = res + plan->first(c); }
		float *rows = nullptr;
		float *esn0(size_t c) override { return rows ? rows + plan->first(c) * ROWS_MAX : nullptr; }
	} hooks;
	hooks.rows = h->esn0_user;
	hooks.plan = &plan;
	hooks.samples = (const char *)d_samples;
	hooks.stride = stride;
	hooks.pay = d_payload;
	hooks.res = (Result *)d_results;
	return finish_call(h, run_pipeline(h, hooks, plan, fmt, channels, spf, stride, d_skip, max_skip));
}

extern "C" int ofdmrx_synchronize(ofdmrx_handle *h)
{
	if (!h)
		return OFDMRX_E_ARG;
	HIP_OK(hipStreamSynchronize(h->stream));
	return 0;
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
	h->cert_chunks = 0;
	h->spans.clear();
	const ChunkPlan plan = plan_chunks(h, n_frames);
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
	const size_t out_bytes = nc * (PAYLOAD_BYTES + sizeof(Result)) + esn0_bytes;
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
			return 0;
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
		int after_back(size_t c, size_t, hipStream_t s) override
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
	HIP_OK(hipStreamSynchronize(h->stream));
	std::memset(&h->timing, 0, sizeof(h->timing));
	for (const auto &sp : h->spans) {
		if (sp.a == (size_t)-1 || sp.b == (size_t)-1)
			continue;
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, h->ev_pool[sp.a], h->ev_pool[sp.b]) == hipSuccess) {
			h->timing.ms[sp.stage] += ms;
			h->timing.launches[sp.stage] += 1;
		}
	}
	*t = h->timing;
	return 0;
}

extern "C" int ofdmrx_debug_dump(ofdmrx_handle *h, int tap, size_t frame, void *dst, size_t dst_bytes)
{
	if (!h || !dst || frame >= (size_t)h->last_n)
		return OFDMRX_E_ARG;
	const void *src = nullptr;
	size_t bytes = 0;
	size_t cap = 0;   // bytes available per frame; the copy is min(dst_bytes, cap)
	switch (tap) {
	case OFDMRX_TAP_HDR_SOFT: src = h->hdr_soft.as<int8_t>() + frame * 256; cap = 255; break;
	case OFDMRX_TAP_CONS_RAW:   /* D5 rotates in place: the raw copy exists only with cfg.flags & 1 */
		if (!(h->cfg.flags & 1))
			return OFDMRX_E_ARG;
		src = h->cons_raw.as<cf>() + frame * CONS_MAX; cap = CONS_MAX * sizeof(cf); break;
	case OFDMRX_TAP_CONS_ROT: src = h->cons_of(h->last_par) + frame * CONS_MAX; cap = CONS_MAX * sizeof(cf); break;
	case OFDMRX_TAP_SLOPE: src = h->slope_of(h->last_par) + frame * ROWS_MAX; cap = ROWS_MAX * 4; break;
	case OFDMRX_TAP_YINT: src = h->yint_of(h->last_par) + frame * ROWS_MAX; cap = ROWS_MAX * 4; break;
	case OFDMRX_TAP_PRECISION: src = h->precision_of(h->last_par) + frame * ROWS_MAX; cap = ROWS_MAX * 4; break;
	case OFDMRX_TAP_LLR: src = h->llr_of(h->last_par) + frame * CODE_LEN; cap = CODE_LEN * 4; break;
	case OFDMRX_TAP_METRIC: src = h->metric.as<float>() + frame * LIST; cap = LIST * 4; break;
	case OFDMRX_TAP_LANE_MESG: src = h->lane_mesg.as<uint8_t>() + frame * LIST * MESG_BYTES; cap = LIST * MESG_BYTES; break;
	case OFDMRX_TAP_ANALYTIC:
		if (!h->last_mono)
			return OFDMRX_E_ARG;
		src = h->z.as<cf>() + frame * (size_t)h->last_spf;
		cap = (size_t)h->last_spf * sizeof(cf);
		break;
	default: return OFDMRX_E_ARG;
	}
	bytes = dst_bytes < cap ? dst_bytes : cap;
	HIP_OK(hipStreamSynchronize(h->stream));
	HIP_OK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
	return 0;
}

// ---- single-stage entry points ------------------------------------------------------
extern "C" int ofdmrx_debug_polar(ofdmrx_handle *h, const float *llr, size_t n, uint8_t *lane_mesg, float *metric)
{
	if (!h || !llr || !n || n > (size_t)h->chunk)
		return OFDMRX_E_ARG;
	HIP_OK(hipSetDevice(h->cfg.device));
	int r = ensure_capacity(h, (int)n, false, 0);
	if (r)
		return r;
	std::vector<SyncState> st(n);
	std::memset(st.data(), 0, n * sizeof(SyncState));
	for (auto &s : st) { s.okay = 1; s.oper_mode = 6; }
	HIP_OK(hipMemcpy(h->st.p, st.data(), n * sizeof(SyncState), hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(h->llr.p, llr, n * CODE_LEN * sizeof(float), hipMemcpyHostToDevice));
	HIP_OK(hipMemsetAsync(h->res.p, 0, n * sizeof(Result), h->stream));
	launch_polar(h->stream, h->list, (int)n, h->polar_grid, h->st.as<SyncState>(), h->llr.as<float>(), h->soft.as<float>(), h->hard.as<uint8_t>(), h->dev, h->metric.as<float>(),
		h->work_counter.as<int>());
	launch_finish(h->stream, h->list, (int)n, h->st.as<SyncState>(), h->llr.as<float>(), h->hard.as<uint8_t>(), h->dev, 0,
		h->lane_mesg.as<uint8_t>(), h->payload.as<uint8_t>(), h->res.as<Result>());
	HIP_OK(hipGetLastError());
	HIP_OK(hipStreamSynchronize(h->stream));
	if (lane_mesg)
		HIP_OK(hipMemcpy(lane_mesg, h->lane_mesg.p, n * LIST * MESG_BYTES, hipMemcpyDeviceToHost));
	if (metric)
		HIP_OK(hipMemcpy(metric, h->metric.p, n * LIST * sizeof(float), hipMemcpyDeviceToHost));
	h->last_n = (int)n;
	return 0;
}

// D5 output -> payload: rotated constellation rows of mode-6 frames through D6-D10 exactly as the pipeline chains them, with the
// syndrome certificate (use_cert != 0: k_back, the list decoder only for the frames it leaves) or without (k_llr, the list decoder
// for every frame); cert_out (nullable) receives the certificate's verdict per frame (1 = finished by it)
extern "C" int ofdmrx_debug_decode_cons(ofdmrx_handle *h, const float *cons, size_t n, int use_cert, uint8_t *payload,
	ofdmrx_frame_result *results, int32_t *cert_out)
{
	if (!h || !cons || !n || n > (size_t)h->chunk || !payload || !results || h->list != 8)
		return OFDMRX_E_ARG;
	HIP_OK(hipSetDevice(h->cfg.device));
	int r = ensure_capacity(h, (int)n, false, 0);
	if (r)
		return r;
	std::vector<SyncState> st(n);
	std::memset(st.data(), 0, n * sizeof(SyncState));
	for (auto &s : st) { s.okay = 1; s.oper_mode = 6; }
	HIP_OK(hipMemcpy(h->st.p, st.data(), n * sizeof(SyncState), hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy2D(h->cons.p, CONS_MAX * sizeof(cf), cons, 21600 * sizeof(cf), 21600 * sizeof(cf), n, hipMemcpyHostToDevice));
	HIP_OK(hipMemsetAsync(h->res.p, 0, n * sizeof(Result), h->stream));
	HIP_OK(hipMemsetAsync(h->slope.p, 0, n * ROWS_MAX * sizeof(float), h->stream));
	HIP_OK(hipMemsetAsync(h->yint.p, 0, n * ROWS_MAX * sizeof(float), h->stream));
	HIP_OK(hipMemsetAsync(h->cert.p, 0, (n + 2) * sizeof(int), h->stream));
	int *cert = use_cert ? h->cert.as<int>() : nullptr;
	SyncState *dst = h->st.as<SyncState>();
	if (cert)
		launch_back(h->stream, h->rate, (int)n, dst, h->cons.as<cf>(), h->slope.as<float>(), h->yint.as<float>(), h->precision.as<float>(),
			h->llr.as<float>(), h->res.as<Result>(), nullptr, h->dev, h->cfg.descramble, h->payload.as<uint8_t>(), cert, nullptr);
	else
		launch_llr(h->stream, h->rate, (int)n, dst, h->cons.as<cf>(), h->slope.as<float>(), h->yint.as<float>(), h->precision.as<float>(),
			h->llr.as<float>(), h->res.as<Result>(), nullptr);
	launch_polar(h->stream, 8, (int)n, h->polar_grid, dst, h->llr.as<float>(), h->soft.as<float>(), h->hard.as<uint8_t>(), h->dev, h->metric.as<float>(),
		h->work_counter.as<int>(), cert);
	launch_finish(h->stream, 8, (int)n, dst, h->llr.as<float>(), h->hard.as<uint8_t>(), h->dev, h->cfg.descramble, nullptr,
		h->payload.as<uint8_t>(), h->res.as<Result>(), cert);
	HIP_OK(hipGetLastError());
	HIP_OK(hipStreamSynchronize(h->stream));
	HIP_OK(hipMemcpy(payload, h->payload.p, n * PAYLOAD_BYTES, hipMemcpyDeviceToHost));
	HIP_OK(hipMemcpy(results, h->res.p, n * sizeof(Result), hipMemcpyDeviceToHost));
	if (cert_out)
		HIP_OK(hipMemcpy(cert_out, h->cert.p, n * sizeof(int), hipMemcpyDeviceToHost));
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
