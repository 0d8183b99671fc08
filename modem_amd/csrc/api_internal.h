// api_internal.h -- what the four files of the C ABI share: the handle, its buffers, the helpers that cross files.
// (round 6: ofdmrx_api.cpp was one file of 1700 lines - api_create.cpp / api_pipeline.cpp / api_debug.cpp / api_tx.cpp now)
#pragma once
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


extern thread_local std::string g_last_error;   // (api_create.cpp)

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
inline long long callsign_value(const char *str)
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
	int sc_top = 1;           // k_sc skips clean nodes of 16384 / 32768 leaves on a lower bound of their share of min_fork (OFDMRX_SC_TOP=0: every figure exact)
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

// ---- helpers that cross the files
int ensure_capacity(ofdmrx_handle *h, int n, bool mono, long samples);                    // api_create.cpp
int host_wait(ofdmrx_handle *h);                                                          // api_create.cpp
void run_sc_pass(ofdmrx_handle *h, hipStream_t s, int n, bool force = true, int chunk_seq = 0);   // api_pipeline.cpp
