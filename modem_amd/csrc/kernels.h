// kernels.h -- structs shared by the host API (api_*.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_common.h"

namespace rx {

// one decode call's view of the raw PCM batch (what DSP::ReadPCM delivers, decode.cc:297-298)
struct FrameBatch {
	const void *samples;           // device pointer, frame 0
	size_t frame_stride_bytes;
	long samples_per_frame;
	int fmt;                       // OFDMRX_FMT_*
	int channels;                  // 1 or 2
};

struct FrontCoef {                 // BlockDC::samples(2*(symbol_len+guard_len)) + Hilbert<cmplx,filter_len> (decode.cc:386,193)
	float dc_a, dc_b;
	float reco, imco[32];          // (filter_len-1)/4 odd-tap pairs: 5 / 10 / 28 / 31 at 8 / 16 / 44.1 / 48 kHz
};

// Mono input (round 4, mono_front.h): the DC blocker's one-pole low pass s[n] = a s[n-1] + g x[n] (y[n] = b x[n] - s[n-1]) with the
// powers of a its blocked scans use, computed once on the host in double, and the Hilbert taps.  ck: the state after every 64th
// sample of every frame (k_mono_carries).
struct MonoArgs {
	const double *ck;              // [n][ck_per_frame]
	int ck_per_frame;
	float a, g, b;
	float apw[8];                  // a^1 .. a^8
	float astep8[6], astep5[6];    // a^(8 2^k), a^(5 2^k): weights of the wave scans over thread chunks of 8 (MonoCover) / 5 (k_demod) samples
	double awave8, awave5;         // a^512, a^320: what a whole wave of such chunks decays by
	FrontCoef co;
};
inline MonoArgs mono_args(const FrontCoef &co, const double *ck, int ck_per_frame)
{
	MonoArgs m;
	m.ck = ck;
	m.ck_per_frame = ck_per_frame;
	const double a = (double)co.dc_a, b = (double)co.dc_b;
	m.a = (float)a; m.b = (float)b; m.g = (float)(b * (1.0 - a));
	double p = a;
	for (int i = 0; i < 8; ++i, p *= a)
		m.apw[i] = (float)p;
	double s8 = 1.0, s5 = 1.0;
	for (int i = 0; i < 8; ++i) s8 *= a;
	for (int i = 0; i < 5; ++i) s5 *= a;
	for (int k = 0; k < 6; ++k) {
		m.astep8[k] = (float)s8; m.astep5[k] = (float)s5;
		s8 *= s8; s5 *= s5;
	}
	m.awave8 = s8; m.awave5 = s5;
	m.co = co;
	return m;
}
inline int mono_ck_per_frame(long samples_per_frame) { return (int)((samples_per_frame + 63) / 64); }
// 8 kHz: the consumers form the analytic signal (launch_sync / _header / _demod with channels = 1).  The other rates (Hilbert filters of
// 41 - 125 taps, symbols of 2560 - 7680 samples) run launch_front_end over the whole stream first and read z like 2-channel input.
inline bool mono_fused(int rate) { return rate == 8000; }

struct SyncState {                 // per frame, across sync rounds (decode.cc:390-448 loop)
	long t_next;                   // next sample time to examine
	long sc_start;                 // stream index of the S&C body
	int active;                    // still searching in this round
	int found;                     // a preamble was accepted in THIS round.  Invariant across kernels: only launch_init_sync and k_header clear
	                               // it (k_header whenever it leaves active = 1 for another round, decode.cc:448), so the scan kernels of a
	                               // round may return early on it; a frame with active = 0 is never looked at again
	int symbol_pos;                // window coordinate, decode.cc:400
	float cfo_rad;
	int rejects;
	int skip_left;                 // decode.cc:448 while (skip_count--)
	int status;                    // OFDMRX_* so far
	int oper_mode;
	unsigned long long call_sign;
	int hdr_rounds;                // header attempts so far
	int okay;
	// a trigger found by the scanning kernel and not yet examined by the accept kernel (rates above 8 kHz, k_sync.hip)
	long pend_g;
	int pend_index_max;
	float pend_phase;
	int pending;
};

struct Tables {                    // device-resident constants, built once per handle
	const cf *tw_sym;              // e^{-j 2 pi m / symbol_len} (1280 at 8 kHz)
	const cf *sc_kern;             // conj(FFT(mls0))/(symbol_len/2), decode.cc:80-82
	const float *mls1_nrz;         // +-1 descrambler, decode.cc:407-409
	const float *mls0_nrz;         // [127] MLS 0b10001001 (transmitter: Schmidl-Cox symbol, encode.cc:144)
	const float *mls2_nrz;         // [512] MLS 0b100101010001 (transmitter: pilot block, encode.cc:134)
	const cf *tw_sym4;             // e^{-j 2 pi m / (4 symbol_len)} (transmitter PAPR step)
	const cf *tw_symc;             // compact per-stage twiddles of the symbol_len-point plan (transmitter: read through L1)
	const uint32_t *frozen;        // [2][2048] words, bit set = frozen: frozen_64800_43072, frozen_64512_43072 (regenerated)
	const uint16_t *info_pos;      // [2][44096] ascending unfrozen positions per table
	const uint32_t *info_compress; // [2][2048][8] per code word: compress move masks, unfrozen mask, first message bit | count << 16 (message_gather)
	const uint8_t *node_lev64;     // [2][1024] the same per 64-leaf block, for k_sc (frozen: 64 or 128 leaves, information: 64 .. 2048)
	const uint32_t *frozen_t;      // [2][16][64][2] frozen bits of a 4096-leaf sub-tree as lane `lane` of k_sc<6> holds it (bit x = leaf s * 4096 + x * 64 + position)
	const uint8_t *node_lev32;     // [2][2048] ... per 32-leaf block (frozen: 32 .. 128 leaves, information: 32 .. 1024)
	const uint8_t *node_lev;       // [2][8192] per 8-leaf group: level of the largest aligned all-frozen (low nibble) /
	                               // all-information (high nibble) node that starts there (frozen: <= 128 leaves, information: <= 2048), 0 = none
	const uint32_t *genmat_bits;   // BCH(255,71) systematic generator, [71][8] words, bit i of row j
	const uint8_t *osd_pairs;      // [2485][2] (a,b)
	const uint8_t *osd_triples;    // [57155][3] (a,b,c) sorted by c (equal d-loop lengths are adjacent)
	const uint32_t *crc32_tab;     // 256-entry byte table of CRC<uint32_t>(0xD419CC15)
	const uint32_t *crc32_shift168; // [4][256]: the CRC state advanced by 168 zero bytes, per state byte (k_finish)
	const uint32_t *crc32_adv;     // [256][32]: bit b of a state advanced by what follows segment k of the 5384 message bytes (crc32_wg256)
	const uint8_t *scramble;       // 5380 bytes of the Xorshift32 stream (decode.cc:613-615)
};

// ---- the list decoder's work queue (round 4).  Frames the syndrome certificate cannot finish are queued by k_back with their
// LLRs - entry e in slot e % cap of the slot arrays (LLRs, partial sums, metrics, ListSlot) - and k_polar / k_finish take
// whole runs of entries: a flush (api_pipeline.cpp) is k_queue_plan | k_polar | k_finish, and it takes nothing until one full
// residency of the list decoder waits, so a few stragglers per chunk no longer cost a decoder round each.  All counters live
// on the device: the host never waits to learn how many frames a chunk left.
struct ListQueue {
	unsigned tail;                 // entries allocated since the call began (k_back: atomicAdd)
	unsigned head;                 // entries taken by the flushes planned so far
	unsigned snap[2];              // tail behind k_back of the chunk with parity 0 / 1 (k_queue_snap): what that chunk's flush may take
	unsigned run_head[2], run_n[2];   // the entries of the flush of the chunk with parity 0 / 1 (k_queue_plan)
	int next_unit[2];              // k_polar's shared work counter, per flush
	int cert_on;                   // adaptive certificate: tried for every frame (1) or for a probe sample (0)
	unsigned tried, certified;     // since the last snapshot: frames the certificate was tried for / that it finished
	unsigned cap;                  // slots
	unsigned done_total;           // SC ring only: frames k_sc_finish has finished since the call began
	unsigned epoch;                // SC ring only: chunks since the call began (k_sc_adapt): the probe sample runs in every fourth
	unsigned probe_tried, probe_done;   // while cert_on = 0: the probe sample's counts, summed over chunks until eight frames have been tried
};
// The SC ring (k_sc.hip) is a second queue of the same shape in front of this one: k_back puts a frame there when the SC pass is
// on (its cert_on; tried / certified count the last run's entries / decided frames), k_sc_plan | k_sc | k_sc_finish run right behind
// k_back of every chunk on the same stream, and what they cannot decide moves on to the list decoder's queue.  Like a flush of the
// list decoder, a run takes WHOLE residencies of k_sc's persistent decoders (round 6: a codeword is one decoder's serial work of
// 1.2 ms, so a run of 8192 entries on the 2560 decoders of that time took four rounds for 3.2 rounds of work; 2048 decoders since the
// clean-node tests, k_sc.hip SC6_WAVES); what is left over waits
// for the next chunk's run, and the last run of a call takes everything.
struct ScStat { float metric, min_fork; int32_t ok, pad; };   // per SC-ring slot: P*'s metric, min_fork, rule holds
struct ListSlot {                  // what k_polar / k_finish need to know about a queued frame
	uint8_t *payload;              // where its 5380 bytes go when the list decoder's flush delivers them (k_finish)
	struct Result *res;            // its record (complete but for best_lane / bit_flips / a payload CRC failure)
	uint8_t *payload_now;          // the same in the chunk's own arrays: where k_sc_finish delivers, which runs right behind k_back -
	struct Result *res_now;        // before a staging buffer of the chunk leaves (launch_back: payload_later)
	int oper_mode;
	int frame;                     // index in its chunk
	int chunk;                     // which chunk of its call: a frame the list-1 pass finishes in a LATER chunk's run goes to payload / res
	int pad;
};
struct ScRing { ListQueue *q; ListSlot *slots; float *llr; };  // what k_back needs of the SC ring (q = nullptr: the pass is off)

struct Result {                    // device mirror of ofdmrx_frame_result (same layout)
	int32_t status;
	int32_t symbol_pos;
	int64_t sc_start;
	float cfo_rad, cfo_fine, sfo_slope;
	int32_t oper_mode;
	uint64_t call_sign;
	int32_t best_lane, bit_flips;
	float esn0_db_last;
	int32_t n_sync_rejects;
};

struct Attempt {                   // device mirror of ofdmrx_attempt: one preamble of a frame's SKIP loop (decode.cc:390-448)
	int32_t status, symbol_pos;
	float cfo_rad;
	int32_t oper_mode;
	uint64_t call_sign;
};
constexpr int ATTEMPTS_MAX = 65;   // OFDMRX_MAX_SKIP + 1

// ---- launch wrappers (defined next to their kernels) ------------------------
// `rate` selects the RateCfg instantiation (8000 / 16000 / 44100 / 48000)
// D1 (mono input, mono_front.h): ck = [n][mono_ck_per_frame()] states of the DC blocker; with mono_fused(rate) the consumers below
// form the analytic signal where they read it (ma.ck = ck) and z ([n][samples_per_frame]) is scratch the sync / header kernels
// write the windows they read into; launch_front_end fills all of z (the other rates; the ANALYTIC tap).  2-channel input: z, ma unused.
void launch_mono_carries(hipStream_t s, int rate, int n, FrameBatch fb, FrontCoef co, double *ck);
void launch_front_end(hipStream_t s, int rate, int n, FrameBatch fb, MonoArgs ma, cf *z);
void launch_sync(hipStream_t s, int rate, int n, FrameBatch fb, cf *z, Tables tb, SyncState *st, cf *scratch, const MonoArgs &ma);
// attempts / attempt_counts (nullable): [n][ATTEMPTS_MAX] records of the preambles examined so far, [n] their number
void launch_header(hipStream_t s, int rate, int n, FrameBatch fb, cf *z, const MonoArgs &ma, Tables tb, SyncState *st, int8_t *hdr_soft,
	Attempt *attempts = nullptr, int32_t *attempt_counts = nullptr);
void launch_osd_only(hipStream_t s, int n, Tables tb, const int8_t *soft, uint8_t *hard, int32_t *unique);
bool demod_forms_cons(int rate);   // cons is complete after k_demod (else k_theil_sen forms the rows from the carriers)
void launch_demod(hipStream_t s, int rate, int n, FrameBatch fb, cf *z, const MonoArgs &ma, Tables tb, const SyncState *st, cf *cons, cf *carr);
void launch_theil_sen(hipStream_t s, int n, const SyncState *st, cf *cons, const cf *carr, float *slope, float *yint, int *chunk_flags);
void launch_theil_sen_raw(hipStream_t s, int rows, int cols, const float *y, float *slope, float *yint);
// D5's rotation + D6-D8 + the syndrome certificate (k_finish.hip: k_back).  cert_mode 0: every frame with a header goes to the list
// decoder's queue; 1: the certificate is tried (adaptively) and finishes the frames it decides (payload + result).
// esn0_rows (nullable): [n][ROWS_MAX] dB values, decode.cc:517-519; slot_of: [n] queue slot per frame, -1 = none.
// payload_later / res_later (nullable: payload / res): where k_finish delivers the frames this chunk leaves to the queue - the
// chunk's own arrays may be a staging buffer that has been copied out and reused by the time their flush comes
void launch_back(hipStream_t s, int rate, int n, int cert_mode, const SyncState *st, const cf *cons, const float *slope, const float *yint,
	float *precision, Result *res, float *esn0_rows, Tables tb, int descramble, uint8_t *payload, ListQueue *q, ListSlot *slots,
	float *llr_q, int *slot_of, uint8_t *payload_later = nullptr, Result *res_later = nullptr, ScRing sc = ScRing{ nullptr, nullptr, nullptr }, int chunk_seq = 0);
void launch_rotate_tap(hipStream_t s, const SyncState *st, const cf *cons, const float *slope, const float *yint, cf *out);   // one frame, CONS_MAX points
void launch_queue_reset(hipStream_t s, ListQueue *q, unsigned cap);
void launch_queue_snap(hipStream_t s, ListQueue *q, int par);
void launch_queue_plan(hipStream_t s, ListQueue *q, int par, unsigned unit, int force);
void launch_queue_fill(hipStream_t s, ListQueue *q, ListSlot *slots, int n, uint8_t *payload, Result *res, int oper_mode);
// the sign-following path alone + its certificate (k_sc.hip): grid = resident decoders (sc_store_bytes() of level store each),
// lb = log2 of the lanes per codeword: 5 (two codewords per wave), 6 (one), 0: both launched, the run's length picks one on the device
// unit: a run takes whole multiples of it (0 / force: everything that waits)
void launch_sc_plan(hipStream_t s, ListQueue *qs, unsigned unit = 0, int force = 1);
void launch_sc(hipStream_t s, int lb, int grid5, int grid6, ListQueue *qs, const ListSlot *slots, const float *llr_q, float *soft, unsigned long long *cw_q,
	unsigned long long *xw_q, ScStat *stat_q, Tables tb, int top_skip = 1);
void launch_sc_finish(hipStream_t s, int max_entries, ListQueue *qs, const ListSlot *slots_s, const float *llr_s, const unsigned long long *cw_q,
	const unsigned long long *xw_q, const ScStat *stat_q, Tables tb, int descramble, ListQueue *ql, ListSlot *slots_l, float *llr_l, int *slot_of, int chunk_seq = 0);
void launch_sc_adapt(hipStream_t s, ListQueue *qs);
size_t sc_store_bytes(int lb);  // level store per resident decoder
int sc_codewords_per_wave(int lb);
// D9 / D10 for the run `par` of the queue; grid = resident decoders, max_entries = an upper bound of the run's length
void launch_polar(hipStream_t s, int list, int grid, ListQueue *q, int par, const ListSlot *slots, const float *llr_q, float *soft, uint8_t *hard_q,
	Tables tb, float *metric_q);
void launch_finish(hipStream_t s, int list, int max_entries, const ListQueue *q, int par, const ListSlot *slots, const float *llr_q,
	const uint8_t *hard_q, Tables tb, int descramble, uint8_t *lane_mesg_q);
void launch_fft_debug(hipStream_t s, int rate, int n, int len, int sign, const cf *in, cf *out, Tables tb);
void launch_awgn_tile(hipStream_t s, const int16_t *base, size_t n_base, int16_t *out, size_t n_out,
	size_t spf, float sigma, uint64_t seed, uint64_t first_frame);
void launch_channel(hipStream_t s, int rate, const int16_t *in, int16_t *out, size_t n, size_t spf, const void *params);
size_t tx_big_scratch_bytes(int rate, int n, int nsym);
void launch_tx(hipStream_t s, int rate, int n, const uint8_t *payload, Tables tb, const void *tp, const cf *tw_sym4,
	uint32_t *code, cf *rowsym, cf *tdom, cf *big_scratch, void *pcm);
// chunk_flags (nullable): per-chunk device flags cleared here ([0]: the largest row count k_theil_sen met, when above 50)
void launch_init_sync(hipStream_t s, int n, SyncState *st, const int32_t *skip_counts, int *chunk_flags = nullptr, int32_t *attempt_counts = nullptr);

}  // namespace rx
