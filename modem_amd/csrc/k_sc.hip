// k_sc.hip -- the list decoder's sign-following path, decoded alone (list size 1), and the certificate that it IS the list
// decoder's answer ("SC dominance") for gfx950.  decode.cc:530-555 needs lane 0's message, its CRC and the flip
// count - not the eight-path search - whenever lane 0 provably is that path.
//
// P* = the path of CODE::PolarListDecoder (decode.cc:201) that takes the sign of its LLR at every information leaf.  k_sc decodes
// P* with one lane's arithmetic of the list decoder - f_minsum, g_add, frozen penalties max(0, -llr) leaf by leaf, an aligned
// all-frozen node of 2..128 leaves in the butterfly order at once (k_polar.hip, oracle/polar.c: scl_node) - and carries
//   M*        P*'s path metric (fp32, same additions in the same order as lane 0's),
//   min_fork  min over the information leaves i of fl(M*(i) + |llr_i|): the metric of the candidate that leaves P* at leaf i.
// Rule: min_fork > M*(final)  =>  P* is lane 0 of the list decoder, for any list size.  Every candidate that is not P* either
// descends from a first deviation at some leaf i - it then carries at least fl(M*(i) + |llr_i|) for ever, fp32 sums of non-negative
// penalties being monotone - or from one of the placeholder paths (metric 1000 at the start) making P*'s own decisions - never
// cheaper than P* (monotone again), and P* wins ties through its candidate index (lane 0).  So P* has the smallest metric, ties
// broken its way, at every fork: never pruned, rank 0 throughout, lane 0 at the end with metric M*.  The syndrome certificate of
// k_back is the case M* = 0.  k_sc_finish then does decode.cc:532-555 for lane 0: CRC-32 of P*'s systematic bits, payload, flip
// count; a frame whose rule or CRC fails goes on to the list decoder's queue unchanged (its LLRs are copied there).
// Checked against the oracle's list decoder by tests/test_oracle_kat.py (the rule, CPU) and tests/test_gpu_parity.py (this kernel).
//
// One wavefront per codeword, persistent.  Where the tree lives: the input arrays of the nodes of 2^15, 2^14, 2^13 leaves in a
// level store in HBM (224 KB per resident decoder; each is written once and read once), of the current 4096-leaf node in LDS (16 KB),
// of the current nodes of 2048..128 leaves in registers (position x * 64 + q of such an array = element x of the lane that holds
// q: every f / g step down to 128 leaves is lane-local), the 64-leaf block below in one register per lane with DPP / permlane
// butterflies.  Partial sums are bits: one 64-bit register per lane for the current 4096-leaf node (bit x = position x * 64 + q),
// published as plain bit-packed words (bit i of the codeword = bit i % 64 of word i / 64) that the upper g steps read back.
// Uniform nodes are decided in one step: all-frozen ones of up to 128 leaves (the penalty sum above), all-information ones of any
// size (the SC decisions of such a node are the signs of its input LLRs; the smallest leaf magnitude on P* is the smallest input
// magnitude, so min_fork takes fl(M* + min |input|)).
// Round 6: CLEAN nodes - any node, of 64 leaves up to half the code, whose input hard decisions are a codeword of its sub-code - are not
// walked either (sc_clean, sc_clean_fork, sc_top_pass below: the proof, what min_fork takes from them, and how the halves and quarters
// are found while the pass that makes them goes by).  A frame with raw bit errors is walked only along the few paths that lead to them:
// 145 k instructions per codeword in place of 323 k at -20 dB (433 raw errors in a frame), a fifth of that on the configs[3] chain (one
// or two), and the level-store traffic of the skipped halves and quarters is never made.  What the store does hold is half arrays where the
// other half can be had from the f array (sc_emit): 1.64 MB per codeword at -20 dB in place of 2.17.
#include "dev_common.h"
#include "kernels.h"
#include "polar_common.h"

namespace rx {

#define SC_WAVE_ORDER() __builtin_amdgcn_wave_barrier()
#ifndef SC_LOADS
#define SC_LOADS 64               // level-store loads a lane keeps in flight in the top passes.  -20 dB, 65 536 frames, one codeword per wave at ten
                                  // decoders per CU: 16 loads 777 k frames/s, 32 807 - 812 k, 64 817 - 821 k (11 registers spilled at 168); with 256
                                  // VGPRs and eight decoders (round 6): 32 and 64 the same, 128: 36 against 26 ms per 65 536 frames
#endif

// ---- how a wave is cut: LB = log2 of the lanes that work on one codeword (6: one codeword per wave, 5: two).  The decoder has
// NO data-dependent control flow - what it does at every node follows from the frozen table alone - so codewords of the same
// table run in lock-step in one wave, and whatever a 2^LB-leaf block costs is shared by 64 >> LB codewords.
template <int LB> struct ScCfg {
	static constexpr int J = 1 << LB, C = 64 >> LB;
	static constexpr int LL = LB + 6;                         // the level whose array lives in LDS: 64 elements per lane
	static constexpr int NSUB = 1 << (16 - LL);               // sub-trees of that size per codeword
	static constexpr int SUB_BYTES = 64 * J * 4;              // one of them in a level array
	static constexpr int STORE_FLOATS = 65536 - (2 << LL);    // levels LL+1 .. 15 of one codeword
	static constexpr int SPARE_WORDS = 2 * (CODE_LEN / 32);   // behind a decoder's level stores: where the second lane group of an unpaired
	                                                          // codeword publishes its (identical) words instead of the codeword's own
	static constexpr int DECODER_FLOATS = C * STORE_FLOATS + SPARE_WORDS;
};
__host__ __device__ constexpr int sc_off(int m) { return (65536 - (2 << m)) * 4; }   // level m <= 15 in a codeword's level store, bytes

// ---- positions and lanes.  Lane l holds position q = j ^ ((j & 4) ? 3 : 0), j = l mod 2^LB, of its codeword's block: then "the
// lane whose position differs in bit 2" is row_half_mirror (l ^ 7), and every butterfly exchange is one DPP / permlane instruction.
__device__ __forceinline__ int sc_pos(int j) { return j ^ ((j & 4) ? 3 : 0); }
template <int H> __device__ __forceinline__ int xpos_i(int v, int lane)
{
	if constexpr (H == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);           // quad_perm [1,0,3,2]
	else if constexpr (H == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
	else if constexpr (H == 4) return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);     // row_half_mirror
	else if constexpr (H == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);     // row_ror:8
	else if constexpr (H == 16) {
		auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
		return (int)((lane & 16) ? r[0] : r[1]);
	} else {
		auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
		return (int)((lane & 32) ? r[0] : r[1]);
	}
}
template <int H> __device__ __forceinline__ float xpos(float v, int lane) { return __int_as_float(xpos_i<H>(__float_as_int(v), lane)); }
template <int H> __device__ __forceinline__ uint32_t xpos(uint32_t v, int lane) { return (uint32_t)xpos_i<H>((int)v, lane); }

constexpr uint32_t SC_SIGN = 0x80000000u;
// what a lane knows about itself: lowsel[L] = the sign bit if its position lies in the LOW half of its level-L node, else 0
struct ScLane {
	int lane, q;
	uint32_t lowsel[7];
};
// the decoder's running figures: M* and min_fork (as its bit pattern: non-negative floats order like unsigned integers)
struct ScAcc {
	float M;
	uint32_t fork;
	__device__ __forceinline__ void info(uint32_t mu_bits)        // an information leaf / all-information node with smallest magnitude mu
	{
		const float cand = M + __uint_as_float(mu_bits);
		fork = min(fork, __float_as_uint(cand));
	}
};

// sum over the 2^LV values of a node held one per lane (duplicated over the other position bits) in the butterfly halving
// order p[i] += p[i + h], h = n/2 .. 1 (oracle/polar.c: scl_node's rate-0 step): both partners form the same sum
template <int LV> __device__ __forceinline__ float sc_pen_sum(float pen, int lane)
{
	if constexpr (LV >= 6) pen = pen + xpos<32>(pen, lane);
	if constexpr (LV >= 5) pen = pen + xpos<16>(pen, lane);
	if constexpr (LV >= 4) pen = pen + xpos<8>(pen, lane);
	if constexpr (LV >= 3) pen = pen + xpos<4>(pen, lane);
	if constexpr (LV >= 2) pen = pen + xpos<2>(pen, lane);
	if constexpr (LV >= 1) pen = pen + xpos<1>(pen, lane);
	return pen;
}
template <int LV> __device__ __forceinline__ uint32_t sc_min_mag(uint32_t mu, int lane)
{
	if constexpr (LV >= 6) mu = min(mu, xpos<32>(mu, lane));
	if constexpr (LV >= 5) mu = min(mu, xpos<16>(mu, lane));
	if constexpr (LV >= 4) mu = min(mu, xpos<8>(mu, lane));
	if constexpr (LV >= 3) mu = min(mu, xpos<4>(mu, lane));
	if constexpr (LV >= 2) mu = min(mu, xpos<2>(mu, lane));
	if constexpr (LV >= 1) mu = min(mu, xpos<1>(mu, lane));
	return mu;
}
// max(0, -llr): what a frozen leaf adds when its LLR is negative (oracle/polar.c: scl_leaf; adding +0 changes nothing)
__device__ __forceinline__ float sc_pen(float v) { return __builtin_fmaxf(-v, 0.f); }
__device__ __forceinline__ uint32_t sc_mag(float v) { return __float_as_uint(v) & 0x7fffffffu; }

// Partial sums inside a block are kept in SIGN-BIT FORM, one register per level: beta = 0x80000000 if the partial sum of the
// level's current node AT THE LANE'S OWN POSITION (mod the node size) is 1, else 0.  Then
//   g step     t = beta_left & lowsel[L];  r' = r ^ t;  child = r' + partner(r')      (b + a or b - a: the low half carries a and
//                                                                                     flips its sign, the sum is the same on both sides)
//   combine    beta_L = beta_right ^ t                                                (low half: left ^ right, high half: right)
//   leaf / all-information node   beta = the sign bit of the LLR;  frozen: 0.
// (A zero LLR at an information leaf carries either sign bit; min_fork then equals M* and the rule cannot hold, see ScAcc.)
template <int LV> __device__ __forceinline__ float sc_f_cross(float r, const ScLane &L) { return f_minsum(r, xpos<(1 << (LV - 1))>(r, L.lane)); }
template <int LV> __device__ __forceinline__ float sc_g_cross(float r, uint32_t t, const ScLane &L)
{
	const float rp = __uint_as_float(__float_as_uint(r) ^ t);
	return rp + xpos<(1 << (LV - 1))>(rp, L.lane);
}

// ---- a sub-tree of 2^LV <= 16 leaves whose frozen pattern FZ is known at compile time: straight-line code, every uniform node
// decided in one step (all frozen: the penalty sum; all information: signs + smallest magnitude), nothing scalar left
template <uint32_t FZ, int LV, int P0> __device__ __forceinline__ uint32_t sc_node_ct(float r, ScAcc &acc, const ScLane &L)
{
	constexpr int N = 1 << LV;
	constexpr uint32_t MASK = (N == 32 ? 0xffffffffu : ((1u << N) - 1u)) << P0, pat = FZ & MASK;
	if constexpr (pat == MASK) {
		acc.M += sc_pen_sum<LV>(sc_pen(r), L.lane);
		return 0u;
	} else if constexpr (pat == 0u) {
		acc.info(sc_min_mag<LV>(sc_mag(r), L.lane));
		return __float_as_uint(r) & SC_SIGN;
	} else {
		const uint32_t bl = sc_node_ct<FZ, LV - 1, P0>(sc_f_cross<LV>(r, L), acc, L);
		const uint32_t t = bl & L.lowsel[LV];
		const uint32_t br = sc_node_ct<FZ, LV - 1, P0 + N / 2>(sc_g_cross<LV>(r, t, L), acc, L);
		return br ^ t;
	}
}
// the same for a pattern only known at run time (wave-uniform): the general walk
template <int LV, int P0> __device__ __forceinline__ uint32_t sc_node_rt(float r, uint32_t fz, ScAcc &acc, const ScLane &L)
{
	constexpr int N = 1 << LV;
	constexpr uint32_t MASK = ((1u << N) - 1u) << P0;
	const uint32_t pat = fz & MASK;
	if (pat == MASK) {
		acc.M += sc_pen_sum<LV>(sc_pen(r), L.lane);
		return 0u;
	}
	if (pat == 0u) {
		acc.info(sc_min_mag<LV>(sc_mag(r), L.lane));
		return __float_as_uint(r) & SC_SIGN;
	}
	if constexpr (LV > 0) {
		const uint32_t bl = sc_node_rt<LV - 1, P0>(sc_f_cross<LV>(r, L), fz, acc, L);
		const uint32_t t = bl & L.lowsel[LV];
		const uint32_t br = sc_node_rt<LV - 1, P0 + N / 2>(sc_g_cross<LV>(r, t, L), fz, acc, L);
		return br ^ t;
	} else
		return 0u;                                                // (a leaf is always uniform)
}
// The mixed 16-leaf patterns of both frozen tables (frozen_64800_43072, frozen_64512_43072: fifteen, the same in both; leaf 0 =
// bit 0) as straight-line code; anything else takes the general walk.  tests/test_abi_cpu.py checks the list against the tables.
#define SC_PATTERNS16(X) \
	X(0x0001u) X(0x0117u) X(0x177fu) X(0x7fffu) X(0x17ffu) X(0x011fu) X(0x0017u) X(0x0003u) \
	X(0x017fu) X(0x037fu) X(0x0007u) X(0x077fu) X(0x3fffu) X(0x1fffu) X(0x013fu)
__device__ __forceinline__ uint32_t sc_block16(float r4, uint32_t pat, ScAcc &acc, const ScLane &L)
{
	switch (pat) {
#define SC_CASE16(P) case P: return sc_node_ct<P, 4, 0>(r4, acc, L);
	SC_PATTERNS16(SC_CASE16)
#undef SC_CASE16
	default: return sc_node_rt<4, 0>(r4, pat, acc, L);
	}
}

// ---- the block of 2^LB leaves one lane group holds: its 16-leaf quarters / halves in turn.  fz0 / fz1: frozen bits of its leaves
// 0..31 / 32..63.  Returns beta of the whole block (the partial sum of the lane's own position, sign-bit form).
template <int LB> __device__ __forceinline__ uint32_t sc_walk_block(float rb, uint32_t fz0, uint32_t fz1, ScAcc &acc, const ScLane &L)
{
	static_assert(LB == 5 || LB == 6, "blocks of 32 or 64 leaves");
	uint32_t beta6 = 0;
	float r5 = rb;
	uint32_t t6 = 0;
	#pragma unroll
	for (int h = 0; h < (LB == 6 ? 2 : 1); ++h) {                 // the 32-leaf halves
		const uint32_t fzw = h ? fz1 : fz0;
		if constexpr (LB == 6) {
			if (h == 0)
				r5 = sc_f_cross<6>(rb, L);
			else {
				t6 = beta6 & L.lowsel[6];                         // (beta6 holds the left half's beta5 here)
				r5 = sc_g_cross<6>(rb, t6, L);
			}
		}
		uint32_t beta5;
		if (fzw == 0xffffffffu) {
			acc.M += sc_pen_sum<5>(sc_pen(r5), L.lane);
			beta5 = 0u;
		} else if (fzw == 0u) {
			acc.info(sc_min_mag<5>(sc_mag(r5), L.lane));
			beta5 = __float_as_uint(r5) & SC_SIGN;
		} else {
			const uint32_t bl = sc_block16(sc_f_cross<5>(r5, L), fzw & 0xffffu, acc, L);
			const uint32_t t5 = bl & L.lowsel[5];
			const uint32_t br = sc_block16(sc_g_cross<5>(r5, t5, L), fzw >> 16, acc, L);
			beta5 = br ^ t5;
		}
		beta6 = (LB == 6 && h == 1) ? (beta5 ^ t6) : beta5;
	}
	return beta6;
}

// One pass over the top of the tree: the array of the sub-tree s (64 * 2^LB leaves per codeword) from the level D above it.
//   KIND 0: f chain from the channel LLRs (s = 0)   KIND 1: g of the channel LLRs, then f (s = NSUB / 2)
//   KIND 2: g of level LL + D of the level store, then f
// A lane owns the columns x * J + j (x = 0..63) of its codeword at EVERY level: the chain below the first step is lane-local,
// each level between is written once (levels > LL to the store, level LL to LDS).  g takes the left child's partial sums from
// the published words (pa / pb: the two 32-bit halves a wave publishes per 64 lanes - the halves of a 64-bit word of one
// codeword, or one word of each of two): lane l fetches word l of each sub-tree once, the loop picks word x with v_readlane.
// KIND 0 also leaves the hard decisions of the channel LLRs (xa / xb, bit-packed like the codeword) for the flip count, and
// checks that every LLR is finite and small enough that no sum of 65536 of them overflows.
template <int LB> struct ScIo {                                   // where the wave's 32-bit half-words live: half h, word w
	uint32_t *cwa, *cwb, *xwa, *xwb;
	__device__ __forceinline__ static int idx(int w) { return LB == 6 ? 2 * w : w; }
};
// The array of a node of 2 * 64 * J leaves (level LL + 1) is stored as HALF an array (round 6).  Its pair (a, b) at a column is needed once
// more, by the g step of the sub-tree that follows - and by then the f of the pair still sits in LDS (the walk only reads its array).
// f = sign(a) sign(b) min(|a|, |b|) holds the smaller magnitude and the product of the signs: stored is only d = the element of the larger
// magnitude (a on a tie) and one bit per column saying which of the two it was (`which`, a lane's 64 columns in one word, stored behind d);
// the other element is |f| with the sign bit of f ^ d - the pair comes back bit for bit (sc_pair_back).  16 KB less written and 16 KB less read
// per pair of sub-trees: 256 KB of the 2.05 MB a codeword's full walk moves.
#ifndef SC_HALF13
#define SC_HALF13 1
#endif
// The same for the two nodes of 32768 leaves (level 15, one codeword per wave): the f of their pairs is the array of the 16384-leaf node below,
// stored in full and still in place when the g step of the other 16384-leaf node reads the pairs (that pass then writes its own level-14
// array over it, column by column behind its own reads).  64 KB less written per half of the tree; the level between stays whole - a half
// array needs the f array whole.
struct ScWhich { unsigned long long w13, w15[4]; };
template <int LB, int LEV, int N> __device__ __forceinline__ void sc_emit(rsrc_t soft, float *lds, int v_dst, int lidx, float (&t)[N], ScWhich &which, int x)
{
	if constexpr (LEV > ScCfg<LB>::LL) {
		if constexpr (SC_HALF13 && LEV == ScCfg<LB>::LL + 1) {
			static_assert(N == 2, "the level above the LDS array holds one pair per column");
			const bool hi = sc_mag(t[1]) > sc_mag(t[0]);
			bstore<2>(soft, v_dst, sc_off(LEV), hi ? t[1] : t[0]);
			which.w13 |= (unsigned long long)(hi ? 1u : 0u) << x;
		} else if constexpr (SC_HALF13 && LB == 6 && LEV == 15) {
			static_assert(N == 8, "the halves of the tree: eight sub-trees each");
			#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const bool hi = sc_mag(t[k + 4]) > sc_mag(t[k]);
				bstore<2>(soft, v_dst, sc_off(LEV) + k * ScCfg<LB>::SUB_BYTES, hi ? t[k + 4] : t[k]);
				which.w15[k] |= (unsigned long long)(hi ? 1u : 0u) << x;
			}
		} else {
			#pragma unroll
			for (int k = 0; k < N; ++k)
				bstore<2>(soft, v_dst, sc_off(LEV) + k * ScCfg<LB>::SUB_BYTES, t[k]);
		}
		float u[N / 2];
		#pragma unroll
		for (int k = 0; k < N / 2; ++k)
			u[k] = f_minsum(t[k], t[k + N / 2]);
		sc_emit<LB, LEV - 1, N / 2>(soft, lds, v_dst, lidx, u, which, x);
	} else
		lds[lidx] = t[0];
}
// the pair (a, b) of sc_emit's half array: d = the larger element, hi = it was b, f = f_minsum(a, b)
__device__ __forceinline__ void sc_pair_back(float d, float f, bool hi, float &a, float &b)
{
	const float small = __uint_as_float(__float_as_uint(f) ^ (__float_as_uint(d) & SC_SIGN));
	a = hi ? small : d;
	b = hi ? d : small;
}
// (one codeword per wave) the pass's first array - the node of NH sub-trees that starts at sub-tree s - is tested for being CLEAN
// (see sc_clean below) while it goes by, when skipping it saves level-store traffic (NH >= 4): a clean one leaves its hard decisions
// in LDS (word k * 64 + lane of the stash = this lane's 64 elements of sub-tree s + k) and the smallest magnitude of its array.
#ifndef SC_CLEAN_TOP
#define SC_CLEAN_TOP 1
#endif
struct ScTop {
	const uint32_t *ft;       // frozen_t of the codeword's table
	bool allow;               // the caller skips a clean node (else nothing is tested: LDS keeps the first sub-tree's array)
	int nskip;                // out: sub-trees covered by the clean node (0: none)
	uint32_t mu;              // out: smallest magnitude of its array
};
// EMIT false: the pass only looks (signs, smallest magnitude, and the channel's hard decisions where it makes them) - nothing goes to the level
// store or to LDS.  A decoder runs the passes that can end in a skip that way when the same pass of its previous codeword found a clean node
// (the store traffic of a skipped node's first pass is for nothing: 224 KB at the halves, 96 KB at the quarters), and runs them again, storing,
// when this time the node is not clean.
template <int LB, int D, int KIND, bool EMIT = true>
__device__ __forceinline__ void sc_top_pass(rsrc_t soft, rsrc_t llr, float *lds, const ScIo<LB> &io, int s, int lane, int v_llr0, int v_soft0, bool &finite,
	ScTop &top, const uint32_t (&lowsel)[7])
{
	using Cf = ScCfg<LB>;
	constexpr bool TOPCHK = LB == 6 && D >= 3 && SC_CLEAN_TOP;
	unsigned long long Hs[TOPCHK ? (1 << (D - 1)) : 1] = {};
	uint32_t mu_top = 0x7f800000u;
	constexpr int NS = 1 << D, NH = NS / 2, XB = NS >= SC_LOADS ? 1 : SC_LOADS / NS, XSTEP = Cf::J * 4;   // SC_LOADS loads in flight per lane
	const rsrc_t src = KIND == 2 ? soft : llr;
	constexpr int src_off = KIND == 2 ? sc_off(Cf::LL + D) : 0;
	uint32_t wa[NH], wb[NH];
	if (KIND) {
		#pragma unroll
		for (int k = 0; k < NH; ++k) {
			const int w = ScIo<LB>::idx((s - NH + k) * 64 + lane);
			wa[k] = io.cwa[w];
			wb[k] = io.cwb[w];
		}
	}
	const int sh = 31 - (lane & 31);
	const bool up = lane >= 32;
	// (sc_emit) the which-word of a lane's 64 columns: behind the half array, at the lane's own 8 bytes
	constexpr bool HALF_IN = SC_HALF13 && KIND == 2 && D == 1, HALF_OUT = SC_HALF13 && EMIT && D >= 2;
	constexpr bool HALF_IN15 = SC_HALF13 && LB == 6 && KIND == 2 && Cf::LL + D == 15, HALF_OUT15 = SC_HALF13 && LB == 6 && EMIT && Cf::LL + D == 16;
	const int v_which = v_soft0 + (lane & (Cf::J - 1)) * 4;
	ScWhich which{ 0ull, { 0ull, 0ull, 0ull, 0ull } };
	auto which_at = [&](int off) {
		return (unsigned long long)__float_as_uint(bload<2>(soft, v_which, off)) | ((unsigned long long)__float_as_uint(bload<2>(soft, v_which, off + 4)) << 32);
	};
	auto which_to = [&](int off, unsigned long long w) {
		bstore<2>(soft, v_which, off, __uint_as_float((uint32_t)w));
		bstore<2>(soft, v_which, off + 4, __uint_as_float((uint32_t)(w >> 32)));
	};
	if constexpr (HALF_IN)
		which.w13 = which_at(sc_off(Cf::LL + 1) + Cf::SUB_BYTES);
	if constexpr (HALF_IN15) {
		#pragma unroll
		for (int k = 0; k < 4; ++k)
			which.w15[k] = which_at(sc_off(15) + 4 * Cf::SUB_BYTES + k * 512);
	}
	int v_src = KIND == 2 ? v_soft0 : v_llr0, v_dst = v_soft0, lidx = (lane >> LB) * (64 * Cf::J) + (lane & (Cf::J - 1));
	#pragma unroll 1
	for (int x0 = 0; x0 < 64; x0 += XB, v_src += XB * XSTEP, v_dst += XB * XSTEP, lidx += XB * Cf::J) {
		float v[XB][NS];
		if constexpr (HALF_IN) {
			float dd[XB];
			#pragma unroll
			for (int xb = 0; xb < XB; ++xb)
				dd[xb] = bload<2>(src, v_src + xb * XSTEP, src_off);
			#pragma unroll
			for (int xb = 0; xb < XB; ++xb)
				sc_pair_back(dd[xb], lds[lidx + xb * Cf::J], ((which.w13 >> (x0 + xb)) & 1ull) != 0ull, v[xb][0], v[xb][1]);
		} else if constexpr (HALF_IN15) {
			float dd[XB][4], ff[XB][4];
			#pragma unroll
			for (int xb = 0; xb < XB; ++xb)
				#pragma unroll
				for (int k = 0; k < 4; ++k) {
					dd[xb][k] = bload<2>(src, v_src + xb * XSTEP, sc_off(15) + k * Cf::SUB_BYTES);
					ff[xb][k] = bload<2>(src, v_src + xb * XSTEP, sc_off(14) + k * Cf::SUB_BYTES);
				}
			#pragma unroll
			for (int xb = 0; xb < XB; ++xb)
				#pragma unroll
				for (int k = 0; k < 4; ++k)
					sc_pair_back(dd[xb][k], ff[xb][k], ((which.w15[k] >> (x0 + xb)) & 1ull) != 0ull, v[xb][k], v[xb][k + 4]);
		} else {
		#pragma unroll
		for (int xb = 0; xb < XB; ++xb)
			#pragma unroll
			for (int k = 0; k < NS; ++k)
				v[xb][k] = KIND == 0 ? bload<0>(src, v_src + xb * XSTEP, src_off + k * Cf::SUB_BYTES) : bload<2>(src, v_src + xb * XSTEP, src_off + k * Cf::SUB_BYTES);
		}
		#pragma unroll
		for (int xb = 0; xb < XB; ++xb) {
			const int x = x0 + xb;
			float t[NH];
			#pragma unroll
			for (int k = 0; k < NH; ++k) {
				if (KIND == 0)
					t[k] = f_minsum(v[xb][k], v[xb][k + NH]);
				else {
					const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)wa[k], x), b = (uint32_t)__builtin_amdgcn_readlane((int)wb[k], x);
					const uint32_t sg = ((up ? b : a) << sh) & SC_SIGN;
					t[k] = __uint_as_float(__float_as_uint(v[xb][k]) ^ sg) + v[xb][k + NH];
				}
			}
			if constexpr (TOPCHK) {
				#pragma unroll
				for (int k = 0; k < NH; ++k) {
					const uint32_t tb = __float_as_uint(t[k]);
					mu_top = min(mu_top, tb & 0x7fffffffu);
					Hs[k] |= (unsigned long long)(tb >> 31) << x;
				}
			}
			if (KIND == 0) {
				// the hard decisions of this column's NS elements: word x of sub-tree k, gathered on lane k and stored from there
				int ma = 0, mb = 0;
				#pragma unroll
				for (int k = 0; k < NS; ++k) {
					const unsigned long long bal = __ballot(v[xb][k] < 0.f);
					if (lane == k) {
						ma = (int)(uint32_t)bal;
						mb = (int)(uint32_t)(bal >> 32);
					}
					finite &= sc_mag(v[xb][k]) < 0x71000000u;         // |llr| < 6e29 (and not a NaN)
				}
				if (lane < NS) {
					const int w = ScIo<LB>::idx(lane * 64 + x);
					io.xwa[w] = (uint32_t)ma;
					io.xwb[w] = (uint32_t)mb;
				}
			}
			if constexpr (EMIT)
				sc_emit<LB, Cf::LL + D - 1, NH>(soft, lds, v_dst + xb * XSTEP, lidx + xb * Cf::J, t, which, x);
		}
	}
	if constexpr (HALF_OUT)
		which_to(sc_off(Cf::LL + 1) + Cf::SUB_BYTES, which.w13);
	if constexpr (HALF_OUT15) {
		#pragma unroll
		for (int k = 0; k < 4; ++k)
			which_to(sc_off(15) + 4 * Cf::SUB_BYTES + k * 512, which.w15[k]);
	}
	if (TOPCHK && top.allow) {
		// u = h F over the node's NH * 4096 positions (sub-tree, element, lane position); zero wherever a leaf is frozen?
		// (the pass holds position `lane` of every block, the walk and its tables position sc_pos(lane): quad_perm [3,2,1,0] where lane & 4)
		unsigned long long U[NH];
		#pragma unroll
		for (int k = 0; k < NH; ++k) {
			const uint32_t a0 = (uint32_t)Hs[k], a1 = (uint32_t)(Hs[k] >> 32);
			const uint32_t s0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a0, 0x1B, 0xf, 0xf, false);
			const uint32_t s1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a1, 0x1B, 0xf, 0xf, false);
			if (lane & 4)
				Hs[k] = (unsigned long long)s0 | ((unsigned long long)s1 << 32);
			U[k] = Hs[k];
		}
		#pragma unroll
		for (int tt = NH / 2; tt >= 1; tt >>= 1)
			#pragma unroll
			for (int k = 0; k < NH; ++k)
				if (!(k & tt))
					U[k] ^= U[k + tt];
		bool bad = mu_top == 0u;
		#pragma unroll
		for (int k = 0; k < NH; ++k) {
			unsigned long long h = U[k];
			h ^= (h >> 32) & 0x00000000ffffffffull;
			h ^= (h >> 16) & 0x0000ffff0000ffffull;
			h ^= (h >> 8) & 0x00ff00ff00ff00ffull;
			h ^= (h >> 4) & 0x0f0f0f0f0f0f0f0full;
			h ^= (h >> 2) & 0x3333333333333333ull;
			h ^= (h >> 1) & 0x5555555555555555ull;
			uint32_t h0 = (uint32_t)h, h1 = (uint32_t)(h >> 32);
			#define SC_XL(H, LV) { \
				const uint32_t m = (uint32_t)((int)lowsel[LV] >> 31); \
				h0 ^= xpos<H>(h0, lane) & m; \
				h1 ^= xpos<H>(h1, lane) & m; }
			SC_XL(32, 6) SC_XL(16, 5) SC_XL(8, 4) SC_XL(4, 3) SC_XL(2, 2) SC_XL(1, 1)
			#undef SC_XL
			const uint2 ft = ((const uint2 *)top.ft)[(s + k) * 64 + lane];
			bad |= ((h0 & ft.x) | (h1 & ft.y)) != 0u;
		}
		if (__ballot(bad) == 0ull) {
			#pragma unroll
			for (int k = 0; k < NH; ++k)
				((unsigned long long *)lds)[k * 64 + lane] = Hs[k];
			top.nskip = NH;
			top.mu = mu_top;
		}
	}
}

// a uniform node of J * CNT leaves on its register array (element x = position x * J + q)
template <int LB, int CNT> __device__ __forceinline__ void sc_rate0(const float (&r)[CNT], ScAcc &acc, int lane)
{
	float pz[CNT];
	#pragma unroll
	for (int x = 0; x < CNT; ++x)
		pz[x] = sc_pen(r[x]);
	#pragma unroll
	for (int hx = CNT / 2; hx >= 1; hx >>= 1)
		#pragma unroll
		for (int x = 0; x < hx; ++x)
			pz[x] = pz[x] + pz[x + hx];
	acc.M += sc_pen_sum<LB>(pz[0], lane);
}
template <int LB, int CNT> __device__ __forceinline__ uint32_t sc_rate1(const float (&r)[CNT], ScAcc &acc, int lane)
{
	uint32_t mu = 0x7f800000u, bits = 0;
	#pragma unroll
	for (int x = 0; x < CNT; ++x) {
		mu = min(mu, sc_mag(r[x]));
		bits |= (__float_as_uint(r[x]) >> 31) << x;
	}
	acc.info(sc_min_mag<LB>(mu, lane));
	return bits;
}
template <int CNT> __device__ __forceinline__ void sc_f_half(float (&dst)[CNT], const float (&src)[2 * CNT])
{
	#pragma unroll
	for (int x = 0; x < CNT; ++x)
		dst[x] = f_minsum(src[x], src[x + CNT]);
}
template <int CNT> __device__ __forceinline__ void sc_g_half(float (&dst)[CNT], const float (&src)[2 * CNT], uint32_t hb)
{
	#pragma unroll
	for (int x = 0; x < CNT; ++x)
		dst[x] = __uint_as_float(__float_as_uint(src[x]) ^ ((hb << (31 - x)) & SC_SIGN)) + src[x + CNT];
}

// ---- CLEAN nodes (round 6; one codeword per wave only).  A node is clean when the hard decisions h of its input LLRs are a codeword of its
// sub-code (h F vanishes on the node's frozen leaves) and no input is zero.  Then successive cancellation returns h and no frozen leaf of
// the node adds a penalty, without the node being walked: by induction over the two halves (a, b) of the array - the left child's input
// f(a, b) has the hard decisions ha ^ hb, which are its part of the codeword, so it is clean and returns them; the right child's input
// is then b + a where ha = hb and b - a where they differ: a sum of two numbers of b's sign, hard decisions hb, magnitudes |a| + |b|, clean
// again - down to leaves whose hard decision is the message bit and, where frozen, zero (a positive LLR: no penalty; an all-frozen node
// below sees positive inputs only: penalty +0).  No magnitude on the way is smaller than the smallest input, so no zero turns up.
// What min_fork needs from such a node is the smallest leaf magnitude over its information leaves: magnitudes obey |f| = min(|a|, |b|),
// |g| = fl(|a| + |b|) (an fp32 sum of two numbers of one sign is the sum of their magnitudes), the partial sums no longer matter, and the
// whole sub-tree of magnitudes is evaluated level by level, every lane busy at every level (sc_clean_fork) - the same fp32 values the walk
// would have met, so M*, min_fork and the rule's outcome are the walk's, bit for bit.  At -20 dB 94 % of the leaves sit under a clean
// node of 64 .. 4096 leaves, in configs[3] 99.9 % (tools/experiments/sc_clean_nodes.py).
// The node's array: element x of the lane at position q = position x * 64 + q, CNT elements per lane; fz: bit x = that position frozen.
#ifndef SC_CLEAN_MIN
#define SC_CLEAN_MIN 0            // the smallest node tested: 64 << SC_CLEAN_MIN leaves (7: none)
#endif
template <int CNT> struct ScWord { typedef uint32_t T; };
template <> struct ScWord<64> { typedef unsigned long long T; };
template <int CNT> __device__ __forceinline__ bool sc_clean(const float (&r)[CNT], unsigned long long fz, const ScLane &L, typename ScWord<CNT>::T &hard)
{
	typedef typename ScWord<CNT>::T W;
	uint32_t mu = 0x7f800000u;
	W h = 0;
	#pragma unroll
	for (int x = 0; x < CNT; ++x) {
		mu = min(mu, sc_mag(r[x]));
		h |= (W)(__float_as_uint(r[x]) >> 31) << x;
	}
	hard = h;
	// u = h F: the element levels inside the lane's word, the position levels across the lanes
	if constexpr (CNT > 32) h ^= (h >> 32) & (W)0x00000000ffffffffull;
	if constexpr (CNT > 16) h ^= (h >> 16) & (W)0x0000ffff0000ffffull;
	if constexpr (CNT > 8) h ^= (h >> 8) & (W)0x00ff00ff00ff00ffull;
	if constexpr (CNT > 4) h ^= (h >> 4) & (W)0x0f0f0f0f0f0f0f0full;
	if constexpr (CNT > 2) h ^= (h >> 2) & (W)0x3333333333333333ull;
	if constexpr (CNT > 1) h ^= (h >> 1) & (W)0x5555555555555555ull;
	uint32_t h0 = (uint32_t)h, h1 = CNT > 32 ? (uint32_t)((unsigned long long)h >> 32) : 0u;
	#define SC_XL(H, LV) { \
		const uint32_t m = (uint32_t)((int)L.lowsel[LV] >> 31); \
		h0 ^= xpos<H>(h0, L.lane) & m; \
		if constexpr (CNT > 32) h1 ^= xpos<H>(h1, L.lane) & m; }
	SC_XL(32, 6) SC_XL(16, 5) SC_XL(8, 4) SC_XL(4, 3) SC_XL(2, 2) SC_XL(1, 1)
	#undef SC_XL
	const W keep = CNT == 64 ? ~(W)0 : (W)(((unsigned long long)1 << (CNT & 63)) - 1ull);
	const bool bad = ((h0 & (uint32_t)fz & (uint32_t)keep) | (CNT > 32 ? h1 & (uint32_t)(fz >> 32) : 0u)) != 0u || mu == 0u;
	return __ballot(bad) == 0ull;
}
// the smallest leaf magnitude over the information leaves of a clean node (in place on its array)
template <int H, int LV, int CNT> __device__ __forceinline__ void sc_mag_cross(float (&a)[CNT], const ScLane &L)
{
	const bool low = L.lowsel[LV] != 0u;
	#pragma unroll
	for (int x = 0; x < CNT; ++x) {
		const float p = xpos<H>(a[x], L.lane);
		a[x] = low ? __builtin_fminf(a[x], p) : a[x] + p;
	}
}
template <int CNT> __device__ __forceinline__ uint32_t sc_clean_fork(float (&a)[CNT], unsigned long long fz, const ScLane &L)
{
	#pragma unroll
	for (int x = 0; x < CNT; ++x)
		a[x] = __builtin_fabsf(a[x]);
	#pragma unroll
	for (int t = CNT / 2; t >= 1; t >>= 1)
		#pragma unroll
		for (int x = 0; x < CNT; ++x)
			if (!(x & t)) {
				const float lo = __builtin_fminf(a[x], a[x + t]), hi = a[x] + a[x + t];
				a[x] = lo;
				a[x + t] = hi;
			}
	sc_mag_cross<32, 6>(a, L);
	sc_mag_cross<16, 5>(a, L);
	sc_mag_cross<8, 4>(a, L);
	sc_mag_cross<4, 3>(a, L);
	sc_mag_cross<2, 2>(a, L);
	sc_mag_cross<1, 1>(a, L);
	uint32_t mu = 0x7f800000u;
	#pragma unroll
	for (int x = 0; x < CNT; ++x)
		mu = min(mu, ((fz >> x) & 1ull) ? 0x7f800000u : __float_as_uint(a[x]));
	return sc_min_mag<6>(mu, L.lane);
}
// a node on a register array: clean -> its bits, its share of min_fork
template <int CNT> __device__ __forceinline__ bool sc_try_clean(float (&r)[CNT], unsigned long long fz, ScAcc &acc, const ScLane &L, uint32_t &bits)
{
	typename ScWord<CNT>::T h;
	if (!sc_clean(r, fz, L, h))
		return false;
	bits = (uint32_t)h;
	acc.info(sc_clean_fork(r, fz, L));
	return true;
}

// Persistent grid: workgroup = one wave = one decoder (of 64 >> LB codewords) with its own level stores; decoders take units of
// 64 >> LB consecutive entries from the run's counter.  Entries of a unit that share the frozen table are decoded side by side;
// a unit whose entries do not (a mixed-mode batch), or whose second entry does not exist, is decoded one entry at a time with
// the lanes of the other codeword doing the same work on the same data.
#ifndef SC_WAVES_PER_SIMD
#define SC_WAVES_PER_SIMD 2       // register budget of the two-codewords layout: 2 = 256 VGPRs (it needs 200 - 250), 3 = 168 (40 spilled)
#endif
// One codeword per wave: 256 VGPRs, two waves per SIMD, eight decoders per CU.  Until the clean-node tests (round 6) the kernel fitted 168
// (three waves per SIMD: ten decoders, what the 16 KB of LDS each allows); with them 120 registers spill at 168 and the smaller residency
// is the faster one (-20 dB, 65 536 frames: 27.9 against 26.1 ms; 128 loads in flight: 36 ms).
#ifndef SC6_WAVES
#define SC6_WAVES 2
#endif
template <int LB>
__global__ __launch_bounds__(64, LB == 6 ? SC6_WAVES : SC_WAVES_PER_SIMD) void k_sc(ListQueue *__restrict__ q, const ListSlot *__restrict__ slots, const float *__restrict__ llr_q,
	float *__restrict__ soft_all, uint32_t *__restrict__ cw_q, uint32_t *__restrict__ xw_q, ScStat *__restrict__ stat_q,
	const uint32_t *__restrict__ frozen2, const uint8_t *__restrict__ node_lev_blk, const uint32_t *__restrict__ frozen_t, int small_run, int top_skip)
{
	using Cf = ScCfg<LB>;
	constexpr int J = Cf::J, C = Cf::C, NBLK = CODE_LEN / J;
	const int lane = threadIdx.x, c = lane >> LB, j = lane & (J - 1);
	ScLane L;
	L.lane = lane;
	L.q = sc_pos(j);
	L.lowsel[0] = 0;
	#pragma unroll
	for (int lv = 1; lv <= 6; ++lv)
		L.lowsel[lv] = ((L.q >> (lv - 1)) & 1) ? 0u : SC_SIGN;
	__shared__ float lds[C * 64 * J];                             // the array of the current sub-tree of each codeword, [c][x][position]
	const int par = 0;
	const int run_n = (int)q->run_n[par];
	const unsigned run_head = q->run_head[par], cap = q->cap;
	if (run_n == 0)
		return;
	// both layouts are launched behind every k_back; the run's length (known on the device only) picks one: a short run is a
	// matter of one codeword's latency, and one codeword per wave has the shorter one (1.2 against 1.9 ms); a long one of throughput
	if (small_run > 0 && (run_n <= small_run) != (LB == 6))
		return;
	const int n_units = (run_n + C - 1) / C;
	float *const my_store = soft_all + (size_t)blockIdx.x * Cf::DECODER_FLOATS;
	const rsrc_t soft = make_rsrc(my_store, C * Cf::STORE_FLOATS * 4);
	const int v_soft0 = c * (Cf::STORE_FLOATS * 4) + j * 4;
	uint32_t look_first = 0;                                      // bit s: this decoder's last codeword had a clean node at the pass of sub-tree s
	for (;;) {
		int unit = 0;
		if (lane == 0)
			unit = atomicAdd(&q->next_unit[par], 1);
		unit = __builtin_amdgcn_readfirstlane(unit);
		if (unit >= n_units)
			break;
		// the unit's entries; which of them go side by side
		int slot_of_c[2], n_pass = 1;
		slot_of_c[0] = (int)((run_head + (unsigned)(unit * C)) % cap);
		slot_of_c[1] = slot_of_c[0];
		if (C == 2 && unit * C + 1 < run_n) {
			slot_of_c[1] = (int)((run_head + (unsigned)(unit * C + 1)) % cap);
			const long dist = ((long)slot_of_c[1] - (long)slot_of_c[0]) * CODE_LEN * 4;
			if ((slots[slot_of_c[0]].oper_mode >= 10) != (slots[slot_of_c[1]].oper_mode >= 10) || dist < 0 || dist >= (1l << 31))
				n_pass = 2;                                       // different tables (or slots a ring's wrap apart): one after the other
		}
		for (int pass = 0; pass < n_pass; ++pass) {
		const int sa = n_pass == 2 ? slot_of_c[pass] : slot_of_c[0], sb = n_pass == 2 ? slot_of_c[pass] : slot_of_c[1];
		const int my_slot = c ? sb : sa;
		const int tab = slots[sa].oper_mode >= 10;                     // decode.cc:312,344
		const uint32_t *frozen = frozen2 + (tab ? 2048 : 0);
		const uint8_t *nlev = node_lev_blk + (tab ? NBLK : 0);
		const rsrc_t llr = make_rsrc(llr_q + (size_t)sa * CODE_LEN, (sb - sa + 1) * CODE_LEN * 4);
		const int v_llr0 = (my_slot - sa) * (CODE_LEN * 4) + j * 4;
		ScIo<LB> io;
		if (LB == 6) {                                                // halves of one codeword's 64-bit words
			io.cwa = cw_q + (size_t)sa * (CODE_LEN / 32);
			io.cwb = io.cwa + 1;
			io.xwa = xw_q + (size_t)sa * (CODE_LEN / 32);
			io.xwb = io.xwa + 1;
		} else {                                                      // 32-bit words of two codewords
			io.cwa = cw_q + (size_t)sa * (CODE_LEN / 32);
			io.xwa = xw_q + (size_t)sa * (CODE_LEN / 32);
			if (sb != sa) {
				io.cwb = cw_q + (size_t)sb * (CODE_LEN / 32);
				io.xwb = xw_q + (size_t)sb * (CODE_LEN / 32);
			} else {                                                  // alone: both lane groups decode it; the second one's words go aside
				io.cwb = (uint32_t *)(my_store + C * Cf::STORE_FLOATS);
				io.xwb = io.cwb + CODE_LEN / 32;
			}
		}
		// A clean node of 16384 / 32768 leaves (sc_top_pass) is skipped with the smallest magnitude of its ARRAY in place of the smallest
		// leaf magnitude over its information leaves - a lower bound: a rule that holds with it holds (min_fork is no smaller), one that fails
		// with it says nothing, and the codeword is decoded once more without such skips (attempt 1: every figure exact).
		#pragma unroll 1
		for (int attempt = 0; attempt < 2; ++attempt) {
		bool weak = false;
		if (attempt)
			look_first = 0;
		int skip_left = 0, skip_lg = 0, skip_k = 0;
		ScAcc acc{ 0.f, 0x7f800000u };
		bool finite = true;
		#pragma unroll 1
		for (int s = 0; s < Cf::NSUB; ++s) {
			// ---------------- the array of this sub-tree into LDS, through the level store
			if (skip_left == 0) {
				const int D = s ? __builtin_ctz(s) + 1 : 16 - Cf::LL;
				ScTop top{ frozen_t + tab * (16 * 64 * 2), attempt == 0 && top_skip != 0, 0, 0u };
				#define SC_PASS(DD, KK, EE) sc_top_pass<LB, DD, KK, EE>(soft, llr, lds, io, s, lane, v_llr0, v_soft0, finite, top, L.lowsel)
				if constexpr (LB == 6 && SC_CLEAN_TOP) {
					if (D >= 3 && top.allow && ((look_first >> s) & 1u)) {   // look before storing (see sc_top_pass)
						if (s == 0) SC_PASS(4, 0, false);
						else if (s == Cf::NSUB / 2) SC_PASS(4, 1, false);
						else SC_PASS(3, 2, false);
						if (!top.nskip) {
							look_first &= ~(1u << s);
							top.allow = false;
						}
					}
				}
				if (!top.nskip) {
					if (s == 0) SC_PASS(16 - Cf::LL, 0, true);
					else if (s == Cf::NSUB / 2) SC_PASS(16 - Cf::LL, 1, true);
					else if (D == 1) SC_PASS(1, 2, true);
					else if (D == 2) SC_PASS(2, 2, true);
					else if (D == 3) SC_PASS(3, 2, true);
					else if constexpr (16 - Cf::LL > 4) SC_PASS(4, 2, true);
					if (top.nskip)
						look_first |= 1u << s;
				}
				#undef SC_PASS
				if (top.nskip) {
					skip_left = top.nskip;
					skip_lg = 31 - __builtin_clz(top.nskip);
					skip_k = 0;
					acc.info(sc_min_mag<6>(top.mu, lane));
					weak = true;
				}
			}
			SC_WAVE_ORDER();
			// this sub-tree's 64 table bytes and frozen words, one block per lane; the block loop reads them with v_readlane
			const int blk0 = s * 64;
			const int nlv = nlev[blk0 + lane];
			uint32_t fzl, fzh = 0;
			if (LB == 6) {
				fzl = frozen[(blk0 + lane) * 2];
				fzh = frozen[(blk0 + lane) * 2 + 1];
			} else
				fzl = frozen[blk0 + lane];
			const float *my = lds + c * (64 * J) + L.q;
			unsigned long long HR = 0;                                // partial sums: bit x = position x * J + q
			// the sub-tree itself clean?  (frozen bits the way this lane holds the sub-tree: tables.cpp frozen_t)
			unsigned long long FT = 0;
			bool clean12 = false;
			int m_first = 1;                                          // the first level of partial-sum combines on the published words below
			if (skip_left) {                                          // inside a clean node: its hard decisions are the words of its sub-trees as they stand
				HR = ((const unsigned long long *)lds)[skip_k * 64 + lane];
				clean12 = true;
				++skip_k;
				m_first = --skip_left ? 99 : skip_lg + 1;
			} else
			if constexpr (LB == 6 && SC_CLEAN_MIN <= 6) {
				const uint2 ft = ((const uint2 *)frozen_t)[(tab * 16 + s) * 64 + lane];
				FT = (unsigned long long)ft.x | ((unsigned long long)ft.y << 32);
				float V[64];
				#pragma unroll
				for (int x = 0; x < 64; ++x)
					V[x] = my[x * J];
				if (sc_clean(V, FT, L, HR)) {
					acc.info(sc_clean_fork(V, FT, L));
					clean12 = true;
				} else
					HR = 0;
			}
			float R5[32], R4[16], R3[8], R2[4], R1[2], R0[1];
			#pragma unroll 1
			for (int b = 0, adv = 1; b < 64 && !clean12; b += adv) {
				adv = 1;
				const int zb = b ? __builtin_ctz(b) : 6;              // the one g step of this block produces the array of J << zb leaves (6: none)
				const int nl = __builtin_amdgcn_readlane(nlv, b), nl0 = nl & 15, nl1 = nl >> 4;
				const int Lt = (nl0 > nl1 ? nl0 : nl1) - LB;          // the largest uniform node that starts here: J << Lt leaves (< 0: none)
				const bool frz = nl0 > nl1;
				const uint32_t hb = zb < 6 ? (uint32_t)(HR >> (b - (1 << zb))) : 0u;
				int L2 = -1;
				uint32_t bits = 0;
				if (zb >= 5) {
					if (zb == 5) {
						#pragma unroll
						for (int x = 0; x < 32; ++x)
							R5[x] = __uint_as_float(__float_as_uint(my[x * J]) ^ ((hb << (31 - x)) & SC_SIGN)) + my[(x + 32) * J];
					} else {
						#pragma unroll
						for (int x = 0; x < 32; ++x)
							R5[x] = f_minsum(my[x * J], my[(x + 32) * J]);
					}
					if (Lt == 5) { bits = sc_rate1<LB>(R5, acc, lane); L2 = 5; }
					else if constexpr (LB == 6 && SC_CLEAN_MIN <= 5) { if (sc_try_clean(R5, FT >> b, acc, L, bits)) L2 = 5; }
				}
				if (zb >= 4 && L2 < 0) {
					if (zb == 4) sc_g_half(R4, R5, hb); else sc_f_half(R4, R5);
					if (Lt == 4) { bits = sc_rate1<LB>(R4, acc, lane); L2 = 4; }
					else if constexpr (LB == 6 && SC_CLEAN_MIN <= 4) { if (sc_try_clean(R4, FT >> b, acc, L, bits)) L2 = 4; }
				}
				if (zb >= 3 && L2 < 0) {
					if (zb == 3) sc_g_half(R3, R4, hb); else sc_f_half(R3, R4);
					if (Lt == 3) { bits = sc_rate1<LB>(R3, acc, lane); L2 = 3; }
					else if constexpr (LB == 6 && SC_CLEAN_MIN <= 3) { if (sc_try_clean(R3, FT >> b, acc, L, bits)) L2 = 3; }
				}
				if (zb >= 2 && L2 < 0) {
					if (zb == 2) sc_g_half(R2, R3, hb); else sc_f_half(R2, R3);
					if (Lt == 2) {
						if (frz) sc_rate0<LB>(R2, acc, lane); else bits = sc_rate1<LB>(R2, acc, lane);
						L2 = 2;
					} else if constexpr (LB == 6 && SC_CLEAN_MIN <= 2) { if (sc_try_clean(R2, FT >> b, acc, L, bits)) L2 = 2; }
				}
				if (zb >= 1 && L2 < 0) {
					if (zb == 1) sc_g_half(R1, R2, hb); else sc_f_half(R1, R2);
					if (Lt == 1) {
						if (frz) sc_rate0<LB>(R1, acc, lane); else bits = sc_rate1<LB>(R1, acc, lane);
						L2 = 1;
					} else if constexpr (LB == 6 && SC_CLEAN_MIN <= 1) { if (sc_try_clean(R1, FT >> b, acc, L, bits)) L2 = 1; }
				}
				if (L2 < 0) {
					if (zb == 0) sc_g_half(R0, R1, hb); else sc_f_half(R0, R1);
					if (Lt == 0) {
						if (frz) sc_rate0<LB>(R0, acc, lane); else bits = sc_rate1<LB>(R0, acc, lane);
						L2 = 0;
					} else if constexpr (LB == 6 && SC_CLEAN_MIN <= 0) { if (sc_try_clean(R0, FT >> b, acc, L, bits)) L2 = 0; }
				}
				if (L2 >= 0) {
					adv = 1 << L2;
					HR |= (unsigned long long)bits << b;
				} else {
					const uint32_t fz0 = (uint32_t)__builtin_amdgcn_readlane((int)fzl, b), fz1 = LB == 6 ? (uint32_t)__builtin_amdgcn_readlane((int)fzh, b) : 0u;
					HR |= (unsigned long long)(sc_walk_block<LB>(R0[0], fz0, fz1, acc, L) >> 31) << b;
				}
				// partial-sum combines of the nodes of 2 J .. 64 J leaves that end here: left half ^= right half
				const int bn = b + adv;
				for (int m = L2 >= 0 ? L2 + 1 : 1; m <= 6 && (bn & ((1 << m) - 1)) == 0; ++m) {
					const int half = 1 << (m - 1), b0 = bn - 2 * half;
					const unsigned long long lmask = ((1ull << half) - 1ull) << b0;
					HR = (HR & ~lmask) | ((HR ^ (HR >> half)) & lmask);
				}
			}
			// publish the sub-tree's partial sums as plain words: position q sits on lane j ^ ((j & 4) ? 3 : 0); bring the bits to their
			// natural lanes (quad_perm [3,2,1,0] on the upper half of every 8), then one ballot per element x: 64 lanes = one 64-bit
			// word of one codeword / one 32-bit word of each of two
			{
				uint32_t h0 = (uint32_t)HR, h1 = (uint32_t)(HR >> 32);
				const uint32_t s0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h0, 0x1B, 0xf, 0xf, false);
				const uint32_t s1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h1, 0x1B, 0xf, 0xf, false);
				if (lane & 4) { h0 = s0; h1 = s1; }
				unsigned long long mine = 0;
				#pragma unroll
				for (int u = 0; u < 64; ++u) {
					const unsigned long long bal = __ballot(((u < 32 ? h0 >> u : h1 >> (u - 32)) & 1u) != 0u);
					if (lane == u)
						mine = bal;
				}
				const int w = ScIo<LB>::idx(s * 64 + lane);
				io.cwa[w] = (uint32_t)mine;
				io.cwb[w] = (uint32_t)(mine >> 32);
			}
			// combines of the larger nodes that end here, on the published words (every word stays with its lane)
			const int sn = s + 1;
			for (int m = m_first; m <= 16 - Cf::LL && (sn & ((1 << m) - 1)) == 0; ++m) {
				const int halfw = 64 << (m - 1), w0 = sn * 64 - 2 * halfw;
				for (int w = lane; w < halfw; w += 64) {
					const int wl = ScIo<LB>::idx(w0 + w), wr = ScIo<LB>::idx(w0 + halfw + w);
					io.cwa[wl] ^= io.cwa[wr];
					io.cwb[wl] ^= io.cwb[wr];
				}
			}
			SC_WAVE_ORDER();
		}
		const unsigned long long unf = __ballot(!finite);
		if (LB == 6 && attempt == 0 && weak && !(unf == 0ull && __uint_as_float(acc.fork) > acc.M))
			continue;                                                 // (wave-uniform: one codeword per wave)
		if (j == 0 && (C == 1 || c == 0 || sb != sa)) {
			const bool all_finite = ((unf >> (c * J)) & (J == 64 ? ~0ull : ((1ull << J) - 1ull))) == 0;
			ScStat st;
			st.metric = acc.M;
			st.min_fork = __uint_as_float(acc.fork);
			st.ok = (all_finite && __uint_as_float(acc.fork) > acc.M) ? 1 : 0;   // the rule (a NaN compares false)
			st.pad = 0;
			stat_q[my_slot] = st;
		}
		SC_WAVE_ORDER();
		break;
		}   // attempt
		}   // second entry of a unit that could not be paired
	}
}

// ---------------------------------------------------------------- k_sc_finish: decode.cc:532-555 for the frames k_sc decided
// One workgroup per entry of the run.  Rule holds: P* is lane 0 - its systematic bits (P*'s codeword at the unfrozen positions,
// decode.cc:254-261), their CRC-32 (decode.cc:533-541); CRC zero: the reference takes lane 0 - payload (descrambled,
// decode.cc:613-615), best_lane 0, the flip count against the channel's hard decisions (decode.cc:546-555): the frame is finished.
// Rule or CRC fails: the frame takes a slot of the list decoder's queue and its LLRs are copied there - the general path, unchanged.
// slot_of[frame]: -2 - (slot in the SC ring) for a frame finished here (its LLRs stay there until the next chunk), else its slot
// in the list decoder's queue.
__global__ __launch_bounds__(256) void k_sc_finish(ListQueue *__restrict__ qs, const ListSlot *__restrict__ slots_s, const float *__restrict__ llr_s,
	const unsigned long long *__restrict__ cw_q, const unsigned long long *__restrict__ xw_q, const ScStat *__restrict__ stat_q, Tables tb, int descramble,
	ListQueue *__restrict__ ql, ListSlot *__restrict__ slots_l, float *__restrict__ llr_l, int *__restrict__ slot_of, int chunk_seq)
{
	const int rel = blockIdx.x, tid = threadIdx.x;
	const unsigned run_n = qs->run_n[0], run_head = qs->run_head[0], cap = qs->cap;
	if ((unsigned)rel >= run_n)
		return;
	const int slot = (int)((run_head + (unsigned)rel) % cap);
	const ListSlot ls = slots_s[slot];
	const ScStat st = stat_q[slot];
	// an entry a run left over for this one (k_sc_plan): its chunk's own arrays (a staging buffer, when the outputs go to the host) have
	// left, its outputs go where the list decoder's would, and the per-chunk slot table is another chunk's by now
	const bool late = ls.chunk != chunk_seq;
	uint8_t *const pay_dst = late ? ls.payload : ls.payload_now;
	Result *const res_dst = late ? ls.res : ls.res_now;
	__shared__ uint32_t bits[CODE_LEN / 32];
	__shared__ uint32_t mesg32[MESG_BYTES_MAX / 4];               // the systematic message, little-endian words
	uint8_t *const mesg = (uint8_t *)mesg32;
	__shared__ uint32_t ctab[256], cpart[4];
	__shared__ int flips_red[4], slot_sh;
	const ModeDesc md = mode_desc(ls.oper_mode);
	bool done = false;
	if (st.ok) {
		const uint32_t *cw = (const uint32_t *)(cw_q + (size_t)slot * (CODE_LEN / 64));
		for (int w = tid; w < CODE_LEN / 32; w += 256)
			bits[w] = cw[w];
		for (int w = tid; w < MESG_BYTES_MAX / 4; w += 256)
			mesg32[w] = 0;
		ctab[tid] = tb.crc32_tab[tid];
		__syncthreads();
		message_gather(bits, mesg32, tb.info_compress + (md.table ? 2048 * 8 : 0), tid);   // decode.cc:254-261
		__syncthreads();
		done = crc32_wg256(mesg, ctab, tb.crc32_adv, cpart, tid) == 0;   // decode.cc:533-541 (dev_common.h)
	}
	if (done) {
		// decode.cc:546-554: received hard decision against decoded bit over the data bits = the unfrozen positions below the one
		// of message bit DATA_BITS
		const uint32_t *xwp = (const uint32_t *)(xw_q + (size_t)slot * (CODE_LEN / 64));
		const uint32_t *frozen = tb.frozen + (md.table ? CODE_LEN / 32 : 0);
		const int p_end = (tb.info_pos + (md.table ? MESG_BITS_MAX : 0))[DATA_BITS];
		int flips = 0;
		for (int w = tid; w * 32 < p_end; w += 256) {
			uint32_t m = ~frozen[w];
			if (p_end - w * 32 < 32)
				m &= (1u << (p_end - w * 32)) - 1u;
			flips += __builtin_popcount((bits[w] ^ xwp[w]) & m);
		}
		#pragma unroll
		for (int m = 32; m; m >>= 1)
			flips += __shfl_xor(flips, m);
		if ((tid & 63) == 0)
			flips_red[tid >> 6] = flips;
		for (int i = tid; i < PAYLOAD_BYTES; i += 256)
			pay_dst[i] = mesg[i] ^ (descramble ? tb.scramble[i] : (uint8_t)0);
		__syncthreads();
		if (tid == 0) {
			res_dst->best_lane = 0;
			res_dst->bit_flips = flips_red[0] + flips_red[1] + flips_red[2] + flips_red[3];
			if (!late)
				slot_of[ls.frame] = -2 - slot;
			atomicAdd(&qs->certified, 1u);
			atomicAdd(&qs->done_total, 1u);
		}
		return;
	}
	if (tid == 0) {
		const unsigned e = atomicAdd(&ql->tail, 1u);
		const int lslot = (int)(e % ql->cap);
		slots_l[lslot] = ls;
		if (!late)
			slot_of[ls.frame] = lslot;
		slot_sh = lslot;
	}
	__syncthreads();
	const float4 *src = (const float4 *)(llr_s + (size_t)slot * CODE_LEN);
	float4 *dst = (float4 *)(llr_l + (size_t)slot_sh * CODE_LEN);
	for (int i = tid; i < CODE_LEN / 4; i += 256)
		dst[i] = src[i];
}

// behind k_back of a chunk: the run of k_sc = everything that chunk put into the SC ring; behind k_sc_finish: the adaptive switch.
// Where k_sc decides few frames (below about -18.5 dB every path metric outgrows min_fork) its pass is spent for nothing:
// k_back then sends a probe sample only (one frame in sixteen of every fourth chunk: a run that small still costs a codeword's
// latency, 2 ms, with the machine idle) and the rest straight to the list decoder, and all of them again when an eighth of the
// sample is decided.  (Frames whose Es/N0 estimate rules the pass out never come here at all: k_back.)  Either way every decision is exact: the list decoder is the general path.
// A run takes whole multiples of `unit` (one residency of k_sc's persistent decoders: a codeword is one decoder's serial work, so the
// tail of a run is a round in which most decoders idle) and leaves the rest to the next chunk's run; force (the last chunk of a
// call, calls whose outputs leave chunk by chunk) and the probe sample take everything.
__global__ void k_sc_plan(ListQueue *__restrict__ qs, unsigned unit, int force)
{
	const unsigned head = qs->head, waiting = qs->tail - head;
	const unsigned n = (force || !qs->cert_on || unit <= 1) ? waiting : (waiting / unit) * unit;
	qs->run_head[0] = head;
	qs->run_n[0] = n;
	qs->head = head + n;
	qs->next_unit[0] = 0;
	qs->tried = n;
	qs->certified = 0;
}
__global__ void k_sc_adapt(ListQueue *__restrict__ qs)
{
	const unsigned tried = qs->tried, done = qs->certified;
	if (qs->cert_on) {
		if (tried >= 64 && done * 8 < tried) {
			qs->cert_on = 0;
			qs->probe_tried = qs->probe_done = 0;
		}
	} else {
		// (a chunk of fewer than 128 frames sends fewer than eight probe frames: the sample is summed over chunks - round-5 advisor)
		const unsigned pt = qs->probe_tried + tried, pd = qs->probe_done + done;
		if (pt >= 8) {
			if (pd * 8 >= pt)
				qs->cert_on = 1;
			qs->probe_tried = qs->probe_done = 0;
		} else {
			qs->probe_tried = pt;
			qs->probe_done = pd;
		}
	}
	qs->epoch += 1;
}

// grid5 / grid6 = resident decoders (waves) of the two layouts; lb = 5 / 6: that layout alone; 0: both are launched and the run's
// length decides on the device (at most `small_run` entries: one codeword per wave)
constexpr int SC_SMALL_RUN = 5120;
int sc_codewords_per_wave(int lb) { return lb == 6 ? ScCfg<6>::C : ScCfg<5>::C; }
size_t sc_store_bytes(int lb)                                     // level store per resident decoder
{
	const size_t b5 = (size_t)ScCfg<5>::DECODER_FLOATS * sizeof(float), b6 = (size_t)ScCfg<6>::DECODER_FLOATS * sizeof(float);
	return lb == 6 ? b6 : (lb == 5 ? b5 : (b5 > b6 ? b5 : b6));
}
void launch_sc(hipStream_t s, int lb, int grid5, int grid6, ListQueue *q, const ListSlot *slots, const float *llr_q, float *soft, unsigned long long *cw_q,
	unsigned long long *xw_q, ScStat *stat_q, Tables tb, int top_skip)
{
	const int small_run = lb == 0 ? SC_SMALL_RUN : 0;
	if (lb != 5)
		hipLaunchKernelGGL(k_sc<6>, dim3(grid6), dim3(64), 0, s, q, slots, llr_q, soft, (uint32_t *)cw_q, (uint32_t *)xw_q, stat_q, tb.frozen, tb.node_lev64, tb.frozen_t, small_run, top_skip);
	if (lb != 6)
		hipLaunchKernelGGL(k_sc<5>, dim3(grid5), dim3(64), 0, s, q, slots, llr_q, soft, (uint32_t *)cw_q, (uint32_t *)xw_q, stat_q, tb.frozen, tb.node_lev32, tb.frozen_t, small_run, 0);
}
void launch_sc_finish(hipStream_t s, int max_entries, ListQueue *qs, const ListSlot *slots_s, const float *llr_s, const unsigned long long *cw_q,
	const unsigned long long *xw_q, const ScStat *stat_q, Tables tb, int descramble, ListQueue *ql, ListSlot *slots_l, float *llr_l, int *slot_of, int chunk_seq)
{
	hipLaunchKernelGGL(k_sc_finish, dim3(max_entries), dim3(256), 0, s, qs, slots_s, llr_s, cw_q, xw_q, stat_q, tb, descramble, ql, slots_l, llr_l, slot_of, chunk_seq);
}
void launch_sc_plan(hipStream_t s, ListQueue *qs, unsigned unit, int force) { hipLaunchKernelGGL(k_sc_plan, dim3(1), dim3(1), 0, s, qs, unit, force); }
void launch_sc_adapt(hipStream_t s, ListQueue *qs) { hipLaunchKernelGGL(k_sc_adapt, dim3(1), dim3(1), 0, s, qs); }

}  // namespace rx
