/*
 * oracle/polar.c -- TEST INFRASTRUCTURE (see modem_oracle.h header).
 *
 * Polar code pieces of the receive path.  polar_tables.hh is pure data and is
 * NOT copied: the masks are regenerated with freezer.cc's recipe and pinned by
 * the SHA-256 of the reference's own table (tests/test_oracle_polar.py).
 * PolarSysEnc / PolarEncoder / PolarListDecoder live in the absent
 * aicodix/code headers ("parity unpinned"): restated from the call sites
 * decode.cc:200-203,245-261,530-541 and encode.cc:48,302.
 */
#include "modem_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- frozen-bit table: freezer.cc:14-32 ---------------------------------- */
/* CODE::PolarCodeConst0<16> is absent; it is the BEC (Bhattacharyya)
 * construction: z <- [p]; 16x { z[2i] = 2z-z^2 (degraded), z[2i+1] = z^2 }
 * in long double, then the K most reliable (smallest z) are unfrozen.  This
 * reproduces both reference tables with 0 mismatches. */
typedef struct { long double z; int idx; } zi_t;
static int zi_cmp(const void *a, const void *b)
{
	const zi_t *x = (const zi_t *)a, *y = (const zi_t *)b;
	if (x->z < y->z) return -1;
	if (x->z > y->z) return 1;
	return x->idx - y->idx;   /* stable */
}
void orc_frozen_table(int table, uint32_t *frozen)
{
	const int M = 16, LEN = 1 << M;
	int N = table ? 64512 : 64800;     /* freezer.cc:36-37 */
	int K = 43040 + 32;
	/* freezer.cc:17-24 */
	long double erasure_probability = (long double)(N - K) / N;
	double design_SNR = 10 * log10(-log((double)erasure_probability));
	double better_SNR = design_SNR + 1.59175;
	long double better_probability = expl(-(long double)pow(10.0, better_SNR / 10));
	int Kp = K + LEN - N;              /* freezer.cc:25 */
	long double *z = (long double *)malloc(sizeof(long double) * LEN * 2);
	long double *a = z, *b = z + LEN;
	a[0] = better_probability;
	for (int m = 0, len = 1; m < M; ++m, len *= 2) {
		for (int i = 0; i < len; ++i) {
			b[2 * i] = 2 * a[i] - a[i] * a[i];
			b[2 * i + 1] = a[i] * a[i];
		}
		long double *t = a; a = b; b = t;
	}
	zi_t *zi = (zi_t *)malloc(sizeof(zi_t) * LEN);
	for (int i = 0; i < LEN; ++i) { zi[i].z = a[i]; zi[i].idx = i; }
	qsort(zi, LEN, sizeof(zi_t), zi_cmp);
	for (int i = 0; i < LEN / 32; ++i)
		frozen[i] = 0xffffffffu;
	for (int i = 0; i < Kp; ++i)
		frozen[zi[i].idx / 32] &= ~(1u << (zi[i].idx % 32));
	free(zi);
	free(z);
}
const uint32_t *orc_frozen_get(int table)
{
	static uint32_t tabs[2][2048];
	static int have[2];
	table = !!table;
	#pragma omp critical(orc_frozen)
	{
		if (!have[table]) {
			orc_frozen_table(table, tabs[table]);
			have[table] = 1;
		}
	}
	return tabs[table];
}

static inline int is_frozen(const uint32_t *frozen, int i) { return (frozen[i / 32] >> (i % 32)) & 1; }

/* ---- CODE::PolarEncoder: x = u F^{(x)n}, NRZ products, natural order ------ */
/* decode.cc:256 call; frozen inputs = +1 */
void orc_polar_enc(int8_t *code, const int8_t *mesg, const uint32_t *frozen, int level)
{
	int length = 1 << level;
	for (int i = 0; i < length; ++i)
		code[i] = is_frozen(frozen, i) ? 1 : *mesg++;
	for (int h = 1; h < length; h *= 2)
		for (int i = 0; i < length; i += 2 * h)
			for (int j = i; j < i + h; ++j)
				code[j] = (int8_t)(code[j] * code[j + h]);
}
/* ---- CODE::PolarSysEnc<int8_t> (encode.cc:302): systematic --------------- */
/* encode -> force frozen positions to +1 -> encode again.  The decoder relies
 * on code[unfrozen_i] == mesg[i] (decode.cc:547-553); the systematic codeword
 * is unique, so any correct construction gives the same bits. */
void orc_polar_sysenc(int8_t *code, const int8_t *mesg, const uint32_t *frozen, int level)
{
	int length = 1 << level;
	orc_polar_enc(code, mesg, frozen, level);
	for (int i = 0; i < length; ++i)
		if (is_frozen(frozen, i))
			code[i] = 1;
	for (int h = 1; h < length; h *= 2)
		for (int i = 0; i < length; i += 2 * h)
			for (int j = i; j < i + h; ++j)
				code[j] = (int8_t)(code[j] * code[j + h]);
}

/* ---- CODE::PolarListDecoder<SIMD<float,L>,16> (decode.cc:201,530) -------- */
/*
 * Successive-cancellation list decoding, min-sum, L lanes (L = SIMD width of
 * the reference build, decode.cc:164-169: 8 with AVX2, else 4).
 *   soft[(n+i)*L+k]  LLRs of the size-n node being decoded (reference layout
 *                    soft[N+i] with one SIMD vector per entry)
 *   hard[i*L+k]      partial sums, +1/-1
 *   f(a,b) = sign(a)sign(b)min(|a|,|b|);  g(a,b,u) = u*a + b
 *   left subtree = lower indices; each subtree returns the lane map it applied
 *   frozen leaf: metric += |llr| when llr < 0, hard = +1, identity map
 *   (an aligned all-frozen node of 2..128 leaves is charged in one step, see scl_node)
 *   info leaf  : fork 2L candidates, keep the L smallest metrics.
 * Tie rule (std::nth_element is implementation-defined there): candidates are
 * ordered by (metric, candidate index 2k+u) and survivors are stored in that
 * sorted order, so the result is deterministic.  Initial metrics: lane 0 = 0,
 * others = 1000, so the list fills from one path.
 */
enum { ORC_RATE0_MIN = 1, ORC_RATE0_MAX = 7 };
static unsigned g_numerics;      /* ORC_NUM_* flags, see modem_oracle.h */
void orc_set_numerics(unsigned flags) { g_numerics = flags; }
unsigned orc_get_numerics(void) { return g_numerics; }   /* rate-0 nodes of 2..128 leaves are charged in one step */

typedef struct {
	int L, count;
	float *soft;       /* 2N*L */
	int8_t *hard;      /* N*L */
	uint8_t *maps;     /* count*L */
	int8_t *mesg;      /* count*L */
	float metric[ORC_MAX_LIST];
	const uint32_t *frozen;
} scl_t;

static inline float prod(float a, float b)
{
	float m = fminf(fabsf(a), fabsf(b));
	return ((a < 0.f) != (b < 0.f)) ? -m : m;
}

static void scl_leaf(scl_t *s, int index, uint8_t *map)
{
	const int L = s->L;
	float *sft = s->soft + 1 * L;
	int8_t *hrd = s->hard + (size_t)index * L;
	if (is_frozen(s->frozen, index)) {
		for (int k = 0; k < L; ++k) {
			if (sft[k] < 0.f)
				s->metric[k] -= sft[k];
			hrd[k] = 1;
			map[k] = (uint8_t)k;
		}
		return;
	}
	float fork[2 * ORC_MAX_LIST];
	int perm[2 * ORC_MAX_LIST];
	for (int k = 0; k < L; ++k)
		fork[2 * k] = fork[2 * k + 1] = s->metric[k];
	for (int k = 0; k < L; ++k) {
		if (sft[k] < 0.f)
			fork[2 * k] -= sft[k];
		else
			fork[2 * k + 1] += sft[k];
	}
	/* rank by (value, index): insertion sort of 2L entries */
	for (int c = 0; c < 2 * L; ++c) {
		int j = c;
		while (j > 0 && fork[perm[j - 1]] > fork[c]) {
			perm[j] = perm[j - 1];
			--j;
		}
		perm[j] = c;
	}
	if (g_numerics & ORC_NUM_SURVIVORS_UNSORTED) {
		/* same survivor SET (the L best by (metric, index)), left in candidate order */
		for (int a = 1; a < L; ++a) {
			int v = perm[a], j = a;
			while (j > 0 && perm[j - 1] > v) {
				perm[j] = perm[j - 1];
				--j;
			}
			perm[j] = v;
		}
	}
	for (int k = 0; k < L; ++k) {
		s->metric[k] = fork[perm[k]];
		map[k] = (uint8_t)(perm[k] >> 1);
		hrd[k] = (int8_t)(1 - 2 * (perm[k] & 1));
	}
	memcpy(s->mesg + (size_t)s->count * L, hrd, (size_t)L);
	memcpy(s->maps + (size_t)s->count * L, map, (size_t)L);
	++s->count;
}

/* decode the size-n (n = 1<<m) node whose first leaf is 'index'; its input
 * LLRs are soft[(n+i)*L+k]; returns accumulated lane map in 'map' */
/* all 1<<m leaves from 'index' frozen? */
static int all_frozen(const uint32_t *frozen, int index, int m)
{
	for (int i = 0; i < (1 << m); ++i)
		if (!is_frozen(frozen, index + i))
			return 0;
	return 1;
}

static void scl_node(scl_t *s, int m, int index, uint8_t *map)
{
	const int L = s->L;
	if (m == 0) {
		scl_leaf(s, index, map);
		return;
	}
	if (m >= ORC_RATE0_MIN && m <= ORC_RATE0_MAX && !(g_numerics & ORC_NUM_RATE0_LEAFWALK) && all_frozen(s->frozen, index, m)) {
		/* Rate-0 node (2..128 leaves; the recursion is top-down, so this is the largest such node) in one step.  With min-sum, the frozen-leaf penalties of a sub-tree add up
		 * to sum_i max(0, -llr_i) over the node's OWN input LLRs (f keeps the smaller magnitude with the
		 * product sign, g with u = 0 is a + b: case by case max(0,-f(a,b)) + max(0,-(a+b)) =
		 * max(0,-a) + max(0,-b), then induction over the levels), so the leaf walk is not needed.  The
		 * reference is built -Ofast (Makefile:2) and may reassociate this sum anyway; the order fixed here
		 * is the butterfly halving p[i] += p[i + h], h = n/2 .. 1 - the order a wave reduces it in - and the
		 * total is added to the path metric once.  Partial sums are all +1, the lane map is the identity. */
		const int n = 1 << m;
		for (int k = 0; k < L; ++k) {
			float p[1 << ORC_RATE0_MAX];
			for (int i = 0; i < n; ++i) {
				float v = s->soft[(size_t)(n + i) * L + k];
				p[i] = v < 0.f ? -v : 0.f;
			}
			for (int h = n / 2; h >= 1; h /= 2)
				for (int i = 0; i < h; ++i)
					p[i] = p[i] + p[i + h];
			s->metric[k] += p[0];
			map[k] = (uint8_t)k;
		}
		for (int i = 0; i < n; ++i)
			for (int k = 0; k < L; ++k)
				s->hard[(size_t)(index + i) * L + k] = 1;
		return;
	}
	const int n = 1 << m, h = n / 2;
	float *soft = s->soft;
	int8_t *hard = s->hard + (size_t)index * L;
	uint8_t lmap[ORC_MAX_LIST], rmap[ORC_MAX_LIST];
	for (int i = 0; i < h; ++i)
		for (int k = 0; k < L; ++k)
			soft[(h + i) * L + k] = prod(soft[(n + i) * L + k], soft[(n + h + i) * L + k]);
	scl_node(s, m - 1, index, lmap);
	for (int i = 0; i < h; ++i)
		for (int k = 0; k < L; ++k)
			soft[(h + i) * L + k] = (float)hard[i * L + k] * soft[(n + i) * L + lmap[k]]
				+ soft[(n + h + i) * L + lmap[k]];
	scl_node(s, m - 1, index + h, rmap);
	for (int i = 0; i < h; ++i) {
		int8_t t[ORC_MAX_LIST];
		for (int k = 0; k < L; ++k)
			t[k] = (int8_t)(hard[i * L + rmap[k]] * hard[(h + i) * L + k]);
		memcpy(hard + i * L, t, (size_t)L);
	}
	for (int k = 0; k < L; ++k)
		map[k] = lmap[rmap[k]];
}

int orc_polar_list_decode(float *metric_out, int8_t *mesg_out, const float *llr,
	const uint32_t *frozen, int level, int L)
{
	const int N = 1 << level;
	scl_t s;
	s.L = L;
	s.count = 0;
	s.frozen = frozen;
	s.soft = (float *)malloc(sizeof(float) * 2 * (size_t)N * L);
	s.hard = (int8_t *)malloc((size_t)N * L);
	s.maps = (uint8_t *)malloc((size_t)N * L);
	s.mesg = mesg_out;
	s.metric[0] = 0.f;
	for (int k = 1; k < L; ++k)
		s.metric[k] = 1000.f;
	for (int i = 0; i < N; ++i)
		for (int k = 0; k < L; ++k)
			s.soft[(size_t)(N + i) * L + k] = llr[i];
	uint8_t map[ORC_MAX_LIST];
	scl_node(&s, level, 0, map);
	/* back-trace: compose the lane maps from the last leaf to the first */
	int count = s.count;
	if (count > 0) {
		uint8_t acc[ORC_MAX_LIST], nxt[ORC_MAX_LIST];
		memcpy(acc, s.maps + (size_t)(count - 1) * L, (size_t)L);
		for (int i = count - 2; i >= 0; --i) {
			int8_t t[ORC_MAX_LIST];
			for (int k = 0; k < L; ++k)
				t[k] = mesg_out[(size_t)i * L + acc[k]];
			memcpy(mesg_out + (size_t)i * L, t, (size_t)L);
			for (int k = 0; k < L; ++k)
				nxt[k] = s.maps[(size_t)i * L + acc[k]];
			memcpy(acc, nxt, (size_t)L);
		}
	}
	if (metric_out)
		for (int k = 0; k < L; ++k)
			metric_out[k] = s.metric[k];
	free(s.soft);
	free(s.hard);
	free(s.maps);
	return count;
}

/* test helper: D9 + systematic() (decode.cc:530-531, 254-261) from raw LLRs:
 * per-lane systematic message bits, LE packed, lane_mesg[L][mesg_bytes] */
int orc_polar_lane_mesg(const float *llr, const uint32_t *frozen, int level, int L,
	uint8_t *lane_mesg, int mesg_bytes, float *metric)
{
	const int N = 1 << level;
	int8_t *mesg = (int8_t *)malloc((size_t)N * L);
	int8_t *u = (int8_t *)malloc((size_t)N);
	int8_t *x = (int8_t *)malloc((size_t)N);
	int count = orc_polar_list_decode(metric, mesg, llr, frozen, level, L);
	memset(lane_mesg, 0, (size_t)L * mesg_bytes);
	for (int k = 0; k < L; ++k) {
		for (int i = 0; i < count; ++i)
			u[i] = mesg[(size_t)i * L + k];
		orc_polar_enc(x, u, frozen, level);
		for (int i = 0, j = 0; i < N && j < count; ++i)
			if (!is_frozen(frozen, i)) {
				if (x[i] < 0)
					lane_mesg[(size_t)k * mesg_bytes + j / 8] |= (uint8_t)(1 << (j % 8));
				++j;
			}
	}
	free(mesg);
	free(u);
	free(x);
	return count;
}

/* ---- the sign-following path of the list decoder, alone (list size 1) -----------------------------------------
 * NOT a restatement of reference code: the checker of the build's own "SC dominance" certificate.
 * The path P* that takes the sign of its LLR at every information leaf, decoded with exactly the arithmetic
 * scl_node runs for one lane (same prod / g, same order of the penalty sums incl. the rate-0 grouping and the
 * numerics switches), carrying
 *   metric    = P*'s path metric M* (its frozen-leaf penalties)
 *   min_fork  = min over the information leaves i of fl(M*(i) + |llr_i|): what the sibling candidate at i costs
 * Claim checked by tests/test_oracle_kat.py: if min_fork > metric (final), orc_polar_list_decode ends with P* as
 * lane 0 - same message, same metric - for any list size.
 * hard[N]: P*'s re-encoded codeword (+1 / -1), i.e. the root's partial sums.  Returns the number of information leaves. */
typedef struct {
	float *soft;       /* 2N */
	int8_t *hard;      /* N */
	float metric, min_fork;
	int count;
	const uint32_t *frozen;
} sc1_t;

static void sc1_node(sc1_t *s, int m, int index)
{
	if (m == 0) {
		float v = s->soft[1];
		if (is_frozen(s->frozen, index)) {
			if (v < 0.f)
				s->metric -= v;
			s->hard[index] = 1;
			return;
		}
		float fork = s->metric + fabsf(v);            /* scl_leaf: the candidate that does NOT follow the sign */
		if (!(fork >= s->min_fork))                   /* (a NaN sticks) */
			s->min_fork = fork;
		s->hard[index] = (int8_t)(v < 0.f ? -1 : 1);
		++s->count;
		return;
	}
	if (m >= ORC_RATE0_MIN && m <= ORC_RATE0_MAX && !(g_numerics & ORC_NUM_RATE0_LEAFWALK) && all_frozen(s->frozen, index, m)) {
		const int n = 1 << m;
		float p[1 << ORC_RATE0_MAX];
		for (int i = 0; i < n; ++i) {
			float v = s->soft[n + i];
			p[i] = v < 0.f ? -v : 0.f;
		}
		for (int h = n / 2; h >= 1; h /= 2)
			for (int i = 0; i < h; ++i)
				p[i] = p[i] + p[i + h];
		s->metric += p[0];
		for (int i = 0; i < n; ++i)
			s->hard[index + i] = 1;
		return;
	}
	const int n = 1 << m, h = n / 2;
	float *soft = s->soft;
	int8_t *hard = s->hard + index;
	for (int i = 0; i < h; ++i)
		soft[h + i] = prod(soft[n + i], soft[n + h + i]);
	sc1_node(s, m - 1, index);
	for (int i = 0; i < h; ++i)
		soft[h + i] = (float)hard[i] * soft[n + i] + soft[n + h + i];
	sc1_node(s, m - 1, index + h);
	for (int i = 0; i < h; ++i)
		hard[i] = (int8_t)(hard[i] * hard[h + i]);
}

int orc_polar_sc_path(const float *llr, const uint32_t *frozen, int level, int8_t *hard, float *metric, float *min_fork)
{
	const int N = 1 << level;
	sc1_t s;
	s.soft = (float *)malloc(sizeof(float) * 2 * (size_t)N);
	s.hard = hard;
	s.metric = 0.f;
	s.min_fork = INFINITY;
	s.count = 0;
	s.frozen = frozen;
	memcpy(s.soft + N, llr, sizeof(float) * (size_t)N);
	sc1_node(&s, level, 0);
	if (metric)
		*metric = s.metric;
	if (min_fork)
		*min_fork = s.min_fork;
	free(s.soft);
	return s.count;
}
