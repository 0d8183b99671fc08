/*
 * oracle/bch_osd.c -- TEST INFRASTRUCTURE (see modem_oracle.h header).
 *
 * BCH(255,71) encoder / generator matrix and the order-4 ordered-statistics
 * decoder of the header symbol.  All three live in the absent aicodix/code
 * headers ("parity unpinned"); restated from the call sites
 * encode.cc:47,164,272-278 and decode.cc:199,378-384,417.
 */
#include "modem_oracle.h"
#include <stdlib.h>
#include <string.h>

enum { N = ORC_BCH_N, K = ORC_BCH_K, NP = N - K };

/* the 24 minimal polynomials, decode.cc:379-384 == encode.cc:273-278 */
static const int minpolys[24] = {
	0x11d, 0x177, 0x1f3, 0x169,   /* 0b100011101, 0b101110111, 0b111110011, 0b101101001 */
	0x1bd, 0x1e7, 0x12b, 0x1d7,   /* 0b110111101, 0b111100111, 0b100101011, 0b111010111 */
	0x013, 0x165, 0x18b, 0x163,   /* 0b000010011, 0b101100101, 0b110001011, 0b101100011 */
	0x11b, 0x13f, 0x18d, 0x12d,   /* 0b100011011, 0b100111111, 0b110001101, 0b100101101 */
	0x15f, 0x1f9, 0x1c3, 0x139,   /* 0b101011111, 0b111111001, 0b111000011, 0b100111001 */
	0x1a9, 0x01f, 0x187, 0x1b1    /* 0b110101001, 0b000011111, 0b110000111, 0b110110001 */
};

/* g[d] = coefficient of x^d, degree NP = 184 */
static void genpoly(uint8_t *g)
{
	memset(g, 0, NP + 1);
	g[0] = 1;
	int deg = 0;
	for (int p = 0; p < 24; ++p) {
		int m = minpolys[p], md = 0;
		while (m >> (md + 1))
			++md;
		uint8_t t[NP + 1];
		memset(t, 0, sizeof(t));
		for (int i = 0; i <= deg; ++i)
			if (g[i])
				for (int j = 0; j <= md; ++j)
					if ((m >> j) & 1)
						t[i + j] ^= 1;
		deg += md;
		memcpy(g, t, NP + 1);
	}
}

/* encode.cc:164 bchenc(data, parity): systematic cyclic encoding,
 * parity(x) = data(x) x^NP mod g(x); bit 0 (BE) = highest-degree coefficient */
void orc_bch_encode(const uint8_t *data, uint8_t *parity)
{
	uint8_t g[NP + 1], r[NP];
	genpoly(g);
	memset(r, 0, sizeof(r));   /* r[d] = coeff of x^d */
	for (int i = 0; i < K; ++i) {
		int fb = orc_get_be_bit(data, i) ^ r[NP - 1];
		memmove(r + 1, r, NP - 1);
		r[0] = 0;
		if (fb)
			for (int d = 0; d < NP; ++d)
				r[d] ^= g[d];
	}
	memset(parity, 0, (NP + 7) / 8);
	for (int i = 0; i < NP; ++i)
		orc_set_be_bit(parity, i, r[NP - 1 - i]);
}

/* decode.cc:378-384 BoseChaudhuriHocquenghemGenerator<255,71>::matrix(genmat,
 * systematic=true, ...): genmat[N*j+i], K rows, systematic [I | P].  The
 * systematic generator of a code with information set {0..K-1} is unique:
 * row j = codeword of the unit message e_j. */
void orc_bch_genmat(int8_t *genmat)
{
	for (int j = 0; j < K; ++j) {
		uint8_t data[(K + 7) / 8] = { 0 }, parity[(NP + 7) / 8];
		orc_set_be_bit(data, j, 1);
		orc_bch_encode(data, parity);
		for (int i = 0; i < K; ++i)
			genmat[N * j + i] = (int8_t)(i == j);
		for (int i = 0; i < NP; ++i)
			genmat[N * j + K + i] = (int8_t)orc_get_be_bit(parity, i);
	}
}

/* ---- OSD(255,71) order 4 -------------------------------------------------- */
typedef struct { uint64_t w[4]; } row_t;
static inline int rget(const row_t *r, int i) { return (int)((r->w[i >> 6] >> (i & 63)) & 1); }
static inline void rflip(row_t *r, int i) { r->w[i >> 6] ^= 1ull << (i & 63); }
static inline void rxor(row_t *a, const row_t *b) { for (int i = 0; i < 4; ++i) a->w[i] ^= b->w[i]; }
static void swap_cols(row_t *G, int a, int b)
{
	for (int j = 0; j < K; ++j)
		if (rget(&G[j], a) != rget(&G[j], b)) {
			rflip(&G[j], a);
			rflip(&G[j], b);
		}
}

int orc_osd_decode(uint8_t *hard, const int8_t *soft, const int8_t *genmat)
{
	int perm[N];
	int8_t rel[N];
	/* reliabilities |max(soft,-127)|, most reliable first (stable) */
	for (int i = 0; i < N; ++i) {
		int v = soft[i] < -127 ? -127 : soft[i];
		rel[i] = (int8_t)(v < 0 ? -v : v);
	}
	for (int c = 0; c < N; ++c) {
		int j = c;
		while (j > 0 && rel[perm[j - 1]] < rel[c]) {
			perm[j] = perm[j - 1];
			--j;
		}
		perm[j] = c;
	}
	row_t G[K];
	memset(G, 0, sizeof(G));
	for (int j = 0; j < K; ++j)
		for (int i = 0; i < N; ++i)
			if (genmat[N * j + perm[i]])
				rflip(&G[j], i);
	/* row echelon with column swaps when a column has no pivot */
	for (int k = 0; k < K; ++k) {
		for (int j = k; j < K; ++j) {
			if (rget(&G[j], k)) {
				if (j != k) { row_t t = G[j]; G[j] = G[k]; G[k] = t; }
				break;
			}
		}
		for (int j = k + 1; !rget(&G[k], k) && j < N; ++j) {
			for (int h = k; h < K; ++h) {
				if (rget(&G[h], j)) {
					int t = perm[k]; perm[k] = perm[j]; perm[j] = t;
					swap_cols(G, k, j);
					if (h != k) { row_t r = G[h]; G[h] = G[k]; G[k] = r; }
					break;
				}
			}
		}
		for (int j = k + 1; j < K; ++j)
			if (rget(&G[j], k))
				rxor(&G[j], &G[k]);
	}
	/* systematic: clear above the diagonal */
	for (int k = K - 1; k; --k)
		for (int j = 0; j < k; ++j)
			if (rget(&G[j], k))
				rxor(&G[j], &G[k]);
	int x[256];
	for (int i = 0; i < N; ++i) {
		int v = soft[perm[i]];
		x[i] = v < -127 ? -127 : v;
	}
	x[255] = 0;
	/* byte-sliced lookup: T[g][b] = sum of x over the set bits of byte g */
	static _Thread_local int T[32][256];
	int X = 0;
	for (int g = 0; g < 32; ++g) {
		T[g][0] = 0;
		for (int b = 1; b < 256; ++b) {
			int low = b & -b, bit = __builtin_ctz((unsigned)b);
			T[g][b] = T[g][b ^ low] + x[8 * g + bit];
		}
		X += T[g][255];
	}
	#define METRIC(c) (X - 2 * metric_sum(&(c)))
	/* order-0 codeword: hard decisions on the K most reliable positions */
	row_t cw;
	memset(&cw, 0, sizeof(cw));
	for (int i = 0; i < K; ++i)
		if (x[i] < 0)
			rxor(&cw, &G[i]);
	row_t cand = cw, bestcw = cw;
	int best, next = -1;
	{
		int s = 0;
		const uint8_t *p = (const uint8_t *)cw.w;
		for (int g = 0; g < 32; ++g) s += T[g][p[g]];
		best = X - 2 * s;
	}
	#define UPDATE() do { \
		int s = 0; const uint8_t *p = (const uint8_t *)cand.w; \
		for (int g = 0; g < 32; ++g) s += T[g][p[g]]; \
		int met = X - 2 * s; \
		if (met > best) { next = best; best = met; bestcw = cand; } \
		else if (met > next) { next = met; } \
	} while (0)
	for (int a = 0; a < K; ++a) {
		rxor(&cand, &G[a]);
		UPDATE();
		for (int b = a + 1; b < K; ++b) {
			rxor(&cand, &G[b]);
			UPDATE();
			for (int c = b + 1; c < K; ++c) {
				rxor(&cand, &G[c]);
				UPDATE();
				for (int d = c + 1; d < K; ++d) {
					rxor(&cand, &G[d]);
					UPDATE();
					rxor(&cand, &G[d]);
				}
				rxor(&cand, &G[c]);
			}
			rxor(&cand, &G[b]);
		}
		rxor(&cand, &G[a]);
	}
	#undef UPDATE
	#undef METRIC
	memset(hard, 0, 32);
	for (int i = 0; i < N; ++i)
		orc_set_be_bit(hard, perm[i], rget(&bestcw, i));
	return best != next;
}
