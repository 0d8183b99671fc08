// tables.cpp -- host-side construction of the constant tables the kernels read.
// Everything is generated from first principles (nothing is copied from the reference's
// polar_tables.hh; nothing links against oracle/).
#include "tables.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

namespace rx {

// ---- CODE::MLS (decode.cc:238,407): Galois LFSR, register starts at 1
struct Mls {
	int poly, test, reg;
	explicit Mls(int p) : poly(p), reg(1)
	{
		unsigned n = (unsigned)p;
		n |= n >> 1; n |= n >> 2; n |= n >> 4; n |= n >> 8; n |= n >> 16;
		test = (int)((n ^ (n >> 1)) >> 1);
	}
	int next()
	{
		int fb = (reg & test) != 0;
		reg <<= 1;
		reg ^= fb * poly;
		return fb;
	}
};

// ---- frozen mask: freezer.cc:14-32 recipe (BEC construction in long double, the K most
// reliable synthetic channels are unfrozen).  N = 64800: polar_tables.hh:2 ; N = 64512: polar_tables.hh:1
static void frozen_mask(uint32_t *frozen, int N)
{
	const int M = 16, LEN = 1 << M, K = 43040 + 32;
	long double erasure_probability = (long double)(N - K) / N;
	double design_SNR = 10 * std::log10(-std::log((double)erasure_probability));
	double better_SNR = design_SNR + 1.59175;
	long double p = expl(-(long double)std::pow(10.0, better_SNR / 10));
	std::vector<long double> a(1, p), b;
	for (int m = 0; m < M; ++m) {
		b.resize(a.size() * 2);
		for (size_t i = 0; i < a.size(); ++i) {
			b[2 * i] = 2 * a[i] - a[i] * a[i];
			b[2 * i + 1] = a[i] * a[i];
		}
		a.swap(b);
	}
	std::vector<int> idx(LEN);
	std::iota(idx.begin(), idx.end(), 0);
	std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return a[x] < a[y]; });
	for (int i = 0; i < LEN / 32; ++i)
		frozen[i] = 0xffffffffu;
	for (int i = 0; i < K + LEN - N; ++i)
		frozen[idx[i] / 32] &= ~(1u << (idx[i] % 32));
}

// ---- BCH(255,71): generator polynomial = product of the 24 minimal polynomials of
// decode.cc:379-384; systematic generator rows = [e_j | (x^184 e_j(x)) mod g(x)]
static void bch_genmat_bits(uint32_t *rows /*71*8*/)
{
	static const int minpolys[24] = {
		0x11d, 0x177, 0x1f3, 0x169, 0x1bd, 0x1e7, 0x12b, 0x1d7, 0x013, 0x165, 0x18b, 0x163,
		0x11b, 0x13f, 0x18d, 0x12d, 0x15f, 0x1f9, 0x1c3, 0x139, 0x1a9, 0x01f, 0x187, 0x1b1 };
	const int NP = 184, K = 71;
	std::vector<uint8_t> g(NP + 1, 0);
	g[0] = 1;
	int deg = 0;
	for (int p = 0; p < 24; ++p) {
		int m = minpolys[p], md = 0;
		while (m >> (md + 1)) ++md;
		std::vector<uint8_t> t(NP + 1, 0);
		for (int i = 0; i <= deg; ++i)
			if (g[i])
				for (int j = 0; j <= md; ++j)
					if ((m >> j) & 1) t[i + j] ^= 1;
		deg += md;
		g = t;
	}
	std::memset(rows, 0, sizeof(uint32_t) * K * 8);
	for (int j = 0; j < K; ++j) {
		// message bit j = coefficient of x^(K-1-j); remainder of x^(NP + K-1-j) mod g
		std::vector<uint8_t> r(NP, 0);
		for (int i = 0; i < K; ++i) {
			int fb = (i == j) ^ r[NP - 1];
			for (int d = NP - 1; d > 0; --d) r[d] = r[d - 1];
			r[0] = 0;
			if (fb)
				for (int d = 0; d < NP; ++d) r[d] ^= g[d];
		}
		auto set = [&](int col) { rows[j * 8 + (col >> 5)] |= 1u << (col & 31); };
		set(j);
		for (int i = 0; i < NP; ++i)
			if (r[NP - 1 - i]) set(K + i);
	}
}

static double bessel_i0(double x)
{
	double sum = 1.0, term = 1.0;
	for (int k = 1; k < 64; ++k) {
		term *= (x / (2.0 * k)) * (x / (2.0 * k));
		sum += term;
		if (term < 1e-20 * sum) break;
	}
	return sum;
}
static double kaiser(double a, int n, int N)
{
	double t = 2.0 * n / (double)(N - 1) - 1.0;
	return bessel_i0(M_PI * a * std::sqrt(1.0 - t * t)) / bessel_i0(M_PI * a);
}

void build_tables(HostTables &t, int rate)
{
	const int SL = rate_symbol_len(rate), HS = SL / 2;
	auto roots = [](std::vector<cf> &v, int n) {
		v.resize(n);
		for (int k = 0; k < n; ++k) {
			double a = -2.0 * M_PI * k / (double)n;
			v[k].re = (float)std::cos(a);
			v[k].im = (float)std::sin(a);
		}
	};
	roots(t.tw_sym, SL);
	roots(t.tw_sym4, 4 * SL);
	// the compact table of the symbol_len-point radix plan (dev_common.h FftPlan: all 5s, then 7s, 3s, 4s, a final 2):
	// stage (P, R) owns (R - 1) P consecutive entries, entry (t - 1) P + k = w^(t k symbol_len / (P R))
	t.tw_symc.clear();
	for (int rem = SL, P = 1; rem > 1;) {
		const int R = rem % 5 == 0 ? 5 : rem % 7 == 0 ? 7 : rem % 3 == 0 ? 3 : rem % 4 == 0 ? 4 : 2;
		if (P > 1)
			for (int i = 0; i < (R - 1) * P; ++i)
				t.tw_symc.push_back(t.tw_sym[(size_t)((i / P + 1) * (i % P)) * (size_t)(SL / (P * R))]);
		P *= R;
		rem /= R;
	}
	// decode.cc:236-244 mls0_seq + decode.cc:80-82: kern = conj(FFT_{symbol_len/2}(seq)) / (symbol_len/2)
	{
		std::vector<double> seq(HS, 0.0);
		Mls m0(0x89);
		const int mls0_off = -127 + 1;
		for (int i = 0; i < 127; ++i)
			seq[(i + mls0_off / 2 + HS) % HS] = 1 - 2 * m0.next();
		t.sc_kern.resize(HS);
		for (int k = 0; k < HS; ++k) {
			double re = 0, im = 0;
			for (int n = 0; n < HS; ++n) {
				if (seq[n] == 0.0)
					continue;
				double a = -2.0 * M_PI * ((long)k * n % HS) / (double)HS;
				re += seq[n] * std::cos(a);
				im += seq[n] * std::sin(a);
			}
			t.sc_kern[k].re = (float)(re / (double)HS);
			t.sc_kern[k].im = (float)(-im / (double)HS);
		}
	}
	{
		Mls m1(0x12b);
		t.mls1_nrz.resize(256, 1.f);
		for (int i = 0; i < 255; ++i)
			t.mls1_nrz[i] = (float)(1 - 2 * m1.next());
	}
	{
		Mls m0(0x89), m2(0x951);
		t.mls0_nrz.resize(128, 1.f);
		for (int i = 0; i < 127; ++i)
			t.mls0_nrz[i] = (float)(1 - 2 * m0.next());
		t.mls2_nrz.resize(512);
		for (int i = 0; i < 512; ++i)
			t.mls2_nrz[i] = (float)(1 - 2 * m2.next());
	}
	t.frozen.resize(2 * 2048);
	frozen_mask(t.frozen.data(), 64800);
	frozen_mask(t.frozen.data() + 2048, 64512);
	// uniform sub-trees for the list decoder: for every aligned group of 8 leaves the level of the largest aligned
	// node that starts there and is all frozen (low nibble, 3..7) or all information (high nibble, 3..11); 0 = neither
	t.node_lev.assign(2 * 8192, 0);
	for (int tab = 0; tab < 2; ++tab)
		for (int t8 = 0; t8 < 8192; ++t8) {
			const int t0 = t8 * 8;
			int lev[2] = { 0, 0 };
			for (int kind = 0; kind < 2; ++kind)
				for (int L = 3; L <= (kind == 0 ? 7 : 11); ++L) {
					if ((t0 & ((1 << L) - 1)) || (L > 7 && t0 == 0))
						break;
					bool uniform = true;
					for (int i = t0; i < t0 + (1 << L) && uniform; ++i) {
						const int fr = (t.frozen[tab * 2048 + i / 32] >> (i % 32)) & 1;
						uniform = kind == 0 ? fr == 1 : fr == 0;
					}
					if (!uniform)
						break;
					lev[kind] = L;
				}
			t.node_lev[tab * 8192 + t8] = (uint8_t)(lev[0] | (lev[1] << 4));
		}
	// the same for k_sc's blocks of 64 / 32 leaves (one or two codewords per wave): all frozen up to 128 leaves (the rate-0 step's
	// limit), all information up to the level below the sub-tree k_sc keeps in LDS (4096 / 2048 leaves)
	for (int lb = 5; lb <= 6; ++lb) {
		std::vector<uint8_t> &nl = lb == 6 ? t.node_lev64 : t.node_lev32;
		const int nblk = 65536 >> lb;
		nl.assign(2 * nblk, 0);
		for (int tab = 0; tab < 2; ++tab)
			for (int b = 0; b < nblk; ++b) {
				const int t0 = b << lb;
				int lev[2] = { 0, 0 };
				for (int kind = 0; kind < 2; ++kind)
					for (int L = lb; L <= (kind == 0 ? 7 : lb + 5); ++L) {
						if (t0 & ((1 << L) - 1))
							break;
						bool uniform = true;
						for (int i = t0; i < t0 + (1 << L) && uniform; ++i) {
							const int fr = (t.frozen[tab * 2048 + i / 32] >> (i % 32)) & 1;
							uniform = kind == 0 ? fr == 1 : fr == 0;
						}
						if (!uniform)
							break;
						lev[kind] = L;
					}
				nl[tab * nblk + b] = (uint8_t)(lev[0] | (lev[1] << 4));
			}
	}
	// frozen_t[tab][s][lane] (two words each): the frozen bits of k_sc's sub-tree s (4096 leaves) the way a lane of the one-codeword layout
	// holds that sub-tree - bit x = leaf s * 4096 + x * 64 + q, q = the position lane `lane` holds (k_sc.hip: sc_pos) - for its clean-node test
	t.frozen_t.assign(2 * 16 * 64 * 2, 0);
	for (int tab = 0; tab < 2; ++tab)
		for (int s = 0; s < 16; ++s)
			for (int lane = 0; lane < 64; ++lane) {
				const int q = lane ^ ((lane & 4) ? 3 : 0);
				for (int x = 0; x < 64; ++x) {
					const int i = s * 4096 + x * 64 + q;
					if ((t.frozen[tab * 2048 + i / 32] >> (i % 32)) & 1)
						t.frozen_t[(size_t)((tab * 16 + s) * 64 + lane) * 2 + x / 32] |= 1u << (x % 32);
				}
			}
	t.info_pos.assign(2 * 44096, 0);
	for (int tab = 0; tab < 2; ++tab) {
		int n = 0;
		for (int i = 0; i < 65536; ++i)
			if (!((t.frozen[tab * 2048 + i / 32] >> (i % 32)) & 1))
				t.info_pos[tab * 44096 + n++] = (uint16_t)i;
	}
	// info_compress[tab][w][8]: what message_gather (dev_common.h) needs to take the message bits out of code word w in one go: the
	// word's mask of unfrozen positions m, the five move masks of the parallel-suffix "compress" of x & m (Hacker's Delight 7-4: they depend
	// on m alone), the message bit the word's first unfrozen position is (low 16 bits) and how many it holds (high 16)
	t.info_compress.assign(2 * 2048 * 8, 0);
	for (int tab = 0; tab < 2; ++tab) {
		uint32_t off = 0;
		for (int w = 0; w < 2048; ++w) {
			uint32_t *rec = &t.info_compress[(size_t)(tab * 2048 + w) * 8];
			uint32_t m = ~t.frozen[tab * 2048 + w];
			rec[5] = m;
			uint32_t mk = ~m << 1;
			for (int i = 0; i < 5; ++i) {
				uint32_t mp = mk ^ (mk << 1);
				mp ^= mp << 2; mp ^= mp << 4; mp ^= mp << 8; mp ^= mp << 16;
				const uint32_t mv = mp & m;
				rec[i] = mv;
				m = (m ^ mv) | (mv >> (1 << i));
				mk &= ~mp;
			}
			const uint32_t cnt = (uint32_t)__builtin_popcount(rec[5]);
			rec[6] = off | (cnt << 16);
			off += cnt;
		}
	}
	t.genmat_bits.resize(71 * 8);
	bch_genmat_bits(t.genmat_bits.data());
	t.osd_pairs.clear();
	for (int a = 0; a < 71; ++a)
		for (int b = a + 1; b < 71; ++b) {
			t.osd_pairs.push_back((uint8_t)a);
			t.osd_pairs.push_back((uint8_t)b);
		}
	t.osd_triples.clear();
	for (int c = 2; c < 71; ++c)
		for (int a = 0; a < c; ++a)
			for (int b = a + 1; b < c; ++b) {
				t.osd_triples.push_back((uint8_t)a);
				t.osd_triples.push_back((uint8_t)b);
				t.osd_triples.push_back((uint8_t)c);
			}
	t.crc32_tab.resize(256);
	for (uint32_t j = 0; j < 256; ++j) {
		uint32_t c = j;
		for (int i = 0; i < 8; ++i)
			c = (c >> 1) ^ ((c & 1) * 0xD419CC15u);   // CRC<uint32_t>(0xD419CC15), decode.cc:198
		t.crc32_tab[j] = c;
	}
	t.crc32_shift168.resize(4 * 256);
	for (int b = 0; b < 4; ++b)
		for (uint32_t v = 0; v < 256; ++v) {
			uint32_t c = v << (8 * b);
			for (int i = 0; i < 168; ++i)
				c = (c >> 8) ^ t.crc32_tab[c & 255];
			t.crc32_shift168[b * 256 + v] = c;
		}
	// crc32_adv[k][b]: bit b of a CRC state advanced by the bytes that follow segment k of crc32_wg256's split of the 5384 message
	// bytes (decode.cc:533-541: 43072 bits) - 255 segments of 21 bytes and a last one of 29: the advance is linear over GF(2), so a
	// segment's CRC from a zero state, advanced bit by bit through this table, is its share of the whole CRC
	t.crc32_adv.resize(256 * 32);
	{
		uint32_t cur[32];
		for (int b = 0; b < 32; ++b)
			cur[b] = 1u << b;                                     // behind the last segment: nothing follows
		for (int k = 255; k >= 0; --k) {
			for (int b = 0; b < 32; ++b)
				t.crc32_adv[k * 32 + b] = cur[b];
			const int len = k == 255 ? 29 : 21;                   // the bytes of segment k itself follow segment k - 1
			for (int b = 0; b < 32; ++b)
				for (int i = 0; i < len; ++i)
					cur[b] = (cur[b] >> 8) ^ t.crc32_tab[cur[b] & 255];
		}
	}
	t.scramble.resize(5380);
	{
		uint32_t y = 2463534242u;   // CODE::Xorshift32 default seed, decode.cc:613
		for (int i = 0; i < 5380; ++i) {
			y ^= y << 13; y ^= y >> 17; y ^= y << 5;
			t.scramble[i] = (uint8_t)y;
		}
	}
	// BlockDC::samples(2*(symbol_len+guard_len)) decode.cc:386 ; Hilbert<cmplx,filter_len> decode.cc:172,193
	const float s = (float)(2 * (SL + SL / 8));
	t.front.dc_a = (s - 1.f) / s;
	t.front.dc_b = (1.f + t.front.dc_a) / 2.f;
	const int TAPS = rate_filter_len(rate);
	t.front.reco = (float)kaiser(2.0, (TAPS - 1) / 2, TAPS);
	for (int i = 0; i < 32; ++i)
		t.front.imco[i] = 0.f;
	for (int i = 0; i < (TAPS - 1) / 4; ++i)
		t.front.imco[i] = (float)(kaiser(2.0, (2 * i + 1) + (TAPS - 1) / 2, TAPS) * 2.0 / ((2 * i + 1) * M_PI));
}

}  // namespace rx
