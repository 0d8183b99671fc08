// tables.h -- host-side constant tables (see tables.cpp)
#pragma once
#include <cstdint>
#include <vector>
#include "kernels.h"

namespace rx {

struct HostTables {
	std::vector<cf> tw_sym, sc_kern, tw_sym4;   // symbol_len roots, S&C kernel (symbol_len/2), 4*symbol_len roots
	std::vector<cf> tw_symc;                    // the symbol_len plan's compact stage tables (dev_common.h: FftPlan::fill)
	std::vector<float> mls1_nrz, mls0_nrz, mls2_nrz;
	std::vector<uint32_t> frozen, genmat_bits, crc32_tab, crc32_shift168, crc32_adv, info_compress, frozen_t;
	std::vector<uint16_t> info_pos;
	std::vector<uint8_t> osd_pairs, osd_triples, scramble, node_lev, node_lev64, node_lev32;
	FrontCoef front;
};

void build_tables(HostTables &t, int rate);

}  // namespace rx
