// polar_common.h -- what the two polar decoders share (k_polar.hip: the list decoder; k_sc.hip: its sign-following path alone):
// the min-sum kernels of decode.cc:201's PolarListDecoder and the raw buffer accesses of their level stores.
#pragma once
#include "dev_common.h"

namespace rx {

// three VALU: v_xor, v_med3_f32 with |.| modifiers (median of (|a|, |b|, 0) = the smaller magnitude; unlike
// fminf no canonicalising v_max is emitted), v_and_or.  A zero result may carry a minus sign; no consumer can tell.
__device__ __forceinline__ float f_minsum(float a, float b)
{
	const uint32_t sgn = (__float_as_uint(a) ^ __float_as_uint(b)) & 0x80000000u;
	const float m = __builtin_amdgcn_fmed3f(fabsf(a), fabsf(b), 0.f);
	return __uint_as_float(sgn | __float_as_uint(m));
}
__device__ __forceinline__ float g_add(float a, float b, int u) { return u ? b - a : a + b; }

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, int bytes)
{
	return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
template <int AUX = 0> __device__ __forceinline__ float bload(rsrc_t r, int voff, int soff) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX)); }
template <int AUX = 0> __device__ __forceinline__ void bstore(rsrc_t r, int voff, int soff, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, AUX); }
__device__ __forceinline__ int bload_u8(rsrc_t r, int voff, int soff) { return (int)__builtin_amdgcn_raw_buffer_load_b8(r, voff, soff, 0); }

}  // namespace rx
