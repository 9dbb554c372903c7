// brick.hpp — helpers shared by the brick-tiled force kernels (kernels_force_lj.hip, kernels_force_ms.hip).
#pragma once
#include "common.hpp"

namespace ls1 {

// block-wide exclusive scan of `n` (<= NT*4) u32 values held in LDS array v[0..n) -> v becomes exclusive prefix,
// v[n] = total.  All threads must call.
template <int NT>
__device__ __forceinline__ void block_scan_lds(uint32_t* v, int n, uint32_t* wsum) {
	const int t = threadIdx.x;
	uint32_t a[4], s = 0;
	for (int k = 0; k < 4; ++k) {
		const int i = t * 4 + k;
		a[k] = (i < n) ? v[i] : 0u;
		s += a[k];
	}
	uint32_t inc = s;
	const int lane = t & 63, w = t >> 6;
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t u = __shfl_up(inc, o);
		if (lane >= o) inc += u;
	}
	if (lane == 63) wsum[w] = inc;
	__syncthreads();
	uint32_t base = 0;
	for (int i = 0; i < w; ++i) base += wsum[i];
	uint32_t ex = base + inc - s;
	__syncthreads();
	for (int k = 0; k < 4; ++k) {
		const int i = t * 4 + k;
		if (i < n) v[i] = ex;
		ex += a[k];
	}
	if (t == NT - 1) v[n] = ex;  // last thread's running value = total (its items beyond n contribute 0)
	__syncthreads();
}


// number of workgroups of a brick traversal (multiple of 8 for the XCD-aware order); fills p.brick_list / p.n_list for
// the inner (which = 1) / boundary (2) passes from the host-built lists (kernels_force_lj.hip)
long plan_bricks(ForceParams& p, BrickLists* bl, int BX, int BY, int BZ, int nbx, int nby, int nbz);

}  // namespace ls1
