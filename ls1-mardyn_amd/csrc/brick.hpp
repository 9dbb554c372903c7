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


// Which brick does this workgroup own?  XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so every XCD
// (own 4 MB L2) gets a contiguous run of bricks whose shells overlap; the grid is 8 * ceil(nb / 8) workgroups.  With a
// brick list (inner / boundary passes) only the listed bricks are launched: an empty 512-thread, 80 KB workgroup still
// costs ~20 ns of dispatch, which made the boundary pass (18 % of the bricks) 2x too slow.  Without a list the
// inner / boundary filter is evaluated here ("inner": no halo cell inside the brick's shell).
struct BrickSel {
	int x0, y0, z0;  // brick origin in grid cell coordinates
	int ex, ey, ez;  // extent in cells (edge bricks are partial)
	int id;          // linear brick index (bz * nby + by) * nbx + bx — independent of the launch order / pass
	int did;         // index of the brick's neighbour-list data (ForceParams::did_mode)
	bool live;
};
// vb / vgrid: (virtual) workgroup index and grid size — blockIdx.x / gridDim.x for one brick per workgroup; persistent
// kernels walk vb = blockIdx.x + k * gridDim.x over a virtual grid (gridDim.x a multiple of 8 keeps vb % 8 = the XCD)
template <int HW, int BX, int BY, int BZ>
__device__ __forceinline__ BrickSel brick_select_v(const ForceParams& P, int nbx, int nby, int nbz, int vb, int vgrid) {
	const int nb = P.inner_box ? P.inner_n[0] * P.inner_n[1] * P.inner_n[2] : (P.brick_list ? (int)P.n_list : nbx * nby * nbz);
	const int chunk = vgrid / 8;
	const int slot = (vb % 8) * chunk + vb / 8;
	BrickSel b;
	b.live = vb < vgrid && slot < nb;
	b.did = slot;
	if (b.live && P.did_mode == 2) b.did = (int)P.brick_did[slot];
	int bx = 0, by = 0, bz = 0;
	if (b.live && P.inner_box) {
		bx = P.inner_lo[0] + slot % P.inner_n[0];
		by = P.inner_lo[1] + (slot / P.inner_n[0]) % P.inner_n[1];
		bz = P.inner_lo[2] + slot / (P.inner_n[0] * P.inner_n[1]);
	} else if (b.live) {
		const int brick = P.brick_list ? (int)P.brick_list[slot] : slot;
		bx = brick % nbx;
		by = (brick / nbx) % nby;
		bz = brick / (nbx * nby);
	}
	b.id = (bz * nby + by) * nbx + bx;
	if (P.did_mode == 0) b.did = b.id;
	b.x0 = HW + bx * BX;
	b.y0 = HW + by * BY;
	b.z0 = HW + bz * BZ;
	b.ex = min(BX, P.g.dims[0] - HW - b.x0);
	b.ey = min(BY, P.g.dims[1] - HW - b.y0);
	b.ez = min(BZ, P.g.dims[2] - HW - b.z0);
	if (b.live && P.which != 0 && !P.brick_list && !P.inner_box) {
		const bool inner = b.x0 >= 2 * HW && b.y0 >= 2 * HW && b.z0 >= 2 * HW && b.x0 + b.ex <= P.g.dims[0] - 2 * HW &&
						   b.y0 + b.ey <= P.g.dims[1] - 2 * HW && b.z0 + b.ez <= P.g.dims[2] - 2 * HW;
		b.live = (P.which == 1) ? inner : !inner;
	}
	return b;
}
template <int HW, int BX, int BY, int BZ>
__device__ __forceinline__ BrickSel brick_select(const ForceParams& P, int nbx, int nby, int nbz) {
	return brick_select_v<HW, BX, BY, BZ>(P, nbx, nby, nbz, (int)blockIdx.x, (int)gridDim.x);
}

// Cell table of the brick's region (brick + cutoff shell) in region-linear order: gbeg[c] = global index of the first
// molecule of region cell c, cstart[c] = its count (turned into the exclusive prefix by block_scan_lds afterwards).
template <int NT, int HW, int RX, int RY, int RZ>
__device__ __forceinline__ void brick_region_table(const ForceParams& P, const BrickSel& b, uint32_t* cstart, uint32_t* gbeg) {
	for (int c = threadIdx.x; c < RX * RY * RZ; c += NT) {
		const int rx = c % RX, ry = (c / RX) % RY, rz = c / (RX * RY);
		const int gx = b.x0 - HW + rx, gy = b.y0 - HW + ry, gz = b.z0 - HW + rz;
		uint32_t beg = 0, n = 0;
		if (gx < P.g.dims[0] && gy < P.g.dims[1] && gz < P.g.dims[2]) {  // lower bounds are >= 0 by construction
			const int gc = cell_index(P.g, gx, gy, gz);
			beg = P.cell_begin[gc];
			n = P.cell_end[gc] - beg;
		}
		gbeg[c] = beg;
		cstart[c] = n;
	}
}

// number of workgroups of a brick traversal (multiple of 8 for the XCD-aware order); fills p.brick_list / p.n_list for
// the inner (which = 1) / boundary (2) passes from the host-built lists (kernels_force_lj.hip)
long plan_bricks(ForceParams& p, BrickLists* bl, int BX, int BY, int BZ, int nbx, int nby, int nbz, bool blocked_order = false);

}  // namespace ls1
