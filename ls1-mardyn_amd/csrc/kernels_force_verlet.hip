// kernels_force_verlet.hip — neighbour-list ("Verlet list") variant of the single-centre LJ fast path.
//
// The candidate SEARCH is what the per-step kernels (kernels_force_lj.hip) spend about half of their instructions on,
// and they redo it every step.  Here it is done once per list lifetime:
//   * the cell grid is built with cutoff rc + skin, molecules are binned and the halo copies generated as usual;
//   * BUILD (vl_mode 1): per brick, every owned molecule's neighbours within rc + skin are stored as u16 LDS byte
//     offsets of the brick's staged region (the region-linear staging order is a pure function of the cell table, which
//     stays FROZEN until the next rebuild);
//   * REUSE (vl_mode 2), every step: the brick region is staged into LDS exactly as at build time (same order, current
//     positions), every lane walks its stored list and evaluates the exact FP64 pair test r^2 < rc^2 and the LJ body.
//     No search, no re-binning, no halo regeneration (the halo positions are refreshed from their source molecules).
// The lists stay complete while no molecule has moved more than skin / 2 since the build; the fused epilogue reports
// max |v| of the step, the reduction accumulates the displacement bound sum(dt * vmax) on the device and publishes
// "rebuild needed" to the host (ls1hip_run polls it) — results never depend on a guessed rebuild interval.
// Precedent in the reference: the Verlet-list containers of AutoPas with a rebuild frequency and skin
// (particleContainer/AutoPasContainer.cpp:281-346); pair arithmetic = VectorizedCellProcessor::_loopBodyLJ
// (particleContainer/adapter/VectorizedCellProcessor.cpp:173-226), masks as in kernels_force_lj.hip.
//
// Mapping (CDNA4): one 512-thread workgroup per brick of 4 x 4 x 2 cells (~500 owned molecules at liquid density with
// rc + skin cells), region 6 x 6 x 4 cells (~2260 molecules, 54 KB of FP64 x / y / z in LDS, two workgroups per CU).
// ONE lane per owned molecule, owned molecules enumerated densely over the brick (no per-cell tile padding); a wave =
// a TILE of 64 consecutive owned molecules.  List layout in HBM: [brick][tile][word][lane] u64, a word = 4 list
// entries: a wave reads one word row with ONE coalesced 512-B load per 4 pairs per lane, prefetched two rows ahead.
// The lists of the 64 lanes of a tile have nearly equal length in a liquid (72 +- 3 at skin = 0.12 rc), so the
// wave-maximum trip count wastes < 10 % — against 40 % (quarter lists) + 22 % (cell-padded owned tiles) in the
// per-step MFMA kernel.  Short lanes are padded with an offset that points at a far-away dummy position (masked by the
// exact cutoff test like any other out-of-range entry).
#include "common.hpp"
#include "brick.hpp"

namespace ls1 {

constexpr int VBX = 4, VBY = 4, VBZ = 2;
constexpr int VNT = 512;
constexpr int VCAPJ = 2816;  // staged molecules per brick region (67.8 KB of x, y, z)
constexpr int VMAXT = 10;    // tiles per brick with stored lists (640 owned molecules); further tiles: direct evaluation
constexpr int VMAXW = 24;    // words per lane = 96 list entries
constexpr double VFAR = 1.0e30;  // dummy position: r^2 ~ 1e60 fails every cutoff test, all LJ terms underflow to 0

void verlet_geometry(const Grid& g, long* nbricks, size_t* words_per_brick, size_t* tiles_per_brick) {
	const long nbx = (g.box[0] + VBX - 1) / VBX, nby = (g.box[1] + VBY - 1) / VBY, nbz = (g.box[2] + VBZ - 1) / VBZ;
	*nbricks = nbx * nby * nbz;
	*words_per_brick = (size_t)VMAXT * VMAXW * 64;
	*tiles_per_brick = VMAXT;
}

__device__ __forceinline__ double v_rcp(double d) {
	double x = __builtin_amdgcn_rcp(d);
	double e = fma(-d, x, 1.0);
	x = fma(x, e, x);
	e = fma(-d, x, 1.0);
	return fma(x, e, x);
}

struct VAcc {
	double fx, fy, fz, slj, vir;
	uint32_t nin;
};

// listed pair: exact strict cutoff test; a pair outside gets r2 := 1e300 and every LJ term underflows to exactly 0
template <bool SHIFT>
__device__ __forceinline__ void v_pair(double xi, double yi, double zi, double xj, double yj, double zj, double rc2, double eps24,
									   double sig2, VAcc& a) {
	const double dx = xi - xj, dy = yi - yj, dz = zi - zj;
	const double r2 = fma(dz, dz, fma(dy, dy, dx * dx));
	const bool in = r2 < rc2;
	const double inv = v_rcp(in ? r2 : 1.0e300);
	const double lj2 = sig2 * inv;
	const double lj6 = lj2 * lj2 * lj2;
	const double lj12 = lj6 * lj6;
	const double lj12m6 = lj12 - lj6;
	const double fac = eps24 * inv * (lj12 + lj12m6);
	a.fx = fma(fac, dx, a.fx);
	a.fy = fma(fac, dy, a.fy);
	a.fz = fma(fac, dz, a.fz);
	a.slj += lj12m6;
	if (SHIFT) a.nin += in ? 1u : 0u;
	a.vir = fma(fac, r2, a.vir);
}

// masked pair for the direct (list-free) fallbacks
__device__ __forceinline__ void v_pair_direct(double xi, double yi, double zi, double xj, double yj, double zj, double rc2,
											  double eps24, double sig2, VAcc& a) {
	const double dx = xi - xj, dy = yi - yj, dz = zi - zj;
	const double r2 = fma(dz, dz, fma(dy, dy, dx * dx));
	const bool in = r2 < rc2;
	const double inv = v_rcp(in ? r2 : 1.0e300);
	const double lj2 = sig2 * inv;
	const double lj6 = lj2 * lj2 * lj2;
	const double lj12 = lj6 * lj6;
	const double lj12m6 = lj12 - lj6;
	const double fac = eps24 * inv * (lj12 + lj12m6);
	a.fx = fma(fac, dx, a.fx);
	a.fy = fma(fac, dy, a.fy);
	a.fz = fma(fac, dz, a.fz);
	a.slj += lj12m6;
	a.nin += in ? 1u : 0u;
	a.vir = fma(fac, r2, a.vir);
}

__device__ __forceinline__ double v_wave_sum(double v) {
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
	return v;
}
__device__ __forceinline__ double v_wave_max(double v) {
	for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o));
	return v;
}

template <bool BUILD, bool SHIFT>
__global__ void __launch_bounds__(VNT, 4) k_force_lj_verlet(ForceParams P, int nbx, int nby, int nbz) {
	constexpr int HW = 1, BX = VBX, BY = VBY, BZ = VBZ, NT = VNT;
	constexpr int RX = BX + 2, RY = BY + 2, RZ = BZ + 2;
	constexpr int NRC = RX * RY * RZ;
	constexpr int NBC = BX * BY * BZ;
	constexpr int NW = NT / 64;
	constexpr int CAPS = VCAPJ + 8;  // +8: the dummy slot and the overrun of unrolled row reads
	static_assert(NRC <= NT * 4, "region too large for the block scan");
	static_assert(2 * CAPS * 8 < 65536, "y / z are addressed as constant offsets from the x entry");
	__shared__ double spos[3 * CAPS];
	double* const sx = spos;
	double* const sy = spos + CAPS;
	double* const sz = spos + 2 * CAPS;
	__shared__ uint32_t cstart[NRC + 1];
	__shared__ uint32_t gbeg[NRC];
	__shared__ uint32_t bstart[NBC + 1];
	__shared__ uint32_t wsum[NW];
	__shared__ double red[NW][4];
	__shared__ uint16_t win[BUILD ? 8 * NT : 1];  // BUILD: per-lane window of two list words, slot-major

	const int tid = threadIdx.x;
	const int lane = tid & 63, wv = tid >> 6;
	const BrickSel bs = brick_select<HW, BX, BY, BZ>(P, nbx, nby, nbz);
	if (!bs.live) {  // uniform per workgroup
		if (!BUILD && tid < 4) P.partials[(size_t)blockIdx.x * 4 + tid] = 0.;
		return;
	}
	const int ex = bs.ex, ey = bs.ey, ez = bs.ez;
	brick_region_table<NT, HW, RX, RY, RZ>(P, bs, cstart, gbeg);
	__syncthreads();
	block_scan_lds<NT>(cstart, NRC, wsum);
	const uint32_t total = cstart[NRC];
	for (int c = tid; c < NBC; c += NT) {
		const int cx = c % BX, cy = (c / BX) % BY, cz = c / (BX * BY);
		uint32_t n = 0;
		if (cx < ex && cy < ey && cz < ez) {
			const int rcell = ((cz + HW) * RY + (cy + HW)) * RX + (cx + HW);
			n = cstart[rcell + 1] - cstart[rcell];
		}
		bstart[c] = n;
	}
	__syncthreads();
	block_scan_lds<NT>(bstart, NBC, wsum);
	const uint32_t n_i = bstart[NBC];
	const bool staged = total <= (uint32_t)VCAPJ;
	if (staged) {
		// 16 lanes per region cell: all global loads of a thread are independent (as in k_force_lj_mfma)
		for (int c = tid >> 4; c < NRC; c += NT / 16) {
			const uint32_t n = cstart[c + 1] - cstart[c], s0 = cstart[c], g0 = gbeg[c];
			for (uint32_t k = (uint32_t)tid & 15u; k < n; k += 16u) {
				const uint32_t g = g0 + k, s = s0 + k;
				sx[s] = P.x[g];
				sy[s] = P.y[g];
				sz[s] = P.z[g];
			}
		}
		if (tid < 8) {
			sx[total + tid] = VFAR;
			sy[total + tid] = VFAR;
			sz[total + tid] = VFAR;
		}
	}
	__syncthreads();
	const double rc2 = P.rc2, eps24 = P.eps24, sig2 = P.sig2, shift6 = P.shift6;
	const uint64_t dummy = (uint64_t)(total * 8u) * 0x0001000100010001ull;  // four entries pointing at the far-away slot
	double u6_tot = 0., vir_tot = 0., kin_tot = 0., vmax2 = 0.;

	for (uint32_t base = 0, pass = 0; base < n_i; base += NT, ++pass) {
		const uint32_t it = base + (uint32_t)tid;
		const bool active = it < n_i;
		const uint32_t tile = pass * NW + (uint32_t)wv;  // wave-uniform
		const bool listed = staged && tile < (uint32_t)VMAXT;
		const size_t tile_g = (size_t)bs.id * VMAXT + tile;
		uint32_t ii = total, gi = 0;
		int rowbase = 0;
		if (active) {
			int lo = 0, hi = NBC;
			while (hi - lo > 1) {
				const int mid = (lo + hi) >> 1;
				if (bstart[mid] <= it) lo = mid;
				else hi = mid;
			}
			const int cx = lo % BX, cy = (lo / BX) % BY, cz = lo / (BX * BY);
			const int rcell = ((cz + HW) * RY + (cy + HW)) * RX + (cx + HW);
			const uint32_t k = it - bstart[lo];
			ii = cstart[rcell] + k;
			gi = gbeg[rcell] + k;
			rowbase = (cz * RY + cy) * RX + cx;  // first cell of neighbour row 0 (region coordinates: own cell minus one)
		}
		if (BUILD) {
			// ---- list construction: 9 contiguous candidate rows per molecule, exact FP64 distance test with rc + skin ----
			if (!listed) continue;  // wave-uniform: tiles beyond the list capacity / unstaged bricks are evaluated directly
			uint64_t* const wp = P.vl_words + tile_g * VMAXW * 64 + lane;
			uint32_t cnt = 0;
			if (active) {
				const double xi = sx[ii], yi = sy[ii], zi = sz[ii];
				const double rcs2 = P.vl_rc2;
				char* const wb = reinterpret_cast<char*>(win) + tid * 2;  // slot s of this lane: wb + s * NT * 2
				for (int row = 0; row < 9; ++row) {
					const int r0 = rowbase + (row / 3) * (RY * RX) + (row % 3) * RX;
					const uint32_t jb = cstart[r0], je = cstart[r0 + 3];
					for (uint32_t j0 = jb; j0 < je; j0 += 4) {
						const uint32_t before = cnt;
#pragma unroll
						for (int u = 0; u < 4; ++u) {
							const uint32_t j = j0 + u;  // reads past je stay inside the padded staging arrays
							const double dx = xi - sx[j], dy = yi - sy[j], dz = zi - sz[j];
							const double r2 = fma(dz, dz, fma(dy, dy, dx * dx));
							const bool hit = (r2 < rcs2) & (j < je) & (j != ii);
							*reinterpret_cast<uint16_t*>(wb + (cnt & 7u) * (NT * 2)) = (uint16_t)(j * 8u);
							cnt += hit ? 1u : 0u;
						}
						if ((cnt >> 2) != (before >> 2)) {  // a word of four entries is complete: flush it
							const uint32_t w = before >> 2, s0 = (w & 1u) * 4u;
							const uint64_t e0 = *reinterpret_cast<uint16_t*>(wb + (s0 + 0) * (NT * 2));
							const uint64_t e1 = *reinterpret_cast<uint16_t*>(wb + (s0 + 1) * (NT * 2));
							const uint64_t e2 = *reinterpret_cast<uint16_t*>(wb + (s0 + 2) * (NT * 2));
							const uint64_t e3 = *reinterpret_cast<uint16_t*>(wb + (s0 + 3) * (NT * 2));
							if (w < (uint32_t)VMAXW) wp[(size_t)w * 64] = e0 | (e1 << 16) | (e2 << 32) | (e3 << 48);
						}
					}
				}
				// the incomplete last word, padded with the dummy entry
				const uint32_t w = cnt >> 2, rem = cnt & 3u;
				if (rem && w < (uint32_t)VMAXW) {
					const uint32_t s0 = (w & 1u) * 4u;
					uint64_t word = dummy;
					for (uint32_t u = 0; u < rem; ++u) {
						const uint64_t e = *reinterpret_cast<uint16_t*>(wb + (s0 + u) * (NT * 2));
						word = (word & ~(0xffffull << (16 * u))) | (e << (16 * u));
					}
					wp[(size_t)w * 64] = word;
				}
			}
			// words in use by this tile = wave maximum; shorter lanes are padded with dummy words up to it
			uint32_t mine = (cnt + 3u) >> 2, nw = mine;
			for (int o = 32; o > 0; o >>= 1) nw = max(nw, (uint32_t)__shfl_xor((int)nw, o));
			if (nw > (uint32_t)VMAXW) {
				if (lane == 0) P.vl_nw[tile_g] = 0xff;  // overflow (very dense neighbourhood): direct evaluation every step
			} else {
				for (uint32_t w = mine; w < nw; ++w) wp[(size_t)w * 64] = dummy;
				if (lane == 0) P.vl_nw[tile_g] = (uint8_t)nw;
			}
			continue;
		}
		// ---- force evaluation ------------------------------------------------------------------------------------------
		VAcc acc = {0., 0., 0., 0., 0., 0u};
		uint32_t nw = 0xffu;
		if (listed) nw = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.vl_nw[tile_g]);
		if (nw != 0xffu) {
			const double xi = sx[ii], yi = sy[ii], zi = sz[ii];  // inactive lanes: the dummy slot (their words are all dummies)
			const uint64_t* const wp = P.vl_words + tile_g * VMAXW * 64 + lane;
			const char* const sxb = reinterpret_cast<const char*>(sx);
			uint64_t w0 = nw > 0 ? wp[0] : 0ull, w1 = nw > 1 ? wp[64] : 0ull;
			for (uint32_t k = 0; k < nw; ++k) {
				const uint64_t cur = w0;
				w0 = w1;
				if (k + 2 < nw) w1 = wp[(size_t)(k + 2) * 64];
				const uint32_t lo32 = (uint32_t)cur, hi32 = (uint32_t)(cur >> 32);
				const uint32_t o0 = lo32 & 0xffffu, o1 = lo32 >> 16, o2 = hi32 & 0xffffu, o3 = hi32 >> 16;
				const double x0 = *reinterpret_cast<const double*>(sxb + o0), y0 = *reinterpret_cast<const double*>(sxb + o0 + CAPS * 8),
							 z0 = *reinterpret_cast<const double*>(sxb + o0 + 2 * CAPS * 8);
				const double x1 = *reinterpret_cast<const double*>(sxb + o1), y1 = *reinterpret_cast<const double*>(sxb + o1 + CAPS * 8),
							 z1 = *reinterpret_cast<const double*>(sxb + o1 + 2 * CAPS * 8);
				const double x2 = *reinterpret_cast<const double*>(sxb + o2), y2 = *reinterpret_cast<const double*>(sxb + o2 + CAPS * 8),
							 z2 = *reinterpret_cast<const double*>(sxb + o2 + 2 * CAPS * 8);
				const double x3 = *reinterpret_cast<const double*>(sxb + o3), y3 = *reinterpret_cast<const double*>(sxb + o3 + CAPS * 8),
							 z3 = *reinterpret_cast<const double*>(sxb + o3 + 2 * CAPS * 8);
				v_pair<SHIFT>(xi, yi, zi, x0, y0, z0, rc2, eps24, sig2, acc);
				v_pair<SHIFT>(xi, yi, zi, x1, y1, z1, rc2, eps24, sig2, acc);
				v_pair<SHIFT>(xi, yi, zi, x2, y2, z2, rc2, eps24, sig2, acc);
				v_pair<SHIFT>(xi, yi, zi, x3, y3, z3, rc2, eps24, sig2, acc);
			}
		} else if (active && staged) {
			// no stored list for this tile (list overflow, or more owned molecules than the list capacity covers)
			const double xi = sx[ii], yi = sy[ii], zi = sz[ii];
			for (int row = 0; row < 9; ++row) {
				const int r0 = rowbase + (row / 3) * (RY * RX) + (row % 3) * RX;
				const uint32_t jb = cstart[r0], je = cstart[r0 + 3];
				for (uint32_t j = jb; j < je; ++j)
					if (j != ii) v_pair_direct(xi, yi, zi, sx[j], sy[j], sz[j], rc2, eps24, sig2, acc);
			}
		} else if (active) {
			// shell does not fit the staging area (pathological density): same arithmetic straight from global memory
			const double xi = P.x[gi], yi = P.y[gi], zi = P.z[gi];
			for (int row = 0; row < 9; ++row) {
				const int r0 = rowbase + (row / 3) * (RY * RX) + (row % 3) * RX;
				for (int c = r0; c < r0 + 3; ++c) {
					const uint32_t gb = gbeg[c], n = cstart[c + 1] - cstart[c];
					for (uint32_t j = gb; j < gb + n; ++j)
						if (j != gi) v_pair_direct(xi, yi, zi, P.x[j], P.y[j], P.z[j], rc2, eps24, sig2, acc);
				}
			}
		}
		if (active) {
			const double fx = acc.fx, fy = acc.fy, fz = acc.fz;
			if (!P.fuse) {
				P.Fx[gi] = fx;
				P.Fy[gi] = fy;
				P.Fz[gi] = fz;
			} else {
				// lj_store of kernels_force_lj.hip (upd_postF, then upd_preF of the next step) + the drift speed of this step
				const double k = P.dt_inv2m;
				double vx = P.vx[gi] + k * fx;
				double vy = P.vy[gi] + k * fy;
				double vz = P.vz[gi] + k * fz;
				kin_tot += P.mass * (vx * vx + vy * vy + vz * vz);
				vx += k * fx;
				vy += k * fy;
				vz += k * fz;
				P.vx[gi] = vx;
				P.vy[gi] = vy;
				P.vz[gi] = vz;
				vmax2 = fmax(vmax2, vx * vx + vy * vy + vz * vz);
				P.Fx[gi] = P.x[gi] + P.dt * vx;
				P.Fy[gi] = P.y[gi] + P.dt * vy;
				P.Fz[gi] = P.z[gi] + P.dt * vz;
			}
			// every in-range pair adds eps24 * (lj12 - lj6) + shift6 to 6 U
			u6_tot += fma(eps24, acc.slj, shift6 * (double)acc.nin);  // nin is only counted where shift6 != 0 or in the fallbacks
			vir_tot += acc.vir;
		}
	}
	if (BUILD) return;
	// every ordered pair contributes half of the pair's U and virial (see kernels_force.hip)
	const double u = v_wave_sum(0.5 * u6_tot), v = v_wave_sum(0.5 * vir_tot), kn = v_wave_sum(kin_tot), vm = v_wave_max(vmax2);
	if (lane == 0) {
		red[wv][0] = u;
		red[wv][1] = kn;
		red[wv][2] = vm;
		red[wv][3] = v;
	}
	__syncthreads();
	if (tid == 0) {
		double* out = P.partials + (size_t)blockIdx.x * 4;
		double su = 0., sk = 0., sm = 0., sv = 0.;
		for (int i = 0; i < NW; ++i) {
			su += red[i][0];
			sk += red[i][1];
			sm = fmax(sm, red[i][2]);
			sv += red[i][3];
		}
		out[0] = su;
		out[1] = sk;  // fused mode: sum m v^2 (see k_force_lj_brick)
		out[2] = sm;  // fused mode: max |v_drift|^2 of the brick's molecules (combined by max in the reduction)
		out[3] = sv;
	}
}

bool launch_force_verlet(const ForceParams& p_in, hipStream_t s, uint32_t* nblocks, size_t partials_cap, BrickLists* bl) {
	ForceParams p = p_in;
	const Grid& g = p.g;
	if (g.hw != 1 || !p.vl_words || !p.vl_nw) return false;
	const int nbx = (g.box[0] + VBX - 1) / VBX, nby = (g.box[1] + VBY - 1) / VBY, nbz = (g.box[2] + VBZ - 1) / VBZ;
	if ((long)nbx * nby * nbz <= 0 || (long)nbx * nby * nbz > 0x7ffffff0L) return false;
	if (p.vl_mode == 1) p.which = 0;  // the lists are always built for all bricks
	const long nb = plan_bricks(p, bl, VBX, VBY, VBZ, nbx, nby, nbz);
	if ((size_t)nb > partials_cap) return false;
	*nblocks = (uint32_t)nb;
	if (nb == 0) return true;
	const bool shift = p.shift6 != 0.;
	if (p.vl_mode == 1) {
		hipLaunchKernelGGL((k_force_lj_verlet<true, false>), dim3((uint32_t)nb), dim3(VNT), 0, s, p, nbx, nby, nbz);
	} else if (shift) {
		hipLaunchKernelGGL((k_force_lj_verlet<false, true>), dim3((uint32_t)nb), dim3(VNT), 0, s, p, nbx, nby, nbz);
	} else {
		hipLaunchKernelGGL((k_force_lj_verlet<false, false>), dim3((uint32_t)nb), dim3(VNT), 0, s, p, nbx, nby, nbz);
	}
	return true;
}

}  // namespace ls1
