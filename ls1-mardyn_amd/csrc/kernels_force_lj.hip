// kernels_force_lj.hip — the MI355X fast path: brick-tiled, LDS-staged single-centre Lennard-Jones force kernels.
//
//   k_force_lj_mfma   (default for one cell per cutoff)   FP32 MFMA distance-tile pre-filter + exact FP64 evaluation,
//                                                         4 lanes per molecule — described at its definition below;
//   k_force_lj_brick  (cells-in-cutoff 2, very dense cells) FP64 list kernel, 1 or 2 lanes per molecule — described here.
// Both share the decomposition, the full-shell / no-atomics rule, the XCD-aware brick order, the optional brick lists
// of the inner / boundary passes and the fused integration epilogue (lj_store).
//
// Work decomposition (CDNA4: wave64, 160 KB LDS / CU, FP64 VALU at 4 cycles per wave instruction)
//   * one workgroup per BRICK of BX x BY x BZ cells (about 200 molecules at liquid density);
//   * the brick plus its cutoff shell ((BX+2hw)(BY+2hw)(BZ+2hw) cells, ~1200 molecules) is staged ONCE into LDS
//     in region-linear order, so every x-row of 2hw+1 neighbour cells is one contiguous LDS range — the
//     positions of a molecule are read from HBM once per brick instead of once per neighbour;
//   * list kernel: one lane per owned molecule i (full shell, no atomics, deterministic):
//       phase 1  walks the (2hw+1)^2 neighbour rows and appends the in-range j to a per-lane list in LDS
//                (u16 indices, slot-major so lanes hit consecutive banks) — cheap distance arithmetic only;
//       phase 2  runs the LJ body over the list: every lane of the wave does useful FP64 work in every
//                iteration (about 51 of 335 candidates are in range; without the list the 30-instruction body
//                would run masked-off for the other 85 %).
//   * U/6 and the virial are reduced per workgroup and summed by a deterministic second pass.
// Robustness: a brick whose shell does not fit the LDS staging area, or a lane whose list is full, falls back to
// direct evaluation (same arithmetic), so any density is handled.
//
// Arithmetic per pair = VectorizedCellProcessor::_loopBodyLJ (/root/reference/src/particleContainer/adapter/
// VectorizedCellProcessor.cpp:173-226); masks = CellPairPolicy_/SingleCellPolicy_ (vectorization/
// SIMD_VectorizedCellProcessorHelpers.h:344-422); the reciprocal is v_rcp_f64 + 2 Newton steps (the reference's
// own AVX2 path is rcp_ps + 3 Newton steps, RealVecDouble.h:415-434) — parity is tolerance-based (1e-10).
#include "common.hpp"
#include "brick.hpp"

#include <algorithm>
#include <vector>

namespace ls1 {

constexpr int LTPB = 256;  // threads per workgroup of the list kernel per SPLIT lane (SPLIT = 2 -> 512 threads)
// LDS budget of every variant: two workgroups per CU (2 x 80 KB of the 160 KB).
//   CAPJ  staged molecules per brick region,  CAPL / ROWS  per-lane neighbour list capacity (u16 each)
//   list kernel, SPLIT = 2, hw = 1:  1616 x 24 B = 38.8 KB + 39 rows x 512 x 2 B = 39.9 KB  (4 waves / SIMD)
//   MFMA kernel, 512 threads:        1712 x 28 B = 47.9 KB + 32 rows x 512 x 2 B = 32.8 KB  (4 waves / SIMD)

__device__ __forceinline__ double fast_rcp(double d) {
	double x = __builtin_amdgcn_rcp(d);
	double e = fma(-d, x, 1.0);
	x = fma(x, e, x);
	e = fma(-d, x, 1.0);
	return fma(x, e, x);
}

struct LjAcc {
	double fx, fy, fz, u6, vir;
};

__device__ __forceinline__ void lj_pair(double xi, double yi, double zi, double xj, double yj, double zj, double rc2,
										double eps24, double sig2, double shift6, LjAcc& a) {
	const double dx = xi - xj, dy = yi - yj, dz = zi - zj;
	const double r2 = fma(dz, dz, fma(dy, dy, dx * dx));
	const bool in = (r2 < rc2) & (r2 != 0.0);
	const double inv = fast_rcp(in ? r2 : 1.0);
	const double lj2 = sig2 * inv;
	const double lj6 = lj2 * lj2 * lj2;
	const double lj12 = lj6 * lj6;
	const double lj12m6 = lj12 - lj6;
	double fac = eps24 * inv * (lj12 + lj12m6);
	double ut = fma(eps24, lj12m6, shift6);
	fac = in ? fac : 0.0;
	ut = in ? ut : 0.0;
	a.fx = fma(fac, dx, a.fx);
	a.fy = fma(fac, dy, a.fy);
	a.fz = fma(fac, dz, a.fz);
	a.u6 += ut;
	a.vir = fma(fac, r2, a.vir);
}

// pair known to be in range and distinct: no masks; the potential is accumulated as sum(lj12 - lj6)
__device__ __forceinline__ void lj_pair_in(double xi, double yi, double zi, double xj, double yj, double zj, double eps24,
										   double sig2, LjAcc& a, double& slj) {
	const double dx = xi - xj, dy = yi - yj, dz = zi - zj;
	const double r2 = fma(dz, dz, fma(dy, dy, dx * dx));
	const double inv = fast_rcp(r2);
	const double lj2 = sig2 * inv;
	const double lj6 = lj2 * lj2 * lj2;
	const double lj12 = lj6 * lj6;
	const double lj12m6 = lj12 - lj6;
	const double fac = eps24 * inv * (lj12 + lj12m6);
	a.fx = fma(fac, dx, a.fx);
	a.fy = fma(fac, dy, a.fy);
	a.fz = fma(fac, dz, a.fz);
	slj += lj12m6;
	a.vir = fma(fac, r2, a.vir);
}

// Epilogue of an owned molecule.  Normal mode: store F.  Fused mode (ForceParams::fuse, the reference's reduced-memory
// scheme VCP1CLJRMM.cpp:241-369 + LeapfrogRMM): the force is consumed at once by the post-force kick of this step and
// the pre-force kick + drift of the next (the arithmetic of k_kick_then_kick_drift, operation for operation): v is
// updated in place and the NEW position goes to the F arrays (the old positions are still being read by other bricks),
// from where the next re-binning pass picks it up.  F is never written: -48 B of HBM traffic per molecule and no
// separate integrator pass.
// Returns m v^2 of the molecule after the post-force kick (the summand of Leapfrog::transition2to3's summv2,
// integrators/Leapfrog.cpp:115-131 / FullMolecule::upd_postF :382-385), 0 in the unfused mode.
__device__ __forceinline__ double lj_store(const ForceParams& P, uint32_t gi, const LjAcc& acc) {
	if (!P.fuse) {
		P.Fx[gi] = acc.fx;
		P.Fy[gi] = acc.fy;
		P.Fz[gi] = acc.fz;
		return 0.;
	}
	const double k = P.dt_inv2m;
	double vx = P.vx[gi] + k * acc.fx;  // upd_postF
	double vy = P.vy[gi] + k * acc.fy;
	double vz = P.vz[gi] + k * acc.fz;
	const double mv2 = P.mass * (vx * vx + vy * vy + vz * vz);
	vx += k * acc.fx;  // upd_preF
	vy += k * acc.fy;
	vz += k * acc.fz;
	P.vx[gi] = vx;
	P.vy[gi] = vy;
	P.vz[gi] = vz;
	P.Fx[gi] = P.x[gi] + P.dt * vx;
	P.Fy[gi] = P.y[gi] + P.dt * vy;
	P.Fz[gi] = P.z[gi] + P.dt * vz;
	return mv2;
}

__device__ __forceinline__ double wave_sum_lj(double v) {
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
	return v;
}

template <int HW, int BX, int BY, int BZ, int CAPJ, int CAPL, int SPLIT>
__global__ void __launch_bounds__(LTPB * SPLIT, 2 * SPLIT) k_force_lj_brick(ForceParams P, int nbx, int nby, int nbz) {
	// SPLIT lanes share one molecule: lane `half` scans the neighbour rows half, half+SPLIT, ... (same code, per-lane
	// row offsets), keeps its own list and partial force; the partial forces are combined with one DPP exchange.
	// With SPLIT = 2 the workgroup has 512 threads on the same LDS budget (the lists are half as long), i.e. 4
	// waves per SIMD instead of 2 to hide LDS latency.
	constexpr int NT = LTPB * SPLIT;
	constexpr int RX = BX + 2 * HW, RY = BY + 2 * HW, RZ = BZ + 2 * HW;
	constexpr int NRC = RX * RY * RZ;
	constexpr int NBC = BX * BY * BZ;
	constexpr int NW = 2 * HW + 1;          // cells per neighbour row
	constexpr int NROWS = NW * NW;          // neighbour rows per molecule
	constexpr int OWNROW = (NROWS - 1) / 2; // the row that contains the molecule itself
	constexpr int LSHIFT = (NT == 256) ? 9 : 10;  // log2(bytes per list row)
	static_assert(NT == 256 || NT == 512, "list addressing assumes 256 or 512 threads");
	static_assert(NRC <= NT * 4, "region too large for the block scan");
	__shared__ double sx[CAPJ + 8], sy[CAPJ + 8], sz[CAPJ + 8];  // +8: unrolled reads may run past a row end
	__shared__ uint16_t lst[(CAPL + 1) * NT];  // +1: per-lane dummy slot for branch-free appends
	__shared__ uint32_t cstart[NRC + 1];  // LDS index of the first molecule of every region cell (region-linear order)
	__shared__ uint32_t gbeg[NRC];        // global index of the first molecule of every region cell
	__shared__ uint32_t bstart[NBC + 1];  // prefix over the brick's own cells (i-molecule enumeration)
	__shared__ uint32_t wsum[NT / 64];
	__shared__ double red[NT / 64][3];

	const int tid = threadIdx.x;
	const BrickSel bs = brick_select<HW, BX, BY, BZ>(P, nbx, nby, nbz);
	if (!bs.live) {  // uniform per workgroup
		if (tid < 4) P.partials[(size_t)blockIdx.x * 4 + tid] = 0.;
		return;
	}
	const int ex = bs.ex, ey = bs.ey, ez = bs.ez;

	// ---- 1. region cell table ------------------------------------------------------------------------------------
	brick_region_table<NT, HW, RX, RY, RZ>(P, bs, cstart, gbeg);
	__syncthreads();
	block_scan_lds<NT>(cstart, NRC, wsum);
	const uint32_t total = cstart[NRC];
	// brick cell prefix (own molecules)
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
	const bool staged = total <= (uint32_t)CAPJ;

	// ---- 2. stage positions (HBM -> LDS, each molecule of the shell read once per brick) ------------------------
	if (staged) {
		for (uint32_t s = tid; s < total; s += NT) {
			int lo = 0, hi = NRC;  // largest c with cstart[c] <= s
			while (hi - lo > 1) {
				const int mid = (lo + hi) >> 1;
				if (cstart[mid] <= s) lo = mid;
				else hi = mid;
			}
			const uint32_t g = gbeg[lo] + (s - cstart[lo]);
			sx[s] = P.x[g];
			sy[s] = P.y[g];
			sz[s] = P.z[g];
		}
	}
	__syncthreads();

	const double rc2 = P.rc2, eps24 = P.eps24, sig2 = P.sig2, shift6 = P.shift6;
	const int half = tid % SPLIT;
	const int nmy = (NROWS - half + SPLIT - 1) / SPLIT;  // neighbour rows of this lane
	double u6_tot = 0., vir_tot = 0., kin_tot = 0.;
	for (uint32_t base = 0; base < n_i; base += NT / SPLIT) {  // one pass unless the brick is over-full
		const uint32_t it = base + (uint32_t)(tid / SPLIT);
		const bool active = it < n_i;
		LjAcc acc = {0., 0., 0., 0., 0.};
		uint32_t gi = 0;
		if (active) {
			int lo = 0, hi = NBC;
			while (hi - lo > 1) {
				const int mid = (lo + hi) >> 1;
				if (bstart[mid] <= it) lo = mid;
				else hi = mid;
			}
			const int cx = lo % BX, cy = (lo / BX) % BY, cz = lo / (BX * BY);
			const int rxc = cx + HW, ryc = cy + HW, rzc = cz + HW;
			const int rcell = (rzc * RY + ryc) * RX + rxc;
			const uint32_t k = it - bstart[lo];
			const uint32_t ii = cstart[rcell] + k;
			gi = gbeg[rcell] + k;
			const int rowbase = ((rzc - HW) * RY + (ryc - HW)) * RX + (rxc - HW);  // row 0 of the neighbourhood
			if (staged) {
				const double xi = sx[ii], yi = sy[ii], zi = sz[ii];
				uint32_t cnt = 0;  // hits found (may exceed CAPL: then the list is incomplete and the lane re-evaluates directly)
				// ---- phase 1: candidate rows -> per-lane list.  U candidates per trip, the next trip's 3U LDS reads are
				// issued before the current trip is processed (register double buffer) so LDS latency overlaps the
				// distance arithmetic; the append is branch-free (misses / overflow write to a per-lane dummy slot).
				constexpr int U = 4;
				const uint32_t lane_off = (uint32_t)tid * 2u;  // byte offset of this lane inside a list row
				char* const lst_bytes = reinterpret_cast<char*>(lst);
				for (int kk = 0; kk < nmy; ++kk) {
					const int row = kk * SPLIT + half;
					const int r0 = rowbase + (row / NW) * (RY * RX) + (row % NW) * RX;
					const uint32_t jb = cstart[r0], je = cstart[r0 + NW];
					const uint32_t self = (row == OWNROW) ? ii : 0xffffffffu;  // the molecule itself lives in one row only
					double ax[U], ay[U], az[U], bxv[U], byv[U], bzv[U];
#pragma unroll
					for (int u = 0; u < U; ++u) {  // reads past `je` stay inside the padded staging arrays
						ax[u] = sx[jb + u];
						ay[u] = sy[jb + u];
						az[u] = sz[jb + u];
					}
					for (uint32_t j0 = jb; j0 < je; j0 += U) {
#pragma unroll
						for (int u = 0; u < U; ++u) {
							bxv[u] = sx[j0 + U + u];
							byv[u] = sy[j0 + U + u];
							bzv[u] = sz[j0 + U + u];
						}
						const uint32_t nvalid = je - j0;
#pragma unroll
						for (int u = 0; u < U; ++u) {
							const double dx = xi - ax[u], dyy = yi - ay[u], dzz = zi - az[u];
							const double r2 = fma(dzz, dzz, fma(dyy, dyy, dx * dx));
							const bool hit = (r2 < rc2) & ((uint32_t)u < nvalid) & (j0 + u != self);
							const uint32_t slot = min(hit ? cnt : (uint32_t)CAPL, (uint32_t)CAPL);
							*reinterpret_cast<uint16_t*>(lst_bytes + ((slot << LSHIFT) | lane_off)) = (uint16_t)(j0 + u);
							cnt += hit ? 1u : 0u;
						}
#pragma unroll
						for (int u = 0; u < U; ++u) {
							ax[u] = bxv[u];
							ay[u] = byv[u];
							az[u] = bzv[u];
						}
					}
				}
				if (cnt <= (uint32_t)CAPL) {
					// ---- phase 2: dense LJ evaluation over the list (every entry is in range, no masks), 4 per trip
					double slj = 0.;
					uint32_t s2 = 0;
					for (; s2 + 4 <= cnt; s2 += 4) {
						uint32_t jj[4];
#pragma unroll
						for (int u = 0; u < 4; ++u) jj[u] = *reinterpret_cast<uint16_t*>(lst_bytes + (((s2 + u) << LSHIFT) | lane_off));
						double px[4], py[4], pz[4];
#pragma unroll
						for (int u = 0; u < 4; ++u) {
							px[u] = sx[jj[u]];
							py[u] = sy[jj[u]];
							pz[u] = sz[jj[u]];
						}
#pragma unroll
						for (int u = 0; u < 4; ++u) lj_pair_in(xi, yi, zi, px[u], py[u], pz[u], eps24, sig2, acc, slj);
					}
					for (; s2 < cnt; ++s2) {
						const uint32_t j = *reinterpret_cast<uint16_t*>(lst_bytes + ((s2 << LSHIFT) | lane_off));
						lj_pair_in(xi, yi, zi, sx[j], sy[j], sz[j], eps24, sig2, acc, slj);
					}
					acc.u6 = fma(eps24, slj, shift6 * (double)cnt);
				} else {
					// list overflow (very dense neighbourhood): direct evaluation of this lane's rows
					for (int kk = 0; kk < nmy; ++kk) {
						const int row = kk * SPLIT + half;
						const int r0 = rowbase + (row / NW) * (RY * RX) + (row % NW) * RX;
						const uint32_t jb = cstart[r0], je = cstart[r0 + NW];
						for (uint32_t j = jb; j < je; ++j)
							if (j != ii) lj_pair(xi, yi, zi, sx[j], sy[j], sz[j], rc2, eps24, sig2, shift6, acc);
					}
				}
			} else {
				// shell does not fit LDS (pathological density): same arithmetic straight from global memory
				const double xi = P.x[gi], yi = P.y[gi], zi = P.z[gi];
				for (int kk = 0; kk < nmy; ++kk) {
					const int row = kk * SPLIT + half;
					const int r0 = rowbase + (row / NW) * (RY * RX) + (row % NW) * RX;
					for (int c = r0; c < r0 + NW; ++c) {
						const uint32_t gb = gbeg[c], n = cstart[c + 1] - cstart[c];
						for (uint32_t j = gb; j < gb + n; ++j)
							if (j != gi) lj_pair(xi, yi, zi, P.x[j], P.y[j], P.z[j], rc2, eps24, sig2, shift6, acc);
					}
				}
			}
		}
		// combine the partial forces of the SPLIT lanes that share a molecule (adjacent lanes); lane `half == 0` stores
		if (SPLIT == 2) {
			acc.fx += __shfl_xor(acc.fx, 1);
			acc.fy += __shfl_xor(acc.fy, 1);
			acc.fz += __shfl_xor(acc.fz, 1);
		}
		if (active && half == 0) kin_tot += lj_store(P, gi, acc);
		u6_tot += acc.u6;
		vir_tot += acc.vir;
	}
	// every ordered pair contributes half of the pair's U and virial (see kernels_force.hip)
	double u = wave_sum_lj(0.5 * u6_tot), v = wave_sum_lj(0.5 * vir_tot), kn = wave_sum_lj(kin_tot);
	const int lane = tid & 63, w = tid >> 6;
	if (lane == 0) {
		red[w][0] = u;
		red[w][1] = v;
		red[w][2] = kn;
	}
	__syncthreads();
	if (tid == 0) {
		double* out = P.partials + (size_t)blockIdx.x * 4;
		double su = 0., sv = 0., sk = 0.;
		for (int i = 0; i < NT / 64; ++i) {
			su += red[i][0];
			sv += red[i][1];
			sk += red[i][2];
		}
		out[0] = su;
		out[1] = sk;  // fused mode: sum m v^2 of the brick's molecules (the uX slot is unused by single-centre LJ)
		out[2] = 0.;
		out[3] = sv;
	}
}

// ======================================================================================================================
// k_force_lj_mfma — phase 1 as a dense r^2 tile on the matrix pipe.
//
// The candidate search is the one dense contraction on this path: for 16 owned molecules j and 16 candidates i,
//     D[i][j] = sum_k A[i][k] B[k][j],   A[i] = (x_i, y_i, z_i, |r_i|^2),   B[.][j] = (-2x_j, -2y_j, -2z_j, 1)
// is r_ij^2 - |r_j|^2 (K = 4): ONE v_mfma_f32_16x16x4_f32 evaluates 256 pair distances (FP32, brick-relative
// coordinates), against ~8 VALU instructions per 64 distances in the scalar formulation.  It is used as a conservative
// PRE-FILTER only (threshold rc^2 + rounding margin); phase 2 recomputes every listed pair in FP64 with the exact
// strict mask, so forces / U / virial are those of the FP64 kernels.
//
// Mapping (v_mfma_f32_16x16x4_f32 register layout): lane l supplies A[i = l%16][k = l/16] and B[k = l/16][j = l%16] and
// receives D[i = 4*(l/16) + r][j = l%16], r = 0..3.  So the four lanes {jo, jo+16, jo+32, jo+48} own molecule jo of the
// tile and each sees a different quarter of every 16-candidate tile: the hit test needs ONE per-lane threshold
// (rc^2 + margin - |r_j|^2), the per-lane lists are a quarter as long, and there is no cross-lane mask traffic.
// Owned tile = the molecules of one cell (padded to 16); pencil bricks (1 x BY x BZ cells) make the 9 neighbour cells
// of every z-plane ONE contiguous LDS range, so a cell has 3 candidate ranges of ~112 molecules = 7 tiles each.
//
// LDS: x, y, z are staged once, in FP64 (absolute), plus the FP32 |r_rel|^2: 28 B per molecule; the A operand is
// converted on load ((float)(x_k[i] - origin_k), 3 VALU per tile).  List entries are BYTE offsets (8 * index, u16) so phase 2 uses
// them as LDS addresses directly.  The append is an unconditional store to slot `cnt` followed by cnt += hit
// (v_cmp + v_addc): a miss is overwritten by the next store; cnt is clamped once per tile, which costs 4 spare rows.
// ======================================================================================================================
typedef float floatx4 __attribute__((ext_vector_type(4)));

// masked pair for the MFMA kernel: out-of-range (and self, r2 == 0) pairs get r2 := 1e300, which drives every LJ term
// to exactly 0 by underflow — one select instead of masking force and potential separately.
__device__ __forceinline__ void lj_pair_m(double xi, double yi, double zi, double xj, double yj, double zj, double rc2,
										  double eps24, double sig2, LjAcc& a, double& slj, uint32_t& nin) {
	const double dx = xi - xj, dy = yi - yj, dz = zi - zj;
	const double r2 = fma(dz, dz, fma(dy, dy, dx * dx));
	const bool in = (r2 < rc2) & (r2 != 0.0);
	const double inv = fast_rcp(in ? r2 : 1.0e300);
	const double lj2 = sig2 * inv;
	const double lj6 = lj2 * lj2 * lj2;
	const double lj12 = lj6 * lj6;
	const double lj12m6 = lj12 - lj6;
	const double fac = eps24 * inv * (lj12 + lj12m6);
	a.fx = fma(fac, dx, a.fx);
	a.fy = fma(fac, dy, a.fy);
	a.fz = fma(fac, dz, a.fz);
	slj += lj12m6;
	nin += in ? 1u : 0u;
	a.vir = fma(fac, r2, a.vir);
}

template <int NT, int BY, int BZ, int CAPJ, int ROWS>
__global__ void __launch_bounds__(NT, NT / 128) k_force_lj_mfma(ForceParams P, int nbx, int nby, int nbz) {
	constexpr int HW = 1, BX = 1;
	constexpr int RX = BX + 2 * HW, RY = BY + 2 * HW, RZ = BZ + 2 * HW;
	constexpr int NRC = RX * RY * RZ;
	constexpr int NBC = BX * BY * BZ;
	constexpr int NW = NT / 64;
	constexpr int PLANE = 3 * RX;  // 9 cells of one z-plane of the neighbourhood are contiguous (RX == 3)
	constexpr int PAD = 48;  // phase 1 prefetches the A operand up to two tiles past the end of a range (index < je + 48)
	constexpr int CAPS = CAPJ + PAD;
	constexpr int RSH = (NT == 512) ? 10 : 9;   // list row stride = NT * 2 bytes
	constexpr uint32_t CAPX = ROWS - 4;          // cnt is clamped to CAPX after every tile; cnt >= CAPX means overflow
	static_assert(NT == 256 || NT == 512, "row stride shift");
	static_assert(NBC % NW == 0, "cells must divide evenly over the waves");
	static_assert(NRC <= NT * 4, "region too large for the block scan");
	static_assert(CAPS * 8 <= 65536, "list entries are u16 byte offsets");
	// staged molecules: x, y, z absolute (FP64, three arrays of CAPS) followed by |r_rel|^2 (FP32, CAPS): 28 B / molecule
	__shared__ __attribute__((aligned(16))) char sraw[CAPS * 28 + 8];
	double* const sx = reinterpret_cast<double*>(sraw);
	double* const sy = sx + CAPS;
	double* const sz = sy + CAPS;
	float* const sq = reinterpret_cast<float*>(sz + CAPS);
	__shared__ uint16_t lst[ROWS * NT];      // per-lane candidate lists, slot-major
	__shared__ uint32_t cstart[NRC + 1];
	__shared__ uint32_t gbeg[NRC];
	__shared__ uint32_t wsum[NW];
	__shared__ double red[NW][3];

	const int tid = threadIdx.x;
	const BrickSel bs = brick_select<HW, BX, BY, BZ>(P, nbx, nby, nbz);
	if (!bs.live) {  // uniform per workgroup
		if (tid < 4) P.partials[(size_t)blockIdx.x * 4 + tid] = 0.;
		return;
	}
	const int x0 = bs.x0, y0 = bs.y0, z0 = bs.z0, ex = bs.ex, ey = bs.ey, ez = bs.ez;
	// ---- region cell table + staging (as in k_force_lj_brick) ---------------------------------------------------------
	brick_region_table<NT, HW, RX, RY, RZ>(P, bs, cstart, gbeg);
	__syncthreads();
	block_scan_lds<NT>(cstart, NRC, wsum);
	const uint32_t total = cstart[NRC];
	const bool staged = total <= (uint32_t)CAPJ;
	// brick-relative origin: lower corner of the region's first cell
	const double ox = P.g.bmin[0] + (double)(x0 - 2 * HW) * P.g.clen[0];
	const double oy = P.g.bmin[1] + (double)(y0 - 2 * HW) * P.g.clen[1];
	const double oz = P.g.bmin[2] + (double)(z0 - 2 * HW) * P.g.clen[2];
	if (staged) {
		// 16 lanes per region cell (4 cells per wave and trip): no search for the cell of a staged index, and all global
		// loads of a thread are independent (the per-molecule binary search was a chain of 7 dependent LDS reads)
		for (int c = tid >> 4; c < NRC; c += NT / 16) {
			const uint32_t n = cstart[c + 1] - cstart[c], s0 = cstart[c], g0 = gbeg[c];
			for (uint32_t k = (uint32_t)tid & 15u; k < n; k += 16u) {
				const uint32_t g = g0 + k, s = s0 + k;
				const double x = P.x[g], y = P.y[g], z = P.z[g];
				sx[s] = x;
				sy[s] = y;
				sz[s] = z;
				const float fx = (float)(x - ox), fy = (float)(y - oy), fz = (float)(z - oz);
				sq[s] = fx * fx + fy * fy + fz * fz;
			}
		}
		if (tid < PAD) {  // padding behind the last molecule: far away, never within the cutoff
			const uint32_t s = total + (uint32_t)tid;
			sx[s] = ox;
			sy[s] = oy;
			sz[s] = oz;
			sq[s] = 3.0e30f;
		}
	}
	__syncthreads();
	const double rc2 = P.rc2, eps24 = P.eps24, sig2 = P.sig2, shift6 = P.shift6;
	// conservative FP32 threshold: |D + |r_j|^2 - r^2| <= ~8 ulp of the largest term (2 E^2, E = region diagonal)
	const float ext2 = (float)((RX * P.g.clen[0]) * (RX * P.g.clen[0]) + (RY * P.g.clen[1]) * (RY * P.g.clen[1]) +
							   (RZ * P.g.clen[2]) * (RZ * P.g.clen[2]));
	const float rc2m = (float)rc2 * (1.0f + 1e-6f) + 4e-6f * ext2;
	const int lane = tid & 63, wv = tid >> 6;
	const int jo = lane & 15, grp = lane >> 4;
	const uint32_t lane_off = (uint32_t)tid * 2u;
	char* const lst_bytes = reinterpret_cast<char*>(lst);
	const char* const sxb = reinterpret_cast<const char*>(sx);
	const char* const syb = reinterpret_cast<const char*>(sy);
	const char* const szb = reinterpret_cast<const char*>(sz);
	const double o_l = (grp == 0) ? ox : (grp == 1) ? oy : (grp == 2) ? oz : 0.0;  // origin of this lane's A component
	// this lane's A component as a strided byte stream: FP64 coordinates for k = 0..2, the FP32 |r|^2 for k = 3.  Both
	// are fetched with the same two-dword LDS read (4-byte aligned), the k = 3 lanes use the low dword as is.
	const uint32_t astride = (grp == 3) ? 4u : 8u;
	const char* const arow = sraw + (size_t)grp * CAPS * 8 + (size_t)jo * astride;
	typedef uint64_t u64_align4 __attribute__((aligned(4)));  // one ds_read2_b32 for every lane, no branch on grp
	auto load_a = [&](const char* pa) -> float {
		const uint64_t raw = *reinterpret_cast<const u64_align4*>(pa);
		const float conv = (float)(__longlong_as_double((long long)raw) - o_l);  // meaningless (but harmless) for k = 3
		return (grp == 3) ? __uint_as_float((uint32_t)raw) : conv;
	};
	// cnt += (d < thr): v_cmp + v_addc, kept as a 2-instruction chain (the compiler's version trades a third instruction
	// per result for a shorter dependency chain, which 4 waves per SIMD hide anyway)
	auto count_hit = [](uint32_t& cnt, float d, float thr) {
		asm("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt) : "v"(d), "v"(thr) : "vcc");
	};
	double u6_tot = 0., vir_tot = 0., kin_tot = 0.;

	for (int cc = 0; cc < NBC / NW; ++cc) {              // every wave owns NBC/NW cells of the brick
		const int c = wv * (NBC / NW) + cc;
		const int cy = c % BY, cz = c / BY;                // BX == 1
		if (cy >= ey || cz >= ez || ex < 1) continue;       // wave-uniform
		const int rcell = ((cz + HW) * RY + (cy + HW)) * RX + HW;
		const uint32_t ob = cstart[rcell], n_c = cstart[rcell + 1] - ob;
		const uint32_t gb = gbeg[rcell];
		for (uint32_t tb = 0; tb < n_c; tb += 16) {        // owned tiles of 16 molecules (one unless the cell is crowded)
			const bool valid = tb + (uint32_t)jo < n_c;
			const uint32_t oi = ob + tb + (uint32_t)jo;     // LDS index of the owned molecule of this lane
			const uint32_t gi = gb + tb + (uint32_t)jo;     // its global index
			LjAcc acc = {0., 0., 0., 0., 0.};
			if (staged) {
				// B operand (owned side) and per-lane threshold
				float bval = (grp < 3) ? 0.f : 1.f;
				float thr = -3.0e38f;
				if (valid) {
					const float own = (float)(sx[(grp < 3 ? grp : 0) * CAPS + oi] - o_l);
					bval = (grp < 3) ? -2.f * own : 1.f;
					thr = rc2m - sq[oi];
				}
				uint32_t cnt = 0;
				// ---- phase 1: MFMA distance tiles -> per-lane lists --------------------------------------------------
				const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
				for (int dz = -1; dz <= 1; ++dz) {
					const int r0 = ((cz + HW + dz) * RY + (cy + HW - 1)) * RX;  // first cell of the plane's 9-cell range
					const uint32_t jb = (uint32_t)__builtin_amdgcn_readfirstlane((int)cstart[r0]);
					const uint32_t je = (uint32_t)__builtin_amdgcn_readfirstlane((int)cstart[r0 + PLANE]);
					if (jb >= je) continue;  // wave-uniform
					const char* ap = arow + jb * astride;
					uint32_t cb8 = (jb + 4u * (uint32_t)grp) * 8u;  // byte offset of this lane's first candidate of the tile
					// software pipeline: the MFMA of tile t+1 is issued (and the A operand of tile t+2 loaded) before the
					// results of tile t are tested, so the matrix-pipe latency (8 passes) and the LDS read overlap the
					// compare / append work instead of stalling in front of it
					float a_next = load_a(ap + 16u * astride);
					floatx4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(load_a(ap), bval, zero, 0, 0, 0);
					ap += 32u * astride;
					// The last tile of a range may run past `je` into the next cells of the region.  Normally those are
					// not neighbour cells of the owned cell (one full cell >= r_c away, or the far-away padding): at worst
					// the FP32 slack lists one and phase 2 rejects it, so no bound test is needed.  The exception are
					// ranges followed by fewer than 16 molecules before the NEXT plane's range starts (partial bricks at
					// the upper domain faces, empty regions): there the overrun would reach true neighbours and list
					// them twice, so the last tile is tested against `je` (wave-uniform choice).
					const bool safe = dz == 1 ||
									  (uint32_t)__builtin_amdgcn_readfirstlane((int)cstart[r0 + RY * RX]) - je >= 16u;
					const uint32_t je_fast = safe ? je : jb + ((je - jb) & ~15u);
					for (uint32_t t = jb; t < je_fast; t += 16u) {
						const floatx4 dn = __builtin_amdgcn_mfma_f32_16x16x4f32(a_next, bval, zero, 0, 0, 0);
						a_next = load_a(ap);
						ap += 16u * astride;
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							*reinterpret_cast<uint16_t*>(lst_bytes + ((cnt << RSH) | lane_off)) = (uint16_t)(cb8 + 8u * r);
							count_hit(cnt, d[r], thr);
						}
						cnt = min(cnt, CAPX);
						cb8 += 128u;
						d = dn;
					}
					if (je_fast < je) {  // checked last tile (rare)
						const uint32_t je8 = je * 8u;
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							const uint32_t c8 = cb8 + 8u * r;
							*reinterpret_cast<uint16_t*>(lst_bytes + ((cnt << RSH) | lane_off)) = (uint16_t)c8;
							cnt += ((d[r] < thr) & (c8 < je8)) ? 1u : 0u;
						}
						cnt = min(cnt, CAPX);
					}
				}
				// ---- phase 2: exact FP64 evaluation of the listed pairs (strict mask; self pair has r2 == 0) --------
				if (valid) {
					const double xi = sx[oi], yi = sy[oi], zi = sz[oi];
					if (cnt < CAPX) {
						double slj = 0.;
						uint32_t nin = 0;
						uint32_t s2 = 0;
						for (; s2 + 2 <= cnt; s2 += 2) {
							const uint32_t ja = *reinterpret_cast<uint16_t*>(lst_bytes + ((s2 << RSH) | lane_off));
							const uint32_t jc = *reinterpret_cast<uint16_t*>(lst_bytes + (((s2 + 1) << RSH) | lane_off));
							const double xa = *reinterpret_cast<const double*>(sxb + ja);
							const double ya = *reinterpret_cast<const double*>(syb + ja);
							const double za = *reinterpret_cast<const double*>(szb + ja);
							const double xb = *reinterpret_cast<const double*>(sxb + jc);
							const double yb = *reinterpret_cast<const double*>(syb + jc);
							const double zb = *reinterpret_cast<const double*>(szb + jc);
							lj_pair_m(xi, yi, zi, xa, ya, za, rc2, eps24, sig2, acc, slj, nin);
							lj_pair_m(xi, yi, zi, xb, yb, zb, rc2, eps24, sig2, acc, slj, nin);
						}
						if (s2 < cnt) {
							const uint32_t ja = *reinterpret_cast<uint16_t*>(lst_bytes + ((s2 << RSH) | lane_off));
							lj_pair_m(xi, yi, zi, *reinterpret_cast<const double*>(sxb + ja),
									  *reinterpret_cast<const double*>(syb + ja), *reinterpret_cast<const double*>(szb + ja),
									  rc2, eps24, sig2, acc, slj, nin);
						}
						acc.u6 = fma(eps24, slj, shift6 * (double)nin);
					} else {
						// list overflow (very dense neighbourhood): this lane evaluates its quarter of every tile directly
						for (int dz = -1; dz <= 1; ++dz) {
							const int r0 = ((cz + HW + dz) * RY + (cy + HW - 1)) * RX;
							const uint32_t jb = cstart[r0], je = cstart[r0 + PLANE];
							for (uint32_t t = jb; t < je; t += 16)
								for (uint32_t r = 0; r < 4; ++r) {
									const uint32_t j = t + 4u * (uint32_t)grp + r;
									if (j < je) lj_pair(xi, yi, zi, sx[j], sy[j], sz[j], rc2, eps24, sig2, shift6, acc);
								}
						}
					}
				}
			} else if (valid) {
				// shell does not fit LDS (pathological density): quarter of the neighbour cells per lane, from global memory
				const double xi = P.x[gi], yi = P.y[gi], zi = P.z[gi];
				for (int dz = -1; dz <= 1; ++dz) {
					const int r0 = ((cz + HW + dz) * RY + (cy + HW - 1)) * RX;
					for (int k = r0 + grp; k < r0 + PLANE; k += 4) {
						const uint32_t g0 = gbeg[k], n = cstart[k + 1] - cstart[k];
						for (uint32_t j = g0; j < g0 + n; ++j)
							if (j != gi) lj_pair(xi, yi, zi, P.x[j], P.y[j], P.z[j], rc2, eps24, sig2, shift6, acc);
					}
				}
			}
			// the four lanes {jo, jo+16, jo+32, jo+48} hold partial forces of the same molecule
			acc.fx += __shfl_xor(acc.fx, 16);
			acc.fy += __shfl_xor(acc.fy, 16);
			acc.fz += __shfl_xor(acc.fz, 16);
			acc.fx += __shfl_xor(acc.fx, 32);
			acc.fy += __shfl_xor(acc.fy, 32);
			acc.fz += __shfl_xor(acc.fz, 32);
			if (valid && grp == 0) kin_tot += lj_store(P, gi, acc);
			u6_tot += acc.u6;
			vir_tot += acc.vir;
		}
	}
	double u = wave_sum_lj(0.5 * u6_tot), v = wave_sum_lj(0.5 * vir_tot), kn = wave_sum_lj(kin_tot);
	if (lane == 0) {
		red[wv][0] = u;
		red[wv][1] = v;
		red[wv][2] = kn;
	}
	__syncthreads();
	if (tid == 0) {
		double* out = P.partials + (size_t)blockIdx.x * 4;
		double su = 0., sv = 0., sk = 0.;
		for (int i = 0; i < NW; ++i) {
			su += red[i][0];
			sv += red[i][1];
			sk += red[i][2];
		}
		out[0] = su;
		out[1] = sk;  // fused mode: sum m v^2 (see k_force_lj_brick)
		out[2] = 0.;
		out[3] = sv;
	}
}

// Lists of the inner / boundary bricks of a grid for one brick shape ("inner": no halo cell inside the brick's shell),
// built on the host when the grid or the shape changes (a few 10^4 entries).
static bool brick_lists_for(BrickLists* bl, const Grid& g, int BX, int BY, int BZ, int nbx, int nby, int nbz) {
	if (!bl) return false;
	const int hw = g.hw;
	if (bl->d[0] && bl->shape[0] == BX && bl->shape[1] == BY && bl->shape[2] == BZ && bl->hw == hw &&
		bl->dims[0] == g.dims[0] && bl->dims[1] == g.dims[1] && bl->dims[2] == g.dims[2])
		return true;
	// Launch order = blocks of BB[0] x BB[1] x BB[2] bricks, x fastest inside a block and over the blocks: the bricks an XCD
	// works on at the same time (a contiguous run of this order, brick_select_v) then overlap in all three dimensions and
	// find each other's shell in the XCD's L2; in plain x-y-z order the z neighbours are a whole plane of bricks apart.
	int BB[3] = {4, 4, 8};  // measured on the 10^8 box: 4x4x8 9.53, 4x4x4 9.50, 8x8x4 9.48, 46x4x4 9.46, plain order 9.41 (10^9 updates/s)
	if (const char* e = getenv("LS1_BRICK_BLOCK")) {
		int a, b, c;
		if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a > 0 && b > 0 && c > 0) BB[0] = a, BB[1] = b, BB[2] = c;
	}
	std::vector<uint32_t> lst[5];
	for (int Z = 0; Z < nbz; Z += BB[2])
		for (int Y = 0; Y < nby; Y += BB[1])
			for (int X = 0; X < nbx; X += BB[0])
				for (int bz = Z; bz < std::min(nbz, Z + BB[2]); ++bz)
					for (int by = Y; by < std::min(nby, Y + BB[1]); ++by)
						for (int bx = X; bx < std::min(nbx, X + BB[0]); ++bx) {
							const int x0 = hw + bx * BX, y0 = hw + by * BY, z0 = hw + bz * BZ;
							const int ex = std::min(BX, g.dims[0] - hw - x0), ey = std::min(BY, g.dims[1] - hw - y0),
									  ez = std::min(BZ, g.dims[2] - hw - z0);
							const bool inner = x0 >= 2 * hw && y0 >= 2 * hw && z0 >= 2 * hw && x0 + ex <= g.dims[0] - 2 * hw &&
											   y0 + ey <= g.dims[1] - 2 * hw && z0 + ez <= g.dims[2] - 2 * hw;
							const uint32_t id = (uint32_t)((bz * nby + by) * nbx + bx);
							lst[inner ? 0 : 1].push_back(id);
							lst[inner ? 3 : 4].push_back((uint32_t)lst[2].size());
							lst[2].push_back(id);
						}
	for (int k = 0; k < 5; ++k) {
		if (bl->d[k]) (void)hipFree(bl->d[k]);
		bl->d[k] = nullptr;
		bl->n[k] = (uint32_t)lst[k].size();
		if (hipMalloc(&bl->d[k], std::max<size_t>(lst[k].size(), 1) * sizeof(uint32_t)) != hipSuccess) {
			bl->d[k] = nullptr;
			for (int j = 0; j < k; ++j) {
				(void)hipFree(bl->d[j]);
				bl->d[j] = nullptr;
			}
			return false;
		}
		if (!lst[k].empty() &&
			hipMemcpy(bl->d[k], lst[k].data(), lst[k].size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
			return false;
	}
	bl->shape[0] = BX; bl->shape[1] = BY; bl->shape[2] = BZ;
	bl->hw = hw;
	for (int d = 0; d < 3; ++d) bl->dims[d] = g.dims[d];
	return true;
}

// number of workgroups for a traversal; fills p.brick_list / p.n_list for the inner (1) / boundary (2) passes
long plan_bricks(ForceParams& p, BrickLists* bl, int BX, int BY, int BZ, int nbx, int nby, int nbz, bool blocked_order) {
	long n = (long)nbx * nby * nbz;
	p.brick_list = nullptr;
	p.n_list = 0;
	p.inner_box = 0;
	p.did_mode = 0;
	p.brick_did = nullptr;
	if (p.which == 1 && blocked_order && brick_lists_for(bl, p.g, BX, BY, BZ, nbx, nby, nbz)) {
		p.brick_list = bl->d[0];  // the inner bricks in the blocked launch order
		p.n_list = bl->n[0];
		p.did_mode = 2;
		p.brick_did = bl->d[3];
		n = p.n_list;
	} else if (p.which == 1) {
		// "inner" is separable per dimension: bricks [lo, lo + cnt) whose cells lie in [2hw, dims - 2hw)
		const int B[3] = {BX, BY, BZ}, nbd[3] = {nbx, nby, nbz};
		const int hw = p.g.hw;
		n = 1;
		for (int d = 0; d < 3; ++d) {
			int lo = nbd[d], hi = -1;
			for (int b = 0; b < nbd[d]; ++b) {
				const int x0 = hw + b * B[d], e = std::min(B[d], p.g.dims[d] - hw - x0);
				if (x0 >= 2 * hw && x0 + e <= p.g.dims[d] - 2 * hw) {
					lo = std::min(lo, b);
					hi = std::max(hi, b);
				}
			}
			p.inner_lo[d] = lo;
			p.inner_n[d] = hi >= lo ? hi - lo + 1 : 0;
			n *= p.inner_n[d];
		}
		p.inner_box = 1;
	} else if (p.which == 2 && brick_lists_for(bl, p.g, BX, BY, BZ, nbx, nby, nbz)) {
		p.brick_list = bl->d[1];
		p.n_list = bl->n[1];
		if (blocked_order) {
			p.did_mode = 2;
			p.brick_did = bl->d[4];
		}
		n = p.n_list;
	} else if (p.which == 0 && blocked_order && brick_lists_for(bl, p.g, BX, BY, BZ, nbx, nby, nbz)) {
		p.brick_list = bl->d[2];  // every brick, in the blocked launch order
		p.n_list = bl->n[2];
		p.did_mode = 1;
	}
	return 8 * ((n + 7) / 8);
}

template <int NT, int BY, int BZ, int CAPJ, int ROWS>
static bool launch_mfma(ForceParams p, BrickLists* bl, hipStream_t s, uint32_t* nblocks, size_t partials_cap) {
	const Grid& g = p.g;
	const int nbx = g.box[0], nby = (g.box[1] + BY - 1) / BY, nbz = (g.box[2] + BZ - 1) / BZ;
	if ((long)nbx * nby * nbz <= 0 || (long)nbx * nby * nbz > 0x7ffffff0L) return false;
	const long nb = plan_bricks(p, bl, 1, BY, BZ, nbx, nby, nbz);
	if ((size_t)nb > partials_cap) return false;
	*nblocks = (uint32_t)nb;
	if (nb == 0) return true;
	hipLaunchKernelGGL((k_force_lj_mfma<NT, BY, BZ, CAPJ, ROWS>), dim3((uint32_t)nb), dim3(NT), 0, s, p, nbx, nby, nbz);
	return true;
}

template <int HW, int BX, int BY, int BZ, int CAPJ, int CAPL, int SPLIT>
static bool launch_brick(ForceParams p, BrickLists* bl, hipStream_t s, uint32_t* nblocks, size_t partials_cap) {
	const Grid& g = p.g;
	const int nbx = (g.box[0] + BX - 1) / BX, nby = (g.box[1] + BY - 1) / BY, nbz = (g.box[2] + BZ - 1) / BZ;
	if ((long)nbx * nby * nbz <= 0 || (long)nbx * nby * nbz > 0x7ffffff0L) return false;
	const long nb = plan_bricks(p, bl, BX, BY, BZ, nbx, nby, nbz);
	if ((size_t)nb > partials_cap) return false;
	*nblocks = (uint32_t)nb;
	if (nb == 0) return true;
	hipLaunchKernelGGL((k_force_lj_brick<HW, BX, BY, BZ, CAPJ, CAPL, SPLIT>), dim3((uint32_t)nb), dim3(LTPB * SPLIT), 0, s,
					   p, nbx, nby, nbz);
	return true;
}

// `split` selects the variant (option "lj_split"): 1 / 2 = list kernel with that many lanes per molecule,
// 4 = MFMA pre-filter kernel, 1x4x4-cell bricks, 512 threads; 5 = same with 256 threads (larger staging area);
// 6 = MFMA kernel with 1x4x2-cell bricks (denser systems); 0 = choose from the mean cell occupancy.
bool launch_force_lj(const ForceParams& p, hipStream_t s, uint32_t* nblocks, double* partials, size_t partials_cap,
					 int split, double mean_per_cell, BrickLists* bl) {
	(void)partials;
	if (split == 0) {
		// staging capacity with 8 % headroom for density fluctuations; a brick that still overflows falls back (slowly)
		// to global memory inside the kernel, so the choice only affects speed
		split = 2;
		if (p.g.hw == 1) {
			if (mean_per_cell * 108. * 1.08 <= 1664.) split = 4;
			else if (mean_per_cell * 72. * 1.08 <= 1664.) split = 6;
		}
	}
	if (p.g.hw == 1) {
		if (split == 4) return launch_mfma<512, 4, 4, 1664, 32>(p, bl, s, nblocks, partials_cap);
		if (split == 5) return launch_mfma<256, 4, 4, 2248, 32>(p, bl, s, nblocks, partials_cap);
		if (split == 6) return launch_mfma<512, 4, 2, 1664, 32>(p, bl, s, nblocks, partials_cap);
		if (split == 2) return launch_brick<1, 4, 2, 2, 1616, 38, 2>(p, bl, s, nblocks, partials_cap);
		return launch_brick<1, 4, 2, 2, 1656, 71, 1>(p, bl, s, nblocks, partials_cap);
	}
	if (p.g.hw == 2) {
		if (split == 2 || split >= 4) return launch_brick<2, 8, 4, 4, 1440, 38, 2>(p, bl, s, nblocks, partials_cap);
		return launch_brick<2, 8, 4, 4, 1528, 63, 1>(p, bl, s, nblocks, partials_cap);
	}
	return false;
}

}  // namespace ls1
