// kernels_force_ms.hip — brick-tiled, LDS-staged force kernel for ANY component set (multi-centre LJ, charges, dipoles,
// quadrupoles): the multi-site counterpart of k_force_lj_brick.
//
//   * one 256-thread workgroup per brick of BX x BY x BZ cells; the brick and its cutoff shell are staged ONCE into LDS
//     in region-linear order: centre position (FP64 x, y, z), NORMALISED quaternion (FullMolecule::setupSoACache
//     normalises q before rotating, FullMolecule.cpp:720 — done once per staged molecule here instead of once per pair)
//     and component id (u8): 57 B per molecule, every x-row of 3 neighbour cells is one contiguous LDS range;
//   * one lane per owned molecule, lanes enumerate the brick's molecules densely (no per-cell tiles: the multi-site
//     fixtures have 2-4 molecules per cell), full shell, no atomics, deterministic;
//   * phase 1: centre-distance test over the 9 neighbour rows (LJ and electrostatics both cut on the CENTRE distance,
//     VectorizedCellProcessor.cpp:967-968,1013-1024), in-range j appended branch-free to a per-lane u16 list in LDS;
//   * phase 2: the molecule-pair body (mol_pair: all ten site-type combinations with on-the-fly site rotation, torque and
//     per-molecule virial) over the list, every lane busy.  Candidates are visited in the order of the generic kernel
//     (dz, dy, then x), so forces, torques and sums are bitwise those of k_force_generic;
//   * list overflow and shells larger than the staging area fall back to direct evaluation (same arithmetic).
// Against k_force_generic (neighbours fetched by every lane from global memory: 27 dependent, uncoalesced cell walks per
// molecule) positions arrive coalesced once per brick and the search runs out of LDS.
#include <type_traits>

#include "common.hpp"
#include "brick.hpp"

namespace ls1 {

constexpr int MTPB = 256;

// 4-double block reduction -> partials[blockIdx.x][4]
__device__ __forceinline__ void ms_block_reduce4(double v0, double v1, double v2, double v3, double* partials, double (*red)[4]) {
	double v[4] = {v0, v1, v2, v3};
	for (int k = 0; k < 4; ++k)
		for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o);
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if (lane == 0)
		for (int k = 0; k < 4; ++k) red[w][k] = v[k];
	__syncthreads();
	if (threadIdx.x < 4) {
		double s = 0.;
		for (int i = 0; i < MTPB / 64; ++i) s += red[i][threadIdx.x];
		partials[(size_t)blockIdx.x * 4 + threadIdx.x] = s;
	}
}

template <int BX, int BY, int BZ, int CAPJ, int CAPL, bool WITH_VI, bool HAS_ROT, bool ONEC>
__global__ void __launch_bounds__(MTPB) k_force_ms_brick(ForceParams P, const CompTable* __restrict__ ctab, int nbx, int nby, int nbz) {
	// (the component table as its own noalias argument: its loads are then known not to be clobbered by the force stores, and
	// with uniform indices become scalar loads)
	constexpr int HW = 1, NT = MTPB;
	constexpr int RX = BX + 2 * HW, RY = BY + 2 * HW, RZ = BZ + 2 * HW;
	constexpr int NRC = RX * RY * RZ;
	constexpr int NBC = BX * BY * BZ;
	constexpr int NW = 3, NROWS = 9;
	constexpr int QN = HAS_ROT ? CAPJ : 1;
	static_assert(NRC <= NT * 4 && NBC <= NT * 4, "region too large for the block scan");
	__shared__ double sx[CAPJ], sy[CAPJ], sz[CAPJ];
	__shared__ double sq0[QN], sq1[QN], sq2[QN], sq3[QN];  // normalised quaternion
	__shared__ uint8_t scid[CAPJ];
	__shared__ uint16_t lst[(CAPL + 1) * NT];  // slot-major; row CAPL = dummy target of misses / overflow
	__shared__ uint32_t cstart[NRC + 1];
	__shared__ uint32_t gbeg[NRC];
	__shared__ uint32_t bstart[NBC + 1];
	__shared__ uint32_t wsum[NT / 64];
	__shared__ double red[NT / 64][4];
	constexpr int OWNCAP = 2 * NT;                  // bricks up to 512 owned molecules get the component-ordered lane map
	__shared__ uint16_t oord[OWNCAP];               // lane slot -> owned enumeration index, grouped by component
	__shared__ uint32_t ccnt[2][NT / 64][MAXC];     // per (round, wave, component) counts, then exclusive bases

	const int tid = threadIdx.x;
	const BrickSel bs = brick_select<HW, BX, BY, BZ>(P, nbx, nby, nbz);
	if (!bs.live) {  // uniform per workgroup
		if (tid < 4) P.partials[(size_t)blockIdx.x * 4 + tid] = 0.;
		return;
	}
	const int ex = bs.ex, ey = bs.ey, ez = bs.ez;
	// ---- region cell table, brick cell prefix --------------------------------------------------------------------------
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
	const bool staged = total <= (uint32_t)CAPJ;
	// ---- stage centres, normalised quaternions, component ids --------------------------------------------------------
	if (staged) {
		for (uint32_t s = tid; s < total; s += NT) {
			int lo = 0, hi = NRC;
			while (hi - lo > 1) {
				const int mid = (lo + hi) >> 1;
				if (cstart[mid] <= s) lo = mid;
				else hi = mid;
			}
			const uint32_t g = gbeg[lo] + (s - cstart[lo]);
			sx[s] = P.x[g];
			sy[s] = P.y[g];
			sz[s] = P.z[g];
			scid[s] = (uint8_t)P.cid[g];
			if (HAS_ROT) {
				const double w = P.q0[g], x = P.q1[g], y = P.q2[g], z = P.q3[g];
				const double inv = 1. / sqrt(w * w + x * x + y * y + z * z);
				sq0[s] = w * inv;
				sq1[s] = x * inv;
				sq2[s] = y * inv;
				sq3[s] = z * inv;
			}
		}
	}
	__syncthreads();

	const double rc2 = ctab->rc2, rclj2 = ctab->rclj2;
	const int ncomp = ctab->ncomp;
	// owned enumeration index -> (region cell, rank in cell)
	auto locate = [&](uint32_t it, int& lo_out) {
		int lo = 0, hi = NBC;
		while (hi - lo > 1) {
			const int mid = (lo + hi) >> 1;
			if (bstart[mid] <= it) lo = mid;
			else hi = mid;
		}
		lo_out = lo;
	};
	// ---- component-ordered lane map -------------------------------------------------------------------------------------
	// With several components the molecule-pair body branches on (ci, cj); lanes of a wave that hold different components
	// walk every branch.  The owned molecules of the brick are therefore handed to the lanes grouped by component
	// (deterministic: stable counting sort by wave ballots), so that ci is (nearly) wave-uniform in phase 2.  Bucketing each
	// lane's candidates by THEIR component as well was tried: the per-bucket wave-max trip counts (42 vs 29 trips) ate the
	// gain.  Upper bound with both uniform (components in slabs): 24.1 -> 11.6 ms for the 10^7-molecule five-component set.
	const bool by_comp = staged && ncomp > 1 && n_i <= (uint32_t)OWNCAP;
	if (by_comp) {
		const int lane = tid & 63, wv = tid >> 6;
		int myc[2];
		for (int r = 0; r < 2; ++r) {
			const uint32_t it = (uint32_t)(r * NT + tid);
			int c = -1;
			if (it < n_i) {
				int lo;
				locate(it, lo);
				const int cx = lo % BX, cy = (lo / BX) % BY, cz = lo / (BX * BY);
				c = scid[cstart[((cz + HW) * RY + (cy + HW)) * RX + (cx + HW)] + (it - bstart[lo])];
			}
			myc[r] = c;
			for (int c8 = 0; c8 < ncomp; ++c8) {
				const unsigned long long m = __ballot(c == c8);
				if (lane == 0) ccnt[r][wv][c8] = (uint32_t)__popcll(m);
			}
		}
		__syncthreads();
		if (tid == 0) {
			uint32_t run = 0;
			for (int c8 = 0; c8 < ncomp; ++c8)
				for (int r = 0; r < 2; ++r)
					for (int w = 0; w < NT / 64; ++w) {
						const uint32_t t = ccnt[r][w][c8];
						ccnt[r][w][c8] = run;
						run += t;
					}
		}
		__syncthreads();
		for (int r = 0; r < 2; ++r)
			for (int c8 = 0; c8 < ncomp; ++c8) {
				const unsigned long long m = __ballot(myc[r] == c8);
				if (myc[r] == c8) oord[ccnt[r][wv][c8] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(r * NT + tid);
			}
		__syncthreads();
	}
	double u6_t = 0., uX_t = 0., rf_t = 0., vir_t = 0.;
	uint16_t* const mylist = lst + tid;
	for (uint32_t base = 0; base < n_i; base += NT) {  // one pass unless the brick holds more than 256 molecules
		const uint32_t slot_i = base + (uint32_t)tid;
		if (slot_i >= n_i) continue;
		const uint32_t it = by_comp ? (uint32_t)oord[slot_i] : slot_i;
		int lo = 0, hi = NBC;
		while (hi - lo > 1) {
			const int mid = (lo + hi) >> 1;
			if (bstart[mid] <= it) lo = mid;
			else hi = mid;
		}
		const int cx = lo % BX, cy = (lo / BX) % BY, cz = lo / (BX * BY);
		const int rcell = ((cz + HW) * RY + (cy + HW)) * RX + (cx + HW);
		const uint32_t k = it - bstart[lo];
		const uint32_t ii = cstart[rcell] + k;
		const uint32_t gi = gbeg[rcell] + k;
		const int rowbase = (cz * RY + cy) * RX + cx;  // first cell of row 0 of the neighbourhood
		MolAcc acc;
		acc.F = {0., 0., 0.};
		acc.M = {0., 0., 0.};
		acc.Vi = {0., 0., 0.};
		acc.u6 = acc.uX = acc.rf = acc.vir = 0.;
		if (staged) {
			const V3 ri = {sx[ii], sy[ii], sz[ii]};
			const int ci = ONEC ? 0 : (int)scid[ii];  // one component: every table index is a compile-time constant -> scalar loads
			const Rot Ri = HAS_ROT ? rot_of(sq0[ii], sq1[ii], sq2[ii], sq3[ii]) : rot_of(1., 0., 0., 0.);
			auto pair = [&](uint32_t j) {  // candidate j (LDS index): exact test done by the caller
				const V3 rj = {sx[j], sy[j], sz[j]};
				const V3 drm = ri - rj;
				const Rot Rj = HAS_ROT ? rot_of(sq0[j], sq1[j], sq2[j], sq3[j]) : rot_of(1., 0., 0., 0.);
				mol_pair<WITH_VI>(*ctab, ci, ri, Ri, ONEC ? 0 : (int)scid[j], rj, Rj, drm, dot(drm, drm) < rclj2, 0.5, acc);
			};
			// Phase 1 / phase 2 in windows of CAPL hits (one window unless the neighbourhood is very dense): the walk over
			// the 9 neighbour rows appends the hits number [done, done + CAPL) to the per-lane list, the molecule-pair body
			// runs over the list.  The body is instantiated ONCE (it is large: a second inlined copy for an overflow path
			// doubled the kernel's time through instruction-cache misses).  Hits are processed in candidate order, i.e.
			// the generic kernel's summation order: results are bitwise the same.
			uint32_t done = 0, seen;
			do {
				seen = 0;
				uint32_t cnt = 0;
				for (int row = 0; row < NROWS; ++row) {
					const int r0 = rowbase + (row / NW) * (RY * RX) + (row % NW) * RX;
					const uint32_t jb = cstart[r0], je = cstart[r0 + NW];
					for (uint32_t j = jb; j < je; ++j) {
						const double dx = ri.x - sx[j], dy = ri.y - sy[j], dz = ri.z - sz[j];
						const double dd = dx * dx + dy * dy + dz * dz;
						const bool hit = (dd < rc2) & (dd != 0.);  // dd == 0: the molecule itself
						const bool take = hit & (seen >= done) & (seen < done + (uint32_t)CAPL);
						mylist[(take ? seen - done : (uint32_t)CAPL) * NT] = (uint16_t)j;
						cnt += take ? 1u : 0u;
						seen += hit ? 1u : 0u;
					}
				}
				for (uint32_t s = 0; s < cnt; ++s) pair(mylist[s * NT]);
				done += cnt;
			} while (seen > done);
		} else {
			// shell does not fit the staging area: same walk straight from global memory
			const V3 ri = {P.x[gi], P.y[gi], P.z[gi]};
			const int ci = ONEC ? 0 : P.cid[gi];
			Rot Ri = rot_of(1., 0., 0., 0.);
			if (HAS_ROT) {
				const double w = P.q0[gi], x = P.q1[gi], y = P.q2[gi], z = P.q3[gi];
				const double inv = 1. / sqrt(w * w + x * x + y * y + z * z);
				Ri = rot_of(w * inv, x * inv, y * inv, z * inv);
			}
			for (int row = 0; row < NROWS; ++row) {
				const int r0 = rowbase + (row / NW) * (RY * RX) + (row % NW) * RX;
				for (int c = r0; c < r0 + NW; ++c) {
					const uint32_t g0 = gbeg[c], n = cstart[c + 1] - cstart[c];
					for (uint32_t j = g0; j < g0 + n; ++j) {
						if (j == gi) continue;
						const V3 rj = {P.x[j], P.y[j], P.z[j]};
						const V3 drm = ri - rj;
						const double dd = dot(drm, drm);
						if (!(dd < rc2) || dd == 0.) continue;
						Rot Rj = rot_of(1., 0., 0., 0.);
						if (HAS_ROT) {
							const double w = P.q0[j], x = P.q1[j], y = P.q2[j], z = P.q3[j];
							const double inv = 1. / sqrt(w * w + x * x + y * y + z * z);
							Rj = rot_of(w * inv, x * inv, y * inv, z * inv);
						}
						mol_pair<WITH_VI>(*ctab, ci, ri, Ri, ONEC ? 0 : P.cid[j], rj, Rj, drm, dd < rclj2, 0.5, acc);
					}
				}
			}
		}
		P.Fx[gi] = acc.F.x;
		P.Fy[gi] = acc.F.y;
		P.Fz[gi] = acc.F.z;
		if (HAS_ROT) {
			P.Mx[gi] = acc.M.x;
			P.My[gi] = acc.M.y;
			P.Mz[gi] = acc.M.z;
		}
		if (WITH_VI) {
			P.Vix[gi] = acc.Vi.x;
			P.Viy[gi] = acc.Vi.y;
			P.Viz[gi] = acc.Vi.z;
		}
		u6_t += acc.u6;
		uX_t += acc.uX;
		rf_t += acc.rf;
		vir_t += acc.vir;
	}
	ms_block_reduce4(u6_t, uX_t, rf_t, vir_t, P.partials, red);
}

template <int BX, int BY, int BZ, int CAPJ_ROT, int CAPJ_NOROT, int CAPL>
static bool launch_ms(ForceParams p, BrickLists* bl, bool with_vi, bool has_rot, bool onec, hipStream_t s, uint32_t* nblocks,
					  size_t partials_cap) {
	const Grid& g = p.g;
	const int nbx = (g.box[0] + BX - 1) / BX, nby = (g.box[1] + BY - 1) / BY, nbz = (g.box[2] + BZ - 1) / BZ;
	if ((long)nbx * nby * nbz <= 0 || (long)nbx * nby * nbz > 0x7ffffff0L) return false;
	const long nb = plan_bricks(p, bl, BX, BY, BZ, nbx, nby, nbz);
	if ((size_t)nb > partials_cap) return false;
	*nblocks = (uint32_t)nb;
	if (nb == 0) return true;
	const dim3 grid((uint32_t)nb), block(MTPB);
	auto go = [&](auto vi, auto rot, auto one) {
		constexpr bool VI = decltype(vi)::value, ROT = decltype(rot)::value, ONE = decltype(one)::value;
		hipLaunchKernelGGL((k_force_ms_brick<BX, BY, BZ, ROT ? CAPJ_ROT : CAPJ_NOROT, CAPL, VI, ROT, ONE>), grid, block, 0, s, p, p.ct, nbx, nby, nbz);
	};
	auto pick = [&](auto vi, auto rot) {
		if (onec) go(vi, rot, std::true_type{});
		else go(vi, rot, std::false_type{});
	};
	if (has_rot) {
		if (with_vi) pick(std::true_type{}, std::true_type{});
		else pick(std::false_type{}, std::true_type{});
	} else {
		if (with_vi) pick(std::true_type{}, std::false_type{});
		else pick(std::false_type{}, std::false_type{});
	}
	return true;
}

// LDS budget 80 KB per workgroup (2 per CU = 2 waves / SIMD, the occupancy the ~220-VGPR pair body allows anyway).
// Staging costs 57 B / molecule with quaternions, 25 B without; a list row is 256 x 2 B.
//   sparse variants (<= 4.6 molecules / cell): 31 list rows (16 KB), CAPJ 1080 / 2470
//   dense  variants:                           63 list rows (32 KB), CAPJ  800 / 1800
// The brick must also hold about 256 owned molecules: with two 80 KB workgroups per CU, idle waves are lost occupancy
// (measured: ethane at 2 molecules per cell in 4x2x2-cell bricks = 32 owned molecules per workgroup ran no faster than
// the generic kernel).  So the brick shape follows the mean cell occupancy: the largest shape whose shell still fits the
// staging area with 8 % headroom.  Returns false (-> k_force_generic) when even the smallest brick would overflow.
bool launch_force_ms(const ForceParams& p, bool with_vi, bool has_rot, bool onec, hipStream_t s, uint32_t* nblocks,
					 size_t partials_cap, double mean_per_cell, BrickLists* bl) {
	if (p.g.hw != 1 || p.ct == nullptr) return false;
	const double m = mean_per_cell * 1.08;
	const double cs = has_rot ? 1050. : 2400., cd = has_rot ? 770. : 1730.;
	if (m * (10 * 6 * 6) <= cs) return launch_ms<8, 4, 4, 1050, 2400, 31>(p, bl, with_vi, has_rot, onec, s, nblocks, partials_cap);
	if (m * (6 * 6 * 6) <= cs) return launch_ms<4, 4, 4, 1050, 2400, 31>(p, bl, with_vi, has_rot, onec, s, nblocks, partials_cap);
	if (m * (6 * 6 * 4) <= cd) return launch_ms<4, 4, 2, 770, 1730, 63>(p, bl, with_vi, has_rot, onec, s, nblocks, partials_cap);
	if (m * (6 * 4 * 4) <= cd) return launch_ms<4, 2, 2, 770, 1730, 63>(p, bl, with_vi, has_rot, onec, s, nblocks, partials_cap);
	if (m * (4 * 4 * 4) <= cd) return launch_ms<2, 2, 2, 770, 1730, 63>(p, bl, with_vi, has_rot, onec, s, nblocks, partials_cap);
	return false;
}

}  // namespace ls1
