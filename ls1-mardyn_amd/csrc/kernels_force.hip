// kernels_force.hip — the generic pair-force kernel (any component set, any density).
//
// One lane per owned molecule i ("full shell": the lane accumulates everything its molecule receives, so there
// are no force atomics and no cross-lane reductions in the hot loop; the result is deterministic).  The lane walks
// the (2*hw+1)^3 cell neighbourhood of its cell through cell_begin/cell_end and reads neighbour data straight from
// the cell-sorted SoA in HBM/L2.  This is the always-correct path: the faster LDS-tiled 1CLJ kernel
// (kernels_force_lj.hip) falls back to it for pathological densities, and the multi-site / electrostatic
// component sets use it directly.
//
// Reference semantics restated (see also molpair.hpp):
//   pair set + halo policy   C08BasedTraversals::processBaseCell (LinkedCellTraversals/C08BasedTraversals.h:48-99),
//                            VectorizedCellProcessor::processCell/processCellPair (VectorizedCellProcessor.cpp:2734-2821)
//   masks                    strict r^2 < rc^2 on molecule centres, r^2 != 0 (SIMD_VectorizedCellProcessorHelpers.h:344-422)
//   macroscopic sums         every unordered pair once: here each ordered pair (i owned, j owned or halo copy)
//                            contributes 1/2, which sums to exactly the reference's "halo cell with larger index"
//                            rule because a pair crossing a periodic/rank boundary is seen once from either side.
//   endTraversal             upot = u6/6 + uX + rf, virial = vir + 3 rf (VectorizedCellProcessor.cpp:155-156)
#include "common.hpp"

namespace ls1 {

constexpr int FTPB = 128;

__device__ __forceinline__ double wave_max(double v) {
	for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o));
	return v;
}
__device__ __forceinline__ double wave_sum(double v) {
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
	return v;
}

// block reduction of 4 doubles -> partials[blockIdx.x][4]
__device__ __forceinline__ void block_reduce4(double v0, double v1, double v2, double v3, double* partials, int nwaves) {
	__shared__ double red[16][4];
	v0 = wave_sum(v0);
	v1 = wave_sum(v1);
	v2 = wave_sum(v2);
	v3 = wave_sum(v3);
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if (lane == 0) {
		red[w][0] = v0;
		red[w][1] = v1;
		red[w][2] = v2;
		red[w][3] = v3;
	}
	__syncthreads();
	if (threadIdx.x < 4) {
		double s = 0.;
		for (int i = 0; i < nwaves; ++i) s += red[i][threadIdx.x];
		partials[(size_t)blockIdx.x * 4 + threadIdx.x] = s;
	}
}

__device__ __forceinline__ bool cell_is_innermost(const Grid& g, int cx, int cy, int cz) {
	// no halo cell in the (2hw+1)^3 neighbourhood: LinkedCells "innermost" cells (CellBorderAndFlagManager.h:96-130)
	const int lo = 2 * g.hw;
	return cx >= lo && cy >= lo && cz >= lo && cx < g.dims[0] - lo && cy < g.dims[1] - lo && cz < g.dims[2] - lo;
}

// Evaluate everything molecule j does to molecule i (centre distance already tested: dd < rc2, dd != 0).
template <bool ONE_CLJ, bool WITH_VI, bool HAS_ROT>
__device__ __forceinline__ void generic_pair(const ForceParams& P, int ci, const V3& ri, const Rot& Ri, uint32_t j, const V3& rj,
											  const V3& drm, double dd, double rclj2, MolAcc& acc) {
	if (ONE_CLJ) {
		if (dd < rclj2) {
			V3 f;
			double u;
			lj(drm, dd, P.eps24, P.sig2, f, u);
			acc.F = acc.F + f;
			acc.u6 += 0.5 * (u + P.shift6);
			acc.vir += 0.5 * dot(drm, f);
			if (WITH_VI) {
				acc.Vi.x += 0.5 * drm.x * f.x;
				acc.Vi.y += 0.5 * drm.y * f.y;
				acc.Vi.z += 0.5 * drm.z * f.z;
			}
		}
	} else {
		const int cj = P.cid[j];
		Rot Rj;
		if (HAS_ROT) {
			double w = P.q0[j], x = P.q1[j], y = P.q2[j], z = P.q3[j];
			const double inv = 1. / sqrt(w * w + x * x + y * y + z * z);
			Rj = rot_of(w * inv, x * inv, y * inv, z * inv);
		} else {
			Rj = rot_of(1., 0., 0., 0.);
		}
		mol_pair<WITH_VI>(*P.ct, ci, ri, Ri, cj, rj, Rj, drm, dd < rclj2, 0.5, acc);
	}
}

// One lane per owned molecule, any component set.  Two phases, as in the LJ fast path: the (cheap) centre-distance test
// over the (2hw+1)^3 neighbour cells appends the in-range j to a per-lane list in LDS; the (expensive) molecule-pair
// body — up to 16 x 16 site interactions with on-the-fly rotations — then runs over the list with every lane busy.
// Evaluating the body inside the search loop made the whole wave execute it whenever ANY lane had a hit, i.e. almost
// every iteration at ~15 % useful lanes (measured: ethane 9.8 M molecules 18.5 ms -> see DESIGN.md).  The list keeps the
// candidate order, so the sums are bitwise those of the single loop.  Neighbourhoods with more hits than the list holds
// are processed in several windows of the same two phases.
constexpr int GCAP = 79;  // list entries per lane: (GCAP + 1) x 128 x 4 B = 40 KB -> 4 workgroups (8 waves) / CU, the VGPR limit of the multi-site body
// SPLIT = 4 (dense multi-site neighbourhoods): four adjacent lanes share a molecule, lane `part` walks the neighbour cells
// k = part, part + 4, ...; the four quarter lists together hold ~4 x GCAP hits in ONE window (a single lane would re-scan
// the ~1000 candidates of liquid ethane once per window of 79 hits), the partial sums are combined by two DPP exchanges.
template <bool ONE_CLJ, bool WITH_VI, bool HAS_ROT, int SPLIT = 1>
__global__ void __launch_bounds__(FTPB) k_force_generic(ForceParams P) {
	static_assert(SPLIT == 1 || (SPLIT == 4 && !ONE_CLJ), "SPLIT = 4 is for the multi-site instantiations");
	// slot-major; row GCAP = dummy target of misses / overflow.  The single-centre LJ body is as cheap as the list
	// bookkeeping (measured 5.5 -> 10.5 ms with a list), so ONE_CLJ keeps the single loop and no LDS.
	__shared__ uint32_t glist[ONE_CLJ ? 1 : (GCAP + 1) * FTPB];
	const uint32_t p = (blockIdx.x * FTPB + threadIdx.x) / SPLIT;
	const int part = (int)(threadIdx.x % SPLIT);
	const uint32_t n_real = P.n_fixed ? P.n_fixed : P.cnt->n_real;
	bool active = p < n_real;
	int cx = 0, cy = 0, cz = 0;
	if (active) {
		cell_coords(P.g, (int)P.ckey[p], cx, cy, cz);
		if (P.which == 3) {
			active = !cell_is_halo(P.g, cx, cy, cz);
		} else if (P.which != 0) {
			const bool inner = cell_is_innermost(P.g, cx, cy, cz);
			active = (P.which == 1) ? inner : !inner;
		}
	}
	MolAcc acc;
	acc.F = {0., 0., 0.};
	acc.M = {0., 0., 0.};
	acc.Vi = {0., 0., 0.};
	acc.u6 = acc.uX = acc.rf = acc.vir = 0.;
	unsigned long long nchk = 0, nhit = 0;
	if (active) {
		const V3 ri = {P.x[p], P.y[p], P.z[p]};
		int ci = 0;
		Rot Ri;
		if (!ONE_CLJ) {
			ci = P.cid[p];
			if (HAS_ROT) {
				// FullMolecule::setupSoACache normalises q before rotating (FullMolecule.cpp:720)
				double w = P.q0[p], x = P.q1[p], y = P.q2[p], z = P.q3[p];
				const double inv = 1. / sqrt(w * w + x * x + y * y + z * z);
				Ri = rot_of(w * inv, x * inv, y * inv, z * inv);
			} else {
				Ri = rot_of(1., 0., 0., 0.);
			}
		}
		const double rc2 = ONE_CLJ ? P.rc2 : P.ct->rc2;
		const double rclj2 = ONE_CLJ ? P.rc2 : P.ct->rclj2;
		const int hw = P.g.hw;
		if (ONE_CLJ) {
			for (int dz = -hw; dz <= hw; ++dz)
				for (int dy = -hw; dy <= hw; ++dy)
					for (int dx = -hw; dx <= hw; ++dx) {
						const int c2 = cell_index(P.g, cx + dx, cy + dy, cz + dz);
						const uint32_t jb = P.cell_begin[c2], je = P.cell_end[c2];
						for (uint32_t j = jb; j < je; ++j) {
							if (j == p) continue;
							const V3 rj = {P.x[j], P.y[j], P.z[j]};
							const V3 drm = ri - rj;
							const double dd = dot(drm, drm);
							++nchk;
							if (!(dd < rc2) || dd == 0.) continue;
							++nhit;
							generic_pair<ONE_CLJ, WITH_VI, HAS_ROT>(P, ci, ri, Ri, j, rj, drm, dd, rclj2, acc);
						}
					}
		} else {
			// Windows of GCAP hits (one window unless the neighbourhood is very dense): phase 1 appends the hits number
			// [done, done + GCAP) to the per-lane list, phase 2 runs the molecule-pair body over it.  ONE instantiation of
			// the (large) body: a second inlined copy for an overflow path costs more in instruction-cache misses than
			// the windowing does.  Candidate order is kept.
			uint32_t* const mylist = glist + threadIdx.x;
			uint32_t done = 0, seen;
			do {
				seen = 0;
				uint32_t cnt = 0;
				int kcell = 0;
				for (int dz = -hw; dz <= hw; ++dz)
					for (int dy = -hw; dy <= hw; ++dy)
						for (int dx = -hw; dx <= hw; ++dx, ++kcell) {
							if (SPLIT > 1 && kcell % SPLIT != part) continue;
							const int c2 = cell_index(P.g, cx + dx, cy + dy, cz + dz);
							const uint32_t jb = P.cell_begin[c2], je = P.cell_end[c2];
							for (uint32_t j = jb; j < je; ++j) {
								const double ex = ri.x - P.x[j], ey = ri.y - P.y[j], ez = ri.z - P.z[j];
								const double dd = ex * ex + ey * ey + ez * ez;
								const bool hit = (dd < rc2) & (dd != 0.) & (j != p);
								const bool take = hit & (seen >= done) & (seen < done + (uint32_t)GCAP);
								mylist[(take ? seen - done : (uint32_t)GCAP) * FTPB] = j;
								cnt += take ? 1u : 0u;
								seen += hit ? 1u : 0u;
								nchk += (done == 0) ? 1u : 0u;
							}
						}
				for (uint32_t s = 0; s < cnt; ++s) {
					const uint32_t j = mylist[s * FTPB];
					const V3 rj = {P.x[j], P.y[j], P.z[j]};
					const V3 drm = ri - rj;
					generic_pair<ONE_CLJ, WITH_VI, HAS_ROT>(P, ci, ri, Ri, j, rj, drm, dot(drm, drm), rclj2, acc);
				}
				done += cnt;
			} while (seen > done);
			nhit = seen;
		}
	}
	if (SPLIT > 1) {
		// the SPLIT adjacent lanes of a molecule (all active or all inactive) combine their partial sums
		auto sum4 = [](double v) {
			v += __shfl_xor(v, 1);
			v += __shfl_xor(v, 2);
			return v;
		};
		acc.F = {sum4(acc.F.x), sum4(acc.F.y), sum4(acc.F.z)};
		if (HAS_ROT) acc.M = {sum4(acc.M.x), sum4(acc.M.y), sum4(acc.M.z)};
		if (WITH_VI) acc.Vi = {sum4(acc.Vi.x), sum4(acc.Vi.y), sum4(acc.Vi.z)};
		// the macroscopic sums below add every lane's own partial value: no combination needed
	}
	if (active && part == 0) {
		P.Fx[p] = acc.F.x;
		P.Fy[p] = acc.F.y;
		P.Fz[p] = acc.F.z;
		if (HAS_ROT) {
			P.Mx[p] = acc.M.x;
			P.My[p] = acc.M.y;
			P.Mz[p] = acc.M.z;
		}
		if (WITH_VI) {
			P.Vix[p] = acc.Vi.x;
			P.Viy[p] = acc.Vi.y;
			P.Viz[p] = acc.Vi.z;
		}
	}
	block_reduce4(acc.u6, acc.uX, acc.rf, acc.vir, P.partials, FTPB / 64);
	if (P.count_pairs) {
		// wave-aggregated by the compiler; diagnostics only
		atomicAdd(&P.cnt->dist_checks, nchk);
		atomicAdd(&P.cnt->pairs_in_range, nhit);
	}
}

template <bool A, bool B, bool C, int SPLIT = 1>
static void launch_g(const ForceParams& p, dim3 grid, hipStream_t s) {
	hipLaunchKernelGGL((k_force_generic<A, B, C, SPLIT>), grid, dim3(FTPB), 0, s, p);
}

// expected_neighbours: mean number of molecules within the cutoff (density x 4/3 pi rc^3); above ~60 the multi-site
// instantiations run with four lanes per molecule (see k_force_generic)
void launch_force_generic(const ForceParams& p, bool one_clj, bool with_vi, bool has_rot, hipStream_t s,
						  uint32_t* nblocks, double expected_neighbours) {
	const bool split4 = !one_clj && expected_neighbours > 0.75 * GCAP;
	const uint32_t nb = (uint32_t)(((size_t)p.n_real_cap * (split4 ? 4 : 1) + FTPB - 1) / FTPB);
	*nblocks = nb;
	if (nb == 0) return;
	const dim3 grid(nb);
	if (one_clj) {
		if (with_vi) launch_g<true, true, false>(p, grid, s);
		else launch_g<true, false, false>(p, grid, s);
	} else if (has_rot) {
		if (split4) {
			if (with_vi) launch_g<false, true, true, 4>(p, grid, s);
			else launch_g<false, false, true, 4>(p, grid, s);
		} else {
			if (with_vi) launch_g<false, true, true>(p, grid, s);
			else launch_g<false, false, true>(p, grid, s);
		}
	} else {
		if (split4) {
			if (with_vi) launch_g<false, true, false, 4>(p, grid, s);
			else launch_g<false, false, false, 4>(p, grid, s);
		} else {
			if (with_vi) launch_g<false, true, false>(p, grid, s);
			else launch_g<false, false, false>(p, grid, s);
		}
	}
}

__global__ void k_clear_macro(DevCounters* cnt) {
	for (int k = 0; k < 4; ++k) cnt->macro[k] = 0.;
	cnt->dist_checks = 0;
	cnt->pairs_in_range = 0;
}
void launch_clear_macro(DevCounters* cnt, hipStream_t s) { hipLaunchKernelGGL(k_clear_macro, dim3(1), dim3(1), 0, s, cnt); }

// deterministic reduction of the per-workgroup partials in two fixed-shape stages (RED_BLOCKS x 256 threads, then one
// block), ADDED to cnt->macro; the summation order depends only on the number of partials, never on timing.
constexpr int RED_BLOCKS = 128;
__global__ void __launch_bounds__(256) k_force_reduce1(const double* partials, uint32_t nblocks, double* stage, int max2) {
	double v[4] = {0., 0., 0., 0.};
	for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b < nblocks; b += RED_BLOCKS * 256)
		for (int k = 0; k < 4; ++k) {
			const double x = partials[(size_t)b * 4 + k];
			v[k] = (k == 2 && max2) ? fmax(v[k], x) : v[k] + x;
		}
	__shared__ double red[4][4];
	for (int k = 0; k < 4; ++k) v[k] = (k == 2 && max2) ? wave_max(v[k]) : wave_sum(v[k]);
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if (lane == 0)
		for (int k = 0; k < 4; ++k) red[w][k] = v[k];
	__syncthreads();
	if (threadIdx.x < 4) {
		const int k = threadIdx.x;
		stage[blockIdx.x * 4 + k] = (k == 2 && max2) ? fmax(fmax(red[0][k], red[1][k]), fmax(red[2][k], red[3][k]))
													  : red[0][k] + red[1][k] + red[2][k] + red[3][k];
	}
}
struct Reduce2Args {
	int overwrite, kin_in_slot1, vmax_in_slot2, last_pass, lists_rebuilt, local_criterion;
	double dt, limit;
	uint32_t seq;
	volatile uint32_t* flag;
	double* log;
	double target_T;
};
__global__ void __launch_bounds__(RED_BLOCKS) k_force_reduce2(DevCounters* cnt, const double* stage, Reduce2Args m) {
	double v[4];
	for (int k = 0; k < 4; ++k) v[k] = stage[threadIdx.x * 4 + k];
	__shared__ double red[RED_BLOCKS / 64][4];
	for (int k = 0; k < 4; ++k) v[k] = (k == 2 && m.vmax_in_slot2) ? wave_max(v[k]) : wave_sum(v[k]);
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if (lane == 0)
		for (int k = 0; k < 4; ++k) red[w][k] = v[k];
	__syncthreads();
	if (threadIdx.x == 0) {
		double s[4];
		for (int k = 0; k < 4; ++k) {
			s[k] = 0.;
			for (int i = 0; i < RED_BLOCKS / 64; ++i) s[k] = (k == 2 && m.vmax_in_slot2) ? fmax(s[k], red[i][k]) : s[k] + red[i][k];
		}
		if (m.kin_in_slot1) {
			// fused force + integration pass: slot 1 is sum m v^2 after the post-force kick (Leapfrog.cpp:115-131)
			cnt->kin[0] = m.overwrite ? s[1] : cnt->kin[0] + s[1];
			cnt->kin[1] = 0.;
			cnt->kin_n = cnt->n_real;
			cnt->kin_rotdof = 0;
			s[1] = 0.;
			if (m.target_T > 0. && cnt->kin_n > 0) {  // thermostat 0 of Domain::calculateGlobalValues, as k_kin_reduce
				cnt->beta[0] = pow(3.0 * (double)cnt->kin_n * m.target_T / cnt->kin[0], 0.4);
				cnt->beta[1] = 1.0;
			}
		}
		if (m.vmax_in_slot2) {
			cnt->vmax2 = m.overwrite ? s[2] : fmax(cnt->vmax2, s[2]);
			s[2] = 0.;
			if (m.last_pass) {
				// every molecule moves by at most dt * vmax in this step: the sum over the steps since the lists were built
				// bounds the displacement of any molecule (triangle inequality); the lists hold all pairs within rc + skin at
				// build time, so they stay complete while 2 * bound <= skin
				const double b = (m.lists_rebuilt ? 0. : cnt->vl_bound) + m.dt * sqrt(cnt->vmax2);
				cnt->vl_bound = b;
				// a fused pass WITHOUT the per-brick bookkeeping counts with the global speed for every brick (as unfused drifts do)
				if (!m.local_criterion) cnt->vl_base = (m.lists_rebuilt ? 0. : cnt->vl_base) + m.dt * sqrt(cnt->vmax2);
				// local criterion (k_bound_local, kernels_force_verlet.hip): a brick neighbourhood's pair bound decides; it is never
				// earlier than the global one
				bool rebuild = b > m.limit;
				if (m.local_criterion) {
					rebuild = rebuild && cnt->vl_local_excess != 0u;
					cnt->vl_local_excess = 0u;
				}
				if (m.flag) {
					__threadfence_system();
					*m.flag = (m.seq << 1) | (rebuild ? 1u : 0u);
					__threadfence_system();
				}
			}
		}
		for (int k = 0; k < 4; ++k) cnt->macro[k] = m.overwrite ? s[k] : cnt->macro[k] + s[k];  // first pass of a traversal starts the sums
		if (m.log) {
			// VectorizedCellProcessor::endTraversal (VectorizedCellProcessor.cpp:155-156) + the kinetic sums of this step
			m.log[0] = cnt->macro[0] / 6.0 + cnt->macro[1] + cnt->macro[2];
			m.log[1] = cnt->macro[3] + 3.0 * cnt->macro[2];
			if (m.kin_in_slot1) {
				m.log[2] = cnt->kin[0];
				m.log[3] = 0.;
				m.log[4] = (double)cnt->kin_n;
				m.log[5] = 0.;
			}
		}
	}
}

void launch_force_reduce(DevCounters* cnt, const double* partials, uint32_t nblocks, double* stage, hipStream_t s, const ReduceMode& m) {
	if (nblocks == 0) {
		if (m.overwrite) launch_clear_macro(cnt, s);
		return;
	}
	Reduce2Args a;
	a.overwrite = m.overwrite; a.kin_in_slot1 = m.kin_in_slot1; a.vmax_in_slot2 = m.vmax_in_slot2; a.last_pass = m.last_pass;
	a.lists_rebuilt = m.lists_rebuilt; a.local_criterion = m.local_criterion; a.dt = m.dt; a.limit = m.limit; a.seq = m.seq; a.flag = m.flag; a.log = m.log;
	a.target_T = m.target_T;
	hipLaunchKernelGGL(k_force_reduce1, dim3(RED_BLOCKS), dim3(256), 0, s, partials, nblocks, stage, m.vmax_in_slot2 ? 1 : 0);
	hipLaunchKernelGGL(k_force_reduce2, dim3(1), dim3(RED_BLOCKS), 0, s, cnt, stage, a);
}

}  // namespace ls1
