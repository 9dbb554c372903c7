// kernels_force_mslist.hip — neighbour lists for MULTI-SITE component sets: a type-sorted stream of molecule pairs per wave.
//
// What the counters said about the per-step multi-site kernels (kernels_force_ms.hip, kernels_force_sites.hip):
// at the 2-5 molecules per cell of the reference's multi-site systems less than half of a launch is pair arithmetic; the rest is a
// per-brick chain of staging / scan / search phases redone every step, a search whose nine per-lane row loops run to the wave
// maximum, pair loops at ~50 % lane use (7-18 neighbours per molecule: the longest list of 64 lanes is twice the mean) and, with
// several components, a body that diverges over the component pair of every lane.  This file removes all four:
//
//   * the SEARCH runs once per list lifetime (cell grid, halo shell and lists with rc + skin; the device-side displacement
//     bound of the drift passes decides about the rebuild — the machinery of the single-centre lists, ls1hip_update);
//   * the list is not per molecule but per WAVE: a group of 128 consecutive owned molecules owns one contiguous block of
//     molecule PAIRS (i local, j); a lane of the force pass takes one pair, so every lane has work whatever the neighbour counts
//     of the individual molecules are (blocks are padded to a multiple of 64 only);
//   * within a block the pairs are sorted by COMPONENT PAIR (c_i, c_j): a trip of 64 pairs sees one or two component pairs,
//     each evaluated with wave-uniform component indices — site loops of uniform length, parameter tables through scalar loads;
//   * no staging phases and no barriers: the group's own molecules sit in 13 KB of LDS per wave, partners are gathered through
//     L2 (cell-sorted arrays), and a pair that crosses a periodic face refers to the SOURCE molecule plus a shift index instead
//     of a halo copy — lists survive without any halo refresh (positions AND orientations of images follow their source).
//
// Forces are one-sided as everywhere in this library (each ordered pair evaluated for the molecule that receives the force):
// the pair's force / torque on i is added to i's accumulator in the wave's LDS block (ds_add_f64 by the lanes of ONE wave: the
// order of same-address additions is that of the hardware's lane serialisation, reproducible from run to run — checked by
// tests/test_gpu_multisite_lists.py), written out once per molecule, coalesced.  "Wavefront-segmented force reduction" in
// north_star's words, without global atomics.
//
// Reference semantics: pair set and masks as kernels_force.hip (VectorizedCellProcessor.cpp:2734-2821, centre-of-mass cutoff,
// strict <, r^2 != 0), bodies = mol_pair (molpair.hpp: potforce.h:282-503), macroscopic sums with weight 1/2 per ordered pair.
// Precedent for list reuse in the reference: AutoPasContainer.cpp:281-346.
#include "common.hpp"

namespace ls1 {

constexpr int MSG = 128;                    // molecules per group (= per wave of the force pass)
constexpr uint32_t MSL_IDX = 0x07ffffffu;  // pair entry: bits 0-26 molecule index, bits 27-31 shift index (13 = none)
constexpr int MSL_MAXT = MAXC * MAXC;       // component pairs

int msl_group_size() { return MSG; }

__device__ __forceinline__ uint32_t msl_wave_sum(uint32_t v) {
	for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
	return v;
}
__device__ __forceinline__ double msl_wave_sum_d(double v) {
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
	return v;
}

// Visit every molecule j != p of the 27 cells around p's cell with centre distance < rcl (list cutoff).
template <class F>
__device__ __forceinline__ void msl_walk(const ForceParams& P, uint32_t p, double rcl2, F&& hit) {
	int cx, cy, cz;
	cell_coords(P.g, (int)P.ckey[p], cx, cy, cz);
	const double xi = P.x[p], yi = P.y[p], zi = P.z[p];
	for (int dz = -1; dz <= 1; ++dz)
		for (int dy = -1; dy <= 1; ++dy) {
			// the three cells of an x row are consecutive in the cell table: one run of molecules when they are all owned or
			// all halo cells, else walked cell by cell (owned and halo molecules live in different index segments)
			for (int dx = -1; dx <= 1; ++dx) {
				const int c2 = cell_index(P.g, cx + dx, cy + dy, cz + dz);
				const uint32_t jb = P.cell_begin[c2], je = P.cell_end[c2];
				for (uint32_t j = jb; j < je; ++j) {
					const double ex = xi - P.x[j], ey = yi - P.y[j], ez = zi - P.z[j];
					const double dd = ex * ex + ey * ey + ez * ez;
					if (dd < rcl2 && j != p) hit(j);
				}
			}
		}
}

// ---- BUILD 1: pairs per group -----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(MSG) k_msl_count(ForceParams P, uint32_t* grp_cnt) {
	__shared__ uint32_t wsum[MSG / 64];
	const uint32_t n_real = P.n_fixed ? P.n_fixed : P.cnt->n_real;
	const uint32_t p = blockIdx.x * MSG + threadIdx.x;
	uint32_t cnt = 0;
	if (p < n_real) msl_walk(P, p, P.vl_rc2, [&](uint32_t) { ++cnt; });
	cnt = msl_wave_sum(cnt);
	if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
	__syncthreads();
	if (threadIdx.x == 0) grp_cnt[blockIdx.x] = wsum[0] + wsum[1];
}

// ---- BUILD 2: block offsets (each block padded to a multiple of 64 pairs); off[ngroups] = total -------------------------------
__global__ void __launch_bounds__(1024) k_msl_scan(const uint32_t* grp_cnt, uint32_t ngroups, uint32_t* off, DevCounters* cnt) {
	__shared__ unsigned long long wtot[16];
	__shared__ unsigned long long carry_s;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	if (tid == 0) carry_s = 0ull;
	__syncthreads();
	for (uint32_t base = 0; base < ngroups; base += 1024u) {
		const uint32_t g = base + (uint32_t)tid;
		const unsigned long long mine = g < ngroups ? (unsigned long long)((grp_cnt[g] + 63u) & ~63u) : 0ull;
		unsigned long long incl = mine;
		for (int o = 1; o < 64; o <<= 1) {
			const unsigned long long t = __shfl_up(incl, o);
			if (lane >= o) incl += t;
		}
		if (lane == 63) wtot[wv] = incl;
		__syncthreads();
		unsigned long long pre = carry_s;
		for (int w = 0; w < wv; ++w) pre += wtot[w];
		const unsigned long long excl = pre + incl - mine;
		if (g < ngroups) off[g] = excl > 0xffffffffull ? 0xffffffffu : (uint32_t)excl;
		__syncthreads();
		if (tid == 1023) carry_s = pre + incl;
		__syncthreads();
	}
	if (tid == 0) {
		off[ngroups] = carry_s > 0xffffffffull ? 0xffffffffu : (uint32_t)carry_s;
		cnt->msl_total = carry_s;
	}
}

// ---- BUILD 3: fill the blocks, sorted by (component pair, local molecule, candidate order) -------------------------------------
// Deterministic counting sort without atomics: walk 1 tallies per lane and component pair, a type-major scan turns the tallies
// into private write cursors, walk 2 writes.
__global__ void __launch_bounds__(MSG) k_msl_fill(ForceParams P, const uint32_t* __restrict__ off, const uint32_t* __restrict__ hsrc,
												  const uint8_t* __restrict__ hdir, uint32_t* __restrict__ out_j,
												  uint8_t* __restrict__ out_il, int ncomp) {
	__shared__ uint32_t cur[MSL_MAXT * MSG];  // [type][lane]
	__shared__ uint32_t wsum[2];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t n_real = P.n_fixed ? P.n_fixed : P.cnt->n_real;
	const uint32_t p = blockIdx.x * MSG + (uint32_t)tid;
	const bool active = p < n_real;
	const int ntypes = ncomp * ncomp;
	for (int t = 0; t < ntypes; ++t) cur[t * MSG + tid] = 0;
	const int ci = (active && ncomp > 1) ? P.cid[p] : 0;
	if (active) msl_walk(P, p, P.vl_rc2, [&](uint32_t j) { cur[((ncomp > 1 ? ci * ncomp + P.cid[j] : 0)) * MSG + tid] += 1u; });
	__syncthreads();
	// type-major exclusive scan: cursor[t][lane] = pairs of all earlier types + pairs of type t of earlier lanes
	uint32_t run = 0;
	for (int t = 0; t < ntypes; ++t) {
		const uint32_t mine = cur[t * MSG + tid];
		uint32_t incl = mine;
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
			if (lane >= o) incl += v;
		}
		if (lane == 63) wsum[wv] = incl;
		__syncthreads();
		const uint32_t pre = wv ? wsum[0] : 0u;
		cur[t * MSG + tid] = run + pre + incl - mine;
		run += wsum[0] + wsum[1];
		__syncthreads();
	}
	const uint32_t total = run, padded = (total + 63u) & ~63u, o0 = off[blockIdx.x];
	if (active)
		msl_walk(P, p, P.vl_rc2, [&](uint32_t j) {
			const int t = ncomp > 1 ? ci * ncomp + P.cid[j] : 0;
			const uint32_t at = o0 + cur[t * MSG + tid]++;
			uint32_t e = j | (13u << 27);
			if (j >= n_real) {  // halo copy: a local periodic image refers to its source molecule + the shift of its direction
				const uint32_t k = j - n_real, src = hsrc[k];
				if (src != 0xffffffffu) e = src | ((uint32_t)hdir[k] << 27);
			}
			out_j[at] = e;
			out_il[at] = (uint8_t)tid;
		});
	for (uint32_t k = total + (uint32_t)tid; k < padded; k += MSG) {
		out_j[o0 + k] = 0u | (13u << 27);
		out_il[o0 + k] = 0xffu;
	}
}

// ---- REUSE: forces from the pair stream, one wave per group ----------------------------------------------------------------------
template <bool WITH_ROT>
__global__ void __launch_bounds__(64) k_force_ms_list(ForceParams P, const uint32_t* __restrict__ off, const uint32_t* __restrict__ pj,
													  const uint8_t* __restrict__ pil, const double* __restrict__ shift27) {
	__shared__ double sr[3][MSG];
	__shared__ double sq[WITH_ROT ? 4 : 1][MSG];
	__shared__ int sci[MSG];
	__shared__ double acc[WITH_ROT ? 6 : 3][MSG];
	__shared__ double ssh[27 * 3];
	const int lane = threadIdx.x;
	const uint32_t n_real = P.n_fixed ? P.n_fixed : P.cnt->n_real;
	const uint32_t p0 = blockIdx.x * MSG;
	const CompTable& ct = *P.ct;
	const int ncomp = ct.ncomp;
	for (int k = lane; k < MSG; k += 64) {
		const uint32_t p = p0 + (uint32_t)k;
		const bool ok = p < n_real;
		sr[0][k] = ok ? P.x[p] : 0.;
		sr[1][k] = ok ? P.y[p] : 0.;
		sr[2][k] = ok ? P.z[p] : 0.;
		if (WITH_ROT) {
			// FullMolecule::setupSoACache normalises q before rotating (FullMolecule.cpp:720)
			double w = ok ? P.q0[p] : 1., x = ok ? P.q1[p] : 0., y = ok ? P.q2[p] : 0., z = ok ? P.q3[p] : 0.;
			const double inv = 1. / sqrt(w * w + x * x + y * y + z * z);
			sq[0][k] = w * inv;
			sq[1][k] = x * inv;
			sq[2][k] = y * inv;
			sq[3][k] = z * inv;
		}
		sci[k] = (ok && ncomp > 1) ? P.cid[p] : 0;
		for (int a = 0; a < (WITH_ROT ? 6 : 3); ++a) acc[a][k] = 0.;
	}
	for (int k = lane; k < 81; k += 64) ssh[k] = shift27[k];
	__syncthreads();  // (one wave: orders the LDS writes above before the reads below)
	const double rc2 = ct.rc2, rclj2 = ct.rclj2;
	MolAcc a;
	a.F = {0., 0., 0.};
	a.M = {0., 0., 0.};
	a.Vi = {0., 0., 0.};
	a.u6 = a.uX = a.rf = a.vir = 0.;
	const uint32_t b0 = off[blockIdx.x], b1 = off[blockIdx.x + 1];
	// the next trip's pair record is requested before the current trip's bodies run
	uint32_t e = 0u | (13u << 27);
	uint32_t il = 0xffu;
	if (b0 < b1) {
		e = pj[b0 + lane];
		il = pil[b0 + lane];
	}
	for (uint32_t b = b0; b < b1; b += 64u) {
		const uint32_t e_now = e, il_now = il;
		if (b + 64u < b1) {
			e = pj[b + 64u + lane];
			il = pil[b + 64u + lane];
		}
		const bool valid = il_now != 0xffu;
		const uint32_t j = e_now & MSL_IDX, sh = e_now >> 27, k = valid ? il_now : 0u;
		const V3 ri = {sr[0][k], sr[1][k], sr[2][k]};
		const V3 rj = {P.x[j] + ssh[3 * sh], P.y[j] + ssh[3 * sh + 1], P.z[j] + ssh[3 * sh + 2]};
		const V3 drm = ri - rj;
		const double dd = dot(drm, drm);
		const bool in = valid && dd < rc2 && dd != 0.;
		int tkey = 0;
		Rot Ri = rot_of(1., 0., 0., 0.), Rj = Ri;
		if (in) {
			if (ncomp > 1) tkey = sci[k] * MAXC + P.cid[j];
			if (WITH_ROT) {
				Ri = rot_of(sq[0][k], sq[1][k], sq[2][k], sq[3][k]);
				double w = P.q0[j], x = P.q1[j], y = P.q2[j], z = P.q3[j];
				const double inv = 1. / sqrt(w * w + x * x + y * y + z * z);
				Rj = rot_of(w * inv, x * inv, y * inv, z * inv);
			}
		}
		a.F = {0., 0., 0.};
		a.M = {0., 0., 0.};
		// one component pair at a time, with wave-uniform component indices (the blocks are sorted by component pair: a trip
		// normally holds one or two of them)
		unsigned long long todo = __ballot(in);
		while (todo) {
			const int first = __ffsll((long long)todo) - 1;
			const int t = __builtin_amdgcn_readlane(tkey, first);
			const bool mine = in && tkey == t;
			if (mine) mol_pair<false>(ct, t / MAXC, ri, Ri, t % MAXC, rj, Rj, drm, dd < rclj2, 0.5, a);
			todo &= ~__ballot(mine);
		}
		if (in) {
			unsafeAtomicAdd(&acc[0][k], a.F.x);
			unsafeAtomicAdd(&acc[1][k], a.F.y);
			unsafeAtomicAdd(&acc[2][k], a.F.z);
			if (WITH_ROT) {
				unsafeAtomicAdd(&acc[3][k], a.M.x);
				unsafeAtomicAdd(&acc[4][k], a.M.y);
				unsafeAtomicAdd(&acc[5][k], a.M.z);
			}
		}
	}
	__syncthreads();
	for (int k = lane; k < MSG; k += 64) {
		const uint32_t p = p0 + (uint32_t)k;
		if (p < n_real) {
			P.Fx[p] = acc[0][k];
			P.Fy[p] = acc[1][k];
			P.Fz[p] = acc[2][k];
			if (WITH_ROT) {
				P.Mx[p] = acc[3][k];
				P.My[p] = acc[4][k];
				P.Mz[p] = acc[5][k];
			}
		}
	}
	const double u6 = msl_wave_sum_d(a.u6), uX = msl_wave_sum_d(a.uX), rf = msl_wave_sum_d(a.rf), vir = msl_wave_sum_d(a.vir);
	if (lane == 0) {
		double* out = P.partials + (size_t)blockIdx.x * 4;
		out[0] = u6;
		out[1] = uX;
		out[2] = rf;
		out[3] = vir;
	}
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
uint32_t msl_groups(uint32_t n_real) { return (n_real + MSG - 1) / MSG; }

void launch_msl_count(const ForceParams& p, uint32_t* grp_cnt, uint32_t* off, hipStream_t s) {
	const uint32_t ng = msl_groups(p.n_real_cap);
	if (ng == 0) return;
	hipLaunchKernelGGL(k_msl_count, dim3(ng), dim3(MSG), 0, s, p, grp_cnt);
	hipLaunchKernelGGL(k_msl_scan, dim3(1), dim3(1024), 0, s, grp_cnt, ng, off, p.cnt);
}

void launch_msl_fill(const ForceParams& p, const uint32_t* off, const uint32_t* hsrc, const uint8_t* hdir, uint32_t* out_j,
					 uint8_t* out_il, int ncomp, hipStream_t s) {
	const uint32_t ng = msl_groups(p.n_real_cap);
	if (ng == 0) return;
	hipLaunchKernelGGL(k_msl_fill, dim3(ng), dim3(MSG), 0, s, p, off, hsrc, hdir, out_j, out_il, ncomp);
}

bool launch_force_ms_list(const ForceParams& p, bool has_rot, const uint32_t* off, const uint32_t* pj, const uint8_t* pil,
						  const double* shift27, hipStream_t s, uint32_t* nblocks, size_t partials_cap) {
	const uint32_t ng = msl_groups(p.n_real_cap);
	if ((size_t)ng > partials_cap || p.which != 0) return false;
	*nblocks = ng;
	if (ng == 0) return true;
	if (has_rot) hipLaunchKernelGGL((k_force_ms_list<true>), dim3(ng), dim3(64), 0, s, p, off, pj, pil, shift27);
	else hipLaunchKernelGGL((k_force_ms_list<false>), dim3(ng), dim3(64), 0, s, p, off, pj, pil, shift27);
	return true;
}

}  // namespace ls1
