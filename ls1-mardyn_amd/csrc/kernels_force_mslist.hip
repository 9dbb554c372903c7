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
//   * the list is not per molecule but per WAVE: a group of 128 (one LJ-only component: 64) owned molecules owns one contiguous
//     block of molecule PAIRS (i local, j); the lanes of the force pass work through the block 64 pairs at a time, so every lane has
//     work whatever the neighbour counts of the individual molecules are (blocks are padded to a multiple of 64 only);
//   * within a block the pairs are sorted by COMPONENT PAIR (c_i, c_j), then by local molecule; with several components the groups
//     themselves are formed by component (k_msl_groups: a slot map inside windows of 1024 molecules, the state arrays keep their
//     order), so that a group sees few component pairs with long runs;
//   * round 4 — FILTER and QUEUE: a trip of 64 listed pairs only tests the cutoff (first 32 bytes of the partner's record) and
//     appends the pairs inside to a per-wave queue in LDS; the pair BODIES run over 64 queue entries of ONE component pair —
//     wave-uniform component indices (site loops of uniform length, parameter tables through scalar loads), no lanes idling for
//     the pairs of the skin or for the other component pair of a trip;
//   * no staging phases and no barriers: the group's own molecules sit in 9-15 KB of LDS per wave, partners are gathered through
//     L2 as one 64-byte record per molecule, and a pair that crosses a periodic face refers to the SOURCE molecule plus a shift
//     index instead of a halo copy — lists survive without any halo refresh (positions AND orientations of images follow their
//     source);
//   * single-component rigid sets: the epilogue integrates the group's own molecules (leapfrog_body.hpp) — between the steps of
//     ls1hip_run forces and torques never reach memory.
//
// Forces are one-sided as everywhere in this library (each ordered pair evaluated for the molecule that receives the force):
// the pair's force / torque on i is added to i's accumulator in the wave's LDS block (ds_add_f64 by the lanes of ONE wave: the
// order of same-address additions is that of the hardware's lane serialisation, reproducible from run to run — checked by
// tests/test_gpu_multisite_lists.py), written out once per molecule, coalesced.  "Wavefront-segmented force reduction" in
// north_star's words, without global atomics.
//
// Reference semantics: pair set and masks as kernels_force.hip (VectorizedCellProcessor.cpp:2734-2821, centre-of-mass cutoff,
// strict <, r^2 != 0), bodies = mol_pair (molpair.hpp: potforce.h:282-503), macroscopic sums with weight 1/2 per ordered pair.
// Precedent for list reuse in the reference: AutoPasContainer.cpp:281-346; for compacting the pairs inside the cutoff ahead of the
// force body: the gather variant of vectorization/MaskGatherChooser.h:87-105 (hit indices compacted, bodies over full vectors).
// pair arithmetic of THIS translation unit: FMA contraction on, reciprocals / square roots by the hardware estimate + two
// Newton steps (see pairphys.hpp); results stay within 1e-13 of the IEEE bodies of the other kernels
#pragma clang fp contract(fast)
#include <hip/hip_runtime.h>
namespace ls1 {
__device__ __forceinline__ double msl_rcp(double d) {
	double x = __builtin_amdgcn_rcp(d);
	double e = fma(-d, x, 1.0);
	x = fma(x, e, x);
	e = fma(-d, x, 1.0);
	return fma(x, e, x);
}
__device__ __forceinline__ double msl_rsq(double d) {
	double y = __builtin_amdgcn_rsq(d);
	double e = fma(-d * y, y, 1.0);
	y = fma(0.5 * y, e, y);
	e = fma(-d * y, y, 1.0);
	return fma(0.5 * y, e, y);
}
}  // namespace ls1
#define LS1_PAIR_RCP(x) ::ls1::msl_rcp(x)
#define LS1_PAIR_SQRT(x) ((x) * ::ls1::msl_rsq(x))
#include <type_traits>

#include "common.hpp"
#include "leapfrog_body.hpp"

namespace ls1 {

// molecules per group (= per wave of the force pass): a template parameter MSG of the kernels below — 128, or 64 for one LJ-only
// component (msl_group_size: cheap bodies, the occupancy is worth more than long runs)
constexpr int MSG_MAX = 128;
constexpr uint32_t MSL_IDX = 0x07ffffffu;  // pair entry: bits 0-26 molecule index, bits 27-31 shift index (13 = none)
constexpr int MSL_MAXT = MAXC * MAXC;       // component pairs
static_assert(MSG_MAX <= 128 && MAXC * MAXC <= 512, "queue entry: 7 bits local molecule, 9 bits component pair");

int msl_group_size(bool lj_only, int ncomp) { return lj_only && ncomp == 1 ? 64 : MSG_MAX; }

__device__ __forceinline__ uint32_t msl_wave_sum(uint32_t v) {
	for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
	return v;
}
// Orders the LDS accesses of the ONE wave of a workgroup (queue writes of some lanes, reads by others): LDS instructions of a
// wave execute in order, so no instruction is needed — only the compiler must not move them.  (__syncthreads() here would also
// wait for every global load in flight, vmcnt(0): exactly the prefetched gathers the filter pipeline keeps behind the bodies.)
__device__ __forceinline__ void msl_wave_lds_order() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// ---- run accumulation (LJ-only sets) ---------------------------------------------------------------------------------------
// The pairs of a block are sorted by (component pair, local molecule): the lanes of a trip that feed the SAME accumulator are
// neighbours.  Their contributions are summed across lanes first — a segmented inclusive scan inside every row of 16 lanes with DPP
// row shifts (VALU only: no LDS, no cross-row traffic) — and only the LAST lane of a run (inside its row) adds to the LDS
// accumulator: one ds_add_f64 per run and row instead of one per pair, and what is left hardly ever meets another lane on the same
// address (ethane: runs of ~10 pairs; the same-address adds were 38 % of the kernel's LDS cycles).  Deterministic:
// the association of the sums is fixed by the lane positions.
template <int CTRL>
__device__ __forceinline__ int msl_dpp_i(int old, int v) {
	return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false);  // lanes without a source keep `old`
}
template <int CTRL>
__device__ __forceinline__ double msl_dpp_d(double v) {
	const int lo = msl_dpp_i<CTRL>(0, __double2loint(v)), hi = msl_dpp_i<CTRL>(0, __double2hiint(v));
	return __hiloint2double(hi, lo);
}
template <int CTRL, int N>
__device__ __forceinline__ void msl_seg_step(int key, double (&val)[N]) {
	const bool same = msl_dpp_i<CTRL>(-1, key) == key;  // the lane CTRL's shift away (same row) feeds the same accumulator
	double t[N];
#pragma unroll
	for (int c = 0; c < N; ++c) t[c] = msl_dpp_d<CTRL>(val[c]);  // (every lane takes part in the shifts)
	if (same) {  // one exec mask for the N additions instead of two v_cndmask per value
#pragma unroll
		for (int c = 0; c < N; ++c) val[c] += t[c];
	}
}
// after the call: val = sum over the lane's run up to and including the lane (inside its row of 16); returns whether the lane is
// the last of its run inside the row (the one that stores)
template <int N>
__device__ __forceinline__ bool msl_run_sums(int key, double (&val)[N]) {
	msl_seg_step<0x111>(key, val);  // row_shr:1
	msl_seg_step<0x112>(key, val);  // row_shr:2
	msl_seg_step<0x114>(key, val);  // row_shr:4
	msl_seg_step<0x118>(key, val);  // row_shr:8
	return msl_dpp_i<0x101>(-2, key) != key;  // row_shl:1: the next lane of the row (none: the lane ends its row)
}

__device__ __forceinline__ double msl_wave_sum_d(double v) {
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
	return v;
}

// Visit every molecule j != p of the 27 cells around p's cell with centre distance < rcl (list cutoff).  The three cells of an
// x row are neighbours in the cell table; where their molecule ranges are contiguous in memory (all owned or all halo cells:
// the usual case) the row is ONE run of candidates, and the candidates are fetched four at a time (independent loads) — the
// walk is a chain of dependent loads otherwise (cell table -> positions), 27 times per molecule.
template <class F>
__device__ __forceinline__ void msl_walk(const ForceParams& P, uint32_t p, double rcl2, F&& hit) {
	int cx, cy, cz;
	cell_coords(P.g, (int)P.ckey[p], cx, cy, cz);
	const double xi = P.x[p], yi = P.y[p], zi = P.z[p];
	auto run = [&](uint32_t jb, uint32_t je) {
		uint32_t j = jb;
		for (; j + 4u <= je; j += 4u) {
			double ex[4], ey[4], ez[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				ex[u] = xi - P.x[j + u];
				ey[u] = yi - P.y[j + u];
				ez[u] = zi - P.z[j + u];
			}
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const double dd = ex[u] * ex[u] + ey[u] * ey[u] + ez[u] * ez[u];
				if (dd < rcl2 && j + u != p) hit(j + u);
			}
		}
		for (; j < je; ++j) {
			const double ex = xi - P.x[j], ey = yi - P.y[j], ez = zi - P.z[j];
			const double dd = ex * ex + ey * ey + ez * ez;
			if (dd < rcl2 && j != p) hit(j);
		}
	};
	for (int dz = -1; dz <= 1; ++dz)
		for (int dy = -1; dy <= 1; ++dy) {
			const int c0 = cell_index(P.g, cx - 1, cy + dy, cz + dz);
			const uint32_t b0 = P.cell_begin[c0], e0 = P.cell_end[c0], b1 = P.cell_begin[c0 + 1], e1 = P.cell_end[c0 + 1],
						   b2 = P.cell_begin[c0 + 2], e2 = P.cell_end[c0 + 2];
			if (e0 == b1 && e1 == b2) {
				run(b0, e2);
			} else {
				run(b0, e0);
				run(b1, e1);
				run(b2, e2);
			}
		}
}

// ---- BUILD 0 (several components): which molecule sits in which group slot -----------------------------------------------------
// A body pass of the force kernel evaluates up to 64 pairs of ONE component pair; a group of 128 consecutive molecules of a
// five-component mixture has 25 component pairs of ~95 pairs inside the cutoff each: two passes for 1.5 full ones.  Groups are
// therefore formed inside WINDOWS of 8 groups (1024 consecutive molecules, cell order: a few rows of cells): the window's molecules
// are sorted by component (stable), the groups are slices of 128 of that order — one or two components per group, runs of
// ~300 pairs.  The state arrays keep their order; only the lists and the force pass go through this map (0xffffffff = empty slot).
__device__ __forceinline__ uint32_t msl_slot(const ForceParams& P, uint32_t slot) { return P.msl_gm ? P.msl_gm[slot] : slot; }

constexpr int MSW = 1024;  // slots per window
__global__ void __launch_bounds__(MSW) k_msl_groups(ForceParams P, uint32_t* __restrict__ gm, int ncomp, uint32_t nslots) {
	__shared__ uint32_t wcnt[MAXC][MSW / 64];
	const uint32_t n_real = P.n_fixed ? P.n_fixed : P.cnt->n_real;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t s0 = blockIdx.x * MSW, p = s0 + (uint32_t)tid;
	const bool active = p < n_real;
	const int c = active ? P.cid[p] : -1;
	uint32_t rank = 0;
	for (int k = 0; k < ncomp; ++k) {
		const unsigned long long m = __ballot(c == k);
		if (lane == 0) wcnt[k][wv] = (uint32_t)__popcll(m);
		if (c == k) rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
	}
	__syncthreads();
	uint32_t at = rank, total = 0;
	for (int k = 0; k < ncomp; ++k)
		for (int w = 0; w < MSW / 64; ++w) {
			const uint32_t n = wcnt[k][w];
			total += n;
			if (k < c || (k == c && w < wv)) at += n;
		}
	if (active) gm[s0 + at] = p;
	if ((uint32_t)tid >= total && s0 + (uint32_t)tid < nslots) gm[s0 + (uint32_t)tid] = 0xffffffffu;
}

// ---- BUILD 1: pairs per group; the hits of every molecule are kept (first MSL_CAP of them) for the fill kernel ------------------
constexpr int MSL_CAP = 32;  // captured hits per molecule: scratch[k][p], k < MSL_CAP (coalesced over p); more: the fill kernel walks again
template <int MSG>
__global__ void __launch_bounds__(MSG) k_msl_count(ForceParams P, uint32_t* grp_cnt, uint32_t* __restrict__ scratch,
												   uint16_t* __restrict__ mcnt, uint32_t stride) {
	__shared__ uint32_t wsum[MSG / 64];
	const uint32_t n_real = P.n_fixed ? P.n_fixed : P.cnt->n_real;
	// (always in the order of the state arrays — the lanes of a wave walk the same cells; with several components the group sums are
	// taken through the slot map afterwards, k_msl_group_sums)
	const uint32_t p = blockIdx.x * MSG + threadIdx.x;
	uint32_t cnt = 0;
	if (p < n_real) {
		msl_walk(P, p, P.vl_rc2, [&](uint32_t j) {
			if (cnt < (uint32_t)MSL_CAP) scratch[(size_t)cnt * stride + p] = j;
			++cnt;
		});
		mcnt[p] = (uint16_t)min(cnt, 0xffffu);
	}
	cnt = msl_wave_sum(cnt);
	if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
	__syncthreads();
	if (threadIdx.x == 0 && !P.msl_gm) grp_cnt[blockIdx.x] = wsum[0] + (MSG > 64 ? wsum[MSG / 64 - 1] : 0u);
}
template <int MSG>
__global__ void __launch_bounds__(MSG) k_msl_group_sums(ForceParams P, uint32_t* grp_cnt, const uint16_t* __restrict__ mcnt) {
	__shared__ uint32_t wsum[MSG / 64];
	const uint32_t n_real = P.n_fixed ? P.n_fixed : P.cnt->n_real;
	const uint32_t p = P.msl_gm[blockIdx.x * MSG + threadIdx.x];
	uint32_t cnt = msl_wave_sum(p < n_real ? (uint32_t)mcnt[p] : 0u);
	if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
	__syncthreads();
	if (threadIdx.x == 0) grp_cnt[blockIdx.x] = wsum[0] + (MSG > 64 ? wsum[MSG / 64 - 1] : 0u);
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
// Deterministic counting sort without atomics: pass 1 tallies per lane and component pair, a type-major scan turns the tallies
// into private write cursors, pass 2 writes.  Both passes read the hits the count kernel captured (no second / third walk of the
// cell neighbourhood); a molecule with more hits than the capture holds walks again.
template <class F>
__device__ __forceinline__ void msl_hits(const ForceParams& P, uint32_t p, uint32_t cnt, const uint32_t* __restrict__ scratch,
										 uint32_t stride, F&& hit) {
	if (cnt <= (uint32_t)MSL_CAP) {
		for (uint32_t k = 0; k < cnt; ++k) hit(scratch[(size_t)k * stride + p]);
	} else {
		msl_walk(P, p, P.vl_rc2, hit);
	}
}

template <int MSG>
__global__ void __launch_bounds__(MSG) k_msl_fill(ForceParams P, const uint32_t* __restrict__ off, const uint32_t* __restrict__ hsrc,
												  const uint8_t* __restrict__ hdir, uint32_t* __restrict__ out_j,
												  uint8_t* __restrict__ out_il, int ncomp, const uint32_t* __restrict__ scratch,
												  const uint16_t* __restrict__ mcnt, uint32_t stride) {
	__shared__ uint32_t cur[MSL_MAXT * MSG];  // [type][lane]
	__shared__ uint32_t wsum[2];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t n_real = P.n_fixed ? P.n_fixed : P.cnt->n_real;
	const uint32_t p = msl_slot(P, blockIdx.x * MSG + (uint32_t)tid);
	const bool active = p < n_real;
	const int ntypes = ncomp * ncomp;
	const int ci = (active && ncomp > 1) ? P.cid[p] : 0;
	const uint32_t cnt = active ? (uint32_t)mcnt[p] : 0u;
	if (ncomp > 1) {
		for (int t = 0; t < ntypes; ++t) cur[t * MSG + tid] = 0;
		if (active) msl_hits(P, p, cnt, scratch, stride, [&](uint32_t j) { cur[(ci * ncomp + P.cid[j]) * MSG + tid] += 1u; });
	} else {
		uint32_t n = cnt;
		if (active && cnt > (uint32_t)MSL_CAP) {  // (the 16-bit tally saturates: count again)
			n = 0;
			msl_walk(P, p, P.vl_rc2, [&](uint32_t) { ++n; });
		}
		cur[tid] = n;
	}
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
		run += wsum[0] + (MSG > 64 ? wsum[1] : 0u);
		__syncthreads();
	}
	const uint32_t total = run, padded = (total + 63u) & ~63u, o0 = off[blockIdx.x];
	if (active)
		msl_hits(P, p, cnt, scratch, stride, [&](uint32_t j) {
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

// ---- per step: one 64-byte record per owned molecule {x, y, z, q0, q1, q2, q3 (normalised), component id} ---------------------------
// (FullMolecule::setupSoACache normalises q before rotating, FullMolecule.cpp:720 — done here once per molecule and step)
__global__ void __launch_bounds__(256) k_msl_pack(ForceParams P, double* __restrict__ pk, int with_rot, int ncomp) {
	const uint32_t n_real = P.n_fixed ? P.n_fixed : P.cnt->n_real;
	const uint32_t p = blockIdx.x * 256u + threadIdx.x;
	if (p >= n_real) return;
	msl_write_record(pk, p, P.x[p], P.y[p], P.z[p], with_rot ? P.q0[p] : 1., with_rot ? P.q1[p] : 0., with_rot ? P.q2[p] : 0.,
					 with_rot ? P.q3[p] : 0., with_rot != 0, ncomp > 1 ? P.cid[p] : 0);
}

// ---- REUSE: forces from the pair stream, one wave per group ----------------------------------------------------------------------
// LJ_ONLY: no electrostatic sites in the component set — the multipole bodies are compiled out, the kernel needs half the
// registers (four waves per SIMD instead of two: the gathers of a latency-bound pair stream want the occupancy)
// LINEAR (with LJ_ONLY): every site of the set lies on the body z axis (ethane, the 2CLJ family): orientations are carried as the
// rotated z axis (RotAxis, pairphys.hpp) — 24 VGPRs fewer than two rotation matrices (112 instead of 136), a third of the rotation
// arithmetic.
// MSG = 64 (one LJ-only component): 8 KB of LDS per wave instead of 15 — the registers, not the LDS, then set the occupancy.
template <bool WITH_ROT, bool LJ_ONLY, bool LINEAR = false, int MSG = MSG_MAX>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) k_force_ms_list(ForceParams P, const uint32_t* __restrict__ off, const uint32_t* __restrict__ pj,
													  const uint8_t* __restrict__ pil, const double* __restrict__ shift27,
													  const double* __restrict__ pk, const CompTable* __restrict__ ctab, uint32_t ngroups) {
	__shared__ double sr[3][MSG];
	// the multipole body (3 waves / SIMD by its registers) keeps the own quaternions out of the LDS: a body pass gathers them from
	// the records like the partner's (consecutive lanes share the molecule: L1 hits) — 11 instead of 15 KB per wave, 12 waves per CU
	constexpr bool Q_FROM_PK = WITH_ROT && !LJ_ONLY;
	__shared__ double sq[(WITH_ROT && !Q_FROM_PK) ? 4 : 1][(WITH_ROT && !Q_FROM_PK) ? MSG : 1];
	__shared__ uint32_t smol[Q_FROM_PK ? MSG : 1];  // (molecule of every slot)
	__shared__ uint8_t sci[MSG];
	__shared__ double acc[WITH_ROT ? 6 : 3][MSG];
	__shared__ double ssh[27 * 3];
	// queue of the pairs inside the cutoff: < 64 left over + 64 per filtered trip (NC trips per iteration, see below)
	constexpr int NC = LJ_ONLY ? 2 : 1;
	constexpr uint32_t MSQ = 128u * NC;
	__shared__ uint32_t sque_e[MSQ];
	__shared__ uint16_t sque_m[MSQ];
	const int lane = threadIdx.x;
	const uint32_t n_real = P.n_fixed ? P.n_fixed : P.cnt->n_real;
	// workgroups are dealt round-robin to the 8 XCDs (each with its own L2): XCD x takes the x-th CHUNK of consecutive groups, so
	// that neighbouring groups — which gather the same partners — share one L2
#if defined(LS1_BUILD_VARIANT) && defined(LS1_X_NO_XCD_CHUNKS)
	const uint32_t grp = blockIdx.x;  // (A/B: consecutive groups on different XCDs)
#else
	const uint32_t chunk = gridDim.x >> 3, grp = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
#endif
	if (grp >= ngroups) return;
	const uint32_t p0 = grp * MSG;
	// the component table as its OWN restrict-qualified kernel argument: its loads are then provably not clobbered by the force
	// stores and, the component indices being wave-uniform, become scalar loads (through P.ct they were ~60 dependent vector
	// loads per trip — that, not the arithmetic, was the trip time of the first versions)
	typedef const CompTable __attribute__((address_space(4))) ConstCompTable;  // constant address space: invariant for the launch
	ConstCompTable& ct = *(ConstCompTable*)(uintptr_t)ctab;
	const int ncomp = ct.ncomp;
	for (int k = lane; k < MSG; k += 64) {
		const uint32_t p = msl_slot(P, p0 + (uint32_t)k);
		const bool ok = p < n_real;
		const double2* const rec = reinterpret_cast<const double2*>(pk + (size_t)8 * (ok ? p : 0u));
		const double2 d0 = rec[0], d1 = rec[1], d2 = rec[2], d3 = rec[3];
		sr[0][k] = d0.x;
		sr[1][k] = d0.y;
		sr[2][k] = d1.x;
		if constexpr (Q_FROM_PK) {
			smol[k] = ok ? p : 0u;
		} else if constexpr (WITH_ROT) {
			sq[0][k] = d2.x;
			sq[1][k] = d2.y;
			sq[2][k] = d3.x;
			sq[3][k] = d3.y;
		}
		sci[k] = (uint8_t)((ok && ncomp > 1) ? __double2loint(d1.y) : 0);
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
	const uint32_t b0 = off[grp], b1 = off[grp + 1];
	// Two stages per wave.  FILTER: a trip of 64 listed pairs gathers the first 32 bytes of the partners' records (position,
	// component id), applies the periodic shift and the centre-of-mass cutoff, and appends the pairs INSIDE the cutoff to a ring
	// queue in LDS (order kept: component pair, local molecule).  BODIES: whenever the queue holds 64 pairs, or a complete run
	// of one component pair, those lanes gather the full 64-byte records and evaluate the molecule pair.  Every lane of a body
	// pass has work and one component pair — scalar component indices, no divergence — whereas a lane per LISTED pair idles
	// for the pairs of the skin ((r_c / (r_c + skin))^3 = 70-78 % inside) and a trip of 64 that straddles two component pairs
	// costs two passes (five components, 128 molecules: ~120 listed pairs per component pair — 2.9 passes for 1.5 full ones).
	// Filter pipeline, two trips deep: the pair words of trip t + 2 and the partners' positions of trip t + 1 are requested before
	// trip t is looked at; they stay in flight behind the body passes.  (The loaded words are carried RAW into the next trip: any
	// arithmetic on them would make the wave wait for the gather before the bodies instead of behind them.)
	struct Cand {
		uint32_t e, il;
		double2 d0, d1;
	};
	auto load_record = [&](uint32_t b, uint32_t& e, uint32_t& il) {
		e = 0u | (13u << 27);
		il = 0xffu;
		if (b < b1) {
			e = pj[b + lane];
			il = pil[b + lane];
		}
	};
	auto load_cand = [&](uint32_t e, uint32_t il) {
		Cand n;
		const double2* const rec = reinterpret_cast<const double2*>(pk + (size_t)8 * (e & MSL_IDX));
		n.d0 = rec[0];
		n.d1 = rec[1];
		n.e = e;
		n.il = il;
		return n;
	};
	// LJ-only sets (cheap bodies: the gathers are what has to be hidden): an iteration filters TWO trips (two candidates per lane)
	// before it drains the queue — 128 listed pairs leave >= 64 inside the cutoff, so that (nearly) every iteration runs a body
	// pass between the issue of the next iteration's gathers and their use.  The multipole body has no registers to spare for that.
	uint32_t e2[NC], il2[NC];
	Cand nxt[NC];
#pragma unroll
	for (int h = 0; h < NC; ++h) {
		load_record(b0 + 64u * h, e2[h], il2[h]);
		nxt[h] = load_cand(e2[h], il2[h]);
	}
#pragma unroll
	for (int h = 0; h < NC; ++h) load_record(b0 + 64u * NC + 64u * h, e2[h], il2[h]);
	uint32_t qhead = 0, qn = 0;  // wave-uniform: ring of MSQ entries {pair word; local molecule | component pair << 7}
	for (uint32_t b = b0;; b += 64u * NC) {
		const bool more = b < b1;
		if (more) {
			Cand raw[NC];
#pragma unroll
			for (int h = 0; h < NC; ++h) {
				raw[h] = nxt[h];
				nxt[h] = load_cand(e2[h], il2[h]);  // the next iteration's trips (padding records past the end: molecule 0, never queued)
			}
#pragma unroll
			for (int h = 0; h < NC; ++h) load_record(b + 128u * NC + 64u * h, e2[h], il2[h]);  // the trips after those
#pragma unroll
			for (int h = 0; h < NC; ++h) {
				const bool valid = raw[h].il != 0xffu;
				const uint32_t k = valid ? raw[h].il : 0u, sh = raw[h].e >> 27;
				const V3 drm = {sr[0][k] - (raw[h].d0.x + ssh[3 * sh]), sr[1][k] - (raw[h].d0.y + ssh[3 * sh + 1]),
								sr[2][k] - (raw[h].d1.x + ssh[3 * sh + 2])};
				const double dd = dot(drm, drm);
				const bool in = valid && dd < rc2 && dd != 0.;
				const uint32_t key = ncomp > 1 ? (uint32_t)((int)sci[k] * MAXC + __double2loint(raw[h].d1.y)) : 0u;
				const unsigned long long m = __ballot(in);
				const uint32_t at = (qhead + qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) & (MSQ - 1);
				if (in) {
					sque_e[at] = raw[h].e;
					sque_m[at] = (uint16_t)(raw[h].il | (key << 7));
				}
				qn += (uint32_t)__popcll(m);
			}
			msl_wave_lds_order();  // the queue writes above before the reads below
		}
		while (qn >= 64u || (!more && qn > 0u)) {
			const uint32_t qat = (qhead + (uint32_t)lane) & (MSQ - 1);
			const uint32_t ent_e = sque_e[qat], ent_m = sque_m[qat];
			const uint32_t key0 = __builtin_amdgcn_readfirstlane(ent_m >> 7);
			// the leading run of one component pair among the first 64 entries (the stream is sorted: equal keys are contiguous)
			const uint32_t cnt = (uint32_t)__popcll(__ballot((uint32_t)lane < qn && (ent_m >> 7) == key0));
			const bool act = (uint32_t)lane < cnt;
			const uint32_t k = act ? (ent_m & 0x7fu) : 0u, sh = act ? (ent_e >> 27) : 13u;
			const double2* const rec = reinterpret_cast<const double2*>(pk + (size_t)8 * (act ? (ent_e & MSL_IDX) : 0u));
			const double2 d0 = rec[0], d1 = rec[1];
			double2 d2 = make_double2(0., 0.), d3 = d2, qi2 = d2, qi3 = d2;
			if (WITH_ROT) {
				d2 = rec[2];
				d3 = rec[3];
			}
			if constexpr (Q_FROM_PK) {
				const double2* const reci = reinterpret_cast<const double2*>(pk + (size_t)8 * smol[k]);
				qi2 = reci[2];
				qi3 = reci[3];
			}
			const V3 ri = {sr[0][k], sr[1][k], sr[2][k]};
			const V3 rj = {d0.x + ssh[3 * sh], d0.y + ssh[3 * sh + 1], d1.x + ssh[3 * sh + 2]};
			const V3 drm = ri - rj;
			const double dd = dot(drm, drm);
			using RotT = typename std::conditional<LINEAR, RotAxis, Rot>::type;
			a.F = {0., 0., 0.};
			a.M = {0., 0., 0.};
			if (act) {
				RotT Ri, Rj;
				if constexpr (LINEAR) {
					Ri = WITH_ROT ? rot_axis_of(sq[0][k], sq[1][k], sq[2][k], sq[3][k]) : rot_axis_of(1., 0., 0., 0.);
					Rj = WITH_ROT ? rot_axis_of(d2.x, d2.y, d3.x, d3.y) : rot_axis_of(1., 0., 0., 0.);  // (normalised by the writer of the record)
				} else if constexpr (Q_FROM_PK) {
					Ri = rot_of(qi2.x, qi2.y, qi3.x, qi3.y);
					Rj = rot_of(d2.x, d2.y, d3.x, d3.y);
				} else {
					Ri = WITH_ROT ? rot_of(sq[0][k], sq[1][k], sq[2][k], sq[3][k]) : rot_of(1., 0., 0., 0.);
					Rj = WITH_ROT ? rot_of(d2.x, d2.y, d3.x, d3.y) : rot_of(1., 0., 0., 0.);
				}
				// one component pair for the whole pass: scalar indices, parameter tables through scalar loads
				const int ci_u = (int)(key0 / (uint32_t)MAXC), cj_u = (int)(key0 % (uint32_t)MAXC);
				mol_pair<false, ConstCompTable, LJ_ONLY, RotT>(ct, ci_u, ri, Ri, cj_u, rj, Rj, drm, dd < rclj2, 0.5, a);
			}
#if defined(LS1_BUILD_VARIANT) && defined(LS1_X_NO_RUNSUM)
			constexpr bool RUNSUM = false;  // (A/B: one ds_add_f64 per pair, the round-3 form)
#else
			constexpr bool RUNSUM = LJ_ONLY;
#endif
			if constexpr (RUNSUM) {
				// run accumulation (see msl_run_sums): every lane takes part in the row shifts; lanes past the run carry zeros, share
				// the key -1 and never store
				double val[WITH_ROT ? 6 : 3];
				val[0] = act ? a.F.x : 0.;
				val[1] = act ? a.F.y : 0.;
				val[2] = act ? a.F.z : 0.;
				if constexpr (WITH_ROT) {
					val[3] = act ? a.M.x : 0.;
					val[4] = act ? a.M.y : 0.;
					val[5] = act ? a.M.z : 0.;
				}
				const int rkey = act ? (int)k : -1;
				const bool tail = msl_run_sums(rkey, val);
				if (tail && act) {
#pragma unroll
					for (int c = 0; c < (WITH_ROT ? 6 : 3); ++c) unsafeAtomicAdd(&acc[c][k], val[c]);
				}
			} else if (act) {
				unsafeAtomicAdd(&acc[0][k], a.F.x);
				unsafeAtomicAdd(&acc[1][k], a.F.y);
				unsafeAtomicAdd(&acc[2][k], a.F.z);
				if (WITH_ROT) {
					unsafeAtomicAdd(&acc[3][k], a.M.x);
					unsafeAtomicAdd(&acc[4][k], a.M.y);
					unsafeAtomicAdd(&acc[5][k], a.M.z);
				}
			}
			qhead = (qhead + cnt) & (MSQ - 1);
			qn -= cnt;
			msl_wave_lds_order();  // the queue reads above before the next trip's writes
		}
		if (!more) break;
	}
	__syncthreads();
	bool integrated = false;
	if constexpr (WITH_ROT) if (P.fuse == 1) {
		// FUSED: upd_postF of this step and upd_preF of the next for the group's molecules, straight from the accumulators — the
		// forces never go to memory, v and D are read and written once, and the records of the next step are written here (into
		// the OTHER record buffer: this launch still reads the current one).  Same arithmetic as k_kick_then_kick_drift.
		double v2max = 0.;
		for (int k = lane; k < MSG; k += 64) {
			const uint32_t p = msl_slot(P, p0 + (uint32_t)k);
			if (p < n_real) {
				const int c = sci[k];
				LeapState st;
				st.x = sr[0][k]; st.y = sr[1][k]; st.z = sr[2][k];  // (the record's position IS the state's)
				st.vx = P.vx[p]; st.vy = P.vy[p]; st.vz = P.vz[p];
				st.q[0] = P.q0[p]; st.q[1] = P.q1[p]; st.q[2] = P.q2[p]; st.q[3] = P.q3[p];  // (the record's is normalised once more)
				st.D = {P.Dx[p], P.Dy[p], P.Dz[p]};
				const V3 F = {acc[0][k], acc[1][k], acc[2][k]}, M = {acc[3][k], acc[4][k], acc[5][k]};
				const V3 invI = {ct.invI[c][0], ct.invI[c][1], ct.invI[c][2]};
				v2max = fmax(v2max, leap_post_pre<true>(P.dt, ct.mass[c], invI, F, M, st));
				P.ox[p] = st.x; P.oy[p] = st.y; P.oz[p] = st.z;
				P.vx[p] = st.vx; P.vy[p] = st.vy; P.vz[p] = st.vz;
				P.oq0[p] = st.q[0]; P.oq1[p] = st.q[1]; P.oq2[p] = st.q[2]; P.oq3[p] = st.q[3];
				P.Dx[p] = st.D.x; P.Dy[p] = st.D.y; P.Dz[p] = st.D.z;
				msl_write_record(P.msl_pk_out, p, st.x, st.y, st.z, st.q[0], st.q[1], st.q[2], st.q[3], true, c);
			}
		}
		for (int o = 32; o > 0; o >>= 1) v2max = fmax(v2max, __shfl_down(v2max, o));
		if (lane == 0) P.msl_vmax[grp] = v2max;
		integrated = true;
	} else if (P.fuse == 2) {
		// POST-KICK (NVT loops, the driver's armed kick): upd_postF and the kinetic sums of the step for the group's molecules — the
		// separate k_kick pass over v, D, q, F, M disappears.  F and M are stored (the pre-force kick of the next step needs them).
		double mv2s = 0., Iw2s = 0., rdofs = 0.;
		for (int k = lane; k < MSG; k += 64) {
			const uint32_t p = msl_slot(P, p0 + (uint32_t)k);
			if (p < n_real) {
				const int c = sci[k];
				LeapState st;
				st.vx = P.vx[p]; st.vy = P.vy[p]; st.vz = P.vz[p];
				st.q[0] = P.q0[p]; st.q[1] = P.q1[p]; st.q[2] = P.q2[p]; st.q[3] = P.q3[p];
				st.D = {P.Dx[p], P.Dy[p], P.Dz[p]};
				const V3 F = {acc[0][k], acc[1][k], acc[2][k]}, M = {acc[3][k], acc[4][k], acc[5][k]};
				const V3 invI = {ct.invI[c][0], ct.invI[c][1], ct.invI[c][2]}, I = {ct.I[c][0], ct.I[c][1], ct.I[c][2]};
				double mv2, Iw2;
				leap_post<true>(0.5 * P.dt, ct.mass[c], invI, I, F, M, st, mv2, Iw2);
				mv2s += mv2;
				Iw2s += Iw2;
				rdofs += (double)ct.rotdof[c];
				P.vx[p] = st.vx; P.vy[p] = st.vy; P.vz[p] = st.vz;
				P.Dx[p] = st.D.x; P.Dy[p] = st.D.y; P.Dz[p] = st.D.z;
				P.Fx[p] = F.x; P.Fy[p] = F.y; P.Fz[p] = F.z;
				P.Mx[p] = M.x; P.My[p] = M.y; P.Mz[p] = M.z;
			}
		}
		mv2s = msl_wave_sum_d(mv2s);
		Iw2s = msl_wave_sum_d(Iw2s);
		rdofs = msl_wave_sum_d(rdofs);
		if (lane == 0) {
			double* const kp = P.msl_vmax + (size_t)grp * 4;  // (the buffer behind the macroscopic partials: [group][4] in this mode)
			kp[0] = mv2s;
			kp[1] = Iw2s;
			kp[2] = rdofs;
		}
		integrated = true;
	}
	if (!integrated)
	for (int k = lane; k < MSG; k += 64) {
		const uint32_t p = msl_slot(P, p0 + (uint32_t)k);
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
		double* out = P.partials + (size_t)grp * 4;
		out[0] = u6;
		out[1] = uX;
		out[2] = rf;
		out[3] = vir;
	}
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
uint32_t msl_groups(uint32_t n_real, int g) { return (n_real + (uint32_t)g - 1u) / (uint32_t)g; }

int msl_capture_cap() { return MSL_CAP; }

void launch_msl_groups(const ForceParams& p, uint32_t* gm, int ncomp, hipStream_t s) {
	const uint32_t ng = msl_groups(p.n_real_cap, p.msl_g);
	if (ng == 0) return;
	hipLaunchKernelGGL(k_msl_groups, dim3((ng * (uint32_t)p.msl_g + MSW - 1) / MSW), dim3(MSW), 0, s, p, gm, ncomp, ng * (uint32_t)p.msl_g);
}

void launch_msl_count(const ForceParams& p, uint32_t* grp_cnt, uint32_t* off, uint32_t* scratch, uint16_t* mcnt, uint32_t stride,
					  hipStream_t s) {
	const uint32_t ng = msl_groups(p.n_real_cap, p.msl_g);
	if (ng == 0) return;
	if (p.msl_g == 64) hipLaunchKernelGGL(k_msl_count<64>, dim3(ng), dim3(64), 0, s, p, grp_cnt, scratch, mcnt, stride);
	else hipLaunchKernelGGL(k_msl_count<128>, dim3(ng), dim3(128), 0, s, p, grp_cnt, scratch, mcnt, stride);
	if (p.msl_gm && p.msl_g == 64) hipLaunchKernelGGL(k_msl_group_sums<64>, dim3(ng), dim3(64), 0, s, p, grp_cnt, mcnt);
	else if (p.msl_gm) hipLaunchKernelGGL(k_msl_group_sums<128>, dim3(ng), dim3(128), 0, s, p, grp_cnt, mcnt);
	hipLaunchKernelGGL(k_msl_scan, dim3(1), dim3(1024), 0, s, grp_cnt, ng, off, p.cnt);
}

void launch_msl_fill(const ForceParams& p, const uint32_t* off, const uint32_t* hsrc, const uint8_t* hdir, uint32_t* out_j,
					 uint8_t* out_il, int ncomp, const uint32_t* scratch, const uint16_t* mcnt, uint32_t stride, hipStream_t s) {
	const uint32_t ng = msl_groups(p.n_real_cap, p.msl_g);
	if (ng == 0) return;
	if (p.msl_g == 64) hipLaunchKernelGGL(k_msl_fill<64>, dim3(ng), dim3(64), 0, s, p, off, hsrc, hdir, out_j, out_il, ncomp, scratch, mcnt, stride);
	else hipLaunchKernelGGL(k_msl_fill<128>, dim3(ng), dim3(128), 0, s, p, off, hsrc, hdir, out_j, out_il, ncomp, scratch, mcnt, stride);
}

bool launch_force_ms_list(const ForceParams& p, bool has_rot, bool lj_only, bool linear, int ncomp, const uint32_t* off, const uint32_t* pj, const uint8_t* pil,
						  const double* shift27, double* pk, hipStream_t s, uint32_t* nblocks, size_t partials_cap, bool pk_fresh) {
	const uint32_t ng = msl_groups(p.n_real_cap, p.msl_g);
	if ((size_t)ng > partials_cap || p.which != 0) return false;
	*nblocks = ng;
	if (ng == 0) return true;
	const uint32_t grid = ((ng + 7u) >> 3) << 3;  // (8 chunks of consecutive groups, see the kernel)
	// (pk_fresh: the last rigid-body kick + drift pass wrote the records of the current state itself)
	if (!pk_fresh) hipLaunchKernelGGL(k_msl_pack, dim3((p.n_real_cap + 255u) / 256u), dim3(256), 0, s, p, pk, has_rot ? 1 : 0, ncomp);
#define LS1_MSL_LAUNCH(...) hipLaunchKernelGGL((k_force_ms_list<__VA_ARGS__>), dim3(grid), dim3(64), 0, s, p, off, pj, pil, shift27, pk, p.ct, ng)
	if (p.msl_g == 64) {  // (one LJ-only component, msl_group_size)
		if (!lj_only) return false;
		if (has_rot && linear) LS1_MSL_LAUNCH(true, true, true, 64);
		else if (has_rot) LS1_MSL_LAUNCH(true, true, false, 64);
		else LS1_MSL_LAUNCH(false, true, false, 64);
	} else if (has_rot && lj_only && linear) LS1_MSL_LAUNCH(true, true, true);
	else if (has_rot && lj_only) LS1_MSL_LAUNCH(true, true);
	else if (has_rot) LS1_MSL_LAUNCH(true, false);
	else if (lj_only) LS1_MSL_LAUNCH(false, true);
	else LS1_MSL_LAUNCH(false, false);
#undef LS1_MSL_LAUNCH
	return true;
}

}  // namespace ls1
