// kernels_force_verlet.hip — neighbour-list ("Verlet list") variant of the single-centre LJ fast path.
//
// The candidate SEARCH is what the per-step kernels (kernels_force_lj.hip) spend about half of their instructions on,
// and they redo it every step.  Here it is done once per list lifetime:
//   * the cell grid is built with cutoff rc + skin, molecules are binned and the halo copies generated as usual;
//   * BUILD (k_lj_verlet_build): per brick, every owned molecule's neighbours within rc + skin are stored as u16 LDS byte
//     offsets of the brick's staged region (the region-linear staging order is a pure function of the cell table, which
//     stays FROZEN until the next rebuild);
//   * REUSE (k_force_lj_verlet*), every step: the brick region is staged into LDS exactly as at build time (same order,
//     current positions), every lane walks its stored list and evaluates the exact FP64 pair test r^2 < rc^2 and the LJ
//     body.  No search, no re-binning, no halo regeneration (halo positions are refreshed from their source molecules).
// The lists stay complete while no molecule has moved more than skin / 2 since the build; the fused epilogue reports
// max |v| of the step, the reduction accumulates the displacement bound sum(dt * vmax) on the device and publishes
// "rebuild needed" to the host (ls1hip_run polls it) — results never depend on a guessed rebuild interval.
// Precedent in the reference: the Verlet-list containers of AutoPas with a rebuild frequency and skin
// (particleContainer/AutoPasContainer.cpp:281-346); pair arithmetic = VectorizedCellProcessor::_loopBodyLJ
// (particleContainer/adapter/VectorizedCellProcessor.cpp:173-226), masks as in kernels_force_lj.hip.
//
// Mapping (CDNA4): bricks of 4 x 4 x 2 cells (~500 owned molecules at liquid density with rc + skin cells), region
// 6 x 6 x 4 cells (~2260 molecules, 54 KB of FP64 x / y / z in LDS), 512 threads, two workgroups per CU.  ONE lane per
// owned molecule, owned molecules enumerated densely over the brick (no per-cell tile padding); a wave = a TILE of 64
// consecutive owned molecules.  List layout in HBM: [brick][tile][word][lane] u64, a word = 4 list entries: a wave reads
// one word row with ONE coalesced 512-B load per 4 pairs per lane, prefetched four rows ahead.  The lists of the 64
// lanes of a tile have nearly equal length in a liquid (72 +- 3 at skin = 0.12 rc), so the wave-maximum trip count wastes
// < 10 % — against 40 % (quarter lists) + 22 % (cell-padded owned tiles) in the per-step MFMA kernel.  Short lanes are
// padded with an offset that points at a far-away dummy position (masked by the exact cutoff test like any other
// out-of-range entry).
//
// Where the time goes (10^7 molecules, rocprofv3 + timing experiments): the pair loop itself runs at
// ~100 % VALU issue (0.63 ms = 2590 VALU lane-instructions per molecule); the rest of the 1.29 ms is the life of a
// workgroup around it — a chain of dependent memory round trips (cell table -> positions -> velocities / list rows) and
// the dispatch gap between workgroup generations, with both workgroups of a CU in lockstep.  List rows, velocities and
// the position staging are therefore issued as early and as wide as possible (every load of the staging in flight at
// once).  Tried and dropped: persistent workgroups that software-pipeline the bricks (next brick's positions held in
// registers across the pair loop, table / list head one brick further ahead): 128 VGPRs with 87 spilled at two
// workgroups per CU = 2.0 ms; the register budget of a 512-thread workgroup cannot hold 36 prefetch registers next to
// the FP64 pair body.
#include "common.hpp"
#include "brick.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

// Timing variants (phase-decomposition builds whose forces are WRONG BY CONSTRUCTION) do not live in this file: the shipped
// translation unit sees the neutral hooks below; tools/ab_variant.sh compiles with -DLS1_BUILD_VARIANT, which pulls the variant
// bodies from csrc/variants/verlet_timing_hooks.hpp and marks the library (ls1hip_get_option "build_variant", ls1hip_version).
#ifdef LS1_BUILD_VARIANT
#include "variants/verlet_timing_hooks.hpp"
#else
#if defined(LS1_NOLOOP_MOCK) || defined(LS1_NOEPI_MOCK) || defined(LS1_NOSTAGE_MOCK) || defined(LS1_POS_AOS)
#error "variant switches need -DLS1_BUILD_VARIANT (tools/ab_variant.sh): the regular build never carries them"
#endif
#define LS1_HOOK_EPILOGUE(P) true
#define LS1_HOOK_STAGING(P) true
#define LS1_HOOK_LAST_ROW(nw) ((nw) - 1u)
#define LS1_HOOK_ROWS(nw) (nw)
#define LS1_HOOK_FORCE(f, P) (f)
#define LS1_HOOK_EARLY_V true
#define LS1_HOOK_STORE(ptr, val) (*(ptr) = (val))
#define LS1_HOOK_RECORD_OF(did) (did)
#define LS1_HOOK_OWN_FROM_LDS true
#define LS1_HOOK_DMA_STAGING true
#define LS1_HOOK_STAGGER() ((void)0)
#endif

namespace ls1 {

constexpr int VBX = 4, VBY = 4, VBZ = 2;
constexpr int VNT = 512;
constexpr int VNW = VNT / 64;
constexpr int VRX = VBX + 2, VRY = VBY + 2, VRZ = VBZ + 2;
constexpr int VNRC = VRX * VRY * VRZ;  // 144 region cells
constexpr int VNBC = VBX * VBY * VBZ;  // 32 brick cells
#if defined(LS1_POS_AOS)
constexpr int VCAPJ = 2730;            // x y z of a molecule side by side (24 B): a u16 list entry = slot * 24 reaches 2730 slots
#else
constexpr int VCAPJ = 2816;            // staged molecules per brick region (67.8 KB of x, y, z)
#endif
constexpr int VCAPS = VCAPJ + 8;       // +8: the dummy slot and the overrun of unrolled row reads
// layout of the staged positions: element (slot, c) at [VPS * slot + c * VCO]; list entry = byte offset of x = slot * VES (FP64 layout;
// the single-precision pass halves it)
#ifdef LS1_POS_AOS
constexpr int VPS = 3, VCO = 1, VES = 24;
static_assert(VCAPJ * VES < 65536, "list entries are u16 byte offsets");
#else
constexpr int VPS = 1, VCO = VCAPS, VES = 8;
#endif
constexpr int VMAXT = 10;    // tiles per brick with stored lists (640 owned molecules); further tiles: direct evaluation
constexpr int VMAXW = 24;    // words per lane = 96 list entries
// per-brick record written by the build: cstart[VNRC + 1], gbeg[VNRC], owned count, flags (bit 0: every tile listed, staged)
constexpr int VREC_GBEG = VNRC + 1, VREC_NI = 2 * VNRC + 1, VREC_FLAGS = 2 * VNRC + 2, VREC = 2 * VNRC + 8;
constexpr double VFAR = 1.0e30;  // dummy position: r^2 ~ 1e60 fails every cutoff test, all LJ terms underflow to 0
static_assert(VNRC <= VNT * 4, "region too large for the block scan");
static_assert(VCAPS * 8 < 65536, "list entries are u16 byte offsets");

int verlet_region_capacity() { return VCAPJ; }
int verlet_region_cells() { return VNRC; }
void verlet_brick_shape(int shape[3]) {
	shape[0] = VBX;
	shape[1] = VBY;
	shape[2] = VBZ;
}

void verlet_geometry(const Grid& g, long* nbricks, size_t* words_per_brick, size_t* tiles_per_brick) {
	const long nbx = (g.box[0] + VBX - 1) / VBX, nby = (g.box[1] + VBY - 1) / VBY, nbz = (g.box[2] + VBZ - 1) / VBZ;
	*nbricks = nbx * nby * nbz;
	*words_per_brick = (size_t)VMAXT * VMAXW * 64;
	*tiles_per_brick = VMAXT;
}
int verlet_record_words() { return VREC; }

// 1 / d: v_rcp_f64 (measured on gfx950: 4.6e-8 relative, tools/probes/rcp_probe.hip) + ONE Newton step = 2.2e-15 relative;
// the second step (1.1e-16) costs two of the ~27 VALU instructions of a pair and buys nothing at the 1e-10 parity bar
__device__ __forceinline__ double v_rcp(double d) {
	const double x = __builtin_amdgcn_rcp(d);
	const double e = fma(-d, x, 1.0);
	return fma(x, e, x);
}

struct VAcc {
	double fx, fy, fz, slj, vir;
	uint32_t nin;
};

// listed pair: exact strict cutoff test; a pair outside gets r2 := 1e300 and every LJ term underflows to exactly 0.
// COUNT: tally the in-range pairs (they carry the potential shift; also used by the list-free fallbacks).
// SIG1: sigma^2 == 1 (reduced LJ units), sig2 * inv is inv bit for bit — one multiplication less per pair.
template <bool COUNT, bool SIG1 = false>
__device__ __forceinline__ void v_pair(double xi, double yi, double zi, double xj, double yj, double zj, double rc2, double eps24,
									   double sig2, VAcc& a) {
	const double dx = xi - xj, dy = yi - yj, dz = zi - zj;
	const double r2 = fma(dz, dz, fma(dy, dy, dx * dx));
	const bool in = r2 < rc2;
	// out of range: replace only the HIGH dword by that of 1e300 (one v_cndmask instead of two; any low dword will do).  The
	// clamped value also feeds the virial term (fac is exactly 0 there, 0 * 1e300 = 0): r2 is then dead and the select happens
	// in place — no register copy
	const double d = __hiloint2double(in ? __double2hiint(r2) : 0x7E37E43C, __double2loint(r2));
	const double inv = v_rcp(d);
	const double lj2 = SIG1 ? inv : sig2 * inv;
	const double lj6 = lj2 * lj2 * lj2;
	const double lj12m6 = fma(lj6, lj6, -lj6);              // lj12 - lj6
	const double fac = inv * fma(lj6, lj6, lj12m6);          // (lj12 + lj12m6) / r2  // (24 eps applied once per molecule: v_pair_scale)
	a.fx = fma(fac, dx, a.fx);
	a.fy = fma(fac, dy, a.fy);
	a.fz = fma(fac, dz, a.fz);
	a.slj += lj12m6;
	if (COUNT) a.nin += in ? 1u : 0u;
	a.vir = fma(fac, d, a.vir);
}

// Four listed pairs (one list word) with ONE reciprocal: v_rcp_f64 is a quarter-rate instruction (16 cycles against 4 for an FMA),
// so 1 / r2 of four pairs is formed from the reciprocal of their PRODUCT: x = 1 / (a b c d) (v_rcp_f64 + one Newton step), then
// 1 / (a b) = (c d) x, 1 / a = b / (a b), ... — nine multiplications + one reciprocal instead of four reciprocals with a Newton step
// each: 15 issue slots instead of 24 per word (- 9 % of the pair loop's VALU time).  Out-of-range pairs get r2 := 1e75 (only the
// high dword is replaced), so that the product of four stays finite: their terms come out as ~1e-225 instead of exactly 0 —
// absorbed by any non-zero sum.  Relative error of 1 / r2: 2.5e-15 (one reciprocal + Newton: 2.2e-15).
template <bool COUNT, bool SIG1 = false>
__device__ __forceinline__ void v_pair4(double xi, double yi, double zi, const double (&xj)[4], const double (&yj)[4], const double (&zj)[4],
										double rc2, double sig2, VAcc& a) {
	double dx[4], dy[4], dz[4], d[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		dx[k] = xi - xj[k];
		dy[k] = yi - yj[k];
		dz[k] = zi - zj[k];
		const double r2 = fma(dz[k], dz[k], fma(dy[k], dy[k], dx[k] * dx[k]));
		const bool in = r2 < rc2;
		d[k] = __hiloint2double(in ? __double2hiint(r2) : 0x4F810000, __double2loint(r2));  // out of range: ~1e75
		if (COUNT) a.nin += in ? 1u : 0u;
	}
	const double p01 = d[0] * d[1], p23 = d[2] * d[3];
	const double x = v_rcp(p01 * p23);
	const double i01 = p23 * x, i23 = p01 * x;  // 1 / (d0 d1), 1 / (d2 d3)
	const double inv[4] = {d[1] * i01, d[0] * i01, d[3] * i23, d[2] * i23};
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const double lj2 = SIG1 ? inv[k] : sig2 * inv[k];
		const double lj6 = lj2 * lj2 * lj2;
		const double lj12m6 = fma(lj6, lj6, -lj6);
		const double fac = inv[k] * fma(lj6, lj6, lj12m6);
		a.fx = fma(fac, dx[k], a.fx);
		a.fy = fma(fac, dy[k], a.fy);
		a.fz = fma(fac, dz[k], a.fz);
		a.slj += lj12m6;
		a.vir = fma(fac, d[k], a.vir);
	}
}


// the factor 24 eps common to every pair of the molecule
__device__ __forceinline__ void v_pair_scale(VAcc& a, double eps24) {
	a.fx *= eps24;
	a.fy *= eps24;
	a.fz *= eps24;
	a.vir *= eps24;
}

__device__ __forceinline__ double v_wave_sum(double v) {
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
	return v;
}
__device__ __forceinline__ double v_wave_max(double v) {
	for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o));
	return v;
}

// LDS-only barrier: the brick pipeline's barriers order LDS traffic, never global memory, so they must not drain the
// loads in flight for the next brick (a __syncthreads() would wait for every outstanding global access first)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct BrickTab {  // region table of one brick in LDS
	uint32_t* cstart;  // [VNRC + 1] LDS index of the first molecule of every region cell; [VNRC] = total
	uint32_t* gbeg;    // [VNRC]     global index of the first molecule of every region cell
	uint32_t* bstart;  // [VNBC + 1] prefix over the brick's own cells; [VNBC] = owned molecules
};
struct ListHead {  // word count + first two word rows of a wave's first tile, loaded ahead of their use
	uint32_t nw;
	uint64_t w0, w1;
};
struct Totals {
	double u6, vir, kin, vmax2;
	double vmax2b;  // second largest |v_drift|^2 of the lane's molecules (local rebuild criterion)
};

__device__ __forceinline__ uint64_t load_row(const uint64_t* p) { return *p; }

__device__ __forceinline__ ListHead load_list_head(const ForceParams& P, int brick_id, int wv, int lane) {
	const size_t t0 = (size_t)brick_id * VMAXT + (size_t)wv;
	const uint64_t* const wp0 = P.vl_words + t0 * VMAXW * 64 + lane;
	ListHead h;
	h.nw = P.vl_nw[t0];
	h.w0 = load_row(wp0);
	h.w1 = load_row(wp0 + 64);
	return h;
}

// the four pairs of one list word (entries = LDS byte offsets of the partners' x)
template <bool SHIFT, bool SIG1>
__device__ __forceinline__ void pairs_of_word(const char* sxb, double xi, double yi, double zi, uint64_t cur, double rc2, double sig2,
											  VAcc& acc) {
	constexpr int CAPS = VCAPS;
	const uint32_t lo32 = (uint32_t)cur, hi32 = (uint32_t)(cur >> 32);
	const uint32_t o[4] = {lo32 & 0xffffu, lo32 >> 16, hi32 & 0xffffu, hi32 >> 16};
	double xj[4], yj[4], zj[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		xj[k] = *reinterpret_cast<const double*>(sxb + o[k]);
		yj[k] = *reinterpret_cast<const double*>(sxb + o[k] + VCO * 8);
		zj[k] = *reinterpret_cast<const double*>(sxb + o[k] + 2 * VCO * 8);
	}
	v_pair4<SHIFT, SIG1>(xi, yi, zi, xj, yj, zj, rc2, sig2, acc);
}

// Forces of the owned molecules of ONE brick from the stored lists (positions staged in sx / sy / sz, table in T).
template <bool SHIFT, bool SIG1 = false>
__device__ __forceinline__ void brick_forces(const ForceParams& P, const BrickTab& T, const double* sx, const double* sy,
											 const double* sz, int brick_id, bool staged, const ListHead& head, Totals& tot,
											 uint32_t total, uint32_t n_i, const uint16_t* fast_ii = nullptr,
											 const uint32_t* fast_gi = nullptr, uint32_t ii0 = 0, uint32_t gi0 = 0) {
	constexpr int NT = VNT, NW = VNW, RX = VRX, RY = VRY, BX = VBX, BY = VBY, NBC = VNBC, CAPS = VCAPS;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	// total = molecules of the staged region, n_i = owned molecules: from the build's record (scalars) or from the tables in T
	const double rc2 = P.rc2, eps24 = P.eps24, sig2 = P.sig2, shift6 = P.shift6;
	// Split tile (see below): a regular brick has at most VMAXT * 64 = 640 owned molecules, i.e. it is always the SECOND pass.
	// (Requesting its indices and word count ahead of the first pass's pair loop was measured: no gain, three registers.)
	static_assert(VMAXT * 64 <= 2 * VNT, "split tiles: second pass only");
	for (uint32_t base = 0, pass = 0; base < n_i; base += NT, ++pass) {
		const uint32_t it = base + (uint32_t)tid;
		bool active = it < n_i;
		const uint32_t tile = pass * NW + (uint32_t)wv;  // wave-uniform
		const bool listed = staged && tile < (uint32_t)VMAXT;
		const size_t tile_g = (size_t)brick_id * VMAXT + tile;
		uint32_t ii = total, gi = 0;
		int rowbase = 0;
		// LEFTOVER TILES of a regular brick (the molecules beyond the 512th: a brick of the aligned grid owns 512 on average, so
		// every second brick has a handful).  One lane per molecule would run a whole pair loop for them on one wave while the
		// other seven wait: the tile is split instead — 8 / 4 / 2 lanes per molecule, lane (m, s) takes the word rows s, s + S, ...
		// of molecule m's list, the partial sums are combined across the lanes of a molecule, lane (m, 0) runs the epilogue.
		const bool split = pass != 0 && fast_ii != nullptr;
		uint32_t sp_m = 0, sp_s = 0, sp_lg = 0, sp_nw = 4u;  // molecule and share of this lane, log2 of the lanes per molecule
		if (split) {
			const uint32_t wbase = base + (uint32_t)wv * 64u;
			const uint32_t n_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_i);
			if (wbase >= n_u) continue;  // wave-uniform: this wave has no molecule in the tile
			const uint32_t left = min(64u, n_u - wbase);
			sp_lg = left <= 8u ? 3u : left <= 16u ? 2u : left <= 32u ? 1u : 0u;
			sp_m = (uint32_t)lane & ((64u >> sp_lg) - 1u);
			sp_s = (uint32_t)lane >> (6u - sp_lg);
			const bool own = sp_m < left;
			active = own && sp_s == 0u;
			if (own) ii = (uint32_t)fast_ii[wbase + sp_m];
			if (active) gi = fast_gi[wbase + sp_m];
			sp_nw = (uint32_t)P.vl_nw[tile_g];
		} else if (active && fast_ii) {  // regular brick: own LDS slot and global index as recorded by the build (no table search)
			ii = pass == 0 ? ii0 : (uint32_t)fast_ii[it];  // first pass: loaded by the caller ahead of the staging
			gi = pass == 0 ? gi0 : fast_gi[it];
		} else if (active) {
			int lo = 0, hi = NBC;
			while (hi - lo > 1) {
				const int mid = (lo + hi) >> 1;
				if (T.bstart[mid] <= it) lo = mid;
				else hi = mid;
			}
			const int cx = lo % BX, cy = (lo / BX) % BY, cz = lo / (BX * BY);
			const int rcell = ((cz + 1) * RY + (cy + 1)) * RX + (cx + 1);
			const uint32_t k = it - T.bstart[lo];
			ii = T.cstart[rcell] + k;
			gi = T.gbeg[rcell] + k;
			rowbase = (cz * RY + cy) * RX + cx;  // first cell of neighbour row 0 (region coordinates: own cell minus one)
		}
		VAcc acc = {0., 0., 0., 0., 0., 0u};
		// the epilogue's velocity loads are issued before the pair loop (their latency is hidden behind it)
		double vx0 = 0., vy0 = 0., vz0 = 0.;
		const bool epi = LS1_HOOK_EPILOGUE(P);  // (true; timing variants: csrc/variants/verlet_timing_hooks.hpp)
		if (active && P.fuse && epi && LS1_HOOK_EARLY_V) {
			vx0 = P.vx[gi];
			vy0 = P.vy[gi];
			vz0 = P.vz[gi];
		}
		uint32_t nw = 0xffu;
		if (split) nw = (uint32_t)__builtin_amdgcn_readfirstlane((int)sp_nw);  // (a regular brick: the tile is listed)
		else if (listed) nw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pass == 0 ? head.nw : (uint32_t)P.vl_nw[tile_g]));
		if (split) {
			const double xi = sx[VPS * ii], yi = sy[VPS * ii], zi = sz[VPS * ii];  // lanes without a molecule: the dummy slot, dummy words
			const uint64_t* const wpm = P.vl_words + tile_g * VMAXW * 64 + sp_m;
			const char* const sxb = reinterpret_cast<const char*>(sx);
			const uint64_t dummy = (uint64_t)(total * (uint32_t)VES) * 0x0001000100010001ull;
			const uint32_t st = 1u << sp_lg, lastw = nw - 1u, trips = (nw + st - 1u) >> sp_lg;
			uint32_t w0 = sp_s, w1 = sp_s + st;
			// two rows in flight, one loop-carried register each (see below); rows past the end: clamped load, dummy word
			uint64_t c0 = load_row(wpm + (size_t)min(w0, lastw) * 64), c1 = load_row(wpm + (size_t)min(w1, lastw) * 64);
			for (uint32_t k = 0; k < trips; k += 2) {
				pairs_of_word<SHIFT, SIG1>(sxb, xi, yi, zi, w0 < nw ? c0 : dummy, rc2, sig2, acc);
				w0 += 2u * st;
				c0 = load_row(wpm + (size_t)min(w0, lastw) * 64);
				if (k + 1 < trips) pairs_of_word<SHIFT, SIG1>(sxb, xi, yi, zi, w1 < nw ? c1 : dummy, rc2, sig2, acc);
				w1 += 2u * st;
				c1 = load_row(wpm + (size_t)min(w1, lastw) * 64);
			}
			// the shares of a molecule: lanes m, m + 64 / S, ... (fixed order: the same bits on every run)
			if (sp_lg >= 1u) {
				acc.fx += __shfl_xor(acc.fx, 32);
				acc.fy += __shfl_xor(acc.fy, 32);
				acc.fz += __shfl_xor(acc.fz, 32);
				acc.slj += __shfl_xor(acc.slj, 32);
				acc.vir += __shfl_xor(acc.vir, 32);
				acc.nin += (uint32_t)__shfl_xor((int)acc.nin, 32);
			}
			if (sp_lg >= 2u) {
				acc.fx += __shfl_xor(acc.fx, 16);
				acc.fy += __shfl_xor(acc.fy, 16);
				acc.fz += __shfl_xor(acc.fz, 16);
				acc.slj += __shfl_xor(acc.slj, 16);
				acc.vir += __shfl_xor(acc.vir, 16);
				acc.nin += (uint32_t)__shfl_xor((int)acc.nin, 16);
			}
			if (sp_lg >= 3u) {
				acc.fx += __shfl_xor(acc.fx, 8);
				acc.fy += __shfl_xor(acc.fy, 8);
				acc.fz += __shfl_xor(acc.fz, 8);
				acc.slj += __shfl_xor(acc.slj, 8);
				acc.vir += __shfl_xor(acc.vir, 8);
				acc.nin += (uint32_t)__shfl_xor((int)acc.nin, 8);
			}
		} else if (nw != 0xffu) {
			const double xi = sx[VPS * ii], yi = sy[VPS * ii], zi = sz[VPS * ii];  // inactive lanes: the dummy slot (their words are all dummies)
			const uint64_t* const wp = P.vl_words + tile_g * VMAXW * 64 + lane;
			const char* const sxb = reinterpret_cast<const char*>(sx);
			// Four word rows in flight (a row comes from HBM more often than from L2: the lists are read once per step).  Each
			// row register is reloaded right after its content is consumed — ONE loop-carried value per row: with a rotating
			// window (w0 <- w1 <- w2 <- w3) the compiler sank every load to its use and waited vmcnt(0) on it, i.e. a dependent
			// HBM round trip per four pairs; a load inside a branch has the same effect (seen in the ISA).
			const uint32_t last = LS1_HOOK_LAST_ROW(nw);  // nw - 1 (nw >= 4: rows are dummy-padded by the build)
			uint64_t r0 = head.w0, r1 = head.w1;  // first tile: loaded ahead by the caller (rows 0 and 1)
			if (pass != 0) {
				r0 = load_row(wp);
				r1 = load_row(wp + 64);
			}
			uint64_t r2 = load_row(wp + 128), r3 = load_row(wp + 192);
			auto four_pairs = [&](uint64_t cur) {
				const uint32_t lo32 = (uint32_t)cur, hi32 = (uint32_t)(cur >> 32);
				const uint32_t o0 = lo32 & 0xffffu, o1 = lo32 >> 16, o2 = hi32 & 0xffffu, o3 = hi32 >> 16;
				const double x0 = *reinterpret_cast<const double*>(sxb + o0), y0 = *reinterpret_cast<const double*>(sxb + o0 + VCO * 8),
							 z0 = *reinterpret_cast<const double*>(sxb + o0 + 2 * VCO * 8);
				const double x1 = *reinterpret_cast<const double*>(sxb + o1), y1 = *reinterpret_cast<const double*>(sxb + o1 + VCO * 8),
							 z1 = *reinterpret_cast<const double*>(sxb + o1 + 2 * VCO * 8);
				const double x2 = *reinterpret_cast<const double*>(sxb + o2), y2 = *reinterpret_cast<const double*>(sxb + o2 + VCO * 8),
							 z2 = *reinterpret_cast<const double*>(sxb + o2 + 2 * VCO * 8);
				const double x3 = *reinterpret_cast<const double*>(sxb + o3), y3 = *reinterpret_cast<const double*>(sxb + o3 + VCO * 8),
							 z3 = *reinterpret_cast<const double*>(sxb + o3 + 2 * VCO * 8);
				const double xj[4] = {x0, x1, x2, x3}, yj[4] = {y0, y1, y2, y3}, zj[4] = {z0, z1, z2, z3};
				v_pair4<SHIFT, SIG1>(xi, yi, zi, xj, yj, zj, rc2, sig2, acc);
			};
			nw = LS1_HOOK_ROWS(nw);  // (nw)
			for (uint32_t k = 0; k < nw; k += 4) {
				// rows past the end are clamped to the last row and evaluated as what they are after the clamp: skipped
				four_pairs(r0);
				r0 = load_row(wp + (size_t)min(k + 4u, last) * 64);
				if (k + 1 < nw) four_pairs(r1);
				r1 = load_row(wp + (size_t)min(k + 5u, last) * 64);
				if (k + 2 < nw) four_pairs(r2);
				r2 = load_row(wp + (size_t)min(k + 6u, last) * 64);
				if (k + 3 < nw) four_pairs(r3);
				r3 = load_row(wp + (size_t)min(k + 7u, last) * 64);
			}
		} else if (active && staged) {
			// no stored list for this tile (list overflow, or more owned molecules than the list capacity covers)
			const double xi = sx[VPS * ii], yi = sy[VPS * ii], zi = sz[VPS * ii];
			for (int row = 0; row < 9; ++row) {
				const int r0 = rowbase + (row / 3) * (RY * RX) + (row % 3) * RX;
				const uint32_t jb = T.cstart[r0], je = T.cstart[r0 + 3];
				for (uint32_t j = jb; j < je; ++j)
					if (j != ii) v_pair<true>(xi, yi, zi, sx[VPS * j], sy[VPS * j], sz[VPS * j], rc2, eps24, sig2, acc);
			}
		} else if (active) {
			// shell does not fit the staging area (pathological density): same arithmetic straight from global memory
			const double xi = P.x[gi], yi = P.y[gi], zi = P.z[gi];
			for (int row = 0; row < 9; ++row) {
				const int r0 = rowbase + (row / 3) * (RY * RX) + (row % 3) * RX;
				for (int c = r0; c < r0 + 3; ++c) {
					const uint32_t gb = T.gbeg[c], n = T.cstart[c + 1] - T.cstart[c];
					for (uint32_t j = gb; j < gb + n; ++j)
						if (j != gi) v_pair<true>(xi, yi, zi, P.x[j], P.y[j], P.z[j], rc2, eps24, sig2, acc);
				}
			}
		}
		if (active && epi) {
			v_pair_scale(acc, eps24);
			const double fx = LS1_HOOK_FORCE(acc.fx, P), fy = LS1_HOOK_FORCE(acc.fy, P), fz = LS1_HOOK_FORCE(acc.fz, P);  // (the sums themselves)
			if (!P.fuse) {
				P.Fx[gi] = fx;
				P.Fy[gi] = fy;
				P.Fz[gi] = fz;
			} else if (P.fuse == 2) {
				// post-force kick only (upd_postF + sum m v^2, Leapfrog.cpp:66-150): F is kept for the pre-force kick of the
				// next step, which has to wait for the thermostat's scaling factor (NVT) — saves the separate kick pass
				const double k = P.dt_inv2m;
				const double vx = vx0 + k * fx, vy = vy0 + k * fy, vz = vz0 + k * fz;
				const double vv = vx * vx + vy * vy + vz * vz;
				tot.kin += P.mass * vv;
				P.vx[gi] = vx;
				P.vy[gi] = vy;
				P.vz[gi] = vz;
				P.Fx[gi] = fx;
				P.Fy[gi] = fy;
				P.Fz[gi] = fz;
				if (P.vl_top2) {
					// local rebuild criterion of the coming (separate) drift: |beta v + k F| <= max(beta, 1) (|v| + k |F|), and without
					// square roots (two of them cost 0.2 ms per pass at 10^8): (a + b)^2 <= (1 + e) a^2 + (1 + 1 / e) b^2, e = 1 / 50 —
					// k |F| is ~1 % of a fast |v|, the bound is 1 % above the speed
					const double u2 = fma(1.02, vv, (51. * k * k) * (fx * fx + fy * fy + fz * fz));
					tot.vmax2b = fmax(tot.vmax2b, fmin(tot.vmax2, u2));
					tot.vmax2 = fmax(tot.vmax2, u2);
				}
			} else {
				// lj_store of kernels_force_lj.hip (upd_postF, then upd_preF of the next step) + the drift speed of this step
				const double k = P.dt_inv2m;
				if (!LS1_HOOK_EARLY_V) {
					vx0 = P.vx[gi];
					vy0 = P.vy[gi];
					vz0 = P.vz[gi];
				}
				double vx = vx0 + k * fx;
				double vy = vy0 + k * fy;
				double vz = vz0 + k * fz;
				tot.kin += P.mass * (vx * vx + vy * vy + vz * vz);
				vx += k * fx;
				vy += k * fy;
				vz += k * fz;
				LS1_HOOK_STORE(&P.vx[gi], vx);
				LS1_HOOK_STORE(&P.vy[gi], vy);
				LS1_HOOK_STORE(&P.vz[gi], vz);
				{
					const double v2 = vx * vx + vy * vy + vz * vz;
					tot.vmax2b = fmax(tot.vmax2b, fmin(tot.vmax2, v2));
					tot.vmax2 = fmax(tot.vmax2, v2);
				}
				double x0, y0, z0;
				if (staged && LS1_HOOK_OWN_FROM_LDS) {  // (a select between an LDS and a global pointer would become flat loads)
					x0 = sx[VPS * ii];
					y0 = sy[VPS * ii];
					z0 = sz[VPS * ii];
				} else {
					x0 = P.x[gi];
					y0 = P.y[gi];
					z0 = P.z[gi];
				}
				LS1_HOOK_STORE(&P.Fx[gi], x0 + P.dt * vx);
				LS1_HOOK_STORE(&P.Fy[gi], y0 + P.dt * vy);
				LS1_HOOK_STORE(&P.Fz[gi], z0 + P.dt * vz);
			}
			// every in-range pair adds eps24 * (lj12 - lj6) + shift6 to 6 U (nin is tallied where the shift is non-zero and in
			// the fallbacks)
			tot.u6 += fma(eps24, acc.slj, shift6 * (double)acc.nin);
			tot.vir += acc.vir;
		}
	}
}

// workgroup reduction of the totals -> one row of partials {U/2-weighted u6, sum m v^2, max |v_drift|^2, virial}
__device__ __forceinline__ void store_partials(const ForceParams& P, const Totals& tot, double (*red)[4], int brick_id = -1) {
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	// every ordered pair contributes half of the pair's U and virial (see kernels_force.hip); the two largest |v_drift|^2 of the
	// wave (local rebuild criterion) are reduced in the same steps — as an extra loop behind the sums their six dependent
	// cross-lane exchanges sat on the tail of every workgroup (+ 0.2 ms per pass at 10^8)
	double u = 0.5 * tot.u6, v = 0.5 * tot.vir, kn = tot.kin, a = tot.vmax2, b = tot.vmax2b;
	for (int o = 32; o > 0; o >>= 1) {
		const double u2 = __shfl_down(u, o), v2 = __shfl_down(v, o), k2 = __shfl_down(kn, o), a2 = __shfl_down(a, o), b2 = __shfl_down(b, o);
		u += u2;
		v += v2;
		kn += k2;
		b = fmax(fmin(a, a2), fmax(b, b2));
		a = fmax(a, a2);
	}
	const double vm = a;
	const bool top2 = P.vl_top2 != nullptr && brick_id >= 0;
	__shared__ double red2[VNW];
	if (lane == 0) red2[wv] = b;
	if (lane == 0) {
		red[wv][0] = u;
		red[wv][1] = kn;
		red[wv][2] = vm;
		red[wv][3] = v;
	}
	__syncthreads();
	if (tid == 0) {
		double* out = P.partials + (size_t)blockIdx.x * 4;
		double su = 0., sk = 0., sm = 0., sv = 0., sb = 0.;
		for (int i = 0; i < VNW; ++i) {
			su += red[i][0];
			sk += red[i][1];
			if (top2) sb = fmax(fmin(sm, red[i][2]), fmax(sb, red2[i]));
			sm = fmax(sm, red[i][2]);
			sv += red[i][3];
		}
		out[0] = su;
		out[1] = sk;  // fused mode: sum m v^2 (see k_force_lj_brick)
		out[2] = P.fuse == 2 ? 0. : sm;  // fused mode: max |v_drift|^2 of the molecules (combined by max in the reduction); a
		                                 // post-kick pass only feeds vl_top2 (slot 2 is a SUM there: the reaction-field term)
		out[3] = sv;
		if (top2) {
			P.vl_top2[2 * (size_t)brick_id] = sm;
			P.vl_top2[2 * (size_t)brick_id + 1] = sb;
		}
	}
}

// region table + brick prefix of one brick, by the whole workgroup (one-brick-per-workgroup kernels)
__device__ __forceinline__ void block_tables(const ForceParams& P, const BrickSel& bs, const BrickTab& T, uint32_t* wsum) {
	const int tid = threadIdx.x;
	brick_region_table<VNT, 1, VRX, VRY, VRZ>(P, bs, T.cstart, T.gbeg);
	__syncthreads();
	block_scan_lds<VNT>(T.cstart, VNRC, wsum);
	for (int c = tid; c < VNBC; c += VNT) {
		const int cx = c % VBX, cy = (c / VBX) % VBY, cz = c / (VBX * VBY);
		uint32_t n = 0;
		if (cx < bs.ex && cy < bs.ey && cz < bs.ez) {
			const int rcell = ((cz + 1) * VRY + (cy + 1)) * VRX + (cx + 1);
			n = T.cstart[rcell + 1] - T.cstart[rcell];
		}
		T.bstart[c] = n;
	}
	__syncthreads();
	block_scan_lds<VNT>(T.bstart, VNBC, wsum);
}

// Staging: 16 lanes per region cell, 32 cells per round.  ALL global loads of a thread (up to 5 cells x 2 molecules x 3
// coordinates) are issued before the first LDS store waits for one: issued round by round the staging is a chain of ~5
// L2 / HBM latencies per workgroup.
// `after_issue` runs between the issue of the position loads and their LDS stores: loads requested there complete BEHIND the
// positions (vmcnt retires in order), i.e. they do not hold up the staging and still have the stores and the barrier to arrive.
struct NoOp {
	__device__ void operator()() const {}
};
struct TotalOf {  // region total known up front
	uint32_t v;
	__device__ uint32_t operator()() const { return v; }
};
template <class TOT, class F = NoOp>
__device__ __forceinline__ void stage_positions(const ForceParams& P, const BrickTab& T, double* sx, double* sy, double* sz,
												TOT total_of, F after_issue = F()) {
	constexpr int NR = (VNRC + VNT / 16 - 1) / (VNT / 16);  // rounds of cells per 16-lane group
	const int tid = threadIdx.x;
	const uint32_t sub = (uint32_t)tid & 15u;
	double px[NR][2], py[NR][2], pz[NR][2];
	uint32_t sdst[NR][2];
	bool more = false;
	// every cell descriptor of the thread first (with the table in global memory — the build's record — they are one round of
	// loads; fetched round by round, each round's descriptors waited behind the previous round's positions: vmcnt is in order)
	uint32_t ds0[NR], dse[NR], dg0[NR];
#pragma unroll
	for (int j = 0; j < NR; ++j) {
		const int c = min((tid >> 4) + j * (VNT / 16), VNRC - 1);
		ds0[j] = T.cstart[c];
		dse[j] = T.cstart[c + 1];
		dg0[j] = T.gbeg[c];
	}
#pragma unroll
	for (int j = 0; j < NR; ++j) {
		const int c = (tid >> 4) + j * (VNT / 16);
		uint32_t n = 0, s0 = 0, g0 = 0;
		if (c < VNRC) {
			s0 = ds0[j];
			n = dse[j] - s0;
			g0 = dg0[j];
		}
		more |= n > 32u;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const uint32_t k = sub + 16u * h;
			const bool ok = k < n;
			sdst[j][h] = ok ? s0 + k : 0xffffffffu;
			const uint32_t g = ok ? g0 + k : 0u;
			px[j][h] = P.x[g];
			py[j][h] = P.y[g];
			pz[j][h] = P.z[g];
		}
	}
	after_issue();
#pragma unroll
	for (int j = 0; j < NR; ++j)
#pragma unroll
		for (int h = 0; h < 2; ++h)
			if (sdst[j][h] != 0xffffffffu) {
				sx[VPS * sdst[j][h]] = px[j][h];
				sy[VPS * sdst[j][h]] = py[j][h];
				sz[VPS * sdst[j][h]] = pz[j][h];
			}
	if (more) {  // cells with more than 32 molecules (dense clusters): the rest in a plain loop
		for (int c = tid >> 4; c < VNRC; c += VNT / 16) {
			const uint32_t n = T.cstart[c + 1] - T.cstart[c], s0 = T.cstart[c], g0 = T.gbeg[c];
			for (uint32_t k = 32u + sub; k < n; k += 16u) {
				sx[VPS * (s0 + k)] = P.x[g0 + k];
				sy[VPS * (s0 + k)] = P.y[g0 + k];
				sz[VPS * (s0 + k)] = P.z[g0 + k];
			}
		}
	}
	const uint32_t total = total_of();  // (evaluated here: behind the position loads and their stores)
	if (tid < 8) {
		sx[VPS * (total + tid)] = VFAR;
		sy[VPS * (total + tid)] = VFAR;
		sz[VPS * (total + tid)] = VFAR;
	}
}

// LDS-DMA form of the staging for REGULAR bricks (gfx950 global_load_lds_dword: per-lane global source, wave-uniform LDS base +
// lane * 4, no VGPR destination, no ds_write): the region image in LDS is, row by row, exactly the global layout — a region row
// of six x-neighbour cells is one contiguous run of molecules in the cell-sorted arrays (several runs where halo cells, which live
// in their own segment, interrupt it) — so every run is copied dword-wise by one wave: three rows per wave, descriptors through
// scalar loads of the build's record.  8-byte alignment of both sides is enough for the dword form (the 16-byte form would need
// runs padded to even molecule counts on both sides, i.e. another list-entry encoding).  tools/probes/glds_probe.hip checks the
// semantics the copy relies on (EXEC-masked lanes write nothing, unaligned-to-16 runs land bit-exactly).
typedef __attribute__((address_space(3))) void* ls1_lds_ptr;
typedef const __attribute__((address_space(1))) void* ls1_glb_ptr;
__device__ __forceinline__ void stage_positions_dma(const ForceParams& P, const uint32_t* __restrict__ rec, double* sx, double* sy, double* sz,
													uint32_t total) {
	const int tid = threadIdx.x, lane = tid & 63;
	const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	auto copy_run = [&](uint32_t s0, uint32_t g0, uint32_t n) {  // n molecules from global index g0 to LDS slot s0 (all wave-uniform)
		const uint32_t nd = 2u * n;
		const uint32_t* const gx = reinterpret_cast<const uint32_t*>(P.x + g0);
		const uint32_t* const gy = reinterpret_cast<const uint32_t*>(P.y + g0);
		const uint32_t* const gz = reinterpret_cast<const uint32_t*>(P.z + g0);
		uint32_t* const lx = reinterpret_cast<uint32_t*>(sx + VPS * s0);
		uint32_t* const ly = reinterpret_cast<uint32_t*>(sy + VPS * s0);
		uint32_t* const lz = reinterpret_cast<uint32_t*>(sz + VPS * s0);
		for (uint32_t k0 = 0; k0 < nd; k0 += 64u) {
			const uint32_t d = k0 + (uint32_t)lane;
			if (d < nd) {
				__builtin_amdgcn_global_load_lds((ls1_glb_ptr)(gx + d), (ls1_lds_ptr)(lx + k0), 4, 0, 0);
				__builtin_amdgcn_global_load_lds((ls1_glb_ptr)(gy + d), (ls1_lds_ptr)(ly + k0), 4, 0, 0);
				__builtin_amdgcn_global_load_lds((ls1_glb_ptr)(gz + d), (ls1_lds_ptr)(lz + k0), 4, 0, 0);
			}
		}
	};
	static_assert(VPS == 1, "the DMA staging copies the separate x / y / z arrays");
	for (int r = wv; r < VRY * VRZ; r += VNW) {
		const uint32_t* const cs = rec + r * VRX;
		const uint32_t* const gb = rec + VREC_GBEG + r * VRX;
		uint32_t run_s = cs[0], run_g = gb[0], run_n = 0;
#pragma unroll
		for (int c = 0; c < VRX; ++c) {
			const uint32_t n = cs[c + 1] - cs[c], g = gb[c];
			if (g != run_g + run_n) {  // the cell does not follow its predecessor in memory (halo segment / domain face)
				if (run_n) copy_run(run_s, run_g, run_n);
				run_s = cs[c];
				run_g = g;
				run_n = 0;
			}
			run_n += n;
		}
		if (run_n) copy_run(run_s, run_g, run_n);
	}
	if (tid < 8) {
		sx[VPS * (total + tid)] = VFAR;
		sy[VPS * (total + tid)] = VFAR;
		sz[VPS * (total + tid)] = VFAR;
	}
}

// ---- BUILD ------------------------------------------------------------------------------------------------------------
// The list only has to be a SUPERSET of the pairs within rc + skin (the force pass re-tests every entry exactly in FP64),
// so the search runs in FP32 on brick-relative coordinates with a conservative threshold: packed FP32 arithmetic handles
// two candidates per instruction (v_pk_add / v_pk_mul / v_pk_fma_f32: 3 VALU per candidate instead of 6 FP64), the staged
// region shrinks to 12 B per molecule (34 KB: the search kernel is not LDS-capacity bound) and its LDS reads halve.
// Per candidate the bookkeeping is a compare and an add-with-carry that shifts the outcome into a per-lane bit mask; list
// entries are formed for the hits only (see the scan below).
// FP32 slack: coordinates are relative to the region's low corner (< 6 cell lengths), |r^2_fp32 - r^2| <= 2 (|dx| + |dy| +
// |dz|) ulp(extent) + 3 ulp(r^2) ~ 4e-5 at liquid-argon scale; the threshold adds 2e-6 * extent * (rc + skin) + 1e-5 (rc +
// skin)^2, several times that bound.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef f32x2 f32x2_a4 __attribute__((aligned(4)));  // candidate pairs start at any molecule index: ds_read2_b32, not b64

__global__ void __launch_bounds__(VNT, 4) k_lj_verlet_build(ForceParams P, int nbx, int nby, int nbz) {
	constexpr int NT = VNT, RX = VRX, RY = VRY, BX = VBX, BY = VBY, NBC = VNBC, CAPS = VCAPS;
	__shared__ float fpos[3 * CAPS];  // x, y, z of a molecule side by side: a group of four candidates is 12 consecutive dwords,
	                                  // read as six ds_read2_b32 (x0 x1 | y0 y1 | ...) off ONE address register
	__shared__ uint32_t cstart[VNRC + 1];
	__shared__ uint32_t gbeg[VNRC];
	__shared__ uint32_t bstart[NBC + 1];
	__shared__ uint32_t wsum[VNW];
	__shared__ uint32_t ovf;          // some tile of the brick has no stored list
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const BrickSel bs = brick_select<1, VBX, VBY, VBZ>(P, nbx, nby, nbz);
	if (!bs.live) return;  // uniform per workgroup
	const BrickTab T = {cstart, gbeg, bstart};
	block_tables(P, bs, T, wsum);
	const uint32_t total = cstart[VNRC], n_i = bstart[NBC];
	uint32_t* const rec = P.vl_rec + (size_t)bs.did * VREC;
	if (total > (uint32_t)VCAPJ) {  // unstaged brick: evaluated directly every step, no lists
		// an empty record: the force pass stages from the record BEFORE it looks at the flags (nothing to stage here)
		for (int c = tid; c < VREC; c += NT) rec[c] = 0;
		if (tid == 0) atomicAdd(&P.cnt->vl_irregular, 1u);
		return;
	}
	// the brick's record: what the force pass would otherwise recompute every step (cell loads, two scans, six barriers)
	for (int c = tid; c <= VNRC; c += NT) rec[c] = cstart[c];
	for (int c = tid; c < VNRC; c += NT) rec[VREC_GBEG + c] = gbeg[c];
	if (tid == 0) {
		rec[VREC_NI] = n_i;
		ovf = (n_i > (uint32_t)(VMAXT * 64)) ? 1u : 0u;
	}
	// brick-relative FP32 positions (origin = low corner of the region's first cell)
	const double ox = P.g.bmin[0] + (double)(bs.x0 - 2) * P.g.clen[0];
	const double oy = P.g.bmin[1] + (double)(bs.y0 - 2) * P.g.clen[1];
	const double oz = P.g.bmin[2] + (double)(bs.z0 - 2) * P.g.clen[2];
	{
		// 16 lanes per region cell, 32 cells per round; the global loads of a thread are issued in two batches (three + two
		// rounds) before their conversions wait for them — round by round the staging was a chain of five HBM latencies per
		// workgroup; all five rounds at once cost 30 more VGPRs and the third workgroup per CU
		constexpr int NR = (VNRC + NT / 16 - 1) / (NT / 16);
		constexpr int NB = 3;
		const uint32_t sub = (uint32_t)tid & 15u;
		bool more = false;
#pragma unroll
		for (int j0 = 0; j0 < NR; j0 += NB) {
			double px[NB][2], py[NB][2], pz[NB][2];
			uint32_t sdst[NB][2];
#pragma unroll
			for (int jj = 0; jj < NB; ++jj) {
				const int c = (tid >> 4) + (j0 + jj) * (NT / 16);
				uint32_t n = 0, s0 = 0, g0 = 0;
				if (j0 + jj < NR && c < VNRC) {
					s0 = cstart[c];
					n = cstart[c + 1] - s0;
					g0 = gbeg[c];
				}
				more |= n > 32u;
#pragma unroll
				for (int h = 0; h < 2; ++h) {
					const uint32_t k = sub + 16u * h;
					const bool ok = k < n;
					sdst[jj][h] = ok ? s0 + k : 0xffffffffu;
					const uint32_t g = ok ? g0 + k : 0u;
					px[jj][h] = P.x[g];
					py[jj][h] = P.y[g];
					pz[jj][h] = P.z[g];
				}
			}
#pragma unroll
			for (int jj = 0; jj < NB; ++jj)
#pragma unroll
				for (int h = 0; h < 2; ++h)
					if (sdst[jj][h] != 0xffffffffu) {
						fpos[3 * sdst[jj][h]] = (float)(px[jj][h] - ox);
						fpos[3 * sdst[jj][h] + 1] = (float)(py[jj][h] - oy);
						fpos[3 * sdst[jj][h] + 2] = (float)(pz[jj][h] - oz);
					}
		}
		if (more) {  // cells with more than 32 molecules: the rest in a plain loop
			for (int c = tid >> 4; c < VNRC; c += NT / 16) {
				const uint32_t n = cstart[c + 1] - cstart[c], s0 = cstart[c], g0 = gbeg[c];
				for (uint32_t k = 32u + sub; k < n; k += 16u) {
					fpos[3 * (s0 + k)] = (float)(P.x[g0 + k] - ox);
					fpos[3 * (s0 + k) + 1] = (float)(P.y[g0 + k] - oy);
					fpos[3 * (s0 + k) + 2] = (float)(P.z[g0 + k] - oz);
				}
			}
		}
	}
	if (tid < 8) {
		fpos[3 * (total + tid)] = 1.0e18f;  // far-away padding behind the last molecule (r^2 ~ 1e36, finite)
		fpos[3 * (total + tid) + 1] = 1.0e18f;
		fpos[3 * (total + tid) + 2] = 1.0e18f;
	}
	__syncthreads();
	const uint64_t dummy = (uint64_t)(total * (uint32_t)VES) * 0x0001000100010001ull;  // four entries pointing at the far-away slot
	const float ext = (float)fmax(fmax(RX * P.g.clen[0], RY * P.g.clen[1]), VRZ * P.g.clen[2]);
	const float rcs = (float)sqrt(P.vl_rc2);
	const float thr = (float)P.vl_rc2 * (1.0f + 1e-5f) + 2e-6f * ext * rcs;
	for (uint32_t base = 0, pass = 0; base < n_i; base += NT, ++pass) {
		const uint32_t it = base + (uint32_t)tid;
		const bool active = it < n_i;
		const uint32_t tile = pass * VNW + (uint32_t)wv;  // wave-uniform
		if (tile >= (uint32_t)VMAXT) continue;          // tiles beyond the list capacity are evaluated directly
		const size_t tile_g = (size_t)bs.did * VMAXT + tile;
		uint64_t* const wp = P.vl_words + tile_g * VMAXW * 64 + lane;
		uint32_t cnt = 0;
		if (active) {
			int lo = 0, hi = NBC;
			while (hi - lo > 1) {
				const int mid = (lo + hi) >> 1;
				if (bstart[mid] <= it) lo = mid;
				else hi = mid;
			}
			const int cx = lo % BX, cy = (lo / BX) % BY, cz = lo / (BX * BY);
			const int rcell = ((cz + 1) * RY + (cy + 1)) * RX + (cx + 1);
			const uint32_t ii = cstart[rcell] + (it - bstart[lo]);
			const int rowbase = (cz * RY + cy) * RX + cx;
			P.vl_ii[(size_t)bs.did * (VMAXT * 64) + it] = (uint16_t)ii;
			P.vl_gi[(size_t)bs.did * (VMAXT * 64) + it] = gbeg[rcell] + (it - bstart[lo]);
			const f32x2 xi = {fpos[3 * ii], fpos[3 * ii]}, yi = {fpos[3 * ii + 1], fpos[3 * ii + 1]}, zi = {fpos[3 * ii + 2], fpos[3 * ii + 2]};
			// Search in two phases per block of 32 candidates.  SCAN: distance test of every candidate, its outcome shifted into a
			// per-lane bit mask by the compare's carry (v_cmp + v_addc: two instructions of bookkeeping per candidate; the first
			// version stored every candidate into an LDS window and let misses be overwritten: five + one ds_write).  EXTRACT: only
			// the hits (14 % of the candidates) are turned into list entries, highest bit first = ascending candidate order, and
			// collected in a register pair that is stored as a word of four entries when full.
			uint32_t wlo = 0, whi = 0;
			const f32x2 nthr = {-thr, -thr};
			auto put_hit = [&](uint32_t j8) {  // j8 = LDS byte offset of the neighbour's x (< 65536)
				wlo = __builtin_amdgcn_alignbit(whi, wlo, 16);
				whi = __builtin_amdgcn_alignbit(j8, whi, 16);
				++cnt;
				// word cnt / 4 - 1 of the lane: cnt is a multiple of 4 here, so its address is one shift-add (capacity: see below)
				if ((cnt & 3u) == 0u) (wp - 64)[(size_t)cnt * 16] = ((uint64_t)whi << 32) | (uint64_t)wlo;
			};
			// SELF: the molecule itself lives in the middle row only — the other eight rows skip the index test
			auto scan_row = [&](int row, auto self_tag) {
				constexpr bool SELF = decltype(self_tag)::value;
				const int r0 = rowbase + (row / 3) * (RY * RX) + (row % 3) * RX;
				const uint32_t jb = cstart[r0], je = cstart[r0 + 3];
				for (uint32_t jblk = jb; jblk < je; jblk += 32u) {
					const uint32_t nblk = min(32u, je - jblk), ngrp = (nblk + 3u) >> 2;
					uint32_t m = 0;
					// m = 2 m + (r^2 < thr): the threshold is folded into the distance (r^2 - thr, first FMA), its sign bit enters the
					// mask with one v_alignbit
					auto bit = [&m](float t) { m = __builtin_amdgcn_alignbit(m, __float_as_uint(t), 31); };
					// groups of four candidates, two per packed instruction.  The last group of a row may run past the row end: those
					// entries are the next cells of the region in linear order (or the far-away padding behind the last molecule);
					// their bits are dropped below
					uint32_t ca = (uint32_t)(uintptr_t)fpos + 12u * jblk;  // LDS byte address of the group's first candidate
					for (uint32_t g = 0; g < ngrp; ++g, ca += 48u) {
						// (x0 x1) (y0 y1) (z0 z1) (x2 x3) ... straight into register pairs: ds_read2_b32 takes two independent dword
						// offsets.  Written as asm: from C++ the compiler pairs ADJACENT dwords and re-sorts with eight v_mov
						f32x2 xa, ya, za, xb, yb, zb;
						asm volatile(
							"ds_read2_b32 %0, %6 offset1:3\n\t"
							"ds_read2_b32 %1, %6 offset0:1 offset1:4\n\t"
							"ds_read2_b32 %2, %6 offset0:2 offset1:5\n\t"
							"ds_read2_b32 %3, %6 offset0:6 offset1:9\n\t"
							"ds_read2_b32 %4, %6 offset0:7 offset1:10\n\t"
							"ds_read2_b32 %5, %6 offset0:8 offset1:11\n\t"
							"s_waitcnt lgkmcnt(0)"
							: "=&v"(xa), "=&v"(ya), "=&v"(za), "=&v"(xb), "=&v"(yb), "=&v"(zb)
							: "v"(ca)
							: "memory");
						const f32x2 dxa = xi - xa, dya = yi - ya, dza = zi - za, dxb = xi - xb, dyb = yi - yb, dzb = zi - zb;
						const f32x2 ra = __builtin_elementwise_fma(
							dza, dza, __builtin_elementwise_fma(dya, dya, __builtin_elementwise_fma(dxa, dxa, nthr)));
						const f32x2 rb = __builtin_elementwise_fma(
							dzb, dzb, __builtin_elementwise_fma(dyb, dyb, __builtin_elementwise_fma(dxb, dxb, nthr)));
						bit(ra.x);
						bit(ra.y);
						bit(rb.x);
						bit(rb.y);
					}
					// candidate k of the block -> bit 31 - k; bits of candidates past the row end cleared
					m = (m << (32u - 4u * ngrp)) & (0xffffffffu << (32u - nblk));
					if (SELF) {
						const uint32_t d = ii - jblk;
						if (d < nblk) m &= ~(0x80000000u >> d);
					}
					// list capacity, tested once per block instead of once per hit: a lane that would exceed it only counts (its tile
					// is then marked as overflowed below and evaluated directly by the force pass)
					const uint32_t hits = (uint32_t)__builtin_popcount(m);
					if (cnt + hits > (uint32_t)(VMAXW * 4)) {
						cnt += hits;
						m = 0;
					}
					while (m) {
						const uint32_t b = (uint32_t)__builtin_clz(m);
						m &= ~(0x80000000u >> b);
						put_hit((jblk + b) * (uint32_t)VES);
					}
				}
			};
			for (int row = 0; row < 4; ++row) scan_row(row, std::false_type());
			scan_row(4, std::true_type());
			for (int row = 5; row < 9; ++row) scan_row(row, std::false_type());
			// the incomplete last word: its entries sit in the top of the register pair; padded with the dummy entry
			const uint32_t w = cnt >> 2, rem = cnt & 3u;
			if (rem && w < (uint32_t)VMAXW) {
				const uint64_t part = (((uint64_t)whi << 32) | (uint64_t)wlo) >> (16u * (4u - rem));
				wp[(size_t)w * 64] = part | (dummy << (16u * rem));
			}
		}
		// words in use by this tile = wave maximum (at least the four rows the force pass loads unconditionally); shorter
		// lanes are padded with dummy words up to it
		uint32_t mine = (cnt + 3u) >> 2, nw = max(mine, 4u);
		for (int o = 32; o > 0; o >>= 1) nw = max(nw, (uint32_t)__shfl_xor((int)nw, o));
		if (nw > (uint32_t)VMAXW) {
			if (lane == 0) {
				P.vl_nw[tile_g] = 0xff;  // overflow (very dense neighbourhood): direct evaluation every step
				ovf = 1u;
			}
		} else {
			for (uint32_t w = mine; w < nw; ++w) wp[(size_t)w * 64] = dummy;
			if (lane == 0) P.vl_nw[tile_g] = (uint8_t)nw;
		}
	}
	__syncthreads();
	if (tid == 0) {
		rec[VREC_FLAGS] = ovf ? 0u : 1u;
		if (ovf) atomicAdd(&P.cnt->vl_irregular, 1u);
	}
}

// ---- REUSE, one brick per workgroup (reference implementation of the pipeline below; LS1_VL_ONE_BRICK_PER_WG) ---------
template <bool SHIFT, bool SIG1, bool FAST>
__global__ void __launch_bounds__(VNT, 4) k_force_lj_verlet(ForceParams P, int nbx, int nby, int nbz, int vgrid) {
	constexpr int CAPS = VCAPS;
	__shared__ double spos[3 * CAPS];
	double* const sx = spos;
	double* const sy = spos + VCO;
	double* const sz = spos + 2 * VCO;
	__shared__ uint32_t cstart[VNRC + 1];
	__shared__ uint32_t gbeg[VNRC];
	__shared__ uint32_t bstart[VNBC + 1];
	__shared__ uint32_t wsum[VNW];
	__shared__ double red[VNW][4];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	// FAST HEAD (the production launch: every brick, blocked order, per-brick data stored in launch order — did_mode 1): the
	// workgroup's data index is pure arithmetic on blockIdx.x; the brick's identity (one dependent scalar load) and its grid
	// coordinates (three integer divisions) are not needed before the end of a regular brick, so nothing at the head waits for
	// them.  Every other launch (inner / boundary passes, plain order) resolves the brick first, as before.
	constexpr bool fast_head = FAST;  // (chosen by the host from did_mode: one head per instantiation, no register cost)
	BrickSel bs;
	bool live;
	int did, slot = 0;
	if constexpr (fast_head) {
		const int vb = (int)blockIdx.x, chunk = vgrid / 8;
		slot = (vb % 8) * chunk + vb / 8;
		live = slot < (int)P.n_list;
		did = slot;
	} else {
		bs = brick_select_v<1, VBX, VBY, VBZ>(P, nbx, nby, nbz, (int)blockIdx.x, vgrid);
		live = bs.live;
		did = bs.did;
	}
	if (!live) {  // uniform per workgroup
		if (tid < 4) P.partials[(size_t)blockIdx.x * 4 + tid] = 0.;
		return;
	}
	LS1_HOOK_STAGGER();  // (nothing; timing variants)
	ListHead head;
	Totals tot = {0., 0., 0., 0., 0.};
	uint32_t* const rec = P.vl_rec + (size_t)did * VREC;
	// the record's three scalars (region total, owned count, flags; uniform addresses = scalar loads).  Forcing them out at the
	// head (inline asm) and requesting all kernel arguments of the later phases in one batch was measured: no gain (± 0.3 %) —
	// the other workgroup of the CU hides scalar round trips of this size
	const uint32_t rec_total = rec[VNRC], rec_ni = rec[VREC_NI], rec_flags = rec[VREC_FLAGS];
	// own LDS slot / global index of the first pass, issued together with the list head and the flags (reads of valid memory
	// whatever the flags say): nothing the pair loop needs is requested after the staging barrier
	const uint16_t* const f_ii = P.vl_ii + (size_t)did * (VMAXT * 64);
	const uint32_t* const f_gi = P.vl_gi + (size_t)did * (VMAXT * 64);
	uint32_t ii0 = 0, gi0 = 0;
	// Staging straight from the build's record, BEFORE the flags are looked at: the descriptor loads leave first, ahead of the
	// list head and the own indices, instead of behind a dependent flag load (an unstaged brick has an empty record).
	// (Also requesting list rows 2 and 3 up here was measured: slower — vmcnt completes in order, so the staging then waits for
	// two more HBM round trips.)
	{
		uint32_t* const srec = P.vl_rec + (size_t)LS1_HOOK_RECORD_OF(did) * VREC;  // (= rec; the staging descriptors)
		const BrickTab R = {srec, srec + VREC_GBEG, nullptr};
		head = load_list_head(P, did, wv, lane);  // list head and own indices: independent of everything staged below
		ii0 = f_ii[tid];
		gi0 = f_gi[tid];
		if (LS1_HOOK_STAGING(P)) {  // (true)
			if (LS1_HOOK_DMA_STAGING) stage_positions_dma(P, srec, sx, sy, sz, rec_total);
			else stage_positions(P, R, sx, sy, sz, TotalOf{rec_total});
		}
	}
	__syncthreads();
	if (rec_flags & 1u) {  // uniform per workgroup
		// regular brick (staged, every tile has its list): cell table and own indices come from the build's record — no cell
		// loads, no scans, no table search; the only barrier of the workgroup is the one behind the staging
		const BrickTab T = {cstart, gbeg, bstart};  // (only the two totals are read on this path)
		brick_forces<SHIFT, SIG1>(P, T, sx, sy, sz, did, true, head, tot, rec_total, rec_ni, f_ii, f_gi, ii0, gi0);
		store_partials(P, tot, red, fast_head ? (int)P.brick_list[slot] : bs.id);
	} else {
		if constexpr (fast_head) bs = brick_select_v<1, VBX, VBY, VBZ>(P, nbx, nby, nbz, (int)blockIdx.x, vgrid);
		const BrickTab T = {cstart, gbeg, bstart};
		block_tables(P, bs, T, wsum);
		const bool staged = cstart[VNRC] <= (uint32_t)VCAPJ;
		if (staged) stage_positions(P, T, sx, sy, sz, TotalOf{cstart[VNRC]});
		__syncthreads();
		brick_forces<SHIFT, SIG1>(P, T, sx, sy, sz, did, staged, head, tot, cstart[VNRC], bstart[VNBC]);
		store_partials(P, tot, red, bs.id);
	}
}


// ---- REUSE in single precision (the reference's MARDYN_SPSP / MARDYN_SPDP build modes, vectorization/RealVec.h,
// RealAccumVecSPDP.h; its reduced-memory mode runs in them) -----------------------------------------------------------------
// Option "precision" = 1 (SPDP: pair arithmetic in FP32, sums in FP64) | 2 (SPSP: FP32 sums).  The molecule state stays FP64
// in memory and the fused integration epilogue is the FP64 one: only the pair loop changes.  Positions are staged as FP32
// relative to the region's corner (12 B per molecule: 34 KB, four workgroups per CU), two pairs per packed instruction
// (v_pk_add / v_pk_mul / v_pk_fma_f32), v_rcp_f32 (1 ulp) without a Newton step.  The list entries (LDS byte offsets of the FP64
// layout) are halved.  Regular bricks only: the host launches this kernel when the build reported no irregular brick.
struct SAcc2 {
	f32x2 fx, fy, fz, slj, vir;
};
template <bool SHIFT>
__device__ __forceinline__ void sp_pair2(f32x2 xi, f32x2 yi, f32x2 zi, f32x2 xj, f32x2 yj, f32x2 zj, float rc2, float sig2, SAcc2& a,
										 uint32_t& nin) {
	const f32x2 dx = xi - xj, dy = yi - yj, dz = zi - zj;
	const f32x2 r2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
	const bool in0 = r2.x < rc2, in1 = r2.y < rc2;
	// out of range: r2 := 1e30, 1 / r2 = 1e-30 and every LJ term underflows to exactly 0
	const f32x2 inv = {__builtin_amdgcn_rcpf(in0 ? r2.x : 1.0e30f), __builtin_amdgcn_rcpf(in1 ? r2.y : 1.0e30f)};
	const f32x2 lj2 = sig2 * inv;
	const f32x2 lj6 = lj2 * lj2 * lj2;
	const f32x2 lj12m6 = __builtin_elementwise_fma(lj6, lj6, -lj6);
	const f32x2 fac = inv * __builtin_elementwise_fma(lj6, lj6, lj12m6);
	a.fx = __builtin_elementwise_fma(fac, dx, a.fx);
	a.fy = __builtin_elementwise_fma(fac, dy, a.fy);
	a.fz = __builtin_elementwise_fma(fac, dz, a.fz);
	a.slj += lj12m6;
	a.vir = __builtin_elementwise_fma(fac, r2, a.vir);
	if (SHIFT) nin += (in0 ? 1u : 0u) + (in1 ? 1u : 0u);
}

template <bool SHIFT, bool DPACC>
__global__ void __launch_bounds__(VNT, 4) k_force_lj_verlet_sp(ForceParams P, int nbx, int nby, int nbz) {
	constexpr int CAPS = VCAPS, NT = VNT, NW = VNW;
	__shared__ float fpos[3 * CAPS];
	float* const fx = fpos;
	float* const fy = fpos + VCO;
	float* const fz = fpos + 2 * VCO;
	__shared__ double red[VNW][4];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const BrickSel bs = brick_select<1, VBX, VBY, VBZ>(P, nbx, nby, nbz);
	if (!bs.live) {  // uniform per workgroup
		if (tid < 4) P.partials[(size_t)blockIdx.x * 4 + tid] = 0.;
		return;
	}
	const ListHead head = load_list_head(P, bs.did, wv, lane);
	const uint32_t* const rec = P.vl_rec + (size_t)bs.did * VREC;
	const uint16_t* const f_ii = P.vl_ii + (size_t)bs.did * (VMAXT * 64);
	const uint32_t* const f_gi = P.vl_gi + (size_t)bs.did * (VMAXT * 64);
	const uint32_t ii0 = f_ii[tid], gi0 = f_gi[tid];
	const uint32_t total = rec[VNRC], n_i = rec[VREC_NI];
	// ---- staging: FP32 relative to the low corner of the region's first cell (as the build does) ------------------------------
	const double ox = P.g.bmin[0] + (double)(bs.x0 - 2) * P.g.clen[0];
	const double oy = P.g.bmin[1] + (double)(bs.y0 - 2) * P.g.clen[1];
	const double oz = P.g.bmin[2] + (double)(bs.z0 - 2) * P.g.clen[2];
	{
		constexpr int NR = (VNRC + NT / 16 - 1) / (NT / 16);
		const uint32_t sub = (uint32_t)tid & 15u;
		double px[NR][2], py[NR][2], pz[NR][2];
		uint32_t sdst[NR][2];
		bool more = false;
		uint32_t ds0[NR], dse[NR], dg0[NR];  // every descriptor first: one round of loads (see stage_positions)
#pragma unroll
		for (int j = 0; j < NR; ++j) {
			const int c = min((tid >> 4) + j * (NT / 16), VNRC - 1);
			ds0[j] = rec[c];
			dse[j] = rec[c + 1];
			dg0[j] = rec[VREC_GBEG + c];
		}
#pragma unroll
		for (int j = 0; j < NR; ++j) {
			const int c = (tid >> 4) + j * (NT / 16);
			uint32_t n = 0, s0 = 0, g0 = 0;
			if (c < VNRC) {
				s0 = ds0[j];
				n = dse[j] - s0;
				g0 = dg0[j];
			}
			more |= n > 32u;
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const uint32_t k = sub + 16u * h;
				const bool ok = k < n;
				sdst[j][h] = ok ? s0 + k : 0xffffffffu;
				const uint32_t g = ok ? g0 + k : 0u;
				px[j][h] = P.x[g];
				py[j][h] = P.y[g];
				pz[j][h] = P.z[g];
			}
		}
#pragma unroll
		for (int j = 0; j < NR; ++j)
#pragma unroll
			for (int h = 0; h < 2; ++h)
				if (sdst[j][h] != 0xffffffffu) {
					fx[VPS * sdst[j][h]] = (float)(px[j][h] - ox);
					fy[VPS * sdst[j][h]] = (float)(py[j][h] - oy);
					fz[VPS * sdst[j][h]] = (float)(pz[j][h] - oz);
				}
		if (more) {
			for (int c = tid >> 4; c < VNRC; c += NT / 16) {
				const uint32_t s0 = rec[c], n = rec[c + 1] - s0, g0 = rec[VREC_GBEG + c];
				for (uint32_t k = 32u + sub; k < n; k += 16u) {
					fx[VPS * (s0 + k)] = (float)(P.x[g0 + k] - ox);
					fy[VPS * (s0 + k)] = (float)(P.y[g0 + k] - oy);
					fz[VPS * (s0 + k)] = (float)(P.z[g0 + k] - oz);
				}
			}
		}
		if (tid < 8) {
			fx[VPS * (total + tid)] = 1.0e15f;  // far-away dummy slot (r^2 ~ 1e30: fails every cutoff test)
			fy[VPS * (total + tid)] = 1.0e15f;
			fz[VPS * (total + tid)] = 1.0e15f;
		}
	}
	__syncthreads();
	Totals tot = {0., 0., 0., 0., 0.};
	const float rc2 = (float)P.rc2, sig2 = (float)P.sig2;
	const double eps24 = P.eps24, shift6 = P.shift6;
	const char* const fxb = reinterpret_cast<const char*>(fx);
	for (uint32_t base = 0, pass = 0; base < n_i; base += NT, ++pass) {
		const uint32_t it = base + (uint32_t)tid;
		const bool active = it < n_i;
		const uint32_t tile = pass * NW + (uint32_t)wv;  // wave-uniform; < VMAXT (regular brick)
		const size_t tile_g = (size_t)bs.did * VMAXT + tile;
		uint32_t ii = total, gi = 0;
		if (active) {
			ii = pass == 0 ? ii0 : (uint32_t)f_ii[it];
			gi = pass == 0 ? gi0 : f_gi[it];
		}
		// own FP64 state for the epilogue: requested before the pair loop
		double x0 = 0., y0 = 0., z0 = 0., vx0 = 0., vy0 = 0., vz0 = 0.;
		if (active && P.fuse) {
			x0 = P.x[gi]; y0 = P.y[gi]; z0 = P.z[gi];
			vx0 = P.vx[gi]; vy0 = P.vy[gi]; vz0 = P.vz[gi];
		}
		const uint32_t nw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pass == 0 ? head.nw : (uint32_t)P.vl_nw[tile_g]));
		const f32x2 xi = {fx[VPS * ii], fx[VPS * ii]}, yi = {fy[VPS * ii], fy[VPS * ii]}, zi = {fz[VPS * ii], fz[VPS * ii]};
		const uint64_t* const wp = P.vl_words + tile_g * VMAXW * 64 + lane;
		const uint32_t last = nw - 1u;
		uint64_t r0 = head.w0, r1 = head.w1;
		if (pass != 0) {
			r0 = load_row(wp);
			r1 = load_row(wp + 64);
		}
		uint64_t r2 = load_row(wp + 128), r3 = load_row(wp + 192);
		double dfx = 0., dfy = 0., dfz = 0., dslj = 0., dvir = 0.;  // DPACC: FP64 sums
		SAcc2 acc = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
		uint32_t nin = 0;
		auto four_pairs = [&](uint64_t cur) {
			const uint32_t lo32 = (uint32_t)cur, hi32 = (uint32_t)(cur >> 32);
			// entries are byte offsets of the FP64 layout (8 j): halved for the FP32 arrays
			const uint32_t o0 = (lo32 & 0xffffu) >> 1, o1 = lo32 >> 17, o2 = (hi32 & 0xffffu) >> 1, o3 = hi32 >> 17;
			const f32x2 xa = {*reinterpret_cast<const float*>(fxb + o0), *reinterpret_cast<const float*>(fxb + o1)};
			const f32x2 ya = {*reinterpret_cast<const float*>(fxb + o0 + VCO * 4), *reinterpret_cast<const float*>(fxb + o1 + VCO * 4)};
			const f32x2 za = {*reinterpret_cast<const float*>(fxb + o0 + 2 * VCO * 4), *reinterpret_cast<const float*>(fxb + o1 + 2 * VCO * 4)};
			const f32x2 xb = {*reinterpret_cast<const float*>(fxb + o2), *reinterpret_cast<const float*>(fxb + o3)};
			const f32x2 yb = {*reinterpret_cast<const float*>(fxb + o2 + VCO * 4), *reinterpret_cast<const float*>(fxb + o3 + VCO * 4)};
			const f32x2 zb = {*reinterpret_cast<const float*>(fxb + o2 + 2 * VCO * 4), *reinterpret_cast<const float*>(fxb + o3 + 2 * VCO * 4)};
			if (DPACC) {
				SAcc2 w = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
				sp_pair2<SHIFT>(xi, yi, zi, xa, ya, za, rc2, sig2, w, nin);
				sp_pair2<SHIFT>(xi, yi, zi, xb, yb, zb, rc2, sig2, w, nin);
				dfx += (double)(w.fx.x + w.fx.y);
				dfy += (double)(w.fy.x + w.fy.y);
				dfz += (double)(w.fz.x + w.fz.y);
				dslj += (double)(w.slj.x + w.slj.y);
				dvir += (double)(w.vir.x + w.vir.y);
			} else {
				sp_pair2<SHIFT>(xi, yi, zi, xa, ya, za, rc2, sig2, acc, nin);
				sp_pair2<SHIFT>(xi, yi, zi, xb, yb, zb, rc2, sig2, acc, nin);
			}
		};
		for (uint32_t k = 0; k < nw; k += 4) {
			four_pairs(r0);
			r0 = load_row(wp + (size_t)min(k + 4u, last) * 64);
			if (k + 1 < nw) four_pairs(r1);
			r1 = load_row(wp + (size_t)min(k + 5u, last) * 64);
			if (k + 2 < nw) four_pairs(r2);
			r2 = load_row(wp + (size_t)min(k + 6u, last) * 64);
			if (k + 3 < nw) four_pairs(r3);
			r3 = load_row(wp + (size_t)min(k + 7u, last) * 64);
		}
		if (active) {
			if (!DPACC) {
				dfx = (double)(acc.fx.x + acc.fx.y);
				dfy = (double)(acc.fy.x + acc.fy.y);
				dfz = (double)(acc.fz.x + acc.fz.y);
				dslj = (double)(acc.slj.x + acc.slj.y);
				dvir = (double)(acc.vir.x + acc.vir.y);
			}
			const double fxd = eps24 * dfx, fyd = eps24 * dfy, fzd = eps24 * dfz;
			if (!P.fuse) {
				P.Fx[gi] = fxd;
				P.Fy[gi] = fyd;
				P.Fz[gi] = fzd;
			} else if (P.fuse == 2) {  // post-force kick only, F kept (see brick_forces)
				const double k2 = P.dt_inv2m;
				const double vx = vx0 + k2 * fxd, vy = vy0 + k2 * fyd, vz = vz0 + k2 * fzd;
				const double vv = vx * vx + vy * vy + vz * vz;
				tot.kin += P.mass * vv;
				P.vx[gi] = vx;
				P.vy[gi] = vy;
				P.vz[gi] = vz;
				P.Fx[gi] = fxd;
				P.Fy[gi] = fyd;
				P.Fz[gi] = fzd;
				if (P.vl_top2) {  // bounds for the local rebuild criterion of the coming drift (see brick_forces)
					const double u2 = fma(1.02, vv, (51. * k2 * k2) * (fxd * fxd + fyd * fyd + fzd * fzd));
					tot.vmax2b = fmax(tot.vmax2b, fmin(tot.vmax2, u2));
					tot.vmax2 = fmax(tot.vmax2, u2);
				}
			} else {  // the FP64 epilogue of brick_forces
				const double k2 = P.dt_inv2m;
				double vx = vx0 + k2 * fxd;
				double vy = vy0 + k2 * fyd;
				double vz = vz0 + k2 * fzd;
				tot.kin += P.mass * (vx * vx + vy * vy + vz * vz);
				vx += k2 * fxd;
				vy += k2 * fyd;
				vz += k2 * fzd;
				P.vx[gi] = vx;
				P.vy[gi] = vy;
				P.vz[gi] = vz;
				{
					const double v2 = vx * vx + vy * vy + vz * vz;
					tot.vmax2b = fmax(tot.vmax2b, fmin(tot.vmax2, v2));
					tot.vmax2 = fmax(tot.vmax2, v2);
				}
				P.Fx[gi] = x0 + P.dt * vx;
				P.Fy[gi] = y0 + P.dt * vy;
				P.Fz[gi] = z0 + P.dt * vz;
			}
			tot.u6 += fma(eps24, dslj, shift6 * (double)nin);
			tot.vir += eps24 * dvir;
		}
	}
	store_partials(P, tot, red, bs.id);
}

// ---- LOCAL REBUILD CRITERION ------------------------------------------------------------------------------------------------
// A listed pair (i, j) stays valid while d_i + d_j < skin (d = displacement since the build).  Both partners of a pair
// evaluated by brick b are owned by bricks of b's 27-brick neighbourhood (the region is the brick plus one cell, a brick is at
// least one cell wide; across a periodic face the partner is the image of a molecule of the wrapped neighbour brick), so
//     d_i + d_j <= sum over the steps of dt * (s1 + s2),   s1 >= s2 the two largest speeds in that neighbourhood in the step.
// acc[b] accumulates dt * (s1 + s2) / 2 and is compared with skin / 2.  The global criterion (sum of dt * v_max, the fastest of
// 10^8 molecules every step) is the special case s2 = s1 = global maximum: the local one is never earlier, and ~13 % later at
// T* = 0.95.  Single periodic domain only (a remote rank's halo molecules do not report their speeds here).
__global__ void __launch_bounds__(256) k_bound_local(int nbx, int nby, int nbz, const double* __restrict__ top2, double* __restrict__ acc,
												  DevCounters* cnt, double dt, double limit, double speed_factor) {
	const int b = blockIdx.x * 256 + threadIdx.x;
	if (b >= nbx * nby * nbz) return;
	const int bx = b % nbx, by = (b / nbx) % nby, bz = b / (nbx * nby);
	double m1 = 0., m2 = 0.;
	// (with fewer than three bricks along a dimension a neighbour is visited twice: its maximum must not fill both places)
	const int x0 = nbx >= 3 ? bx - 1 : 0, y0 = nby >= 3 ? by - 1 : 0, z0 = nbz >= 3 ? bz - 1 : 0;
	const int cx = nbx >= 3 ? 3 : nbx, cy = nby >= 3 ? 3 : nby, cz = nbz >= 3 ? 3 : nbz;
	for (int k = 0; k < cz; ++k)
		for (int j = 0; j < cy; ++j)
			for (int i = 0; i < cx; ++i) {
				const int qx = (x0 + i + nbx) % nbx, qy = (y0 + j + nby) % nby, qz = (z0 + k + nbz) % nbz;
				const size_t q = ((size_t)qz * nby + qy) * nbx + qx;
				const double a = top2[2 * q], c = top2[2 * q + 1];
				m2 = fmax(fmin(m1, a), fmax(m2, c));
				m1 = fmax(m1, a);
			}
	// acc and vl_base are zeroed by the list build; unfused drifts in between (the first step of a run, NVT steps) add their
	// dt * v_max to vl_base for every brick
	// (unfused drifts: top2 holds bounds of |v| + |dt/2m F| from the pass that did the post-force kick; the drift then moved
	// the molecules with beta v + dt/2m F, |.| <= max(beta, 1) times that)
	const double fac = speed_factor < 0. ? fmax(cnt->beta[0], 1.) : speed_factor;
	const double v = acc[b] + dt * 0.5 * fac * (sqrt(m1) + sqrt(m2));
	acc[b] = v;
	if (v + cnt->vl_base > limit) atomicOr(&cnt->vl_local_excess, 1u);
}
long verlet_brick_count(const Grid& g) {
	return (long)((g.box[0] + VBX - 1) / VBX) * ((g.box[1] + VBY - 1) / VBY) * ((g.box[2] + VBZ - 1) / VBZ);
}
void launch_bound_local(const Grid& g, const double* top2, double* acc, DevCounters* cnt, double dt, double limit, hipStream_t s,
						double speed_factor) {
	const int nbx = (g.box[0] + VBX - 1) / VBX, nby = (g.box[1] + VBY - 1) / VBY, nbz = (g.box[2] + VBZ - 1) / VBZ;
	const long nb = (long)nbx * nby * nbz;
	if (nb <= 0) return;
	hipLaunchKernelGGL(k_bound_local, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, s, nbx, nby, nbz, top2, acc, cnt, dt, limit, speed_factor);
}

bool launch_force_verlet(const ForceParams& p_in, hipStream_t s, uint32_t* nblocks, size_t partials_cap, BrickLists* bl) {
	ForceParams p = p_in;
	const Grid& g = p.g;
	if (g.hw != 1 || !p.vl_words || !p.vl_nw) return false;
	const int nbx = (g.box[0] + VBX - 1) / VBX, nby = (g.box[1] + VBY - 1) / VBY, nbz = (g.box[2] + VBZ - 1) / VBZ;
	if ((long)nbx * nby * nbz <= 0 || (long)nbx * nby * nbz > 0x7ffffff0L) return false;
	if (p.vl_mode == 1) p.which = 0;  // the lists are always built for all bricks
	// blocked launch order of the bricks (kernels_force_lj.hip, brick_lists_for): + 1 % on the 10^8 box; LS1_BRICK_BLOCKED=0
	// restores the plain x-y-z order
	static const bool blocked = [] {
		const char* e = getenv("LS1_BRICK_BLOCKED");
		return e ? atoi(e) != 0 : true;
	}();
	const long nb = plan_bricks(p, bl, VBX, VBY, VBZ, nbx, nby, nbz, blocked);  // multiple of 8
	// the per-brick data is stored in the blocked order: a launch that cannot have it (no brick lists: allocation failure) must
	// not run against data another launch stored that way
	if (blocked && p.did_mode == 0) return false;
	if (nb == 0) {
		*nblocks = 0;
		return true;
	}
	const bool shift = p.shift6 != 0.;
	if (p.vl_mode == 1) {
		*nblocks = 0;
		hipLaunchKernelGGL(k_lj_verlet_build, dim3((uint32_t)nb), dim3(VNT), 0, s, p, nbx, nby, nbz);
		return true;
	}
	if ((size_t)nb > partials_cap) return false;
	*nblocks = (uint32_t)nb;
	if (p.precision == 1) {
		if (shift) hipLaunchKernelGGL((k_force_lj_verlet_sp<true, true>), dim3((uint32_t)nb), dim3(VNT), 0, s, p, nbx, nby, nbz);
		else hipLaunchKernelGGL((k_force_lj_verlet_sp<false, true>), dim3((uint32_t)nb), dim3(VNT), 0, s, p, nbx, nby, nbz);
	} else if (p.precision == 2) {
		if (shift) hipLaunchKernelGGL((k_force_lj_verlet_sp<true, false>), dim3((uint32_t)nb), dim3(VNT), 0, s, p, nbx, nby, nbz);
		else hipLaunchKernelGGL((k_force_lj_verlet_sp<false, false>), dim3((uint32_t)nb), dim3(VNT), 0, s, p, nbx, nby, nbz);
	} else {
		const bool sig1 = p.sig2 == 1.0;  // reduced units: one multiplication less per pair, same bits
		auto go = [&](auto sh, auto s1) {
			const bool fast = p.did_mode == 1;
			if (fast)
				hipLaunchKernelGGL((k_force_lj_verlet<decltype(sh)::value, decltype(s1)::value, true>), dim3((uint32_t)nb), dim3(VNT), 0, s, p, nbx, nby, nbz, (int)nb);
			else
				hipLaunchKernelGGL((k_force_lj_verlet<decltype(sh)::value, decltype(s1)::value, false>), dim3((uint32_t)nb), dim3(VNT), 0, s, p, nbx, nby, nbz, (int)nb);
		};
		if (shift && sig1) go(std::true_type{}, std::true_type{});
		else if (shift) go(std::true_type{}, std::false_type{});
		else if (sig1) go(std::false_type{}, std::true_type{});
		else go(std::false_type{}, std::false_type{});
	}
	return true;
}

}  // namespace ls1
