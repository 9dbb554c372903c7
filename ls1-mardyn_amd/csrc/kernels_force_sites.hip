// kernels_force_sites.hip — multi-site force kernel, second generation ("site kernel"): default for component sets with
// more than one interaction site (multi-centre LJ, charges, dipoles, quadrupoles, any mix of components).
//
// What it computes is what VectorizedCellProcessor::_calculatePairs computes for one molecule and all its neighbours
// (/root/reference/src/particleContainer/adapter/VectorizedCellProcessor.cpp:1192-2720, site-pair physics
// molecules/potforce.h:18-263), in the one-sided full-shell form of this library: every lane owns the accumulators of its
// molecule, no atomics, deterministic.  How it is organised differs from the first multi-site kernel (kernels_force_ms.hip,
// kept as LS1HIP_FK_MS_BRICK: bitwise equal to the generic kernel) in the places its counters showed the time going
// (profiles/r2_ms_*: 8.6 k VALU lane-slots per ethane molecule at 53 % lane use, 219 VGPRs, table reads from global memory):
//
//   * component / site tables live in LDS (2.9 KB + the LJ pair table), not behind 190 global loads in the pair body;
//   * the lane's OWN sites are rotated once per molecule into an LDS cache (offset from the centre, orientation axis),
//     the neighbour's sites once per molecule pair (j-site outer loop, i-site inner loop) instead of once per site pair;
//   * site distances are formed from the centre distance (dr = (r_i - r_j) + d_i - d_j): no absolute site positions;
//   * FMA contraction and Newton-refined v_rcp_f64 / v_rsq_f64 instead of IEEE division and square root (the results
//     stay within 1e-12 of the generic kernel, tests/test_gpu_parity.py);
//   * LPM lanes share one molecule (they split its candidates, then add their partial sums with two DPP shuffles), so the
//     brick can be small enough for the LDS budget at 4-5 molecules per cell and still fill 256 lanes;
//   * the brick shape is a launch parameter (not a template parameter): the host picks the shape whose owned molecules
//     fill the lanes of one pass and whose shell fits the staging area.
//
// Bricks whose shell does not fit the staging area (dense clusters) evaluate straight from global memory — same arithmetic.
#pragma clang fp contract(fast)  // this translation unit only (the Makefile default is off): FMA contraction in the pair bodies

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "brick.hpp"
#include "common.hpp"

namespace ls1 {

constexpr int STPB = 256;     // threads per workgroup
// MPL (template parameter) = owned molecules per lane slot; 2: an expensive and a cheap one (cost-sorted), so lane sums level out
constexpr int SORT_CNT = 64;  // neighbour-count classes of the cost sort
constexpr int SORT_BINS = MAXC * SORT_CNT;

struct SitesGeom {
	int bx, by, bz;   // brick shape in cells
	int capm;         // staging capacity (molecules of brick + shell)
	int capl;         // list rows per lane
	int maxs, maxe;   // most sites / most oriented sites (dipoles + quadrupoles) of one component: i-cache strides
	int nc2;          // LJ pair table entries (ncenters^2)
	uint8_t crank[MAXC];  // component -> cost rank (0 = most sites)
	int dbg;              // diagnostics (LS1HIP_SITES_DBG): 1 = skip the pair bodies (times staging + search + sort)
};

// LDS copy of the per-site part of CompTable
struct SiteTab {
	double ljpos[MAXS][3];
	double chpos[MAXS][3], chq[MAXS];
	double dppos[MAXS][3], dpe[MAXS][3], dpmy[MAXS];
	double qppos[MAXS][3], qpe[MAXS][3], qpQ[MAXS];
	int nlj[MAXC], nc[MAXC], nd[MAXC], nq[MAXC];
	int olj[MAXC], oc[MAXC], od[MAXC], oq[MAXC];
};

__device__ __forceinline__ double rcp_nr(double d) {
	double x = __builtin_amdgcn_rcp(d);
	double e = fma(-d, x, 1.0);
	x = fma(x, e, x);
	e = fma(-d, x, 1.0);
	return fma(x, e, x);
}
// 1 / sqrt(d): v_rsq_f64 + two Newton steps
__device__ __forceinline__ double rsq_nr(double d) {
	double y = __builtin_amdgcn_rsq(d);
	double e = fma(-d * y, y, 1.0);
	y = fma(0.5 * y, e, y);
	e = fma(-d * y, y, 1.0);
	return fma(0.5 * y, e, y);
}

__device__ __forceinline__ V3 fma3(double s, V3 a, V3 b) { return {fma(s, a.x, b.x), fma(s, a.y, b.y), fma(s, a.z, b.z)}; }

// ---- site-pair bodies.  r = (site of i) - (site of j); every body returns what molecule i receives: force f on its site,
// torque t about its site's axis (orientation-dependent part; the lever-arm part d_i x f is added by the caller), energy u.
// The formulas are the multipole expansions of potforce.h (charge / point dipole / linear point quadrupole), written in
// terms of the direction cosines a = e_i.r / r, b = e_j.r / r, g = e_i.e_j.

// LJ 12-6 (potforce.h:18-30); u6 = 6 U without the shift
__device__ __forceinline__ void sp_lj(V3 r, double eps24, double sig2, V3& f, double& u6) {
	const double r2 = dot(r, r);
	const double inv = rcp_nr(r2);
	const double s2 = sig2 * inv;
	const double s6 = s2 * s2 * s2;
	const double s12m6 = fma(s6, s6, -s6);
	u6 = eps24 * s12m6;
	f = (eps24 * inv * fma(s6, s6, s12m6)) * r;
}
// charge - charge (potforce.h:190-199)
__device__ __forceinline__ void sp_cc(V3 r, double qq, V3& f, double& u) {
	const double inv = rsq_nr(dot(r, r));
	u = qq * inv;
	f = (u * inv * inv) * r;
}
// monopole at the origin end of r, axial multipole (axis e) at the other end.  sgn = +1: i carries the charge (r = r_i - r_j,
// e = e_j) -> returns the force on the charge; sgn = -1: i carries the multipole (r = r_j - r_i, e = e_i) -> returns the
// force on and the torque about the multipole.  (potforce.h:205-263)
template <bool I_IS_CHARGE>
__device__ __forceinline__ void sp_charge_dipole(V3 r, V3 e, double mqmy, V3& f, V3& t, double& u) {
	const double inv = rsq_nr(dot(r, r)), inv2 = inv * inv;
	const double c = dot(e, r) * inv;
	const double k = mqmy * inv2;
	u = k * c;
	const double te = k * inv;
	const V3 fa = fma3(-te, e, (3.0 * u * inv2) * r);  // force on the charge
	if (I_IS_CHARGE) {
		f = fa;
		t = {0., 0., 0.};
	} else {
		f = {-fa.x, -fa.y, -fa.z};
		t = te * cross(r, e);
	}
}
template <bool I_IS_CHARGE>
__device__ __forceinline__ void sp_charge_quadrupole(V3 r, V3 e, double qQ05, V3& f, V3& t, double& u) {
	const double inv = rsq_nr(dot(r, r)), inv2 = inv * inv;
	const double c = dot(e, r) * inv;
	const double w = qQ05 * inv * inv2;
	u = w * fma(3.0 * c, c, -1.0);
	const double te = 6.0 * c * w * inv;
	const V3 fa = fma3(-te, e, fma(c * te, inv, 3.0 * u * inv2) * r);
	if (I_IS_CHARGE) {
		f = fa;
		t = {0., 0., 0.};
	} else {
		f = {-fa.x, -fa.y, -fa.z};
		t = te * cross(r, e);
	}
}
// dipole (i) - dipole (j) with reaction field (potforce.h:36-80)
__device__ __forceinline__ void sp_dd(V3 r, V3 ei, V3 ej, double my2, double rffac, V3& f, V3& t, double& u, double& rf) {
	const double inv = rsq_nr(dot(r, r)), inv2 = inv * inv;
	const double w = my2 * inv2 * inv;
	const double a = dot(ei, r) * inv, b = dot(ej, r) * inv, g = dot(ei, ej);
	u = w * fma(-3.0 * a, b, g);
	rf = -rffac * g;
	const double ta = -3.0 * w * b * inv, tb = -3.0 * w * a * inv;  // d u / d a, d u / d b (over r)
	const double fr = fma(fma(a, ta, b * tb), inv, 3.0 * u * inv2);
	f = fma3(-tb, ej, fma3(-ta, ei, fr * r));
	t = fma3(rffac - w, cross(ei, ej), (-ta) * cross(ei, r));
}
// quadrupole (i) - quadrupole (j) (potforce.h:86-133)
__device__ __forceinline__ void sp_qq(V3 r, V3 ei, V3 ej, double q2075, V3& f, V3& t, double& u) {
	const double inv = rsq_nr(dot(r, r)), inv2 = inv * inv;
	const double w = q2075 * inv2 * inv2 * inv;
	const double a = dot(ei, r) * inv, b = dot(ej, r) * inv, g = dot(ei, ej);
	const double a2 = a * a, b2 = b * b;
	const double h = fma(-5.0 * a, b, g);
	u = w * (1.0 - 5.0 * (a2 + b2) - 15.0 * a2 * b2 + 2.0 * h * h);
	const double ta = -10.0 * w * (a + 3.0 * a * b2 + 2.0 * b * h) * inv;
	const double tb = -10.0 * w * (b + 3.0 * a2 * b + 2.0 * a * h) * inv;
	const double fr = fma(fma(a, ta, b * tb), inv, 5.0 * u * inv2);
	f = fma3(-tb, ej, fma3(-ta, ei, fr * r));
	t = fma3(-4.0 * w * h, cross(ei, ej), (-ta) * cross(ei, r));
}
// dipole (axis ea, at the head of r) - quadrupole (axis eb, at the tail of r): r = r_dipole - r_quadrupole (potforce.h:139-184).
// I_IS_DIPOLE: returns force / torque of the dipole; otherwise those of the quadrupole (the caller passes r = r_j - r_i).
template <bool I_IS_DIPOLE>
__device__ __forceinline__ void sp_dq(V3 r, V3 ea, V3 eb, double myq15, V3& f, V3& t, double& u) {
	const double inv = rsq_nr(dot(r, r)), inv2 = inv * inv;
	const double w = myq15 * inv2 * inv2;
	const double a = dot(ea, r) * inv, b = dot(eb, r) * inv, g = dot(ea, eb);
	const double b2 = b * b;
	u = w * fma(-a, fma(5.0, b2, -1.0), 2.0 * g * b);
	const double ta = w * fma(-5.0, b2, 1.0) * inv;
	const double tb = 2.0 * w * fma(-5.0 * a, b, g) * inv;
	const double pg = 2.0 * w * b;
	const double fr = fma(fma(a, ta, b * tb), inv, 4.0 * u * inv2);
	const V3 fd = fma3(-tb, eb, fma3(-ta, ea, fr * r));  // force on the dipole
	const V3 x = cross(ea, eb);
	if (I_IS_DIPOLE) {
		f = fd;
		t = fma3(-pg, x, (-ta) * cross(ea, r));
	} else {
		f = {-fd.x, -fd.y, -fd.z};
		t = fma3(pg, x, (-tb) * cross(eb, r));
	}
}

__device__ __forceinline__ V3 ld3s(const double (*t)[3], int k) { return {t[k][0], t[k][1], t[k][2]}; }

template <int HW>
__device__ __forceinline__ BrickSel brick_select_rt(const ForceParams& P, const SitesGeom& G, int nbx, int nby, int nbz, int vb,
													 int vgrid) {
	const int nb = P.inner_box ? P.inner_n[0] * P.inner_n[1] * P.inner_n[2] : (P.brick_list ? (int)P.n_list : nbx * nby * nbz);
	const int chunk = vgrid / 8;
	const int slot = (vb % 8) * chunk + vb / 8;  // XCD-aware order, see brick.hpp
	BrickSel b;
	b.live = slot < nb;
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
	b.x0 = HW + bx * G.bx;
	b.y0 = HW + by * G.by;
	b.z0 = HW + bz * G.bz;
	b.ex = min(G.bx, P.g.dims[0] - HW - b.x0);
	b.ey = min(G.by, P.g.dims[1] - HW - b.y0);
	b.ez = min(G.bz, P.g.dims[2] - HW - b.z0);
	if (b.live && P.which != 0 && !P.brick_list && !P.inner_box) {
		const bool inner = b.x0 >= 2 * HW && b.y0 >= 2 * HW && b.z0 >= 2 * HW && b.x0 + b.ex <= P.g.dims[0] - 2 * HW &&
						   b.y0 + b.ey <= P.g.dims[1] - 2 * HW && b.z0 + b.ez <= P.g.dims[2] - 2 * HW;
		b.live = (P.which == 1) ? inner : !inner;
	}
	return b;
}

template <int LPM, int MPL, bool WITH_VI, bool HAS_ROT, bool HAS_ES>
__global__ void __launch_bounds__(STPB, HAS_ES ? 2 : 3) k_force_sites(ForceParams P, const CompTable* __restrict__ ctab, SitesGeom G, int nbx,
													   int nby, int nbz, int vgrid) {
	constexpr int HW = 1, NT = STPB, NSLOT = NT / LPM, NOWN = MPL * NSLOT, NW = 3, NROWS = 9;
	extern __shared__ double dyn[];
	const int RX = G.bx + 2 * HW, RY = G.by + 2 * HW, RZ = G.bz + 2 * HW;
	const int NRC = RX * RY * RZ, NBC = G.bx * G.by * G.bz;
	const int CAPM = G.capm, CAPL = G.capl;
	// ---- LDS carve-up (doubles, then u32, u16, u8) -----------------------------------------------------------------------
	double* sx = dyn;
	double* sy = sx + CAPM;
	double* sz = sy + CAPM;
	double* sq0 = sz + CAPM;  // normalised quaternion (HAS_ROT)
	double* sq1 = sq0 + (HAS_ROT ? CAPM : 0);
	double* sq2 = sq1 + (HAS_ROT ? CAPM : 0);
	double* sq3 = sq2 + (HAS_ROT ? CAPM : 0);
	double* icd = sq3 + (HAS_ROT ? CAPM : 0);                      // [3][maxs][NOWN]: own site offsets d_i = R_i p_site
	double* ice = icd + (HAS_ROT ? 3 * G.maxs * NOWN : 0);         // [3][maxe][NOWN]: own site axes e_i
	double* pe24 = ice + (HAS_ROT ? 3 * G.maxe * NOWN : 0);        // LJ pair table
	double* ps2 = pe24 + G.nc2;
	double* psh6 = ps2 + G.nc2;
	double (*red)[4] = reinterpret_cast<double (*)[4]>(psh6 + G.nc2);
	SiteTab& T = *reinterpret_cast<SiteTab*>(&red[NT / 64][0]);
	uint32_t* cstart = reinterpret_cast<uint32_t*>(&T + 1);
	uint32_t* gbeg = cstart + NRC + 1;
	uint32_t* bstart = gbeg + NRC;
	uint32_t* wsum = bstart + NBC + 1;
	uint32_t* m_gi = wsum + NT / 64;            // [NOWN] global index of the owned molecule
	uint32_t* hist = m_gi + NOWN;               // [SORT_BINS + 1] cost histogram / bin starts
	uint32_t* blkflag = hist + SORT_BINS + 1;   // [2]: any molecule left for direct evaluation
	uint16_t* lst = reinterpret_cast<uint16_t*>(blkflag + 2);  // [MPL][CAPL + 1][NT], row CAPL = dummy target of misses
	uint16_t* m_ii = lst + MPL * (CAPL + 1) * NT;  // [NOWN] LDS index of the owned molecule
	uint16_t* m_key = m_ii + NOWN;                 // [NOWN] sort key (component rank, neighbour count); 0xffff = direct evaluation
	uint16_t* order = m_key + NOWN;                // [NOWN] owned molecules by descending cost
	uint16_t* lcnt = order + NOWN;                 // [MPL][NT] list lengths
	uint8_t* scid = reinterpret_cast<uint8_t*>(lcnt + MPL * NT);

	const int tid = threadIdx.x;
	// ---- tables -> LDS ----------------------------------------------------------------------------------------------------
	for (int k = tid; k < MAXS * 3; k += NT) {
		const int s = k / 3, d = k % 3;
		T.ljpos[s][d] = ctab->ljpos[s][d];
		T.chpos[s][d] = ctab->chpos[s][d];
		T.dppos[s][d] = ctab->dppos[s][d];
		T.dpe[s][d] = ctab->dpe[s][d];
		T.qppos[s][d] = ctab->qppos[s][d];
		T.qpe[s][d] = ctab->qpe[s][d];
	}
	if (tid < MAXS) {
		T.chq[tid] = ctab->chq[tid];
		T.dpmy[tid] = ctab->dpmy[tid];
		T.qpQ[tid] = ctab->qpQ[tid];
	}
	if (tid < MAXC) {
		T.nlj[tid] = ctab->nlj[tid]; T.nc[tid] = ctab->nc[tid]; T.nd[tid] = ctab->nd[tid]; T.nq[tid] = ctab->nq[tid];
		T.olj[tid] = ctab->olj[tid]; T.oc[tid] = ctab->oc[tid]; T.od[tid] = ctab->od[tid]; T.oq[tid] = ctab->oq[tid];
	}
	for (int k = tid; k < G.nc2; k += NT) {
		pe24[k] = ctab->eps24[k];
		ps2[k] = ctab->sig2[k];
		psh6[k] = ctab->shift6[k];
	}
	// (A persistent variant — two workgroups per CU walking the brick grid — was measured: the loop-carried scalar state
	// pushed the kernel over 256 VGPRs through SGPR spills, one wave per SIMD, 1.7x slower.)
	const int vb = (int)blockIdx.x;
	do {  // (single trip: `continue` leaves the brick)
	const BrickSel bs = brick_select_rt<HW>(P, G, nbx, nby, nbz, vb, vgrid);
	if (!bs.live) {  // uniform per workgroup
		if (tid < 4) P.partials[(size_t)vb * 4 + tid] = 0.;
		continue;
	}
	const int ex = bs.ex, ey = bs.ey, ez = bs.ez;
	// ---- region cell table, brick cell prefix ---------------------------------------------------------------------------
	for (int c = tid; c < NRC; c += NT) {
		const int rx = c % RX, ry = (c / RX) % RY, rz = c / (RX * RY);
		const int gx = bs.x0 - HW + rx, gy = bs.y0 - HW + ry, gz = bs.z0 - HW + rz;
		uint32_t beg = 0, n = 0;
		if (gx < P.g.dims[0] && gy < P.g.dims[1] && gz < P.g.dims[2]) {
			const int gc = cell_index(P.g, gx, gy, gz);
			beg = P.cell_begin[gc];
			n = P.cell_end[gc] - beg;
		}
		gbeg[c] = beg;
		cstart[c] = n;
	}
	__syncthreads();
	if (G.dbg == 3) {  // tables + region table only
		if (tid < 4) P.partials[(size_t)vb * 4 + tid] = 0.;
		continue;
	}
	block_scan_lds<NT>(cstart, NRC, wsum);
	const uint32_t total = cstart[NRC];
	for (int c = tid; c < NBC; c += NT) {
		const int cx = c % G.bx, cy = (c / G.bx) % G.by, cz = c / (G.bx * G.by);
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
	const bool staged = total <= (uint32_t)CAPM;
	// ---- stage centres, normalised quaternions (FullMolecule.cpp:720), component ids -------------------------------------
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
	if (G.dbg == 4) {  // ... + scans + staging
		if (tid < 4) P.partials[(size_t)vb * 4 + tid] = 0.;
		continue;
	}

	const double rc2 = ctab->rc2, rclj2 = ctab->rclj2, rf_eps = ctab->epsRFInvrc3;
	const int ncen = ctab->ncenters;
	double u6_t = 0., uX_t = 0., rf_t = 0., vir_t = 0.;
	const int h = tid % LPM, slot = tid / LPM;  // LPM adjacent lanes share one owned molecule
	const int estride = G.maxs * NOWN;          // distance between the x / y / z planes of the site cache
	const int astride = G.maxe * NOWN;

	// Owned molecule `it` (brick enumeration index) -> (region cell, LDS index, global index, first neighbour row)
	struct Own {
		uint32_t ii, gi;
		int rowbase;
	};
	auto locate = [&](uint32_t it) __attribute__((always_inline)) {
		int lo = 0, hi = NBC;
		while (hi - lo > 1) {
			const int mid = (lo + hi) >> 1;
			if (bstart[mid] <= it) lo = mid;
			else hi = mid;
		}
		const int cx = lo % G.bx, cy = (lo / G.bx) % G.by, cz = lo / (G.bx * G.by);
		const int rcell = ((cz + HW) * RY + (cy + HW)) * RX + (cx + HW);
		const uint32_t k = it - bstart[lo];
		Own o;
		o.ii = cstart[rcell] + k;
		o.gi = gbeg[rcell] + k;
		o.rowbase = (cz * RY + cy) * RX + cx;
		return o;
	};
	// rotate the sites of owned molecule mi (component ci, normalised quaternion) into the site cache; the LPM lanes split them
	auto fill_cache = [&](int mi, int ci, double w, double x, double y, double z) __attribute__((always_inline)) {
		const Rot Ri = rot_of(w, x, y, z);
		const int nl = T.nlj[ci], nc = T.nc[ci], nd = T.nd[ci], nq = T.nq[ci];
		const int ns = nl + nc + nd + nq;
		for (int s = h; s < ns; s += LPM) {
			V3 p, e = {0., 0., 0.};
			int eidx = -1;
			if (s < nl) p = ld3s(T.ljpos, T.olj[ci] + s);
			else if (s < nl + nc) p = ld3s(T.chpos, T.oc[ci] + s - nl);
			else if (s < nl + nc + nd) {
				const int q = s - nl - nc;
				p = ld3s(T.dppos, T.od[ci] + q);
				e = ld3s(T.dpe, T.od[ci] + q);
				eidx = q;
			} else {
				const int q = s - nl - nc - nd;
				p = ld3s(T.qppos, T.oq[ci] + q);
				e = ld3s(T.qpe, T.oq[ci] + q);
				eidx = nd + q;
			}
			const V3 d = rotate(Ri, p);
			icd[s * NOWN + mi] = d.x;
			icd[estride + s * NOWN + mi] = d.y;
			icd[2 * estride + s * NOWN + mi] = d.z;
			if (eidx >= 0) {
				const V3 a = rotate(Ri, e);
				ice[eidx * NOWN + mi] = a.x;
				ice[astride + eidx * NOWN + mi] = a.y;
				ice[2 * astride + eidx * NOWN + mi] = a.z;
			}
		}
	};

	// ---- per-molecule evaluation state of a lane (one owned molecule at a time) ------------------------------------------
	V3 ri, F, M, Vi;
	double u6, uX, rfs, vir;
	int cmi, nlji, nci, ndi, nqi, oli, oci, odi, oqi;
	auto begin_molecule = [&](int mi, V3 r, int c) __attribute__((always_inline)) {
		ri = r;
		cmi = mi;
		nlji = T.nlj[c]; nci = T.nc[c]; ndi = T.nd[c]; nqi = T.nq[c];
		oli = T.olj[c]; oci = T.oc[c]; odi = T.od[c]; oqi = T.oq[c];
		F = {0., 0., 0.}; M = {0., 0., 0.}; Vi = {0., 0., 0.};
		u6 = uX = rfs = vir = 0.;
	};
	auto own_d = [&](int s) __attribute__((always_inline)) -> V3 {
		if (!HAS_ROT) return {0., 0., 0.};
		return {icd[s * NOWN + cmi], icd[estride + s * NOWN + cmi], icd[2 * estride + s * NOWN + cmi]};
	};
	auto own_e = [&](int q) __attribute__((always_inline)) -> V3 { return {ice[q * NOWN + cmi], ice[astride + q * NOWN + cmi], ice[2 * astride + q * NOWN + cmi]}; };
	// everything molecule j (centre rj, normalised quaternion, component cj) does to the lane's molecule
	auto pair = [&](V3 rj, double q0, double q1, double q2, double q3, int cj) __attribute__((always_inline)) {
		const V3 drm = ri - rj;
		Rot Rj;
		if (HAS_ROT) Rj = rot_of(q0, q1, q2, q3);
		V3 Fp = {0., 0., 0.};
		double pu6 = 0., puX = 0., prf = 0.;
		V3 f, t;
		double u;
		auto add = [&](V3 di, V3 fs) __attribute__((always_inline)) {  // force on a site of i at offset di: total force + lever-arm torque
			Fp = Fp + fs;
			if (HAS_ROT) {
				M.x = fma(di.y, fs.z, fma(-di.z, fs.y, M.x));
				M.y = fma(di.z, fs.x, fma(-di.x, fs.z, M.y));
				M.z = fma(di.x, fs.y, fma(-di.y, fs.x, M.z));
			}
		};
		if (dot(drm, drm) < rclj2) {  // LJ uses the LJ cutoff on the CENTRE distance (VectorizedCellProcessor.cpp:967-968)
			const int nj = T.nlj[cj], oj = T.olj[cj];
			for (int sj = 0; sj < nj; ++sj) {
				const V3 a = HAS_ROT ? drm - rotate(Rj, ld3s(T.ljpos, oj + sj)) : drm;
				for (int si = 0; si < nlji; ++si) {
					const V3 di = own_d(si);
					const int kk = (oli + si) * ncen + oj + sj;
					sp_lj(a + di, pe24[kk], ps2[kk], f, u);
					pu6 += u + psh6[kk];
					add(di, f);
				}
			}
		}
		if (HAS_ES) {  // charges of j
			const int nj = T.nc[cj], oj = T.oc[cj];
			for (int sj = 0; sj < nj; ++sj) {
				const V3 a = HAS_ROT ? drm - rotate(Rj, ld3s(T.chpos, oj + sj)) : drm;
				const double qj = T.chq[oj + sj];
				for (int si = 0; si < nci; ++si) {
					const V3 di = own_d(nlji + si);
					sp_cc(a + di, T.chq[oci + si] * qj, f, u);
					puX += u;
					add(di, f);
				}
				if (!HAS_ROT) continue;  // dipoles / quadrupoles imply HAS_ROT
				for (int si = 0; si < ndi; ++si) {  // dipole of i - charge of j
					const V3 di = own_d(nlji + nci + si);
					const V3 rji = {-(a.x + di.x), -(a.y + di.y), -(a.z + di.z)};
					sp_charge_dipole<false>(rji, own_e(si), -qj * T.dpmy[odi + si], f, t, u);
					puX += u;
					add(di, f);
					M = M + t;
				}
				for (int si = 0; si < nqi; ++si) {  // quadrupole of i - charge of j
					const V3 di = own_d(nlji + nci + ndi + si);
					const V3 rji = {-(a.x + di.x), -(a.y + di.y), -(a.z + di.z)};
					sp_charge_quadrupole<false>(rji, own_e(ndi + si), 0.5 * qj * T.qpQ[oqi + si], f, t, u);
					puX += u;
					add(di, f);
					M = M + t;
				}
			}
		}
		if (HAS_ROT && HAS_ES) {
			{  // dipoles of j
				const int nj = T.nd[cj], oj = T.od[cj];
				for (int sj = 0; sj < nj; ++sj) {
					const V3 a = drm - rotate(Rj, ld3s(T.dppos, oj + sj));
					const V3 ej = rotate(Rj, ld3s(T.dpe, oj + sj));
					const double myj = T.dpmy[oj + sj];
					for (int si = 0; si < nci; ++si) {
						const V3 di = own_d(nlji + si);
						sp_charge_dipole<true>(a + di, ej, -T.chq[oci + si] * myj, f, t, u);
						puX += u;
						add(di, f);
					}
					for (int si = 0; si < ndi; ++si) {
						const V3 di = own_d(nlji + nci + si);
						const double my2 = T.dpmy[odi + si] * myj;
						double r1;
						sp_dd(a + di, own_e(si), ej, my2, my2 * rf_eps, f, t, u, r1);
						puX += u;
						prf += r1;
						add(di, f);
						M = M + t;
					}
					for (int si = 0; si < nqi; ++si) {  // quadrupole of i - dipole of j
						const V3 di = own_d(nlji + nci + ndi + si);
						const V3 rji = {-(a.x + di.x), -(a.y + di.y), -(a.z + di.z)};
						sp_dq<false>(rji, ej, own_e(ndi + si), 1.5 * T.qpQ[oqi + si] * myj, f, t, u);
						puX += u;
						add(di, f);
						M = M + t;
					}
				}
			}
			{  // quadrupoles of j
				const int nj = T.nq[cj], oj = T.oq[cj];
				for (int sj = 0; sj < nj; ++sj) {
					const V3 a = drm - rotate(Rj, ld3s(T.qppos, oj + sj));
					const V3 ej = rotate(Rj, ld3s(T.qpe, oj + sj));
					const double Qj = T.qpQ[oj + sj];
					for (int si = 0; si < nci; ++si) {
						const V3 di = own_d(nlji + si);
						sp_charge_quadrupole<true>(a + di, ej, 0.5 * T.chq[oci + si] * Qj, f, t, u);
						puX += u;
						add(di, f);
					}
					for (int si = 0; si < ndi; ++si) {
						const V3 di = own_d(nlji + nci + si);
						sp_dq<true>(a + di, own_e(si), ej, 1.5 * T.dpmy[odi + si] * Qj, f, t, u);
						puX += u;
						add(di, f);
						M = M + t;
					}
					for (int si = 0; si < nqi; ++si) {
						const V3 di = own_d(nlji + nci + ndi + si);
						sp_qq(a + di, own_e(ndi + si), ej, 0.75 * T.qpQ[oqi + si] * Qj, f, t, u);
						puX += u;
						add(di, f);
						M = M + t;
					}
				}
			}
		}
		F = F + Fp;
		if (WITH_VI) {
			Vi.x = fma(0.5 * drm.x, Fp.x, Vi.x);
			Vi.y = fma(0.5 * drm.y, Fp.y, Vi.y);
			Vi.z = fma(0.5 * drm.z, Fp.z, Vi.z);
		}
		u6 = fma(0.5, pu6, u6);
		uX = fma(0.5, puX, uX);
		rfs = fma(0.5, prf, rfs);
		vir = fma(0.5, dot(drm, Fp), vir);
	};
	// add the LPM partial sums of the molecule, store its results, keep the macroscopic sums of the lane
	auto finish_molecule = [&](bool valid, uint32_t gi) __attribute__((always_inline)) {
		for (int o = 1; o < LPM; o <<= 1) {
			F.x += __shfl_xor(F.x, o); F.y += __shfl_xor(F.y, o); F.z += __shfl_xor(F.z, o);
			if (HAS_ROT) { M.x += __shfl_xor(M.x, o); M.y += __shfl_xor(M.y, o); M.z += __shfl_xor(M.z, o); }
			if (WITH_VI) { Vi.x += __shfl_xor(Vi.x, o); Vi.y += __shfl_xor(Vi.y, o); Vi.z += __shfl_xor(Vi.z, o); }
		}
		if (valid && h == 0) {
			P.Fx[gi] = F.x;
			P.Fy[gi] = F.y;
			P.Fz[gi] = F.z;
			if (HAS_ROT) {
				P.Mx[gi] = M.x;
				P.My[gi] = M.y;
				P.Mz[gi] = M.z;
			}
			if (WITH_VI) {
				P.Vix[gi] = Vi.x;
				P.Viy[gi] = Vi.y;
				P.Viz[gi] = Vi.z;
			}
		}
		if (valid) {
			u6_t += u6;
			uX_t += uX;
			rf_t += rfs;
			vir_t += vir;
		}
	};
	// direct evaluation of owned molecule (mi, o) straight from global memory: bricks whose shell does not fit the staging
	// area, and molecules with more neighbours than list rows.  Same arithmetic as the list path.
	auto direct = [&](bool valid, int mi, const Own& o) __attribute__((always_inline)) {
		if (valid) {
			const uint32_t gi = o.gi;
			double w = 1., x = 0., y = 0., z = 0.;
			if (HAS_ROT) {
				w = P.q0[gi]; x = P.q1[gi]; y = P.q2[gi]; z = P.q3[gi];
				const double inv = 1. / sqrt(w * w + x * x + y * y + z * z);
				w *= inv; x *= inv; y *= inv; z *= inv;
				fill_cache(mi, P.cid[gi], w, x, y, z);
			}
		}
		__builtin_amdgcn_wave_barrier();  // the LPM lanes of a molecule sit in one wave: LDS order is program order
		if (valid) {
			const uint32_t gi = o.gi;
			begin_molecule(mi, V3{P.x[gi], P.y[gi], P.z[gi]}, P.cid[gi]);
			for (int row = 0; row < NROWS; ++row) {
				const int r0 = o.rowbase + (row / NW) * (RY * RX) + (row % NW) * RX;
				for (int c = r0; c < r0 + NW; ++c) {
					const uint32_t g0 = gbeg[c], n = cstart[c + 1] - cstart[c];
					for (uint32_t j = g0 + (uint32_t)h; j < g0 + n; j += LPM) {
						if (j == gi) continue;
						const V3 rj = {P.x[j], P.y[j], P.z[j]};
						const double dx = ri.x - rj.x, dy = ri.y - rj.y, dz = ri.z - rj.z;
						const double dd = fma(dx, dx, fma(dy, dy, dz * dz));
						if (!(dd < rc2)) continue;
						double w = 1., x = 0., y = 0., z = 0.;
						if (HAS_ROT) {
							w = P.q0[j]; x = P.q1[j]; y = P.q2[j]; z = P.q3[j];
							const double inv = 1. / sqrt(w * w + x * x + y * y + z * z);
							w *= inv; x *= inv; y *= inv; z *= inv;
						}
						pair(rj, w, x, y, z, P.cid[j]);
					}
				}
			}
		} else {
			F = {0., 0., 0.}; M = {0., 0., 0.}; Vi = {0., 0., 0.};
		}
		finish_molecule(valid, o.gi);
	};

	for (uint32_t chunk = 0; chunk < n_i; chunk += NOWN) {  // one chunk when the host chose the brick shape well
		const uint32_t n_c = min((uint32_t)NOWN, n_i - chunk);
		if (!staged) {
			for (int r = 0; r < MPL; ++r) {
				const int mi = r * NSLOT + slot;
				const bool valid = (uint32_t)mi < n_c;
				const Own o = locate(valid ? chunk + (uint32_t)mi : 0u);
				direct(valid, mi, o);
			}
			__syncthreads();
			continue;
		}
		// ---- phase 1: every lane searches the candidates of MPL owned molecules (its share: every LPM-th candidate) -------
		for (int k = tid; k <= SORT_BINS; k += NT) hist[k] = 0;
		if (tid < 2) blkflag[tid] = 0;
		__syncthreads();
		for (int r = 0; r < MPL; ++r) {
			const int mi = r * NSLOT + slot;
			const bool valid = (uint32_t)mi < n_c;
			const Own o = locate(valid ? chunk + (uint32_t)mi : 0u);
			uint32_t cnt = 0;
			int c_i = 0;
			if (valid) {
				const uint32_t ii = o.ii;
				c_i = scid[ii];
				if (HAS_ROT) fill_cache(mi, c_i, sq0[ii], sq1[ii], sq2[ii], sq3[ii]);
				const double xi = sx[ii], yi = sy[ii], zi = sz[ii];
				uint16_t* const mylist = lst + (size_t)r * (CAPL + 1) * NT + tid;
				for (int row = 0; row < (G.dbg == 2 ? 0 : NROWS); ++row) {
					const int r0 = o.rowbase + (row / NW) * (RY * RX) + (row % NW) * RX;
					const uint32_t jb = cstart[r0], je = cstart[r0 + NW];
					for (uint32_t j = jb + (uint32_t)h; j < je; j += LPM) {
						const double dx = xi - sx[j], dy = yi - sy[j], dz = zi - sz[j];
						const double dd = fma(dx, dx, fma(dy, dy, dz * dz));
						const bool hit = (dd < rc2) & (j != ii);
						mylist[(hit ? min(cnt, (uint32_t)CAPL) : (uint32_t)CAPL) * NT] = (uint16_t)j;
						cnt += hit ? 1u : 0u;
					}
				}
			}
			lcnt[r * NT + tid] = (uint16_t)min(cnt, (uint32_t)CAPL);
			uint32_t cmax = cnt;
			for (int o2 = 1; o2 < LPM; o2 <<= 1) cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, o2));
			if (h == 0) {
				// sort key: component rank (costly components first), then neighbour count, descending -> ascending bin number
				uint32_t bin = SORT_BINS;  // not there: behind everything
				if (valid) {
					if (cmax > (uint32_t)CAPL) {
						bin = SORT_BINS;  // more neighbours than list rows: evaluated directly after phase 2
						blkflag[0] = 1;
					} else {
						bin = (uint32_t)G.crank[c_i] * SORT_CNT + (SORT_CNT - 1 - min(cmax, (uint32_t)SORT_CNT - 1));
					}
				}
				m_ii[mi] = (uint16_t)o.ii;
				m_gi[mi] = o.gi;
				m_key[mi] = (uint16_t)((valid && cmax > (uint32_t)CAPL) ? 0xffffu : (valid ? bin : 0xfffeu));
				const uint32_t rank = atomicAdd(&hist[bin], 1u);
				order[mi] = (uint16_t)rank;  // rank inside the bin, for now
			}
		}
		__syncthreads();
		// ---- cost order: pass 0 takes the NSLOT most expensive molecules, pass 1 the rest in reverse — lane sums level out ----
		uint32_t myrank[MPL], mybin[MPL];
		for (int r = 0; r < MPL; ++r) {
			const int mi = r * NSLOT + slot;
			myrank[r] = order[mi];
			const uint32_t key = m_key[mi];
			mybin[r] = key >= 0xfffeu ? (uint32_t)SORT_BINS : key;
		}
		__syncthreads();
		block_scan_lds<NT>(hist, SORT_BINS, wsum);  // hist[SORT_BINS] = start of the bin of absent / directly evaluated molecules
		if (h == 0)
			for (int r = 0; r < MPL; ++r) order[hist[mybin[r]] + myrank[r]] = (uint16_t)(r * NSLOT + slot);
		__syncthreads();
		// ---- phase 2: the pair body over the lists, every lane busy ------------------------------------------------------------
		for (int pass = 0; pass < MPL; ++pass) {
			const int mi = order[pass == 0 ? slot : NOWN - 1 - slot];
			const uint32_t key = m_key[mi];
			const bool valid = key < 0xfffeu;
			if (valid) {
				const uint32_t ii = m_ii[mi];
				begin_molecule(mi, V3{sx[ii], sy[ii], sz[ii]}, (int)scid[ii]);
				const int col = (mi % NSLOT) * LPM + h;  // the lane that searched this share of the molecule's candidates
				const uint16_t* const list = lst + (size_t)(mi / NSLOT) * (CAPL + 1) * NT + col;
				const uint32_t cnt = lcnt[(mi / NSLOT) * NT + col];
				for (uint32_t s = 0; s < (G.dbg == 1 ? 0u : cnt); ++s) {
					const uint32_t j = list[s * NT];
					const V3 rj = {sx[j], sy[j], sz[j]};
					if (HAS_ROT) pair(rj, sq0[j], sq1[j], sq2[j], sq3[j], (int)scid[j]);
					else pair(rj, 1., 0., 0., 0., (int)scid[j]);
				}
			} else {
				F = {0., 0., 0.}; M = {0., 0., 0.}; Vi = {0., 0., 0.};
			}
			finish_molecule(valid, m_gi[mi]);
		}
		if (blkflag[0]) {  // uniform: some molecule has more neighbours than list rows
			for (int r = 0; r < MPL; ++r) {
				const int mi = r * NSLOT + slot;
				const bool valid = (uint32_t)mi < n_c && m_key[mi] == 0xffffu;
				const Own o = locate((uint32_t)mi < n_c ? chunk + (uint32_t)mi : 0u);
				direct(valid, mi, o);
			}
		}
		__syncthreads();
	}
	// ---- block reduction of the macroscopic sums -> partials[blockIdx.x][4] ----------------------------------------------------
	double v[4] = {u6_t, uX_t, rf_t, vir_t};
	for (int q = 0; q < 4; ++q)
		for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_down(v[q], o);
	__syncthreads();
	if ((tid & 63) == 0)
		for (int q = 0; q < 4; ++q) red[tid >> 6][q] = v[q];
	__syncthreads();
	if (tid < 4) {
		double s = 0.;
		for (int i = 0; i < NT / 64; ++i) s += red[i][tid];
		P.partials[(size_t)vb * 4 + tid] = s;
	}
	} while (false);
}

// ---- host side: brick shape, LDS budget, launch --------------------------------------------------------------------------
static size_t sites_lds_bytes(const SitesGeom& G, int lpm, int MPL, bool has_rot) {
	const int nslot = STPB / lpm, nown = MPL * nslot;
	const int nrc = (G.bx + 2) * (G.by + 2) * (G.bz + 2), nbc = G.bx * G.by * G.bz;
	size_t b = 0;
	b += (size_t)G.capm * 8 * (has_rot ? 7 : 3);
	if (has_rot) b += (size_t)3 * (G.maxs + G.maxe) * nown * 8;
	b += (size_t)3 * G.nc2 * 8;
	b += (STPB / 64) * 4 * 8 + sizeof(SiteTab);
	b += (size_t)(nrc + 1 + nrc + nbc + 1 + STPB / 64 + nown + SORT_BINS + 1 + 2) * 4;
	b += (size_t)(MPL * (G.capl + 1) * STPB + 3 * nown + MPL * STPB) * 2;
	b += (size_t)G.capm;
	return (b + 15) & ~(size_t)15;
}

// The brick shape follows the mean cell occupancy: owned molecules x LPM should fill the 256 lanes in one pass (a second
// pass with a handful of molecules costs as much as the first), and brick + shell must fit the staging area with 8 %
// headroom for density fluctuations.  Two workgroups per CU (80 KB each) keep one workgroup computing while the other
// stages.  Returns false when nothing fits (-> the first-generation brick kernel / the generic kernel).
bool launch_force_sites(const ForceParams& p, const CompTable& hct, bool with_vi, hipStream_t s, uint32_t* nblocks,
						size_t partials_cap, double mean_per_cell, double mean_neighbours, BrickLists* bl) {
	if (p.g.hw != 1 || p.ct == nullptr) return false;
	const bool has_rot = hct.has_rot != 0;
	SitesGeom G;
	G.maxs = 1;
	G.maxe = 0;
	for (int c = 0; c < hct.ncomp; ++c) {
		G.maxs = std::max(G.maxs, hct.nlj[c] + hct.nc[c] + hct.nd[c] + hct.nq[c]);
		G.maxe = std::max(G.maxe, hct.nd[c] + hct.nq[c]);
	}
	G.nc2 = hct.ncenters * hct.ncenters;
	G.dbg = getenv("LS1HIP_SITES_DBG") ? atoi(getenv("LS1HIP_SITES_DBG")) : 0;
	for (int c = 0; c < MAXC; ++c) G.crank[c] = 0;
	for (int c = 0; c < hct.ncomp; ++c) {  // cost rank: components with more sites first (ties: lower id first)
		const int nsc = hct.nlj[c] + hct.nc[c] + hct.nd[c] + hct.nq[c];
		int rank = 0;
		for (int d = 0; d < hct.ncomp; ++d) {
			const int nsd = hct.nlj[d] + hct.nc[d] + hct.nd[d] + hct.nq[d];
			if (nsd > nsc || (nsd == nsc && d < c)) ++rank;
		}
		G.crank[c] = (uint8_t)rank;
	}
	bool has_es = false;
	for (int c = 0; c < hct.ncomp; ++c) has_es |= (hct.nc[c] + hct.nd[c] + hct.nq[c]) > 0;
	const double m = std::max(mean_per_cell, 1e-3);
	// LDS budget per workgroup: two workgroups per CU (the electrostatic bodies need ~250 VGPRs: two waves per SIMD anyway);
	// LJ-only component sets (~125 VGPRs) may run three
	const size_t budgets[2] = {has_es ? (size_t)80 * 1024 - 256 : (size_t)53 * 1024, (size_t)80 * 1024 - 256};
	static const int shapes[][3] = {{8, 8, 4}, {8, 6, 4}, {8, 4, 4}, {7, 4, 4}, {6, 4, 4}, {5, 4, 4}, {4, 4, 4}, {6, 4, 2}, {5, 4, 2}, {4, 4, 2},
									{6, 2, 2}, {5, 2, 2}, {4, 2, 2}, {3, 2, 2}, {2, 2, 2}, {2, 2, 1}, {2, 1, 1}, {1, 1, 1}};
	auto size_for = [&](SitesGeom& T, int lpm) {
		const int nrc = (T.bx + 2) * (T.by + 2) * (T.bz + 2);
		// staging capacity: mean shell population + 25 % (a real fluid fluctuates more than Poisson; an overflowing brick
		// falls back to global memory, which costs several times more)
		T.capm = (int)(m * nrc * 1.25) + 32;
		T.capm = (T.capm + 7) & ~7;
		// list rows: the lane's share of the neighbours + 6 sigma (Poisson): a molecule with more neighbours than rows is
		// evaluated straight from global memory after phase 2, which must stay a rare event
		const double mu = mean_neighbours / lpm;
		T.capl = std::min(SORT_CNT - 1, std::max(8, (int)(mu + 6. * std::sqrt(mu) + 3.)));
		return nrc;
	};
	int best_lpm = 0, best_mpl = 1;
	double best_score = -1.;
	SitesGeom bestG = G;
	if (const char* e = getenv("LS1HIP_SITES_SHAPE")) {  // diagnostics: "bx,by,bz,lpm,mpl"
		SitesGeom T = G;
		int lpm = 1, mpl = 1;
		if (sscanf(e, "%d,%d,%d,%d,%d", &T.bx, &T.by, &T.bz, &lpm, &mpl) == 5 && (lpm == 1 || lpm == 2 || lpm == 4) &&
			(mpl == 1 || mpl == 2) && size_for(T, lpm) <= STPB * 4 && T.capm <= 0xfff0 && sites_lds_bytes(T, lpm, mpl, has_rot) <= 160 * 1024 - 256) {
			best_lpm = lpm;
			best_mpl = mpl;
			bestG = T;
		}
	}
	for (size_t BUDGET : budgets)
	if (!best_lpm)
		for (int mpl : {1, 2})
			for (int lpm : {1, 2, 4}) {
				const int nslot = STPB / lpm;
				for (int k = 0; k < (int)(sizeof(shapes) / sizeof(shapes[0])); ++k) {
					SitesGeom T = G;
					T.bx = shapes[k][0]; T.by = shapes[k][1]; T.bz = shapes[k][2];
					const double owned = m * T.bx * T.by * T.bz;
					if (owned * 1.06 > mpl * nslot) continue;  // must (almost always) be a single chunk
					const int nrc = size_for(T, lpm);
					if (nrc > STPB * 4 || T.capm > 0xfff0) continue;
					if (sites_lds_bytes(T, lpm, mpl, has_rot) > BUDGET) continue;
					// lanes in use x how little of the staged shell is halo
					const double score = (owned / (mpl * nslot)) * (double)(T.bx * T.by * T.bz) / nrc;
					if (score > best_score) {
						best_score = score;
						best_lpm = lpm;
						best_mpl = mpl;
						bestG = T;
					}
				}
			}
	if (!best_lpm) {
		if (getenv("LS1HIP_DEBUG_SITES")) fprintf(stderr, "[ls1hip] site kernel: no brick shape fits (%.2f molecules per cell, %.1f neighbours)\n", mean_per_cell, mean_neighbours);
		return false;
	}
	G = bestG;
	const Grid& g = p.g;
	const int nbx = (g.box[0] + G.bx - 1) / G.bx, nby = (g.box[1] + G.by - 1) / G.by, nbz = (g.box[2] + G.bz - 1) / G.bz;
	if ((long)nbx * nby * nbz <= 0 || (long)nbx * nby * nbz > 0x7ffffff0L) return false;
	ForceParams q = p;
	const long nb = plan_bricks(q, bl, G.bx, G.by, G.bz, nbx, nby, nbz);
	if ((size_t)nb > partials_cap) return false;
	*nblocks = (uint32_t)nb;
	if (nb == 0) return true;
	const size_t lds = sites_lds_bytes(G, best_lpm, best_mpl, has_rot);
	static const bool debug = getenv("LS1HIP_DEBUG_SITES") != nullptr && atoi(getenv("LS1HIP_DEBUG_SITES")) != 0;
	if (debug)
		fprintf(stderr, "[ls1hip] site kernel: brick %dx%dx%d, %d lanes per molecule, %d molecules per lane slot, staging %d molecules, "
						"%d list rows, %zu B LDS, %ld workgroups (%.2f molecules per cell, %.1f neighbours)\n",
				G.bx, G.by, G.bz, best_lpm, best_mpl, G.capm, G.capl, lds, nb, mean_per_cell, mean_neighbours);
	const dim3 grid((uint32_t)nb), block(STPB);
	auto go = [&](auto lpm, auto mpl, auto vi, auto rot, auto es) {
		constexpr int L = decltype(lpm)::value, MP = decltype(mpl)::value;
		constexpr bool VI = decltype(vi)::value, ROT = decltype(rot)::value, ES = decltype(es)::value;
		// (dynamic LDS above 64 KB needs the attribute; it is per device, so it is simply set with every launch)
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_force_sites<L, MP, VI, ROT, ES>),
								  hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(lds, (size_t)80 * 1024));
		hipLaunchKernelGGL((k_force_sites<L, MP, VI, ROT, ES>), grid, block, lds, s, q, q.ct, G, nbx, nby, nbz, (int)nb);
	};
	auto pick4 = [&](auto lpm, auto mpl, auto vi, auto rot) {
		if (has_es) go(lpm, mpl, vi, rot, std::true_type{});
		else go(lpm, mpl, vi, rot, std::false_type{});
	};
	auto pick3 = [&](auto lpm, auto mpl, auto vi) {
		if (has_rot) pick4(lpm, mpl, vi, std::true_type{});
		else pick4(lpm, mpl, vi, std::false_type{});
	};
	auto pick2 = [&](auto lpm, auto mpl) {
		if (with_vi) pick3(lpm, mpl, std::true_type{});
		else pick3(lpm, mpl, std::false_type{});
	};
	auto pick1 = [&](auto lpm) {
		if (best_mpl == 1) pick2(lpm, std::integral_constant<int, 1>{});
		else pick2(lpm, std::integral_constant<int, 2>{});
	};
	if (best_lpm == 1) pick1(std::integral_constant<int, 1>{});
	else if (best_lpm == 2) pick1(std::integral_constant<int, 2>{});
	else pick1(std::integral_constant<int, 4>{});
	return true;
}

}  // namespace ls1
