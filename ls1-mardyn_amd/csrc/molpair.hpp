// molpair.hpp — component tables and the one-sided molecule-pair routine (all ten site-type combinations).
//
// Restates, for molecule i only, PotForce (/root/reference/src/molecules/potforce.h:282-503) with the parameter
// products of Comp2Param::initialize (/root/reference/src/molecules/Comp2Param.cpp:98-186) formed on the fly from
// per-site values exactly as VectorizedCellProcessor does (adapter/VectorizedCellProcessor.cpp:252,317,391,490 ...).
#pragma once
#include "pairphys.hpp"

namespace ls1 {

constexpr int MAXC = 8;   // components
constexpr int MAXS = 16;  // sites of one type summed over all components

// Plain-old-data, lives in device global memory (read through the scalar/L1 caches; indices are near-uniform).
struct CompTable {
	int ncomp, ncenters, maxsites, has_rot;
	int nlj[MAXC], nc[MAXC], nd[MAXC], nq[MAXC];
	int olj[MAXC], oc[MAXC], od[MAXC], oq[MAXC];
	int rotdof[MAXC];
	double mass[MAXC], I[MAXC][3], invI[MAXC][3];
	double ljpos[MAXS][3];
	double chpos[MAXS][3], chq[MAXS];
	double dppos[MAXS][3], dpe[MAXS][3], dpmy[MAXS];
	double qppos[MAXS][3], qpe[MAXS][3], qpQ[MAXS];
	double eps24[MAXS * MAXS], sig2[MAXS * MAXS], shift6[MAXS * MAXS];  // [ci*ncenters + cj]
	double rc2, rclj2, epsRFInvrc3;
};

struct MolAcc {
	V3 F, M, Vi;
	double u6, uX, rf, vir;
};

template <class T3>
LS1_HD V3 ld3(const T3& t, int k) {  // t: double [n][3] in any address space
	return {t[k][0], t[k][1], t[k][2]};
}

// Accumulate on molecule i everything molecule j does to it.  drm = r_i - r_j (centres); calcLJ per
// VectorizedCellProcessor.cpp:967-968,1013-1024 (LJ uses the LJ cutoff on the CENTRE distance).
// w = weight of the pair's macroscopic contribution seen from i (0.5 in full-shell mode).
// CT: CompTable in any address space (kernels_force_mslist.hip reads it through the constant address space, so that the
// table loads behind its wave-uniform component indices become scalar loads)
// LJ_ONLY: the component set has no charges, dipoles or quadrupoles (their loops — and their registers — are compiled out)
// ROT: Rot (general) or RotAxis (every site of the component set on the body z axis; LJ_ONLY sets only)
template <bool WITH_VI, class CT = CompTable, bool LJ_ONLY = false, class ROT = Rot>
LS1_HD void mol_pair(const CT& ct, int ci, V3 ri, const ROT& Ri, int cj, V3 rj, const ROT& Rj, V3 drm,
					 bool calcLJ, double w, MolAcc& a) {
	static_assert(LJ_ONLY || sizeof(ROT) == sizeof(Rot), "the axis form serves LJ-only component sets");
	V3 Fp = {0., 0., 0.};  // force on i from this pair (for the virial)
	double u6 = 0., uX = 0., rf = 0.;
	V3 f, m1, m2;
	double u;
	const int nlji = ct.nlj[ci], nljj = ct.nlj[cj];
	const int nci = LJ_ONLY ? 0 : ct.nc[ci], ncj = LJ_ONLY ? 0 : ct.nc[cj];
	const int ndi = LJ_ONLY ? 0 : ct.nd[ci], ndj = LJ_ONLY ? 0 : ct.nd[cj];
	const int nqi = LJ_ONLY ? 0 : ct.nq[ci], nqj = LJ_ONLY ? 0 : ct.nq[cj];
	if (calcLJ) {
		for (int si = 0; si < nlji; ++si) {
			const int gi = ct.olj[ci] + si;
			const V3 di = rotate(Ri, ld3(ct.ljpos, gi));
			const V3 pi = ri + di;
			V3 fs = {0., 0., 0.};
			for (int sj = 0; sj < nljj; ++sj) {
				const int gj = ct.olj[cj] + sj;
				const V3 pj = rj + rotate(Rj, ld3(ct.ljpos, gj));
				const V3 dr = pi - pj;
				const int k = gi * ct.ncenters + gj;
				lj(dr, dot(dr, dr), ct.eps24[k], ct.sig2[k], f, u);
				u6 += u + ct.shift6[k];
				fs = fs + f;
			}
			Fp = Fp + fs;
			a.M = a.M + cross(di, fs);
		}
	}
	for (int si = 0; si < nci; ++si) {
		const int gi = ct.oc[ci] + si;
		const V3 di = rotate(Ri, ld3(ct.chpos, gi));
		const V3 pi = ri + di;
		const double qi = ct.chq[gi];
		V3 fs = {0., 0., 0.};
		for (int sj = 0; sj < ncj; ++sj) {  // charge-charge, potforce.h:332-346
			const int gj = ct.oc[cj] + sj;
			const V3 dr = pi - (rj + rotate(Rj, ld3(ct.chpos, gj)));
			charge_charge(dr, dot(dr, dr), qi * ct.chq[gj], f, u);
			uX += u;
			fs = fs + f;
		}
		for (int sj = 0; sj < nqj; ++sj) {  // charge-quadrupole, :347-363
			const int gj = ct.oq[cj] + sj;
			const V3 dr = pi - (rj + rotate(Rj, ld3(ct.qppos, gj)));
			charge_quadrupole(dr, dot(dr, dr), rotate(Rj, ld3(ct.qpe, gj)), 0.5 * qi * ct.qpQ[gj], f, m2, u);
			uX += u;
			fs = fs + f;
		}
		for (int sj = 0; sj < ndj; ++sj) {  // charge-dipole, :364-380
			const int gj = ct.od[cj] + sj;
			const V3 dr = pi - (rj + rotate(Rj, ld3(ct.dppos, gj)));
			charge_dipole(dr, dot(dr, dr), rotate(Rj, ld3(ct.dpe, gj)), -qi * ct.dpmy[gj], f, m2, u);
			uX += u;
			fs = fs + f;
		}
		Fp = Fp + fs;
		a.M = a.M + cross(di, fs);
	}
	for (int si = 0; si < nqi; ++si) {
		const int gi = ct.oq[ci] + si;
		const V3 di = rotate(Ri, ld3(ct.qppos, gi));
		const V3 pi = ri + di;
		const V3 ei = rotate(Ri, ld3(ct.qpe, gi));
		const double Qi = ct.qpQ[gi];
		V3 fs = {0., 0., 0.};
		for (int sj = 0; sj < ncj; ++sj) {  // quadrupole-charge (roles swapped), :387-402
			const int gj = ct.oc[cj] + sj;
			const V3 dr = (rj + rotate(Rj, ld3(ct.chpos, gj))) - pi;
			charge_quadrupole(dr, dot(dr, dr), ei, 0.5 * ct.chq[gj] * Qi, f, m1, u);
			uX += u;
			fs = fs - f;
			a.M = a.M + m1;
		}
		for (int sj = 0; sj < nqj; ++sj) {  // quadrupole-quadrupole, :403-421
			const int gj = ct.oq[cj] + sj;
			const V3 dr = pi - (rj + rotate(Rj, ld3(ct.qppos, gj)));
			quadrupole_quadrupole(dr, dot(dr, dr), ei, rotate(Rj, ld3(ct.qpe, gj)), .75 * Qi * ct.qpQ[gj], f, m1, m2, u);
			uX += u;
			fs = fs + f;
			a.M = a.M + m1;
		}
		for (int sj = 0; sj < ndj; ++sj) {  // quadrupole-dipole (roles swapped), :422-440
			const int gj = ct.od[cj] + sj;
			const V3 dr = (rj + rotate(Rj, ld3(ct.dppos, gj))) - pi;
			dipole_quadrupole(dr, dot(dr, dr), rotate(Rj, ld3(ct.dpe, gj)), ei, 1.5 * Qi * ct.dpmy[gj], f, m2, m1, u);
			uX += u;
			fs = fs - f;
			a.M = a.M + m1;
		}
		Fp = Fp + fs;
		a.M = a.M + cross(di, fs);
	}
	for (int si = 0; si < ndi; ++si) {
		const int gi = ct.od[ci] + si;
		const V3 di = rotate(Ri, ld3(ct.dppos, gi));
		const V3 pi = ri + di;
		const V3 ei = rotate(Ri, ld3(ct.dpe, gi));
		const double myi = ct.dpmy[gi];
		V3 fs = {0., 0., 0.};
		for (int sj = 0; sj < ncj; ++sj) {  // dipole-charge (roles swapped), :445-460
			const int gj = ct.oc[cj] + sj;
			const V3 dr = (rj + rotate(Rj, ld3(ct.chpos, gj))) - pi;
			charge_dipole(dr, dot(dr, dr), ei, -ct.chq[gj] * myi, f, m1, u);
			uX += u;
			fs = fs - f;
			a.M = a.M + m1;
		}
		for (int sj = 0; sj < nqj; ++sj) {  // dipole-quadrupole, :461-478
			const int gj = ct.oq[cj] + sj;
			const V3 dr = pi - (rj + rotate(Rj, ld3(ct.qppos, gj)));
			dipole_quadrupole(dr, dot(dr, dr), ei, rotate(Rj, ld3(ct.qpe, gj)), 1.5 * myi * ct.qpQ[gj], f, m1, m2, u);
			uX += u;
			fs = fs + f;
			a.M = a.M + m1;
		}
		for (int sj = 0; sj < ndj; ++sj) {  // dipole-dipole, :479-497
			const int gj = ct.od[cj] + sj;
			const V3 dr = pi - (rj + rotate(Rj, ld3(ct.dppos, gj)));
			const double my2 = myi * ct.dpmy[gj];
			double rfp;
			dipole_dipole(dr, dot(dr, dr), ei, rotate(Rj, ld3(ct.dpe, gj)), my2, my2 * ct.epsRFInvrc3, f, m1, m2, u, rfp);
			uX += u;
			rf += rfp;
			fs = fs + f;
			a.M = a.M + m1;
		}
		Fp = Fp + fs;
		a.M = a.M + cross(di, fs);
	}
	a.F = a.F + Fp;
	if (WITH_VI) {
		a.Vi.x += 0.5 * drm.x * Fp.x;
		a.Vi.y += 0.5 * drm.y * Fp.y;
		a.Vi.z += 0.5 * drm.z * Fp.z;
	}
	a.u6 += w * u6;
	a.uX += w * uX;
	a.rf += w * rf;
	a.vir += w * (drm.x * Fp.x + drm.y * Fp.y + drm.z * Fp.z);
}

}  // namespace ls1
