// pairphys.hpp — site-site pair bodies for the MI355X force kernels (FP64).
//
// Notation: r = site-site vector, ir1 = 1/|r|, ir2 = 1/r^2, ca / cb = direction cosines of the two axes with r, cg = cosine between
// the axes, w_* = radial prefactors, dUd*_over_r = partial derivatives of the pair energy divided by |r|.
//
// One-sided ("full shell") formulation: every function returns what molecule i receives from site j —
// force f on i's site, torque m on i, pair energy u — so each lane owns its molecule's accumulators and no
// atomics or cross-lane reductions are needed in the hot loop.  The physics is that of the reference's scalar
// bodies (/root/reference/src/molecules/potforce.h:18-263) with the role handling of PotForce (:282-503) and of
// VectorizedCellProcessor's swapped-role call sites (adapter/VectorizedCellProcessor.cpp:1323-1481,1849-2006...).
//
// LS1_HD expands to __host__ __device__ under hipcc and to nothing under a plain C++ compiler, so the CPU test
// suite can compile this header and compare it with the oracle without a GPU (tests/hostshim/).
#pragma once
#include <math.h>

#ifndef LS1_HD
#if defined(__HIPCC__)
#define LS1_HD __host__ __device__ __forceinline__
#else
#define LS1_HD inline
#endif
#endif

// Reciprocal and square root of the pair bodies.  Default: IEEE division / sqrt (what the generic and the brick kernels use:
// their results are bitwise equal to each other, and the host shim compiles the same expressions).  A translation unit may
// define LS1_PAIR_RCP / LS1_PAIR_SQRT before including this header to trade the last bit for instruction count
// (kernels_force_mslist.hip: v_rcp_f64 / v_rsq_f64 with two Newton steps, 1.1e-16 relative — an IEEE division is ~28
// instructions of a 60-instruction LJ site pair).
#ifndef LS1_PAIR_RCP
#define LS1_PAIR_RCP(x) (1. / (x))
#endif
#ifndef LS1_PAIR_SQRT
#define LS1_PAIR_SQRT(x) sqrt(x)
#endif

namespace ls1 {

struct V3 {
	double x, y, z;
};
LS1_HD V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
LS1_HD V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
LS1_HD V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
LS1_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LS1_HD V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// Rotation matrix of a (normalised) quaternion q = (w,x,y,z): Quaternion::rotate
// (/root/reference/src/molecules/Quaternion.cpp:45-61).
struct Rot {
	double m[9];
};
LS1_HD Rot rot_of(double w, double x, double y, double z) {
	const double ww = w * w, xx = x * x, yy = y * y, zz = z * z;
	const double wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
	Rot R;
	R.m[0] = ww + xx - yy - zz;
	R.m[1] = 2. * (xy - wz);
	R.m[2] = 2. * (wy + xz);
	R.m[3] = 2. * (wz + xy);
	R.m[4] = ww - xx + yy - zz;
	R.m[5] = 2. * (yz - wx);
	R.m[6] = 2. * (xz - wy);
	R.m[7] = 2. * (wx + yz);
	R.m[8] = ww - xx - yy + zz;
	return R;
}
LS1_HD V3 rotate(const Rot& R, V3 d) {
	return {R.m[0] * d.x + R.m[1] * d.y + R.m[2] * d.z, R.m[3] * d.x + R.m[4] * d.y + R.m[5] * d.z,
			R.m[6] * d.x + R.m[7] * d.y + R.m[8] * d.z};
}
// LINEAR molecules (every site on the body z axis: the 2CLJ / 2CLJQ / 2CLJD family, ethane): a site offset (0, 0, d) turns into
// d times the third COLUMN of the rotation matrix — three doubles per molecule instead of nine, one multiplication per component
// instead of three (the products with the zero components vanish exactly, so the result is the full matrix's to the last bit
// except for the sign of zeros).  Used by the LJ-only pair-stream kernel (kernels_force_mslist.hip).
struct RotAxis {
	V3 ez;
};
LS1_HD RotAxis rot_axis_of(double w, double x, double y, double z) {
	return {{2. * (w * y + x * z), 2. * (y * z - w * x), w * w - x * x - y * y + z * z}};
}
LS1_HD V3 rotate(const RotAxis& R, V3 d) { return {R.ez.x * d.z, R.ez.y * d.z, R.ez.z * d.z}; }
// Quaternion::rotateinv (Quaternion.cpp:63-81) = transpose.
LS1_HD V3 rotate_inv(const Rot& R, V3 d) {
	return {R.m[0] * d.x + R.m[3] * d.y + R.m[6] * d.z, R.m[1] * d.x + R.m[4] * d.y + R.m[7] * d.z,
			R.m[2] * d.x + R.m[5] * d.y + R.m[8] * d.z};
}

// ---- LJ 12-6: potforce.h:18-30 / VectorizedCellProcessor::_loopBodyLJ (VectorizedCellProcessor.cpp:173-226).
// dr = r_i - r_j (sites).  f = force on i.  u6 = 6*U (without shift).
LS1_HD void lj(V3 dr, double dr2, double eps24, double sig2, V3& f, double& u6) {
	const double ir2 = LS1_PAIR_RCP(dr2);
	double lj6 = sig2 * ir2;
	lj6 = lj6 * lj6 * lj6;
	const double lj12 = lj6 * lj6;
	const double lj12m6 = lj12 - lj6;
	u6 = eps24 * lj12m6;
	const double fac = eps24 * (lj12 + lj12m6) * ir2;
	f = fac * dr;
}

// ---- charge-charge: potforce.h:190-199.
LS1_HD void charge_charge(V3 dr, double dr2, double q1q2, V3& f, double& u) {
	const double ir2 = LS1_PAIR_RCP(dr2);
	const double ir1 = LS1_PAIR_SQRT(ir2);
	u = q1q2 * ir1;
	f = (u * ir2) * dr;
}

// ---- charge (site a) - dipole (site b): potforce.h:237-263.  dr = r_a - r_b, e = dipole axis,
// neg_q_my = -q*my.  fa = force on the charge; mb = torque on the dipole.
LS1_HD void charge_dipole(V3 dr, double dr2, V3 e, double neg_q_my, V3& fa, V3& mb, double& u) {
	const double ir2 = LS1_PAIR_RCP(dr2);
	const double ir1 = LS1_PAIR_SQRT(ir2);
	const double cb = dot(e, dr) * ir1;
	const double k_cd = neg_q_my * ir2;
	u = k_cd * cb;
	const double dUdb_over_r = k_cd * ir1;
	const double fac = 3.0 * u * ir2;
	fa = fac * dr - dUdb_over_r * e;
	mb = dUdb_over_r * cross(dr, e);
}

// ---- charge (a) - quadrupole (b): potforce.h:205-231.  half_qQ = 0.5*q*Q.
LS1_HD void charge_quadrupole(V3 dr, double dr2, V3 e, double half_qQ, V3& fa, V3& mb, double& u) {
	const double ir2 = LS1_PAIR_RCP(dr2);
	const double ir1 = LS1_PAIR_SQRT(ir2);
	const double cb = dot(e, dr) * ir1;
	const double w_cq = half_qQ * ir1 * ir2;
	u = w_cq * (3.0 * cb * cb - 1);
	const double dUdr_over_r = -3.0 * u * ir2;
	const double dUdb_over_r = 6.0 * cb * w_cq * ir1;
	const double fac = cb * dUdb_over_r * ir1 - dUdr_over_r;
	fa = fac * dr - dUdb_over_r * e;
	mb = dUdb_over_r * cross(dr, e);
}

// ---- dipole (i) - dipole (j): potforce.h:36-80.  dr = r_i - r_j.  f = force on i, mi / mj torques,
// rf = reaction-field energy contribution of the pair (MyRF -= rffac*cos gamma).
LS1_HD void dipole_dipole(V3 dr, double dr2, V3 ei, V3 ej, double my2, double rffac, V3& f, V3& mi, V3& mj, double& u,
						  double& rf) {
	const double ir2 = LS1_PAIR_RCP(dr2);
	const double ir1 = LS1_PAIR_SQRT(ir2);
	const double w_dd = my2 * ir2 * ir1;
	double ca = dot(ei, dr), cb = dot(ej, dr);
	const double cg = dot(ei, ej);
	ca *= ir1;
	cb *= ir1;
	u = w_dd * (cg - 3. * ca * cb);
	rf = -rffac * cg;
	const double dUdr_over_r = -3. * u * ir2;
	const double dUda_over_r = -w_dd * 3. * cb * ir1;
	const double dUdb_over_r = -w_dd * 3. * ca * ir1;
	const double dUdg = w_dd;
	const double fac = -dUdr_over_r + (ca * dUda_over_r + cb * dUdb_over_r) * ir1;
	f = fac * dr - dUda_over_r * ei - dUdb_over_r * ej;
	const V3 ea_x_eb = cross(ei, ej);
	mi = (-dUda_over_r) * cross(ei, dr) + (-dUdg + rffac) * ea_x_eb;
	mj = (-dUdb_over_r) * cross(ej, dr) + (dUdg - rffac) * ea_x_eb;
}

// ---- quadrupole (i) - quadrupole (j): potforce.h:86-133.  qq075 = 0.75*Qi*Qj.
LS1_HD void quadrupole_quadrupole(V3 dr, double dr2, V3 ei, V3 ej, double qq075, V3& f, V3& mi, V3& mj, double& u) {
	const double ir2 = LS1_PAIR_RCP(dr2);
	const double ir1 = LS1_PAIR_SQRT(ir2);
	const double w_qq = qq075 * ir2 * ir2 * ir1;
	double ca = dot(ei, dr), cb = dot(ej, dr);
	const double cg = dot(ei, ej);
	ca *= ir1;
	cb *= ir1;
	const double ca2 = ca * ca, cb2 = cb * cb;
	const double term = (cg - 5. * ca * cb);
	u = w_qq * (1. - 5. * (ca2 + cb2) - 15. * ca2 * cb2 + 2. * term * term);
	const double dUdr_over_r = -5. * u * ir2;
	const double dUda_over_r = -w_qq * 10. * (ca + 3. * ca * cb2 + 2. * cb * term) * ir1;
	const double dUdb_over_r = -w_qq * 10. * (cb + 3. * ca2 * cb + 2. * ca * term) * ir1;
	const double dUdg = w_qq * 4. * term;
	const double fac = -dUdr_over_r + (ca * dUda_over_r + cb * dUdb_over_r) * ir1;
	f = fac * dr - dUda_over_r * ei - dUdb_over_r * ej;
	const V3 ea_x_eb = cross(ei, ej);
	mi = (-dUda_over_r) * cross(ei, dr) - dUdg * ea_x_eb;
	mj = (-dUdb_over_r) * cross(ej, dr) + dUdg * ea_x_eb;
}

// ---- dipole (a) - quadrupole (b): potforce.h:139-184.  dr = r_a - r_b.  dq15 = 1.5*my*Q.
// f = force on the dipole site, ma / mb torques on dipole / quadrupole.
LS1_HD void dipole_quadrupole(V3 dr, double dr2, V3 ea, V3 eb, double dq15, V3& f, V3& ma, V3& mb, double& u) {
	const double ir2 = LS1_PAIR_RCP(dr2);
	const double ir1 = LS1_PAIR_SQRT(ir2);
	const double w_dq = dq15 * ir2 * ir2;
	double ca = dot(ea, dr), cb = dot(eb, dr);
	const double cg = dot(ea, eb);
	ca *= ir1;
	cb *= ir1;
	const double cb2 = cb * cb;
	u = w_dq * (-ca * (5. * cb2 - 1.) + 2. * cg * cb);
	const double dUdr_over_r = -4. * u * ir2;
	const double dUda_over_r = w_dq * (-5. * cb2 + 1.) * ir1;
	const double dUdb_over_r = w_dq * 2. * (-5. * ca * cb + cg) * ir1;
	const double dUdg = w_dq * 2. * cb;
	const double fac = -dUdr_over_r + (ca * dUda_over_r + cb * dUdb_over_r) * ir1;
	f = fac * dr - dUda_over_r * ea - dUdb_over_r * eb;
	const V3 ea_x_eb = cross(ea, eb);
	ma = (-dUda_over_r) * cross(ea, dr) - dUdg * ea_x_eb;
	mb = (-dUdb_over_r) * cross(eb, dr) + dUdg * ea_x_eb;
}

}  // namespace ls1
