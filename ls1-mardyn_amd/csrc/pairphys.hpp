// pairphys.hpp — site-site pair bodies for the MI355X force kernels (FP64).
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
// Quaternion::rotateinv (Quaternion.cpp:63-81) = transpose.
LS1_HD V3 rotate_inv(const Rot& R, V3 d) {
	return {R.m[0] * d.x + R.m[3] * d.y + R.m[6] * d.z, R.m[1] * d.x + R.m[4] * d.y + R.m[7] * d.z,
			R.m[2] * d.x + R.m[5] * d.y + R.m[8] * d.z};
}

// ---- LJ 12-6: potforce.h:18-30 / VectorizedCellProcessor::_loopBodyLJ (VectorizedCellProcessor.cpp:173-226).
// dr = r_i - r_j (sites).  f = force on i.  u6 = 6*U (without shift).
LS1_HD void lj(V3 dr, double dr2, double eps24, double sig2, V3& f, double& u6) {
	const double invdr2 = 1. / dr2;
	double lj6 = sig2 * invdr2;
	lj6 = lj6 * lj6 * lj6;
	const double lj12 = lj6 * lj6;
	const double lj12m6 = lj12 - lj6;
	u6 = eps24 * lj12m6;
	const double fac = eps24 * (lj12 + lj12m6) * invdr2;
	f = fac * dr;
}

// ---- charge-charge: potforce.h:190-199.
LS1_HD void charge_charge(V3 dr, double dr2, double q1q2, V3& f, double& u) {
	const double invdr2 = 1.0 / dr2;
	const double invdr = sqrt(invdr2);
	u = q1q2 * invdr;
	f = (u * invdr2) * dr;
}

// ---- charge (site a) - dipole (site b): potforce.h:237-263.  dr = r_a - r_b, e = dipole axis,
// mqmy = -q*my.  fa = force on the charge; mb = torque on the dipole.
LS1_HD void charge_dipole(V3 dr, double dr2, V3 e, double mqmy, V3& fa, V3& mb, double& u) {
	const double invdr2 = 1.0 / dr2;
	const double invdr = sqrt(invdr2);
	const double costj = dot(e, dr) * invdr;
	const double uInvcostj = mqmy * invdr2;
	u = uInvcostj * costj;
	const double partialTjInvdr1 = uInvcostj * invdr;
	const double fac = 3.0 * u * invdr2;
	fa = fac * dr - partialTjInvdr1 * e;
	mb = partialTjInvdr1 * cross(dr, e);
}

// ---- charge (a) - quadrupole (b): potforce.h:205-231.  qQ05 = 0.5*q*Q.
LS1_HD void charge_quadrupole(V3 dr, double dr2, V3 e, double qQ05, V3& fa, V3& mb, double& u) {
	const double invdr2 = 1.0 / dr2;
	const double invdr = sqrt(invdr2);
	const double costj = dot(e, dr) * invdr;
	const double qQinv4dr3 = qQ05 * invdr * invdr2;
	u = qQinv4dr3 * (3.0 * costj * costj - 1);
	const double partialRijInvdr1 = -3.0 * u * invdr2;
	const double partialTjInvdr1 = 6.0 * costj * qQinv4dr3 * invdr;
	const double fac = costj * partialTjInvdr1 * invdr - partialRijInvdr1;
	fa = fac * dr - partialTjInvdr1 * e;
	mb = partialTjInvdr1 * cross(dr, e);
}

// ---- dipole (i) - dipole (j): potforce.h:36-80.  dr = r_i - r_j.  f = force on i, mi / mj torques,
// rf = reaction-field energy contribution of the pair (MyRF -= rffac*cos gamma).
LS1_HD void dipole_dipole(V3 dr, double dr2, V3 ei, V3 ej, double my2, double rffac, V3& f, V3& mi, V3& mj, double& u,
						  double& rf) {
	const double invdr2 = 1. / dr2;
	const double invdr1 = sqrt(invdr2);
	const double myfac = my2 * invdr2 * invdr1;
	double costi = dot(ei, dr), costj = dot(ej, dr);
	const double cosgij = dot(ei, ej);
	costi *= invdr1;
	costj *= invdr1;
	u = myfac * (cosgij - 3. * costi * costj);
	rf = -rffac * cosgij;
	const double partialRijInvdr1 = -3. * u * invdr2;
	const double partialTiInvdr1 = -myfac * 3. * costj * invdr1;
	const double partialTjInvdr1 = -myfac * 3. * costi * invdr1;
	const double partialGij = myfac;
	const double fac = -partialRijInvdr1 + (costi * partialTiInvdr1 + costj * partialTjInvdr1) * invdr1;
	f = fac * dr - partialTiInvdr1 * ei - partialTjInvdr1 * ej;
	const V3 eiXej = cross(ei, ej);
	mi = (-partialTiInvdr1) * cross(ei, dr) + (-partialGij + rffac) * eiXej;
	mj = (-partialTjInvdr1) * cross(ej, dr) + (partialGij - rffac) * eiXej;
}

// ---- quadrupole (i) - quadrupole (j): potforce.h:86-133.  q2075 = 0.75*Qi*Qj.
LS1_HD void quadrupole_quadrupole(V3 dr, double dr2, V3 ei, V3 ej, double q2075, V3& f, V3& mi, V3& mj, double& u) {
	const double invdr2 = 1. / dr2;
	const double invdr1 = sqrt(invdr2);
	const double qfac = q2075 * invdr2 * invdr2 * invdr1;
	double costi = dot(ei, dr), costj = dot(ej, dr);
	const double cosgij = dot(ei, ej);
	costi *= invdr1;
	costj *= invdr1;
	const double cos2ti = costi * costi, cos2tj = costj * costj;
	const double term = (cosgij - 5. * costi * costj);
	u = qfac * (1. - 5. * (cos2ti + cos2tj) - 15. * cos2ti * cos2tj + 2. * term * term);
	const double partialRijInvdr1 = -5. * u * invdr2;
	const double partialTiInvdr1 = -qfac * 10. * (costi + 3. * costi * cos2tj + 2. * costj * term) * invdr1;
	const double partialTjInvdr1 = -qfac * 10. * (costj + 3. * cos2ti * costj + 2. * costi * term) * invdr1;
	const double partialGij = qfac * 4. * term;
	const double fac = -partialRijInvdr1 + (costi * partialTiInvdr1 + costj * partialTjInvdr1) * invdr1;
	f = fac * dr - partialTiInvdr1 * ei - partialTjInvdr1 * ej;
	const V3 eiXej = cross(ei, ej);
	mi = (-partialTiInvdr1) * cross(ei, dr) - partialGij * eiXej;
	mj = (-partialTjInvdr1) * cross(ej, dr) + partialGij * eiXej;
}

// ---- dipole (a) - quadrupole (b): potforce.h:139-184.  dr = r_a - r_b.  myq15 = 1.5*my*Q.
// f = force on the dipole site, ma / mb torques on dipole / quadrupole.
LS1_HD void dipole_quadrupole(V3 dr, double dr2, V3 ea, V3 eb, double myq15, V3& f, V3& ma, V3& mb, double& u) {
	const double invdr2 = 1. / dr2;
	const double invdr1 = sqrt(invdr2);
	const double myqfac = myq15 * invdr2 * invdr2;
	double costi = dot(ea, dr), costj = dot(eb, dr);
	const double cosgij = dot(ea, eb);
	costi *= invdr1;
	costj *= invdr1;
	const double cos2tj = costj * costj;
	u = myqfac * (-costi * (5. * cos2tj - 1.) + 2. * cosgij * costj);
	const double partialRijInvdr1 = -4. * u * invdr2;
	const double partialTiInvdr1 = myqfac * (-5. * cos2tj + 1.) * invdr1;
	const double partialTjInvdr1 = myqfac * 2. * (-5. * costi * costj + cosgij) * invdr1;
	const double partialGij = myqfac * 2. * costj;
	const double fac = -partialRijInvdr1 + (costi * partialTiInvdr1 + costj * partialTjInvdr1) * invdr1;
	f = fac * dr - partialTiInvdr1 * ea - partialTjInvdr1 * eb;
	const V3 eiXej = cross(ea, eb);
	ma = (-partialTiInvdr1) * cross(ea, dr) - partialGij * eiXej;
	mb = (-partialTjInvdr1) * cross(eb, dr) + partialGij * eiXej;
}

}  // namespace ls1
